// gz_api.hip -- the gz* file API (include/zmi355_zlib.h) on top of this library's own stream ABI.
//
// What the reference does behind these symbols: libz-rs-sys/src/gz.rs (32 entry points, gzopen ... gzvprintf) is
// file-descriptor I/O around deflate() / inflate(): a gzip writer, and a reader that accepts concatenated gzip
// members (gz.rs:931-932, 1464-1506) or passes a non-gzip file through unchanged.  SURVEY.md section 8(f)2: this
// is the container sharded output naturally takes.  Host code only; the compression work is whatever deflate() /
// inflate() of zlib_abi.hip do (GPU).  Like there, no exception may cross the C boundary.
//
// The first three members of the handle are public in zlib's ABI (the gzgetc() macro of zlib.h reads them,
// gz.rs:27-41), so they keep their place and meaning: have, next, pos.
#define ZLIB_CONST 1   // the library itself treats next_in / msg as pointers to const
#include "../../include/zmi355_zlib.h"
#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <new>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <unistd.h>
#include <vector>

namespace {
enum { GZ_NONE = 0, GZ_READ = 7247, GZ_WRITE = 31153, GZ_APPEND = 1 };   // the magic numbers of gz.rs:160-166
enum { LOOK = 0, COPY = 1, GZIP = 2 };
const size_t kBufSize = 128u * 1024u;   // gz.rs:175 GZBUFSIZE
// the device decodes per inflate() call: hand it more than the caller-visible buffer size at a time
const size_t kReadChunk = 4u << 20;

struct GzState {
    // public part
    unsigned have = 0;
    unsigned char* next = nullptr;
    int64_t pos = 0;
    // both directions
    int mode = GZ_NONE;
    int fd = -1;
    std::string path;
    size_t want = kBufSize;
    size_t size = 0;                 // 0: buffers not allocated yet
    std::vector<unsigned char> in;   // read: file bytes; write: bytes waiting for deflate()
    std::vector<unsigned char> out;  // read: decoded bytes (second half: room for gzungetc); write: compressed bytes
    bool direct = false;
    // reading
    int how = LOOK;
    int64_t start = 0;
    bool eof = false, past = false;
    // writing
    int level = Z_DEFAULT_COMPRESSION, strategy = Z_DEFAULT_STRATEGY;
    bool reset = false;              // a deflateReset is due after a Z_FINISH
    // seeking
    int64_t skip = 0;
    bool seek = false;
    // errors
    int err = Z_OK;
    std::string msg;
    z_stream strm;
    bool strm_live = false;
};

GzState* st_of(gzFile f) { return (GzState*)(void*)f; }

void gz_error(GzState* s, int err, const char* msg) {   // gz.rs:467-500
    if (err != Z_OK && err != Z_BUF_ERROR) s->have = 0;   // a fatal error: what is buffered is not to be used
    s->err = err;
    s->msg.clear();
    if (msg == nullptr || err == Z_MEM_ERROR) return;
    s->msg = s->path + ": " + msg;
}

void gz_reset(GzState* s) {   // gz.rs:437-455
    s->have = 0;
    if (s->mode == GZ_READ) { s->eof = false; s->past = false; s->how = LOOK; }
    else s->reset = false;
    s->seek = false;
    gz_error(s, Z_OK, nullptr);
    s->pos = 0;
    s->strm.avail_in = 0;
}

gzFile gz_open(const char* path, int fd, const char* mode) {
    if (!mode || (!path && fd == -1)) return nullptr;
    GzState* s = new (std::nothrow) GzState();
    if (!s) return nullptr;
    memset(&s->strm, 0, sizeof(s->strm));
    bool excl = false, cloexec = false;
    for (const char* m = mode; *m; ++m) {   // gz.rs:83-118
        if (*m >= '0' && *m <= '9') s->level = *m - '0';
        else switch (*m) {
            case 'r': s->mode = GZ_READ; break;
            case 'w': s->mode = GZ_WRITE; break;
            case 'a': s->mode = GZ_APPEND; break;
            case '+': delete s; return nullptr;   // no read + write
            case 'e': cloexec = true; break;
            case 'x': excl = true; break;
            case 'f': s->strategy = Z_FILTERED; break;
            case 'h': s->strategy = Z_HUFFMAN_ONLY; break;
            case 'R': s->strategy = Z_RLE; break;
            case 'F': s->strategy = Z_FIXED; break;
            case 'T': s->direct = true; break;
            default: break;
        }
    }
    if (s->mode == GZ_NONE) { delete s; return nullptr; }
    if (s->mode == GZ_READ) {
        if (s->direct) { delete s; return nullptr; }   // "T" is for writing
        s->direct = true;                              // what an empty file reports
    }
    if (path) s->path = path;
    else { char b[32]; snprintf(b, sizeof b, "<fd:%d>", fd); s->path = b; }
    if (path) {
        int flags = s->mode == GZ_READ ? O_RDONLY
                                       : (O_WRONLY | O_CREAT | (excl ? O_EXCL : 0) | (s->mode == GZ_WRITE ? O_TRUNC : O_APPEND));
        if (cloexec) flags |= O_CLOEXEC;
        s->fd = open(path, flags, 0666);
        if (s->fd == -1) { delete s; return nullptr; }
    } else s->fd = fd;
    if (s->mode == GZ_APPEND) {
        (void)lseek(s->fd, 0, SEEK_END);
        s->mode = GZ_WRITE;
    }
    if (s->mode == GZ_READ) {
        const off_t at = lseek(s->fd, 0, SEEK_CUR);
        s->start = at == (off_t)-1 ? 0 : (int64_t)at;
    }
    gz_reset(s);
    return (gzFile)(void*)s;
}

// ------------------------------------------------------------------ reading
int gz_load(GzState* s, unsigned char* buf, size_t len, size_t* got) {   // read() until len bytes or end of file
    *got = 0;
    while (*got < len) {
        const size_t ask = len - *got > (size_t)1 << 30 ? (size_t)1 << 30 : len - *got;
        const ssize_t r = read(s->fd, buf + *got, ask);
        if (r < 0) { if (errno == EINTR) continue; gz_error(s, Z_ERRNO, strerror(errno)); return -1; }
        if (r == 0) { s->eof = true; break; }
        *got += (size_t)r;
    }
    return 0;
}
int gz_avail(GzState* s) {   // more file bytes behind what inflate() has not taken yet
    if (s->err != Z_OK && s->err != Z_BUF_ERROR) return -1;
    if (!s->eof) {
        if (s->strm.avail_in) memmove(s->in.data(), s->strm.next_in, s->strm.avail_in);
        size_t got = 0;
        if (gz_load(s, s->in.data() + s->strm.avail_in, s->in.size() - s->strm.avail_in, &got) == -1) return -1;
        s->strm.avail_in += (uInt)got;
        s->strm.next_in = s->in.data();
    }
    return 0;
}
int gz_look(GzState* s) {   // gz.rs:1296-1385: what comes next -- a gzip member, plain bytes, or nothing
    if (s->size == 0) {
        s->in.resize(s->want > kReadChunk ? s->want : kReadChunk);
        s->out.resize(s->want << 1);
        s->size = s->want;
        memset(&s->strm, 0, sizeof(s->strm));
        if (inflateInit2_(&s->strm, 15 + 16, ZLIB_VERSION, (int)sizeof(z_stream)) != Z_OK) {
            s->size = 0;
            gz_error(s, Z_MEM_ERROR, "out of memory");
            return -1;
        }
        s->strm_live = true;
    }
    if (s->strm.avail_in < 2) {
        if (gz_avail(s) == -1) return -1;
        if (s->strm.avail_in == 0) return 0;
    }
    if (s->strm.avail_in > 1 && s->strm.next_in[0] == 31 && s->strm.next_in[1] == 139) {
        inflateReset(&s->strm);
        s->how = GZIP;
        s->direct = false;
        return 0;
    }
    if (!s->direct) {   // bytes behind a gzip member that are no gzip member: ignored
        s->strm.avail_in = 0;
        s->eof = true;
        s->have = 0;
        return 0;
    }
    // plain file: what has been read goes out as it is
    s->next = s->out.data();
    const size_t n = s->strm.avail_in < s->size ? s->strm.avail_in : s->size;
    memcpy(s->next, s->strm.next_in, n);
    s->have = (unsigned)n;
    s->strm.next_in += n;
    s->strm.avail_in -= (uInt)n;
    s->how = COPY;
    s->direct = true;
    return 0;
}
int gz_decomp(GzState* s) {   // gz.rs:1456-1509: inflate into strm.next_out until it is full or the member ends
    const unsigned had = s->strm.avail_out;
    for (;;) {
        // this library's inflate() may hold decoded bytes queued from input it took earlier: it is asked first, and the
        // file is read only when it has nothing more to give (room left in the output, all input used).  A refill on
        // "avail_in == 0" alone pulled a 4 MiB chunk per small gzread() while a member was still draining.
        const uInt in_before = s->strm.avail_in, out_before = s->strm.avail_out;
        const int rc = inflate(&s->strm, Z_NO_FLUSH);
        if (rc == Z_STREAM_ERROR || rc == Z_NEED_DICT) { gz_error(s, Z_STREAM_ERROR, "internal error: inflate stream corrupt"); return -1; }
        if (rc == Z_MEM_ERROR) { gz_error(s, Z_MEM_ERROR, "out of memory"); return -1; }
        if (rc == Z_DATA_ERROR) { gz_error(s, Z_DATA_ERROR, s->strm.msg ? s->strm.msg : "compressed data error"); return -1; }
        if (rc == Z_STREAM_END) { s->how = LOOK; break; }
        if (s->strm.avail_out == 0) break;
        if (s->strm.avail_in == 0) {   // everything buffered is decoded and handed out: more of the file
            if (gz_avail(s) == -1) return -1;
            if (s->strm.avail_in == 0) { gz_error(s, Z_BUF_ERROR, "unexpected end of file"); break; }
        } else if (in_before == s->strm.avail_in && out_before == s->strm.avail_out) {
            gz_error(s, Z_STREAM_ERROR, "internal error: inflate made no progress");
            return -1;
        }
    }
    s->have = had - s->strm.avail_out;
    s->next = s->strm.next_out - s->have;
    return 0;
}
// nothing more can come: the file is at its end, inflate() has taken all of it, and no member is open that could still
// have decoded bytes queued (an open member that yields nothing any more has been reported as truncated: Z_BUF_ERROR)
bool gz_drained(const GzState* s) { return s->eof && s->strm.avail_in == 0 && (s->how != GZIP || s->err == Z_BUF_ERROR); }
int gz_fetch(GzState* s) {   // something into the output buffer
    do {
        if (s->how == LOOK) {
            if (gz_look(s) == -1) return -1;
            if (s->how == LOOK) return 0;
        } else if (s->how == COPY) {
            // leftovers of the detection read first, then the file
            size_t got = 0;
            if (s->strm.avail_in) {
                got = s->strm.avail_in < (s->size << 1) ? s->strm.avail_in : (s->size << 1);
                memcpy(s->out.data(), s->strm.next_in, got);
                s->strm.next_in += got;
                s->strm.avail_in -= (uInt)got;
            } else if (gz_load(s, s->out.data(), s->size << 1, &got) == -1) return -1;
            s->have = (unsigned)got;
            s->next = s->out.data();
            return 0;
        } else {
            s->strm.avail_out = (uInt)(s->size << 1);
            s->strm.next_out = s->out.data();
            if (gz_decomp(s) == -1) return -1;
        }
    } while (s->have == 0 && !gz_drained(s));
    return 0;
}
int gz_skip(GzState* s, int64_t len) {
    while (len) {
        if (s->have) {
            const unsigned n = (int64_t)s->have > len ? (unsigned)len : s->have;
            s->have -= n; s->next += n; s->pos += n; len -= n;
        } else if (gz_drained(s)) break;
        else if (gz_fetch(s) == -1) return -1;
    }
    return 0;
}
size_t gz_read(GzState* s, unsigned char* buf, size_t len) {   // gz.rs gz_read
    if (len == 0) return 0;
    if (s->seek) { s->seek = false; if (gz_skip(s, s->skip) == -1) return 0; }
    size_t got = 0;
    do {
        unsigned n = len > UINT_MAX ? UINT_MAX : (unsigned)len;
        if (s->have) {
            if (s->have < n) n = s->have;
            memcpy(buf, s->next, n);
            s->next += n;
            s->have -= n;
        } else if (gz_drained(s)) {
            s->past = true;
            break;
        } else if (s->how == LOOK || n < (s->size << 1)) {
            if (gz_fetch(s) == -1) return 0;
            continue;
        } else if (s->how == COPY) {
            size_t g = 0;
            if (s->strm.avail_in) {   // leftovers of the detection read
                g = s->strm.avail_in < n ? s->strm.avail_in : n;
                memcpy(buf, s->strm.next_in, g);
                s->strm.next_in += g;
                s->strm.avail_in -= (uInt)g;
            } else if (gz_load(s, buf, n, &g) == -1) return 0;
            n = (unsigned)g;
        } else {   // large request: decode straight into the caller's buffer
            s->strm.avail_out = n;
            s->strm.next_out = buf;
            if (gz_decomp(s) == -1) return 0;
            n = s->have;
            s->have = 0;
        }
        len -= n;
        buf += n;
        got += n;
        s->pos += n;
    } while (len);
    return got;
}

// ------------------------------------------------------------------ writing
int gz_init(GzState* s) {
    s->in.resize(s->want << 1);
    if (!s->direct) {
        s->out.resize(s->want > kReadChunk ? s->want : kReadChunk);
        memset(&s->strm, 0, sizeof(s->strm));
        if (deflateInit2_(&s->strm, s->level, Z_DEFLATED, 15 + 16, 8, s->strategy, ZLIB_VERSION, (int)sizeof(z_stream)) != Z_OK) {
            gz_error(s, Z_MEM_ERROR, "out of memory");
            return -1;
        }
        s->strm_live = true;
        s->strm.next_in = nullptr;
    }
    s->size = s->want;
    if (!s->direct) {
        s->strm.avail_out = (uInt)s->out.size();
        s->strm.next_out = s->out.data();
        s->next = s->strm.next_out;
    }
    return 0;
}
int gz_put(GzState* s, const unsigned char* p, size_t n) {   // write() all of it
    while (n) {
        const size_t ask = n > (size_t)1 << 30 ? (size_t)1 << 30 : n;
        const ssize_t w = write(s->fd, p, ask);
        if (w < 0) { if (errno == EINTR) continue; gz_error(s, Z_ERRNO, strerror(errno)); return -1; }
        p += w;
        n -= (size_t)w;
    }
    return 0;
}
int gz_comp(GzState* s, int flush) {   // gz.rs gz_comp: strm.next_in / avail_in through deflate() to the file
    if (s->size == 0 && gz_init(s) == -1) return -1;
    if (s->direct) {
        if (gz_put(s, s->strm.next_in, s->strm.avail_in) == -1) return -1;
        s->strm.avail_in = 0;
        return 0;
    }
    if (s->reset) {   // a new member after a Z_FINISH
        if (s->strm.avail_in == 0) return 0;
        deflateReset(&s->strm);
        s->reset = false;
    }
    int rc = Z_OK;
    do {
        if (s->strm.avail_out == 0 || (flush != Z_NO_FLUSH && (flush != Z_FINISH || rc == Z_STREAM_END))) {
            if (s->strm.next_out > s->next && gz_put(s, s->next, (size_t)(s->strm.next_out - s->next)) == -1) return -1;
            if (s->strm.avail_out == 0) { s->strm.avail_out = (uInt)s->out.size(); s->strm.next_out = s->out.data(); }
            s->next = s->strm.next_out;
        }
        const unsigned had = s->strm.avail_out;
        rc = deflate(&s->strm, flush);
        if (rc == Z_STREAM_ERROR) { gz_error(s, Z_STREAM_ERROR, "internal error: deflate stream corrupt"); return -1; }
        if (rc == Z_MEM_ERROR) { gz_error(s, Z_MEM_ERROR, "out of memory"); return -1; }
        if (had == s->strm.avail_out) break;
    } while (true);
    // whatever deflate() produced in the last round still sits in the buffer: out with it when flushing
    if (flush != Z_NO_FLUSH && s->strm.next_out > s->next) {
        if (gz_put(s, s->next, (size_t)(s->strm.next_out - s->next)) == -1) return -1;
        s->next = s->strm.next_out;
    }
    if (flush == Z_FINISH) s->reset = true;
    return 0;
}
int gz_zero(GzState* s, int64_t len) {   // a forward seek while writing: zeros
    if (s->strm.avail_in && gz_comp(s, Z_NO_FLUSH) == -1) return -1;
    bool first = true;
    while (len) {
        const unsigned n = (int64_t)s->size > len ? (unsigned)len : (unsigned)s->size;
        if (first) { memset(s->in.data(), 0, n); first = false; }
        s->strm.avail_in = n;
        s->strm.next_in = s->in.data();
        s->pos += n;
        if (gz_comp(s, Z_NO_FLUSH) == -1) return -1;
        len -= n;
    }
    return 0;
}
size_t gz_write(GzState* s, const unsigned char* buf, size_t len) {
    const size_t put = len;
    if (len == 0) return 0;
    if (s->size == 0 && gz_init(s) == -1) return 0;
    if (s->seek) { s->seek = false; if (gz_zero(s, s->skip) == -1) return 0; }
    if (len < s->size) {   // small writes gather in the input buffer
        do {
            if (s->strm.avail_in == 0) s->strm.next_in = s->in.data();
            const size_t have = (size_t)((s->strm.next_in + s->strm.avail_in) - s->in.data());
            size_t copy = s->size - have;
            if (copy > len) copy = len;
            memcpy(s->in.data() + have, buf, copy);
            s->strm.avail_in += (uInt)copy;
            s->pos += (int64_t)copy;
            buf += copy;
            len -= copy;
            if (len && gz_comp(s, Z_NO_FLUSH) == -1) return 0;
        } while (len);
    } else {   // large ones go to deflate() directly
        if (s->strm.avail_in && gz_comp(s, Z_NO_FLUSH) == -1) return 0;
        s->strm.next_in = buf;
        do {
            const unsigned n = len > UINT_MAX ? UINT_MAX : (unsigned)len;
            s->strm.avail_in = n;
            s->pos += n;
            if (gz_comp(s, Z_NO_FLUSH) == -1) return 0;
            len -= n;
        } while (len);
    }
    return put;
}
}  // namespace

#define GZ_TRY try {
#define GZ_CATCH(ret) } catch (...) { return (ret); }

extern "C" {

gzFile gzopen(const char* path, const char* mode) { GZ_TRY return path ? gz_open(path, -1, mode) : nullptr; GZ_CATCH(nullptr) }
gzFile gzopen64(const char* path, const char* mode) { return gzopen(path, mode); }
gzFile gzdopen(int fd, const char* mode) { GZ_TRY return fd == -1 ? nullptr : gz_open(nullptr, fd, mode); GZ_CATCH(nullptr) }

int gzbuffer(gzFile file, unsigned size) {   // only before the first read / write
    GzState* s = st_of(file);
    if (!s || (s->mode != GZ_READ && s->mode != GZ_WRITE) || s->size != 0) return -1;
    if ((size << 1) < size) return -1;
    s->want = size < 8 ? 8 : size;
    return 0;
}
int gzread(gzFile file, voidp buf, unsigned len) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_READ || (s->err != Z_OK && s->err != Z_BUF_ERROR)) return -1;
    if ((int)len < 0) { gz_error(s, Z_STREAM_ERROR, "request does not fit in an int"); return -1; }
    const size_t got = gz_read(s, (unsigned char*)buf, len);
    if (got == 0 && s->err != Z_OK && s->err != Z_BUF_ERROR) return -1;
    return (int)got;
    GZ_CATCH(-1)
}
z_size_t gzfread(voidp buf, z_size_t size, z_size_t nitems, gzFile file) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_READ || (s->err != Z_OK && s->err != Z_BUF_ERROR) || size == 0) return 0;
    const z_size_t len = nitems * size;
    if (len / size != nitems) { gz_error(s, Z_STREAM_ERROR, "request does not fit in a size_t"); return 0; }
    return len ? gz_read(s, (unsigned char*)buf, len) / size : 0;
    GZ_CATCH(0)
}
#undef gzgetc   // zlib.h's macro form; here the function itself is defined
int gzgetc(gzFile file) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_READ || (s->err != Z_OK && s->err != Z_BUF_ERROR)) return -1;
    if (s->have) { s->have--; s->pos++; return *s->next++; }
    unsigned char c;
    return gz_read(s, &c, 1) < 1 ? -1 : c;
    GZ_CATCH(-1)
}
int gzgetc_(gzFile file) { return gzgetc(file); }
int gzungetc(int c, gzFile file) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_READ) return -1;
    if (s->how == LOOK && s->have == 0) (void)gz_look(s);   // so that the buffers exist
    if (s->err != Z_OK && s->err != Z_BUF_ERROR) return -1;
    if (s->seek) { s->seek = false; if (gz_skip(s, s->skip) == -1) return -1; }
    if (c < 0 || s->size == 0) return -1;
    if (s->have == 0) {   // the pushed-back bytes live at the end of the double-sized buffer
        s->have = 1;
        s->next = s->out.data() + (s->size << 1) - 1;
        s->next[0] = (unsigned char)c;
        s->pos--;
        s->past = false;
        return c;
    }
    if (s->have == (s->size << 1)) { gz_error(s, Z_DATA_ERROR, "out of room to push characters"); return -1; }
    if (s->next == s->out.data()) {   // slide what is there to the end
        memmove(s->out.data() + (s->size << 1) - s->have, s->next, s->have);
        s->next = s->out.data() + (s->size << 1) - s->have;
    }
    s->have++;
    s->next--;
    s->next[0] = (unsigned char)c;
    s->pos--;
    s->past = false;
    return c;
    GZ_CATCH(-1)
}
char* gzgets(gzFile file, char* buf, int len) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || !buf || len < 1 || s->mode != GZ_READ || (s->err != Z_OK && s->err != Z_BUF_ERROR)) return nullptr;
    if (s->seek) { s->seek = false; if (gz_skip(s, s->skip) == -1) return nullptr; }
    char* str = buf;
    unsigned left = (unsigned)len - 1;
    if (left) do {
        if (s->have == 0 && gz_fetch(s) == -1) return nullptr;
        if (s->have == 0) { s->past = true; break; }
        unsigned n = s->have > left ? left : s->have;
        const unsigned char* eol = (const unsigned char*)memchr(s->next, '\n', n);
        if (eol) n = (unsigned)(eol - s->next) + 1;
        memcpy(buf, s->next, n);
        s->have -= n; s->next += n; s->pos += n;
        left -= n; buf += n;
        if (eol) break;
    } while (left);
    if (buf == str) return nullptr;
    buf[0] = 0;
    return str;
    GZ_CATCH(nullptr)
}
int gzdirect(gzFile file) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s) return 0;
    if (s->mode == GZ_READ && s->how == LOOK && s->have == 0) (void)gz_look(s);
    return s->direct ? 1 : 0;
    GZ_CATCH(0)
}
int gzwrite(gzFile file, voidpc buf, unsigned len) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_WRITE || s->err != Z_OK) return 0;
    if ((int)len < 0) { gz_error(s, Z_DATA_ERROR, "requested length does not fit in int"); return 0; }
    return (int)gz_write(s, (const unsigned char*)buf, len);
    GZ_CATCH(0)
}
z_size_t gzfwrite(voidpc buf, z_size_t size, z_size_t nitems, gzFile file) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_WRITE || s->err != Z_OK || size == 0) return 0;
    const z_size_t len = nitems * size;
    if (len / size != nitems) { gz_error(s, Z_STREAM_ERROR, "request does not fit in a size_t"); return 0; }
    return len ? gz_write(s, (const unsigned char*)buf, len) / size : 0;
    GZ_CATCH(0)
}
int gzputc(gzFile file, int c) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_WRITE || s->err != Z_OK) return -1;
    const unsigned char b = (unsigned char)c;
    return gz_write(s, &b, 1) == 1 ? (int)b : -1;
    GZ_CATCH(-1)
}
int gzputs(gzFile file, const char* str) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || !str || s->mode != GZ_WRITE || s->err != Z_OK) return -1;
    const size_t len = strlen(str);
    if ((int)len < 0 || (unsigned)len != len) { gz_error(s, Z_STREAM_ERROR, "string length does not fit in int"); return -1; }
    const size_t put = gz_write(s, (const unsigned char*)str, len);
    return put < len ? -1 : (int)len;
    GZ_CATCH(-1)
}
int gzvprintf(gzFile file, const char* format, va_list va) {
    // gz.rs:2729-2810: the formatted text must fit the buffer (one less than the size given to gzbuffer(), default
    // 128 KiB - 1); longer output writes nothing and returns 0, as in the reference
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || !format) return Z_STREAM_ERROR;
    if (s->mode != GZ_WRITE || s->err != Z_OK) return Z_STREAM_ERROR;
    if (s->size == 0 && gz_init(s) == -1) return s->err;
    if (s->seek) { s->seek = false; if (gz_zero(s, s->skip) == -1) return s->err; }
    std::vector<char> text(s->size);
    const int need = vsnprintf(text.data(), text.size(), format, va);
    if (need <= 0 || (size_t)need >= s->size) return 0;
    return (int)gz_write(s, (const unsigned char*)text.data(), (size_t)need);
    GZ_CATCH(Z_MEM_ERROR)
}
int gzprintf(gzFile file, const char* format, ...) {
    va_list va;
    va_start(va, format);
    const int r = gzvprintf(file, format, va);
    va_end(va);
    return r;
}
int gzflush(gzFile file, int flush) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_WRITE || s->err != Z_OK) return Z_STREAM_ERROR;
    if (flush < 0 || flush > Z_FINISH) return Z_STREAM_ERROR;
    if (s->seek) { s->seek = false; if (gz_zero(s, s->skip) == -1) return s->err; }
    (void)gz_comp(s, flush);
    return s->err;
    GZ_CATCH(Z_MEM_ERROR)
}
int gzsetparams(gzFile file, int level, int strategy) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_WRITE || s->err != Z_OK || s->direct) return Z_STREAM_ERROR;
    if (level == s->level && strategy == s->strategy) return Z_OK;
    if (s->seek) { s->seek = false; if (gz_zero(s, s->skip) == -1) return s->err; }
    if (s->size) {   // what was written so far keeps the old parameters
        if (s->strm.avail_in && gz_comp(s, Z_BLOCK) == -1) return s->err;
        deflateParams(&s->strm, level, strategy);
    }
    s->level = level;
    s->strategy = strategy;
    return Z_OK;
    GZ_CATCH(Z_MEM_ERROR)
}
int gzrewind(gzFile file) {
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_READ || (s->err != Z_OK && s->err != Z_BUF_ERROR)) return -1;
    if (lseek(s->fd, (off_t)s->start, SEEK_SET) == (off_t)-1) return -1;
    gz_reset(s);
    return 0;
}
long long gzseek64(gzFile file, long long offset, int whence) {
    GZ_TRY
    GzState* s = st_of(file);
    if (!s || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return -1;
    if (s->err != Z_OK && s->err != Z_BUF_ERROR) return -1;
    if (whence != SEEK_SET && whence != SEEK_CUR) return -1;
    if (whence == SEEK_SET) offset -= s->pos;
    else if (s->seek) offset += s->skip;
    s->seek = false;
    if (s->mode == GZ_READ && s->how == COPY && s->pos + offset >= 0) {   // a plain file: a real seek
        const off_t r = lseek(s->fd, (off_t)(offset - (long long)s->have - (long long)s->strm.avail_in), SEEK_CUR);
        if (r == (off_t)-1) return -1;
        s->have = 0;
        s->strm.avail_in = 0;
        s->eof = false;
        s->past = false;
        s->seek = false;
        gz_error(s, Z_OK, nullptr);
        s->pos += offset;
        return s->pos;
    }
    if (offset < 0) {   // backwards: from the start again
        if (s->mode != GZ_READ) return -1;
        offset += s->pos;
        if (offset < 0) return -1;
        if (gzrewind(file) == -1) return -1;
    }
    if (s->mode == GZ_READ) {   // what is decoded already is skipped now
        const unsigned n = (long long)s->have > offset ? (unsigned)offset : s->have;
        s->have -= n; s->next += n; s->pos += n;
        offset -= n;
    }
    if (offset) { s->seek = true; s->skip = offset; }
    return s->pos + offset;
    GZ_CATCH(-1)
}
long gzseek(gzFile file, long offset, int whence) { return (long)gzseek64(file, offset, whence); }
long long gztell64(gzFile file) {
    GzState* s = st_of(file);
    if (!s || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return -1;
    return s->pos + (s->seek ? s->skip : 0);
}
long gztell(gzFile file) { return (long)gztell64(file); }
long long gzoffset64(gzFile file) {
    GzState* s = st_of(file);
    if (!s || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return -1;
    off_t at = lseek(s->fd, 0, SEEK_CUR);
    if (at == (off_t)-1) return -1;
    if (s->mode == GZ_READ) at -= s->strm.avail_in;   // read but not used yet
    return (long long)at;
}
long gzoffset(gzFile file) { return (long)gzoffset64(file); }
int gzeof(gzFile file) {
    GzState* s = st_of(file);
    if (!s || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return 0;
    return s->mode == GZ_READ ? (s->past ? 1 : 0) : 0;
}
const char* gzerror(gzFile file, int* errnum) {
    GzState* s = st_of(file);
    if (!s || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return nullptr;
    if (errnum) *errnum = s->err;
    if (s->err == Z_MEM_ERROR) return "out of memory";
    return s->msg.c_str();
}
void gzclearerr(gzFile file) {
    GzState* s = st_of(file);
    if (!s || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return;
    if (s->mode == GZ_READ) { s->eof = false; s->past = false; }
    gz_error(s, Z_OK, nullptr);
}
int gzclose_r(gzFile file) {
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_READ) return Z_STREAM_ERROR;
    if (s->strm_live) inflateEnd(&s->strm);
    const int err = s->err == Z_BUF_ERROR ? Z_BUF_ERROR : Z_OK;
    const int rc = close(s->fd);
    delete s;
    return rc ? Z_ERRNO : err;
}
int gzclose_w(gzFile file) {
    GzState* s = st_of(file);
    if (!s || s->mode != GZ_WRITE) return Z_STREAM_ERROR;
    int ret = Z_OK;
    try {
        if (s->seek) { s->seek = false; if (gz_zero(s, s->skip) == -1) ret = s->err; }
        if (gz_comp(s, Z_FINISH) == -1) ret = s->err;
    } catch (...) { ret = Z_MEM_ERROR; }
    if (s->strm_live) (void)deflateEnd(&s->strm);
    if (close(s->fd) == -1) ret = Z_ERRNO;
    delete s;
    return ret;
}
int gzclose(gzFile file) {
    GzState* s = st_of(file);
    if (!s) return Z_STREAM_ERROR;
    return s->mode == GZ_READ ? gzclose_r(file) : gzclose_w(file);
}

}  // extern "C"
