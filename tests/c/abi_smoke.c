/* abi_smoke.c -- plain C caller of the zlib stream ABI (include/zmi355_zlib.h), the shape of the
 * reference's libz-rs-sys-cdylib/zpipe.c: def() / inf() loops over fixed-size chunks. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zmi355_zlib.h"

#define CHUNK 16384

static int def(const unsigned char* src, size_t n, unsigned char** out, size_t* out_n, int level) {
    z_stream strm;
    unsigned char obuf[CHUNK];
    size_t pos = 0, cap = n + n / 8 + 1024, have = 0;
    int ret, flush;
    *out = malloc(cap);
    memset(&strm, 0, sizeof strm);
    ret = deflateInit(&strm, level);
    if (ret != Z_OK) return ret;
    do {
        size_t k = n - pos < CHUNK ? n - pos : CHUNK;
        strm.next_in = src + pos;
        strm.avail_in = (uInt)k;
        pos += k;
        flush = pos >= n ? Z_FINISH : Z_NO_FLUSH;
        do {
            strm.avail_out = CHUNK;
            strm.next_out = obuf;
            ret = deflate(&strm, flush);
            if (ret == Z_STREAM_ERROR) return ret;
            size_t got = CHUNK - strm.avail_out;
            if (have + got > cap) return Z_BUF_ERROR;
            memcpy(*out + have, obuf, got);
            have += got;
        } while (strm.avail_out == 0);
    } while (flush != Z_FINISH);
    if (ret != Z_STREAM_END) return Z_DATA_ERROR;
    deflateEnd(&strm);
    *out_n = have;
    return Z_OK;
}

static int inf(const unsigned char* src, size_t n, unsigned char* dst, size_t cap, size_t* out_n) {
    z_stream strm;
    size_t pos = 0, have = 0;
    int ret;
    memset(&strm, 0, sizeof strm);
    ret = inflateInit(&strm);
    if (ret != Z_OK) return ret;
    do {
        size_t k = n - pos < CHUNK ? n - pos : CHUNK;
        strm.next_in = src + pos;
        strm.avail_in = (uInt)k;
        pos += k;
        do {
            strm.avail_out = (uInt)(cap - have < CHUNK ? cap - have : CHUNK);
            strm.next_out = dst + have;
            uInt before = strm.avail_out;
            ret = inflate(&strm, pos >= n ? Z_FINISH : Z_NO_FLUSH);
            if (ret == Z_NEED_DICT || ret == Z_DATA_ERROR || ret == Z_MEM_ERROR || ret == Z_STREAM_ERROR) { inflateEnd(&strm); return ret; }
            have += before - strm.avail_out;
        } while (strm.avail_out == 0 && ret != Z_STREAM_END);
    } while (ret != Z_STREAM_END && pos < n);
    inflateEnd(&strm);
    *out_n = have;
    return ret == Z_STREAM_END ? Z_OK : Z_DATA_ERROR;
}

int main(void) {
    size_t n = 3u << 20, i, cn = 0, bn = 0;
    unsigned char* src = malloc(n);
    unsigned char* comp = NULL;
    unsigned char* back = malloc(n);
    unsigned x = 12345;
    for (i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; src[i] = (unsigned char)("etaoin shrdlu"[(x >> 24) % 13]); }
    if (def(src, n, &comp, &cn, 6) != Z_OK) { printf("def failed\n"); return 1; }
    if (inf(comp, cn, back, n, &bn) != Z_OK || bn != n || memcmp(src, back, n)) { printf("inf failed\n"); return 1; }
    uLong a = adler32(1, src, (uInt)n), c = crc32(0, src, (uInt)n);
    printf("abi_smoke ok: %zu -> %zu bytes, adler %08lx crc %08lx, %s\n", n, cn, a, c, zlibVersion());
    return 0;
}
