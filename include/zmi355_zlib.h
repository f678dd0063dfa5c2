/* zmi355_zlib.h -- the zlib stream ABI exported by libz_mi355.so (the drop-in library; the batch engine it runs on is
 * libzmi355.so, include/zmi355.h).
 *
 * Same symbols, same z_stream layout (112 bytes on LP64) and same return codes as the reference's
 * C ABI crate libz-rs-sys, so a C program or a Rust `extern "C"` block written against
 * libz-rs-sys/include/zlib.h links against libz_mi355.so unchanged.  Behind these entry points the
 * deflate / inflate work runs on the MI355X (kernels in zlib_rs_amd/csrc); without a HIP device the
 * *Init* functions fail with Z_MEM_ERROR and msg "no HIP device" -- there is no CPU fallback.
 *
 * Each declaration cites the reference definition it replaces (libz-rs-sys/src/lib.rs:LINE).
 *
 * Stream semantics on the GPU (documented deviations that stay inside zlib's contract):
 *   deflate()  consumes and buffers input; compressed data is produced when the caller flushes or finishes, and under
 *              Z_NO_FLUSH whenever 4 MiB have come in (the reference emits whenever its pending buffer fills,
 *              deflate.rs:2805-2826): a caller sees output as it goes and a stream holds a few MiB, not its input.
 *              The input is compressed in 64 KiB segments (32 KiB when a call brings less than 8 MiB) side by side --
 *              one match-search workgroup and a few encoder waves each, so a 4 MiB call is 128 workgroups on the
 *              chip -- that start byte aligned (the
 *              empty stored block of Z_SYNC_FLUSH, zlib-rs/src/deflate.rs:2733-2738) and keep the window: a
 *              segment matches into the 27 KiB in front of it, like a preset dictionary (deflate.rs:499-564).
 *              Across separate deflate() calls the last 32 KiB of input stay the window; only Z_FULL_FLUSH
 *              forgets them (deflate.rs:2739-2752).
 *              NOT byte-reproducible across call patterns: where the segments are cut and how large the encoder's pieces are
 *              follows from how much input a call brings (32 / 64 KiB segments, 8 KiB pieces in a call of less than 512
 *              segments), so the same input fed in different chunk sizes gives different -- equally valid -- compressed bytes
 *              (sizes within ~1 % of each other).  The reference is deterministic in its input alone; callers that compare or
 *              deduplicate compressed bytes must feed the same chunks.  The same calls always give the same bytes.
 *   inflate()  decodes as far as the input it is given allows, so the output of a flushed packet is there when the call
 *              returns.  The caller's input is taken a piece at a time and only while the decoder can use it; what a pause
 *              (much output queued, Z_NEED_DICT) leaves unread is handed back.  The device decodes from a checkpoint -- the
 *              last block boundary it reached -- with the window of output in front of it as history
 *              (zmi_inflate_resume, include/zmi355.h; the reference's Mode / BitReader / Window,
 *              zlib-rs/src/inflate.rs:288-320): a block is decoded again only while it is incomplete, memory
 *              is bounded by the block size.  Bytes behind the end of the stream are handed back (avail_in / next_in)
 *              exactly, whenever the end is reached.  Wrapper header / trailer are parsed on the host.  Z_BLOCK stops at
 *              the next block boundary and Z_TREES also behind the next block header, with data_type = unused bits |
 *              64 (last block) | 128 (boundary) | 256 (behind a header), as the reference does (zlib-rs/src/inflate.rs:
 *              1276-1284,1323,1369,1772,1856-1873).  The bytes in front of a corrupt
 *              spot are delivered before Z_DATA_ERROR, as the reference does; header, trailer and deflate-data errors
 *              carry the reference's messages ("invalid stored block lengths", "invalid distance too far back", ...:
 *              the decode kernel reports the cause, inflate.rs State::bad).
 *              ZMI_INFLATE_DEFER=BYTES in the environment (with ZMI_TUNING=1; read once, default off) lets inflate(Z_NO_FLUSH) take small
 *              pieces WITHOUT a device decode per call -- zlib's "output latency": the input is decoded once BYTES have come
 *              in, or when a call flushes, brings no input, brings less than the call before, or asks for a block stop.  A
 *              device decode costs a launch (~170 us) whatever it is given; a caller feeding 16-byte pieces needs this, a
 *              caller feeding 64 KiB and more does not.  Its price: Z_STREAM_END may come from a later call than the one
 *              that delivered the last byte (ask again with avail_in = 0), and bytes behind the end of the stream that
 *              arrived in earlier calls cannot be handed back -- readers of concatenated streams leave it off.
 *   windowBits 9..14 bound the back-references (2^windowBits - 262, deflate.rs:1423-1425); deflateBound is the
 *              reference's bound() (deflate.rs:3193-3287: wrapper, gzip header fields, DICTID, small windows);
 *              the first deflate() call writes the wrapper's header even without input (deflate.rs:2543-2627).
 * Preset dictionaries (deflateSetDictionary / inflateSetDictionary, incl. Z_NEED_DICT and the DICTID check) are
 * supported: the dictionary is the window in front of the first segment.
 * gzip header fields (deflateSetHeader / inflateGetHeader), deflateCopy / inflateCopy, *ResetKeep and *GetDictionary
 * work on the host-side stream state.
 * deflatePrime: the bits go out in front of the next compressed data; since every segment of this engine starts on a
 * byte boundary, a bit count that is not a multiple of 8 is followed by an empty stored block (3 bits + padding +
 * 00 00 FF FF), which keeps the stream valid.  inflatePrime is accepted while no undecoded input is buffered (right
 * after init / reset or at a block boundary: the documented uses).  inflateSync, inflateSyncPoint, inflateMark,
 * inflateValidate, inflateUndermine, inflateBack* follow the reference; inflateCodesUsed reports the entries of the device's
 * decode tables for the most recent dynamic block (roots 9 / 8 with exact-fit sub-tables, so 768...1252 where the reference's
 * roots 10 / 9 give other figures for the same quantity; 0 before the first dynamic block, (ulong)-1 without a stream).
 * The gz* file API (csrc/gz_api.hip) is host code around these entry points.
 */
#ifndef ZMI355_ZLIB_H
#define ZMI355_ZLIB_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZLIB_VERSION "1.3.0-zmi355-0.1.0"
#define ZLIB_VERNUM 0x1300
#define ZLIB_VER_MAJOR 1
#define ZLIB_VER_MINOR 3
#define ZLIB_VER_REVISION 0
#define ZLIB_VER_SUBREVISION 0

/* ---- what zlib's zconf.h supplies (libz-rs-sys/include/zconf.h): type names, const-ness and linkage macros that
 * programs written against zlib.h spell out (libz-rs-sys-cdylib/example.c uses z_const and z_off64_t) ---- */
#if defined(ZLIB_CONST) && !defined(z_const)
#  define z_const const          /* libz-rs-sys/include/zlib.h:95,103,1069 */
#elif !defined(z_const)
#  define z_const
#endif
#ifndef ZEXTERN
#  define ZEXTERN extern
#endif
#ifndef ZEXPORT
#  define ZEXPORT
#endif
#ifndef ZEXPORTVA
#  define ZEXPORTVA
#endif
#ifndef Z_EXTERN
#  define Z_EXTERN extern
#endif
#ifndef Z_EXPORT
#  define Z_EXPORT
#endif
#ifndef Z_EXPORTVA
#  define Z_EXPORTVA
#endif
#ifndef OF
#  define OF(args) args
#endif
#ifndef FAR
#  define FAR
#endif
#ifndef MAX_MEM_LEVEL
#  define MAX_MEM_LEVEL 9
#endif
#ifndef MIN_WBITS
#  define MIN_WBITS 8
#endif

typedef unsigned char Byte;
typedef unsigned char Bytef;
typedef unsigned int uInt;
typedef unsigned long uLong;
typedef char charf;
typedef int intf;
typedef uInt uIntf;
typedef uLong uLongf;
typedef void* voidpf;
typedef void* voidp;
typedef const void* voidpc;
typedef size_t z_size_t;
typedef unsigned int z_crc_t;
#ifndef z_off_t
typedef long z_off_t;            /* libz-rs-sys/include/zlib.h:1714,1793,1813: gzseek / gztell / gzoffset, *_combine */
#endif
#ifndef z_off64_t
typedef long long z_off64_t;     /* the *64 variants; same width as z_off_t on LP64 */
#endif

typedef voidpf (*alloc_func)(voidpf opaque, uInt items, uInt size); /* zlib-rs/src/c_api.rs:8 */
typedef void (*free_func)(voidpf opaque, voidpf address);           /* zlib-rs/src/c_api.rs:9 */

struct internal_state;

/* zlib-rs/src/c_api.rs:54-71 */
typedef struct z_stream_s {
    z_const Bytef* next_in;
    uInt avail_in;
    uLong total_in;
    Bytef* next_out;
    uInt avail_out;
    uLong total_out;
    z_const char* msg;
    struct internal_state* state;
    alloc_func zalloc;
    free_func zfree;
    voidpf opaque;
    int data_type;
    uLong adler;
    uLong reserved;
} z_stream;
typedef z_stream* z_streamp;

/* gzip header information (zlib-rs/src/c_api.rs:174-203) */
typedef struct gz_header_s {
    int text;        /* true if compressed data believed to be text */
    uLong time;      /* modification time */
    int xflags;      /* extra flags (not used when writing a gzip file) */
    int os;          /* operating system */
    Bytef* extra;    /* pointer to extra field or Z_NULL if none */
    uInt extra_len;  /* extra field length (valid if extra != Z_NULL) */
    uInt extra_max;  /* space at extra (only when reading header) */
    Bytef* name;     /* pointer to zero-terminated file name or Z_NULL */
    uInt name_max;   /* space at name (only when reading header) */
    Bytef* comment;  /* pointer to zero-terminated comment or Z_NULL */
    uInt comm_max;   /* space at comment (only when reading header) */
    int hcrc;        /* true if there was or will be a header crc */
    int done;        /* true when done reading gzip header (not used when writing a gzip file) */
} gz_header;
typedef gz_header* gz_headerp;

/* zlib-rs/src/c_api.rs:132-166 */
#define Z_NO_FLUSH 0
#define Z_PARTIAL_FLUSH 1
#define Z_SYNC_FLUSH 2
#define Z_FULL_FLUSH 3
#define Z_FINISH 4
#define Z_BLOCK 5
#define Z_TREES 6
#define Z_OK 0
#define Z_STREAM_END 1
#define Z_NEED_DICT 2
#define Z_ERRNO (-1)
#define Z_STREAM_ERROR (-2)
#define Z_DATA_ERROR (-3)
#define Z_MEM_ERROR (-4)
#define Z_BUF_ERROR (-5)
#define Z_VERSION_ERROR (-6)
#define Z_NO_COMPRESSION 0
#define Z_BEST_SPEED 1
#define Z_BEST_COMPRESSION 9
#define Z_DEFAULT_COMPRESSION (-1)
#define Z_FILTERED 1
#define Z_HUFFMAN_ONLY 2
#define Z_RLE 3
#define Z_FIXED 4
#define Z_DEFAULT_STRATEGY 0
#define Z_BINARY 0
#define Z_TEXT 1
#define Z_ASCII Z_TEXT
#define Z_UNKNOWN 2
#define Z_DEFLATED 8
#define Z_NULL 0
#define MAX_WBITS 15

const char* zlibVersion(void);                                                           /* lib.rs:2156 */
uLong zlibCompileFlags(void);                                                            /* lib.rs:2219 */
const char* zError(int err);                                                             /* lib.rs:2115 */

int deflateInit_(z_streamp strm, int level, const char* version, int stream_size);       /* lib.rs:1918 */
int deflateInit2_(z_streamp strm, int level, int method, int windowBits, int memLevel, int strategy,
                  const char* version, int stream_size);                                 /* lib.rs:2005 */
int deflate(z_streamp strm, int flush);                                                  /* lib.rs:1281 */
int deflateEnd(z_streamp strm);                                                          /* lib.rs:1582 */
int deflateReset(z_streamp strm);                                                        /* lib.rs:1611 */
int deflateParams(z_streamp strm, int level, int strategy);                              /* lib.rs:1658 */
int deflateTune(z_streamp strm, int good_length, int max_lazy, int nice_length, int max_chain); /* lib.rs:2063 */
uLong deflateBound(z_streamp strm, uLong sourceLen);                                     /* lib.rs:1364 */
z_size_t deflateBound_z(z_streamp strm, z_size_t sourceLen);                             /* lib.rs:1345 */
int deflatePending(z_streamp strm, unsigned* pending, int* bits);                        /* lib.rs:1757 */
int deflateSetDictionary(z_streamp strm, const Bytef* dictionary, uInt dictLength);      /* lib.rs:1689 */
int deflateGetDictionary(z_streamp strm, Bytef* dictionary, uInt* dictLength);           /* lib.rs:2332 */
int deflateSetHeader(z_streamp strm, gz_headerp head);                                   /* lib.rs:1319 */
int deflateCopy(z_streamp dest, z_streamp source);                                       /* lib.rs:1837 */
int deflateResetKeep(z_streamp strm);                                                    /* lib.rs:1627 */
int deflatePrime(z_streamp strm, int bits, int value);                                   /* lib.rs:1725 */
int deflateUsed(z_streamp strm, int* bits);                                              /* lib.rs:1800 */

int inflateInit_(z_streamp strm, const char* version, int stream_size);                  /* lib.rs:935 */
int inflateInit2_(z_streamp strm, int windowBits, const char* version, int stream_size); /* lib.rs:967 */
int inflate(z_streamp strm, int flush);                                                  /* lib.rs:636 */
int inflateEnd(z_streamp strm);                                                          /* lib.rs:660 */
int inflateReset(z_streamp strm);                                                        /* lib.rs:1055 */
int inflateReset2(z_streamp strm, int windowBits);                                       /* lib.rs:1082 */
int inflateSetDictionary(z_streamp strm, const Bytef* dictionary, uInt dictLength);      /* lib.rs:1121 */
int inflateGetDictionary(z_streamp strm, Bytef* dictionary, uInt* dictLength);           /* lib.rs:2287 */
int inflateGetHeader(z_streamp strm, gz_headerp head);                                   /* lib.rs:1179 */
int inflateCopy(z_streamp dest, z_streamp source);                                       /* lib.rs:815 */
int inflateResetKeep(z_streamp strm);                                                    /* lib.rs:1233 */
int inflateSync(z_streamp strm);                                                         /* lib.rs:884 */
int inflateSyncPoint(z_streamp strm);                                                    /* lib.rs:901 */
int inflatePrime(z_streamp strm, int bits, int value);                                   /* lib.rs:1029 */
long inflateMark(z_streamp strm);                                                        /* lib.rs:850 */
int inflateValidate(z_streamp strm, int check);                                          /* lib.rs:1216 */
int inflateUndermine(z_streamp strm, int subvert);                                       /* lib.rs:1199 */
unsigned long inflateCodesUsed(z_streamp strm);                                          /* lib.rs:1252 */
/* raw deflate through callbacks (zlib-rs/src/inflate/infback.rs) */
typedef unsigned (*in_func)(void* desc, unsigned char** buf);                            /* zlib-rs/src/c_api.rs:12 */
typedef int (*out_func)(void* desc, unsigned char* buf, unsigned len);                   /* zlib-rs/src/c_api.rs:13 */
int inflateBackInit_(z_streamp strm, int windowBits, unsigned char* window, const char* version, int stream_size); /* lib.rs:697 */
int inflateBack(z_streamp strm, in_func in, void* in_desc, out_func out, void* out_desc); /* lib.rs:741 */
int inflateBackEnd(z_streamp strm);                                                      /* lib.rs:780 */

/* ---- gz* file API (libz-rs-sys/src/gz.rs): gzip files through the stream ABI above; the reader accepts
 * concatenated members (gz.rs:931-932, 1464-1506) and passes non-gzip files through unchanged ---- */
#include <stdarg.h>
struct gzFile_s {            /* public part of the handle, read by zlib.h's gzgetc() macro (gz.rs:27-41) */
    unsigned have;
    unsigned char* next;
    long long pos;
};
typedef struct gzFile_s* gzFile;
gzFile gzopen(const char* path, const char* mode);                                       /* gz.rs:230 */
gzFile gzopen64(const char* path, const char* mode);                                     /* gz.rs:208 */
gzFile gzdopen(int fd, const char* mode);                                                /* gz.rs:258 */
int gzbuffer(gzFile file, unsigned size);                                                /* gz.rs:738 */
int gzsetparams(gzFile file, int level, int strategy);                                   /* gz.rs:2467 */
int gzread(gzFile file, voidp buf, unsigned len);                                        /* gz.rs:969 */
z_size_t gzfread(voidp buf, z_size_t size, z_size_t nitems, gzFile file);                /* gz.rs:1029 */
int gzwrite(gzFile file, voidpc buf, unsigned len);                                      /* gz.rs:1537 */
z_size_t gzfwrite(voidpc buf, z_size_t size, z_size_t nitems, gzFile file);              /* gz.rs:1586 */
int gzprintf(gzFile file, const char* format, ...);                                      /* gz.rs:2707 */
int gzvprintf(gzFile file, const char* format, va_list va);                              /* gz.rs:2729 */
int gzputs(gzFile file, const char* s);                                                  /* gz.rs:2137 */
char* gzgets(gzFile file, char* buf, int len);                                           /* gz.rs:2356 */
int gzputc(gzFile file, int c);                                                          /* gz.rs:2079 */
int gzgetc(gzFile file);                                                                 /* gz.rs:2179 */
int gzgetc_(gzFile file);                                                                /* gz.rs:2223 */
int gzungetc(int c, gzFile file);                                                        /* gz.rs:2247 */
int gzflush(gzFile file, int flush);                                                     /* gz.rs:1928 */
z_off_t gzseek(gzFile file, z_off_t offset, int whence);                                 /* gz.rs:2650 */
z_off64_t gzseek64(gzFile file, z_off64_t offset, int whence);                           /* gz.rs:2530 */
int gzrewind(gzFile file);                                                               /* gz.rs:2667 */
z_off_t gztell(gzFile file);                                                             /* gz.rs:2004 */
z_off64_t gztell64(gzFile file);                                                         /* gz.rs:1971 */
z_off_t gzoffset(gzFile file);                                                           /* gz.rs:2064 */
z_off64_t gzoffset64(gzFile file);                                                       /* gz.rs:2024 */
int gzeof(gzFile file);                                                                  /* gz.rs:870 */
int gzdirect(gzFile file);                                                               /* gz.rs:910 */
int gzclose(gzFile file);                                                                /* gz.rs:600 */
int gzclose_r(gzFile file);                                                              /* gz.rs:627 */
int gzclose_w(gzFile file);                                                              /* gz.rs:676 */
const char* gzerror(gzFile file, int* errnum);                                           /* gz.rs:797 */
void gzclearerr(gzFile file);                                                            /* gz.rs:833 */

int compress(Bytef* dest, uLongf* destLen, const Bytef* source, uLong sourceLen);        /* lib.rs:1447 */
int compress2(Bytef* dest, uLongf* destLen, const Bytef* source, uLong sourceLen, int level); /* lib.rs:1529 */
int compress_z(Bytef* dest, z_size_t* destLen, const Bytef* source, z_size_t sourceLen); /* lib.rs:1379 */
int compress2_z(Bytef* dest, z_size_t* destLen, const Bytef* source, z_size_t sourceLen, int level); /* lib.rs:1471 */
uLong compressBound(uLong sourceLen);                                                    /* lib.rs:1561 */
z_size_t compressBound_z(z_size_t sourceLen);                                            /* lib.rs:1553 */
int uncompress(Bytef* dest, uLongf* destLen, const Bytef* source, uLong sourceLen);      /* lib.rs:499 */
int uncompress2(Bytef* dest, uLongf* destLen, const Bytef* source, uLong* sourceLen);    /* lib.rs:583 */
int uncompress_z(Bytef* dest, z_size_t* destLen, const Bytef* source, z_size_t sourceLen); /* lib.rs:433 */
int uncompress2_z(Bytef* dest, z_size_t* destLen, const Bytef* source, z_size_t* sourceLen); /* lib.rs:518 */

uLong adler32(uLong adler, const Bytef* buf, uInt len);                                  /* lib.rs:340 */
uLong adler32_z(uLong adler, const Bytef* buf, z_size_t len);                            /* lib.rs:307 */
uLong adler32_combine(uLong adler1, uLong adler2, z_off_t len2);                         /* lib.rs:372 */
uLong adler32_combine64(uLong adler1, uLong adler2, z_off64_t len2);                     /* lib.rs:412 */
uLong crc32(uLong crc, const Bytef* buf, uInt len);                                      /* lib.rs:183 */
uLong crc32_z(uLong crc, const Bytef* buf, z_size_t len);                                /* lib.rs:150 */
uLong crc32_combine(uLong crc1, uLong crc2, z_off_t len2);                               /* lib.rs:215 */
uLong crc32_combine64(uLong crc1, uLong crc2, z_off64_t len2);                           /* lib.rs:247 */
uLong crc32_combine_gen(z_off_t len2);                                                   /* lib.rs:268 */
uLong crc32_combine_gen64(z_off64_t len2);                                               /* lib.rs:260 */
uLong crc32_combine_op(uLong crc1, uLong crc2, uLong op);                                /* lib.rs:277 */
const uint32_t* get_crc_table(void);                                                     /* lib.rs:253 */

#define deflateInit(strm, level) deflateInit_((strm), (level), ZLIB_VERSION, (int)sizeof(z_stream))
#define deflateInit2(strm, level, method, windowBits, memLevel, strategy) \
    deflateInit2_((strm), (level), (method), (windowBits), (memLevel), (strategy), ZLIB_VERSION, (int)sizeof(z_stream))
#define inflateInit(strm) inflateInit_((strm), ZLIB_VERSION, (int)sizeof(z_stream))
#define inflateInit2(strm, windowBits) inflateInit2_((strm), (windowBits), ZLIB_VERSION, (int)sizeof(z_stream))
#define inflateBackInit(strm, windowBits, window) \
    inflateBackInit_((strm), (windowBits), (window), ZLIB_VERSION, (int)sizeof(z_stream))
#define zlib_version zlibVersion()
/* zlib.h's gzgetc() reads the public part of the handle and falls back to the function (libz-rs-sys/include/zlib.h) */
#define gzgetc(g) ((g)->have ? ((g)->have--, (g)->pos++, *((g)->next)++) : (gzgetc)(g))

#ifdef __cplusplus
}
#endif
#endif
