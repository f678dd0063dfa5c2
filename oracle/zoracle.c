/* zoracle.c -- CPU restatement of the reference's deflate/inflate hot path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call this file; the product (zlib_rs_amd/, libzmi355.so) never does and fails
 * loudly without its HIP library.
 *
 * Parity status: PINNED for inflate, checksums and combine (golden vectors of the reference's own
 * tests, tests/test_oracle.py + tests/golden/), PINNED for the deflate format writer through the
 * byte-exact golden vectors that do not depend on zlib-ng's parse (stored / huffman-only / rle /
 * level-6 `deflate_medium_bypass`, `Ferris`), see the header of zo_deflate below.
 *
 * Each function cites the reference lines (paths relative to /root/reference) whose semantics it
 * restates.  Nothing here is copied from the reference: it is Rust, this is C written from the
 * RFC 1950/1951/1952 formats and the behaviour observed in the cited code.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../zlib_rs_amd/csrc/shardgen.h" /* the workload DEFINITION (not product code paths) */

#define ZO_OK 0
#define ZO_STREAM_END 1
#define ZO_NEED_DICT 2
#define ZO_STREAM_ERROR (-2)
#define ZO_DATA_ERROR (-3)
#define ZO_MEM_ERROR (-4)
#define ZO_BUF_ERROR (-5)

/* ------------------------------------------------------------------------------------------
 * Adler-32: zlib-rs/src/adler32.rs:19-47 (BASE 65521, NMAX 5552 :105-106), adler32/generic.rs:43-81
 * ---------------------------------------------------------------------------------------- */
#define ZO_BASE 65521u
#define ZO_NMAX 5552u

uint32_t zo_adler32(uint32_t adler, const uint8_t* buf, size_t len) {
    uint32_t a = adler & 0xFFFFu, b = (adler >> 16) & 0xFFFFu;
    while (len) {
        size_t k = len < ZO_NMAX ? len : ZO_NMAX;
        len -= k;
        while (k--) {
            a += *buf++;
            b += a;
        }
        a %= ZO_BASE;
        b %= ZO_BASE;
    }
    return (b << 16) | a;
}

/* zlib-rs/src/adler32.rs:58-87 */
uint32_t zo_adler32_combine(uint32_t adler1, uint32_t adler2, uint64_t len2) {
    uint32_t rem = (uint32_t)(len2 % ZO_BASE);
    uint32_t sum1 = adler1 & 0xFFFFu;
    uint32_t sum2 = (rem * sum1) % ZO_BASE;
    sum1 += (adler2 & 0xFFFFu) + ZO_BASE - 1u;
    sum2 += ((adler1 >> 16) & 0xFFFFu) + ((adler2 >> 16) & 0xFFFFu) + ZO_BASE - rem;
    if (sum1 >= ZO_BASE) sum1 -= ZO_BASE;
    if (sum1 >= ZO_BASE) sum1 -= ZO_BASE;
    if (sum2 >= (ZO_BASE << 1)) sum2 -= (ZO_BASE << 1);
    if (sum2 >= ZO_BASE) sum2 -= ZO_BASE;
    return sum1 | (sum2 << 16);
}

/* ------------------------------------------------------------------------------------------
 * CRC-32 (reflected 0xEDB88320): zlib-rs/src/crc32.rs:19-29, crc32/braid.rs:12-86
 * combine: zlib-rs/src/crc32/combine.rs:3-61 (multmodp :26, x2nmodp :49)
 * ---------------------------------------------------------------------------------------- */
#define ZO_POLY 0xEDB88320u
static uint32_t zo_crc_tab[256];
static int zo_crc_ready = 0;
static void zo_crc_init(void) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? ZO_POLY : 0u);
        zo_crc_tab[i] = c;
    }
    zo_crc_ready = 1;
}
uint32_t zo_crc32(uint32_t crc, const uint8_t* buf, size_t len) {
    if (!zo_crc_ready) zo_crc_init();
    crc = ~crc;
    while (len--) crc = zo_crc_tab[(crc ^ *buf++) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}
static uint32_t zo_multmodp(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 31; i >= 0; --i) {
        if ((a >> i) & 1u) p ^= b;
        b = (b >> 1) ^ ((b & 1u) ? ZO_POLY : 0u);
    }
    return p;
}
/* x^(n * 2^k) mod p */
static uint32_t zo_x2nmodp(uint64_t n, uint32_t k) {
    uint32_t p = 0x80000000u; /* x^0 */
    uint32_t sq = 0x40000000u; /* x^1 */
    for (uint32_t i = 0; i < k; ++i) sq = zo_multmodp(sq, sq);
    while (n) {
        if (n & 1u) p = zo_multmodp(sq, p);
        sq = zo_multmodp(sq, sq);
        n >>= 1;
    }
    return p;
}
uint32_t zo_crc32_combine_gen(uint64_t len2) { return zo_x2nmodp(len2, 3); }
uint32_t zo_crc32_combine_op(uint32_t crc1, uint32_t crc2, uint32_t op) { return zo_multmodp(op, crc1) ^ crc2; }
uint32_t zo_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2) {
    return zo_crc32_combine_op(crc1, crc2, zo_crc32_combine_gen(len2));
}

/* ------------------------------------------------------------------------------------------
 * compress_bound: zlib-rs/src/deflate.rs:2975-2991 (+ constants :3157-3176)
 * ---------------------------------------------------------------------------------------- */
uint64_t zo_compress_bound(uint64_t n, int wrap) {
    uint64_t w = wrap == 1 ? 6u : (wrap == 2 ? 18u : 0u);
    return n + (n == 0) + (n < 9) + ((n + 7) >> 3) + 3 + w;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic shards (csrc/shardgen.h) and the reference's low-entropy LCG
 * (test-libz-rs-sys/src/inflate.rs:1981-1993: state = 1664525*state + 1013904223, each output byte
 * repeated `step` times)
 * ---------------------------------------------------------------------------------------- */
void zo_gen_shard(uint64_t seed, uint32_t shard, uint32_t nbytes, uint8_t* out) {
    uint32_t lines = nbytes / ZMI_GEN_LINE;
    for (uint32_t l = 0; l < lines; ++l) zmi_gen_line(seed, shard, l, lines, out + (size_t)ZMI_GEN_LINE * l);
}
void zo_prng_bytes(uint32_t seed, uint8_t* out, size_t len, uint32_t step) {
    uint32_t state = seed;
    size_t i = 0;
    while (i < len) {
        state = state * 1664525u + 1013904223u;
        uint8_t v = (uint8_t)(state >> 24);
        for (uint32_t k = 0; k < step && i < len; ++k) out[i++] = v;
    }
}

/* ------------------------------------------------------------------------------------------
 * Inflate: zlib-rs/src/inflate.rs
 *   wrapper header  :927-1275   block header :1287-1349   stored :1351-1396
 *   dynamic tables  :1604-1777  (+ inflate/inftrees.rs:42-245 validity rules)
 *   symbols         :567-894, 1918-2158   check/length :1398-1430, :1814-1831
 * wrap: 0 raw, 1 zlib, 2 gzip, 3 auto.  Returns the zlib code a one-shot inflate(Z_FINISH) gives:
 * ZO_STREAM_END on success, ZO_DATA_ERROR (+msg), ZO_BUF_ERROR (input or output exhausted),
 * ZO_NEED_DICT.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t* p;
    size_t n, pos;
    uint64_t hold;
    uint32_t bits;
    int eof;
} zo_bits;

static int zo_need(zo_bits* b, uint32_t k) {
    while (b->bits < k) {
        if (b->pos >= b->n) { b->eof = 1; return 0; }
        b->hold |= (uint64_t)b->p[b->pos++] << b->bits;
        b->bits += 8;
    }
    return 1;
}
static uint32_t zo_take(zo_bits* b, uint32_t k) {
    uint32_t v = (uint32_t)(b->hold & ((1ull << k) - 1ull));
    b->hold >>= k;
    b->bits -= k;
    return v;
}

typedef struct {
    uint16_t count[16];
    uint16_t sym[320];
    int nsym_coded;
} zo_huff;

/* returns 0 ok, -1 over-subscribed, +1 incomplete (caller applies the reference's rule:
 * incomplete is an error for the code-length code, and for the others unless max length == 1) */
static int zo_huff_build(zo_huff* h, const uint8_t* lens, int n, int* maxlen) {
    uint16_t offs[16];
    memset(h->count, 0, sizeof h->count);
    for (int i = 0; i < n; ++i) h->count[lens[i]]++;
    h->count[0] = 0;
    int mx = 15;
    while (mx > 0 && h->count[mx] == 0) --mx;
    *maxlen = mx;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= h->count[l];
        if (left < 0) return -1;
    }
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + h->count[l]);
    h->nsym_coded = 0;
    for (int i = 0; i < n; ++i)
        if (lens[i]) { h->sym[offs[lens[i]]++] = (uint16_t)i; h->nsym_coded++; }
    return left > 0 ? 1 : 0;
}
/* canonical decode, one bit at a time; -1 invalid code, -2 out of input */
static int zo_huff_decode(zo_bits* b, const zo_huff* h) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
        if (!zo_need(b, 1)) return -2;
        code |= (int)zo_take(b, 1);
        int c = h->count[l];
        if (code - c < first) return h->sym[index + (code - first)];
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

static const uint16_t zo_lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t zo_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t zo_dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t zo_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

#define ZO_FAIL(code, m) do { *msg = (m); rc = (code); goto done; } while (0)

int zo_inflate(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, int wrap, size_t* out_len, size_t* in_used,
               const char** msg) {
    zo_bits b;
    memset(&b, 0, sizeof b);
    b.p = in;
    b.n = in_len;
    size_t op = 0;
    int rc = ZO_OK;
    const char* dummy;
    if (!msg) msg = &dummy;
    *msg = NULL;
    int kind = wrap;
    if (wrap == 3) kind = (in_len >= 2 && in[0] == 0x1F && in[1] == 0x8B) ? 2 : 1;

    if (kind == 1) { /* inflate.rs:948-985 */
        if (in_len < 2) ZO_FAIL(ZO_BUF_ERROR, NULL);
        uint32_t cmf = in[0], flg = in[1];
        if (((cmf << 8) | flg) % 31u) ZO_FAIL(ZO_DATA_ERROR, "incorrect header check");
        if ((cmf & 15u) != 8u) ZO_FAIL(ZO_DATA_ERROR, "unknown compression method");
        if ((cmf >> 4) + 8u > 15u) ZO_FAIL(ZO_DATA_ERROR, "invalid window size");
        b.pos = 2;
        if (flg & 0x20u) ZO_FAIL(ZO_NEED_DICT, NULL);
    } else if (kind == 2) { /* inflate.rs:990-1275 */
        if (in_len < 10) ZO_FAIL(ZO_BUF_ERROR, NULL);
        if (in[0] != 0x1F || in[1] != 0x8B) ZO_FAIL(ZO_DATA_ERROR, "incorrect header check");
        if (in[2] != 8) ZO_FAIL(ZO_DATA_ERROR, "unknown compression method");
        if (in[3] & 0xE0) ZO_FAIL(ZO_DATA_ERROR, "unknown header flags set");
        uint32_t flg = in[3];
        size_t p = 10;
        if (flg & 4u) {
            if (p + 2 > in_len) ZO_FAIL(ZO_BUF_ERROR, NULL);
            p += 2u + (in[p] | ((size_t)in[p + 1] << 8));
        }
        if (flg & 8u) { while (p < in_len && in[p]) ++p; ++p; }
        if (flg & 16u) { while (p < in_len && in[p]) ++p; ++p; }
        if (flg & 2u) {
            if (p + 2 <= in_len) {
                uint32_t want = in[p] | ((uint32_t)in[p + 1] << 8);
                if ((zo_crc32(0, in, p) & 0xFFFFu) != want) ZO_FAIL(ZO_DATA_ERROR, "header crc mismatch");
            }
            p += 2;
        }
        if (p > in_len) ZO_FAIL(ZO_BUF_ERROR, NULL);
        b.pos = p;
    }

    for (int last = 0; !last;) {
        if (!zo_need(&b, 3)) ZO_FAIL(ZO_BUF_ERROR, NULL);
        last = (int)zo_take(&b, 1);
        uint32_t type = zo_take(&b, 2);
        if (type == 0) { /* inflate.rs:1351-1396 */
            zo_take(&b, b.bits & 7u);
            if (!zo_need(&b, 32)) ZO_FAIL(ZO_BUF_ERROR, NULL);
            uint32_t len = zo_take(&b, 16), nlen = zo_take(&b, 16);
            if ((len ^ 0xFFFFu) != nlen) ZO_FAIL(ZO_DATA_ERROR, "invalid stored block lengths");
            /* hand whole buffered bytes back */
            b.pos -= b.bits >> 3;
            b.bits = 0;
            b.hold = 0;
            { /* Mode::CopyBlock, inflate.rs:1374-1394: min(length, room, input) bytes are copied, whatever is missing */
                uint32_t have = (uint32_t)(b.n - b.pos), room = (uint32_t)(out_cap - op);
                uint32_t copy = len < have ? len : have;
                if (copy > room) copy = room;
                memcpy(out + op, in + b.pos, copy);
                op += copy;
                b.pos += copy;
                if (copy < len) ZO_FAIL(ZO_BUF_ERROR, NULL);
            }
            continue;
        }
        if (type == 3) ZO_FAIL(ZO_DATA_ERROR, "invalid block type");
        zo_huff lh, dh;
        uint8_t lens[320];
        int mx;
        if (type == 1) { /* inflate.rs:1309-1317, inflate/inffixed_tbl.rs */
            for (int i = 0; i < 288; ++i) lens[i] = (uint8_t)(i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)));
            zo_huff_build(&lh, lens, 288, &mx);
            for (int i = 0; i < 30; ++i) lens[i] = 5;
            zo_huff_build(&dh, lens, 30, &mx);
        } else { /* inflate.rs:1604-1777 */
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            if (!zo_need(&b, 14)) ZO_FAIL(ZO_BUF_ERROR, NULL);
            int nlen = (int)zo_take(&b, 5) + 257, ndist = (int)zo_take(&b, 5) + 1, ncode = (int)zo_take(&b, 4) + 4;
            if (nlen > 286 || ndist > 30) ZO_FAIL(ZO_DATA_ERROR, "too many length or distance symbols");
            uint8_t cl[19];
            memset(cl, 0, sizeof cl);
            for (int i = 0; i < ncode; ++i) {
                if (!zo_need(&b, 3)) ZO_FAIL(ZO_BUF_ERROR, NULL);
                cl[order[i]] = (uint8_t)zo_take(&b, 3);
            }
            zo_huff ch;
            if (zo_huff_build(&ch, cl, 19, &mx) != 0) ZO_FAIL(ZO_DATA_ERROR, "invalid code lengths set");
            int have = 0;
            while (have < nlen + ndist) {
                int s = zo_huff_decode(&b, &ch);
                if (s == -2) ZO_FAIL(ZO_BUF_ERROR, NULL);
                if (s < 0) ZO_FAIL(ZO_DATA_ERROR, "invalid code lengths set");
                if (s < 16) { lens[have++] = (uint8_t)s; continue; }
                int rep, val = 0;
                if (s == 16) {
                    if (!zo_need(&b, 2)) ZO_FAIL(ZO_BUF_ERROR, NULL);
                    if (have == 0) ZO_FAIL(ZO_DATA_ERROR, "invalid bit length repeat");
                    val = lens[have - 1];
                    rep = 3 + (int)zo_take(&b, 2);
                } else if (s == 17) {
                    if (!zo_need(&b, 3)) ZO_FAIL(ZO_BUF_ERROR, NULL);
                    rep = 3 + (int)zo_take(&b, 3);
                } else {
                    if (!zo_need(&b, 7)) ZO_FAIL(ZO_BUF_ERROR, NULL);
                    rep = 11 + (int)zo_take(&b, 7);
                }
                if (have + rep > nlen + ndist) ZO_FAIL(ZO_DATA_ERROR, "invalid bit length repeat");
                while (rep--) lens[have++] = (uint8_t)val;
            }
            if (lens[256] == 0) ZO_FAIL(ZO_DATA_ERROR, "invalid code -- missing end-of-block");
            int r = zo_huff_build(&lh, lens, nlen, &mx);
            if (r < 0 || (r > 0 && mx != 1)) ZO_FAIL(ZO_DATA_ERROR, "invalid literal/lengths set");
            r = zo_huff_build(&dh, lens + nlen, ndist, &mx);
            if (r < 0 || (r > 0 && mx != 1 && mx != 0)) ZO_FAIL(ZO_DATA_ERROR, "invalid distances set");
        }
        for (;;) { /* inflate.rs:567-894 */
            int s = zo_huff_decode(&b, &lh);
            if (s == -2) ZO_FAIL(ZO_BUF_ERROR, NULL);
            if (s < 0 || s > 285) ZO_FAIL(ZO_DATA_ERROR, "invalid literal/length code");
            if (s < 256) {
                if (op >= out_cap) ZO_FAIL(ZO_BUF_ERROR, NULL);
                out[op++] = (uint8_t)s;
                continue;
            }
            if (s == 256) break;
            s -= 257;
            if (!zo_need(&b, zo_lext[s])) ZO_FAIL(ZO_BUF_ERROR, NULL);
            uint32_t len = zo_lbase[s] + zo_take(&b, zo_lext[s]);
            int d = zo_huff_decode(&b, &dh);
            if (d == -2) ZO_FAIL(ZO_BUF_ERROR, NULL);
            if (d < 0 || d > 29) ZO_FAIL(ZO_DATA_ERROR, "invalid distance code");
            if (!zo_need(&b, zo_dext[d])) ZO_FAIL(ZO_BUF_ERROR, NULL);
            uint32_t dist = zo_dbase[d] + zo_take(&b, zo_dext[d]);
            if (dist > op) ZO_FAIL(ZO_DATA_ERROR, "invalid distance too far back");
            if (op + len > out_cap) ZO_FAIL(ZO_BUF_ERROR, NULL);
            for (uint32_t i = 0; i < len; ++i, ++op) out[op] = out[op - dist];
        }
    }
    /* trailer: inflate.rs:1398-1430 (check), :1814-1831 (gzip length) */
    b.pos -= b.bits >> 3;
    b.bits = 0;
    if (kind == 1) {
        if (b.pos + 4 > b.n) ZO_FAIL(ZO_BUF_ERROR, NULL);
        uint32_t want = ((uint32_t)in[b.pos] << 24) | ((uint32_t)in[b.pos + 1] << 16) | ((uint32_t)in[b.pos + 2] << 8) | in[b.pos + 3];
        b.pos += 4;
        if (want != zo_adler32(1, out, op)) ZO_FAIL(ZO_DATA_ERROR, "incorrect data check");
    } else if (kind == 2) {
        /* the check value is verified as soon as its 4 bytes are there, the length after 4 more
         * (inflate.rs:1398-1430 then :1814-1831) */
        if (b.pos + 4 > b.n) ZO_FAIL(ZO_BUF_ERROR, NULL);
        uint32_t want = in[b.pos] | ((uint32_t)in[b.pos + 1] << 8) | ((uint32_t)in[b.pos + 2] << 16) | ((uint32_t)in[b.pos + 3] << 24);
        b.pos += 4;
        if (want != zo_crc32(0, out, op)) ZO_FAIL(ZO_DATA_ERROR, "incorrect data check");
        if (b.pos + 4 > b.n) ZO_FAIL(ZO_BUF_ERROR, NULL);
        uint32_t isz = in[b.pos] | ((uint32_t)in[b.pos + 1] << 8) | ((uint32_t)in[b.pos + 2] << 16) | ((uint32_t)in[b.pos + 3] << 24);
        b.pos += 4;
        if (isz != (uint32_t)op) ZO_FAIL(ZO_DATA_ERROR, "incorrect length check");
    }
    rc = ZO_STREAM_END;
done:
    if (out_len) *out_len = op;
    if (in_used) *in_used = b.pos;
    return rc;
}
