// inflate.hip -- batch inflate: one wave per DEFLATE stream (raw / zlib / gzip wrapped).
//
// Reference semantics being reproduced:
//   header / trailer state machine   zlib-rs/src/inflate.rs:927-1275,1398-1430,1814-1831
//   block type, stored, dynamic      zlib-rs/src/inflate.rs:1287-1349,1604-1777
//   code tables                      zlib-rs/src/inflate/inftrees.rs:42-245 (root 10 / 9 / 7; ENOUGH 1332 + 592,
//                                    zlib-rs/src/lib.rs:88-102)
//   hot loop                         zlib-rs/src/inflate.rs:1918-2158 (inflate_fast_help_impl)
//   match copy                       zlib-rs/src/inflate/writer.rs:266-300
// Error behaviour mirrors the reference's Z_DATA_ERROR cases (invalid block type, stored length
// mismatch, too many symbols, invalid code lengths set, missing end-of-block, invalid
// literal/length or distance code, distance too far back) and Z_BUF_ERROR for truncated input
// or a too-small output buffer.
//
// MI355X design: the symbol decode of one stream is serial, so the batch supplies the parallelism --
// one 64-lane wave per stream, its two lookup tables (7.6 KiB) in LDS, ~19 streams resident per CU.
// The bit buffer and all decode state are wave-uniform; the 64 lanes co-operate on table
// construction (replicated entries written lane-parallel), on literal stores and on back-reference
// copies (64 bytes per step, overlap-safe).  Output bytes are produced in place in HBM; a
// back-reference reads what the same wave stored earlier (same-CU L1, in-order memory pipeline).
// Algorithmic HBM traffic: (1/ratio) B read + 1 B written per output byte.
#include "zmi_device.h"
#include "zmi_kernels.h"

// internal status codes resolved by zmi_inflate_verify_kernel (the CRC of the output is only known there)
#define ZMI_TRAILER_SHORT (-1005)     // gzip: CRC present, ISIZE cut off   -> data error if CRC wrong, else buf error
#define ZMI_LENGTH_MISMATCH (-1003)   // gzip: ISIZE wrong                  -> data error either way
#define ZMI_NEED_OUTPUT (-1006)       // output capacity exhausted          -> Z_BUF_ERROR (detail 2)
#define INF_LROOT 10u
#define INF_DROOT 9u
#define INF_LSIZE 1344u
#define INF_DSIZE 592u

// table entry: val << 16 | op << 8 | bits
#define INF_OP_LIT 0x00u
#define INF_OP_BASE 0x10u   // | extra-bit count: length or distance base
#define INF_OP_EOB 0x20u
#define INF_OP_BAD 0x40u
#define INF_OP_LINK 0x80u   // | sub-table index bits; val = sub-table offset
#define INF_ENTRY(val, op, bits) (((uint32_t)(val) << 16) | ((uint32_t)(op) << 8) | (uint32_t)(bits))

struct InfShared {
    uint32_t ltab[INF_LSIZE];
    uint32_t dtab[INF_DSIZE];
    uint16_t sorted[320];
    uint8_t lens[320];
    uint8_t stage[320];
    uint32_t cnt[16];
    uint32_t offs[16];
    uint32_t misc[8];
};

struct InfBits {
    const uint8_t* src;
    uint32_t n;
    uint32_t ipos;   // next unread input byte
    uint64_t hold;
    uint32_t nbits;
};

// 4 input bytes at an arbitrary address: two aligned dword loads + v_alignbyte (the address is
// wave-uniform, so the loads can be served by the scalar cache)
static __device__ __forceinline__ uint32_t inf_load32(const uint8_t* p) {
    const uintptr_t a = (uintptr_t)p;
    const uint32_t* q = (const uint32_t*)(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3u);
    const uint32_t lo = q[0];
    const uint32_t hi = sh ? q[1] : 0u;
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}
// keep at least 33 valid bits in the buffer while input remains
static __device__ __forceinline__ void inf_refill(InfBits& B) {
    if (B.nbits <= 32u) {
        if (B.ipos + 4u <= B.n) {
            B.hold |= (uint64_t)inf_load32(B.src + B.ipos) << B.nbits;
            B.ipos += 4u;
            B.nbits += 32u;
        } else {
            while (B.nbits <= 56u && B.ipos < B.n) {
                B.hold |= (uint64_t)B.src[B.ipos++] << B.nbits;
                B.nbits += 8u;
            }
        }
    }
}
static __device__ __forceinline__ uint32_t inf_peek(const InfBits& B, uint32_t k) {
    return (uint32_t)(B.hold & ((1ull << k) - 1ull));
}
static __device__ __forceinline__ void inf_drop(InfBits& B, uint32_t k) {
    B.hold >>= k;
    B.nbits -= k;
}

static const __device__ uint16_t inf_lbase[31] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27, 31,
                                                  35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 0,  0};
static const __device__ uint8_t inf_lext[31] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                                3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 99, 99};
static const __device__ uint16_t inf_dbase[32] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,
                                                  49,  65,  97,  129, 193, 257,  385,  513,  769,  1025, 1537,
                                                  2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 0,    0};
static const __device__ uint8_t inf_dext[32] = {0, 0, 0,  0,  1,  1,  2,  2,  3,  3,  4,  4,  5,  5,  6,  6,
                                                7, 7, 8,  8,  9,  9,  10, 10, 11, 11, 12, 12, 13, 13, 99, 99};

// kind: 0 = code-length code (symbols are values), 1 = literal/length, 2 = distance
// Builds a two-level lookup table from S->lens[0..nsym).  Returns 0 ok, 1 over-subscribed /
// incomplete, 2 table overflow.  All lanes call; `used` receives the entry count.
static __device__ uint32_t inf_build(InfShared* S, uint32_t kind, uint32_t nsym, uint32_t* tab, uint32_t root,
                                     uint32_t cap) {
    const uint32_t lane = zmi_lane();
    // lane 0: histogram, validity, counting sort of the symbols by (length, index)
    if (lane == 0) {
        for (uint32_t l = 0; l < 16u; ++l) S->cnt[l] = 0;
        for (uint32_t i = 0; i < nsym; ++i) S->cnt[S->lens[i]]++;
        uint32_t maxl = 15u;
        while (maxl > 0u && S->cnt[maxl] == 0u) --maxl;
        int32_t left = 1;
        uint32_t bad = 0;
        for (uint32_t l = 1; l <= 15u; ++l) {
            left <<= 1;
            left -= (int32_t)S->cnt[l];
            if (left < 0) { bad = 1; break; }
        }
        if (!bad && left > 0 && maxl != 0u && (kind == 0u || maxl != 1u)) bad = 1;
        uint32_t o = 0;
        for (uint32_t l = 1; l <= 15u; ++l) { S->offs[l] = o; o += S->cnt[l]; }
        S->offs[0] = 0;
        uint32_t tmp[16];
        for (uint32_t l = 0; l < 16u; ++l) tmp[l] = S->offs[l];
        for (uint32_t i = 0; i < nsym; ++i) {
            uint32_t l = S->lens[i];
            if (l) S->sorted[tmp[l]++] = (uint16_t)i;
        }
        S->misc[0] = bad;
        S->misc[1] = maxl;
        S->misc[2] = o;  // number of coded symbols
    }
    zmi_wave_sync();
    if (S->misc[0]) return 1u;
    const uint32_t maxl = S->misc[1];
    const uint32_t ncoded = S->misc[2];
    const uint32_t rsize = 1u << root;
    for (uint32_t i = lane; i < rsize; i += 64u) tab[i] = INF_ENTRY(0, INF_OP_BAD, 0);
    zmi_wave_sync();
    if (maxl == 0u) return 0u;  // no codes at all: every lookup reports an invalid code

    uint32_t code = 0;       // canonical code of the current symbol (MSB-first)
    uint32_t curlen = 0;
    uint32_t used = rsize;   // next free sub-table slot
    uint32_t sub_prefix = 0xFFFFFFFFu, sub_off = 0, sub_bits = 0;
    for (uint32_t k = 0; k < ncoded; ++k) {
        const uint32_t sym = S->sorted[k];
        const uint32_t l = S->lens[sym];
        code <<= (l - curlen);
        curlen = l;
        // entry payload
        uint32_t ent;
        if (kind == 0u) ent = INF_ENTRY(sym, INF_OP_LIT, l);
        else if (kind == 1u) {
            if (sym < 256u) ent = INF_ENTRY(sym, INF_OP_LIT, l);
            else if (sym == 256u) ent = INF_ENTRY(0, INF_OP_EOB, l);
            else if (sym - 257u < 29u) ent = INF_ENTRY(inf_lbase[sym - 257u], INF_OP_BASE | inf_lext[sym - 257u], l);
            else ent = INF_ENTRY(0, INF_OP_BAD, l);
        } else {
            if (sym < 30u) ent = INF_ENTRY(inf_dbase[sym], INF_OP_BASE | inf_dext[sym], l);
            else ent = INF_ENTRY(0, INF_OP_BAD, l);
        }
        const uint32_t rev = __brev(code) >> (32u - l);  // LSB-first bit pattern
        if (l <= root) {
            const uint32_t nrep = 1u << (root - l);
            for (uint32_t j = lane; j < nrep; j += 64u) tab[rev + (j << l)] = ent;
        } else {
            const uint32_t prefix = rev & (rsize - 1u);
            if (prefix != sub_prefix) {
                // new sub-table: the codes sharing this root prefix are contiguous in canonical
                // order; the longest of them sizes the table
                uint32_t last = l;
                // incremental walk (lengths are non-decreasing in sorted order)
                {
                    uint32_t c = code, cl = l;
                    for (uint32_t k2 = k + 1u; k2 < ncoded; ++k2) {
                        uint32_t l2 = S->lens[S->sorted[k2]];
                        c = (c + 1u) << (l2 - cl);
                        cl = l2;
                        if ((c >> (cl - root)) != (code >> (l - root))) break;
                        last = l2;
                    }
                }
                sub_prefix = prefix;
                sub_bits = last - root;
                sub_off = used;
                used += 1u << sub_bits;
                if (used > cap) return 2u;
                for (uint32_t j = lane; j < (1u << sub_bits); j += 64u) tab[sub_off + j] = INF_ENTRY(0, INF_OP_BAD, 0);
                if (lane == 0) tab[prefix] = INF_ENTRY(sub_off, INF_OP_LINK | sub_bits, root);
                zmi_wave_sync();
            }
            const uint32_t sl = l - root;            // bits of this code inside the sub-table
            const uint32_t srev = rev >> root;
            const uint32_t nrep = 1u << (sub_bits - sl);
            for (uint32_t j = lane; j < nrep; j += 64u) tab[sub_off + srev + (j << sl)] = ent;
        }
        code += 1u;
    }
    zmi_wave_sync();
    return 0u;
}

static __device__ __forceinline__ uint32_t inf_lookup(const uint32_t* tab, uint32_t root, const InfBits& B) {
    uint32_t e = tab[inf_peek(B, root)];
    uint32_t op = (e >> 8) & 0xFFu;
    if (op & INF_OP_LINK) {
        uint32_t sb = op & 0x0Fu;
        e = tab[(e >> 16) + ((uint32_t)(B.hold >> root) & ((1u << sb) - 1u))];
    }
    return e;
}

// wrap: 0 raw, 1 zlib, 2 gzip, 3 auto (zlib or gzip by magic)
__global__ void __launch_bounds__(64) zmi_inflate_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                         const uint32_t* __restrict__ in_len, uint32_t wrap,
                                                         uint8_t* out, const uint64_t* __restrict__ out_off,
                                                         const uint32_t* __restrict__ out_cap,
                                                         uint32_t* __restrict__ out_len, uint32_t* __restrict__ in_used,
                                                         uint32_t* __restrict__ check, int32_t* __restrict__ status) {
    __shared__ InfShared Sh;
    InfShared* S = &Sh;
    const uint32_t lane = zmi_lane();
    const uint32_t s = blockIdx.x;
    InfBits B;
    B.src = in + in_off[s];
    B.n = in_len[s];
    B.ipos = 0;
    B.hold = 0;
    B.nbits = 0;
    uint8_t* dst = out + out_off[s];
    const uint32_t cap = out_cap[s];
    uint32_t opos = 0;
    int32_t st = ZMI_OK;
    uint32_t kind_found = wrap;  // resolved wrapper: 0 raw, 1 zlib, 2 gzip
    uint32_t fixed_ready = 0;

    // ---- wrapper header ----
    if (wrap == 3u) kind_found = (B.n >= 2u && B.src[0] == 0x1Fu && B.src[1] == 0x8Bu) ? 2u : 1u;
    if (kind_found == 1u) {
        if (B.n < 2u) st = ZMI_BUF_ERROR;
        else {
            uint32_t cmf = B.src[0], flg = B.src[1];
            if ((cmf & 0x0Fu) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u) st = ZMI_DATA_ERROR;
            else if (flg & 0x20u) st = 2;  // Z_NEED_DICT: preset dictionaries are not supported in batch mode
            B.ipos = 2;
        }
    } else if (kind_found == 2u) {
        if (B.n < 10u) st = ZMI_BUF_ERROR;
        else if (B.src[0] != 0x1Fu || B.src[1] != 0x8Bu || B.src[2] != 8u || (B.src[3] & 0xE0u)) st = ZMI_DATA_ERROR;
        else {
            uint32_t flg = B.src[3];
            uint32_t p = 10;
            if (flg & 4u) {  // FEXTRA
                if (p + 2u > B.n) st = ZMI_BUF_ERROR;
                else { uint32_t xl = B.src[p] | ((uint32_t)B.src[p + 1u] << 8); p += 2u + xl; }
            }
            if (st == ZMI_OK && (flg & 8u)) {  // FNAME
                while (p < B.n && B.src[p] != 0) ++p;
                ++p;
            }
            if (st == ZMI_OK && (flg & 16u)) {  // FCOMMENT
                while (p < B.n && B.src[p] != 0) ++p;
                ++p;
            }
            if (st == ZMI_OK && (flg & 2u)) p += 2u;  // FHCRC (not verified)
            if (st == ZMI_OK && p > B.n) st = ZMI_BUF_ERROR;
            B.ipos = p;
        }
    }

    // ---- blocks ----
    uint32_t last = 0;
    while (st == ZMI_OK && !last) {
        inf_refill(B);
        if (B.nbits < 3u) { st = ZMI_BUF_ERROR; break; }
        last = inf_peek(B, 1);
        uint32_t type = (inf_peek(B, 3) >> 1);
        inf_drop(B, 3);
        if (type == 0u) {
            // stored: realign to a byte, hand unread bytes back to the input cursor
            inf_drop(B, B.nbits & 7u);
            B.ipos -= B.nbits >> 3;
            B.hold = 0;
            B.nbits = 0;
            if (B.ipos + 4u > B.n) { st = ZMI_BUF_ERROR; break; }
            uint32_t l = B.src[B.ipos] | ((uint32_t)B.src[B.ipos + 1u] << 8);
            uint32_t nl = B.src[B.ipos + 2u] | ((uint32_t)B.src[B.ipos + 3u] << 8);
            B.ipos += 4u;
            if ((l ^ 0xFFFFu) != nl) { st = ZMI_DATA_ERROR; break; }   // "invalid stored block lengths"
            if (B.ipos + l > B.n) { st = ZMI_BUF_ERROR; break; }
            if (opos + l > cap) { st = ZMI_NEED_OUTPUT; break; }
            for (uint32_t i = lane; i < l; i += 64u) dst[opos + i] = B.src[B.ipos + i];
            zmi_wave_sync();
            opos += l;
            B.ipos += l;
            continue;
        }
        if (type == 3u) { st = ZMI_DATA_ERROR; break; }  // "invalid block type"
        if (type == 1u) {
            if (!fixed_ready) {
                for (uint32_t i = lane; i < 288u; i += 64u) S->lens[i] = (uint8_t)(i < 144u ? 8u : (i < 256u ? 9u : (i < 280u ? 7u : 8u)));
                zmi_wave_sync();
                inf_build(S, 1u, 288u, S->ltab, INF_LROOT, INF_LSIZE);
                if (lane < 32u) S->lens[lane] = 5;
                zmi_wave_sync();
                inf_build(S, 2u, 32u, S->dtab, INF_DROOT, INF_DSIZE);
                fixed_ready = 1;
            }
        } else {
            fixed_ready = 0;
            inf_refill(B);
            if (B.nbits < 14u) { st = ZMI_BUF_ERROR; break; }
            uint32_t nlen = inf_peek(B, 5) + 257u; inf_drop(B, 5);
            uint32_t ndist = inf_peek(B, 5) + 1u; inf_drop(B, 5);
            uint32_t ncode = inf_peek(B, 4) + 4u; inf_drop(B, 4);
            if (nlen > 286u || ndist > 30u) { st = ZMI_DATA_ERROR; break; }  // "too many length or distance symbols"
            const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            if (lane < 19u) S->lens[lane] = 0;
            zmi_wave_sync();
            for (uint32_t i = 0; i < ncode; ++i) {
                inf_refill(B);
                if (B.nbits < 3u) { st = ZMI_BUF_ERROR; break; }
                if (lane == 0) S->lens[order[i]] = (uint8_t)inf_peek(B, 3);
                inf_drop(B, 3);
            }
            if (st != ZMI_OK) break;
            zmi_wave_sync();
            // code-length code table lives at the start of dtab (128 entries, root 7)
            if (inf_build(S, 0u, 19u, S->dtab, 7u, INF_DSIZE)) { st = ZMI_DATA_ERROR; break; }  // "invalid code lengths set"
            uint32_t have = 0, prevl = 0;
            const uint32_t total = nlen + ndist;
            while (have < total) {
                inf_refill(B);
                uint32_t e = S->dtab[inf_peek(B, 7)];
                uint32_t eb = e & 0xFFu;
                if (((e >> 8) & 0xFFu) != INF_OP_LIT || eb == 0u) { st = ZMI_DATA_ERROR; break; }
                if (B.nbits < eb) { st = ZMI_BUF_ERROR; break; }
                uint32_t sym = e >> 16;
                uint32_t rep, val;
                if (sym < 16u) {
                    inf_drop(B, eb);
                    if (lane == 0) S->stage[have] = (uint8_t)sym;
                    prevl = sym;
                    ++have;
                    continue;
                }
                uint32_t xb = sym == 16u ? 2u : (sym == 17u ? 3u : 7u);
                if (B.nbits < eb + xb) { st = ZMI_BUF_ERROR; break; }
                inf_drop(B, eb);
                uint32_t x = inf_peek(B, xb);
                inf_drop(B, xb);
                if (sym == 16u) {
                    if (have == 0u) { st = ZMI_DATA_ERROR; break; }  // "invalid bit length repeat"
                    val = prevl; rep = 3u + x;
                } else if (sym == 17u) { val = 0; rep = 3u + x; }
                else { val = 0; rep = 11u + x; }
                if (have + rep > total) { st = ZMI_DATA_ERROR; break; }  // "invalid bit length repeat"
                if (lane == 0) for (uint32_t j = 0; j < rep; ++j) S->stage[have + j] = (uint8_t)val;
                have += rep;
                prevl = val;
            }
            if (st != ZMI_OK) break;
            zmi_wave_sync();
            for (uint32_t i = lane; i < nlen; i += 64u) S->lens[i] = S->stage[i];
            zmi_wave_sync();
            if (S->lens[256] == 0) { st = ZMI_DATA_ERROR; break; }  // "invalid code -- missing end-of-block"
            if (inf_build(S, 1u, nlen, S->ltab, INF_LROOT, INF_LSIZE)) { st = ZMI_DATA_ERROR; break; }  // "invalid literal/lengths set"
            if (lane < ndist) S->lens[lane] = S->stage[nlen + lane];
            zmi_wave_sync();
            if (inf_build(S, 2u, ndist, S->dtab, INF_DROOT, INF_DSIZE)) { st = ZMI_DATA_ERROR; break; }  // "invalid distances set"
        }

        // ---- symbol loop ----
        // Literals are collected (up to 8, in a wave-uniform register) and stored by one masked store;
        // back-references flush them first because they may read those bytes.
        uint64_t litbuf = 0;
        uint32_t nlit = 0;
#define INF_FLUSH_LITS()                                                                  \
        do {                                                                              \
            if (nlit) {                                                                   \
                if (lane < nlit) dst[opos - nlit + lane] = (uint8_t)(litbuf >> (8u * lane)); \
                zmi_wave_sync();                                                          \
                litbuf = 0; nlit = 0;                                                     \
            }                                                                             \
        } while (0)
        for (;;) {
            inf_refill(B);
            uint32_t e = inf_lookup(S->ltab, INF_LROOT, B);
            uint32_t eb = e & 0xFFu, op = (e >> 8) & 0xFFu;
            if (op == INF_OP_BAD || eb == 0u) { st = (B.nbits < 15u && B.ipos >= B.n) ? ZMI_BUF_ERROR : ZMI_DATA_ERROR; break; }  // "invalid literal/length code"
            if (eb > B.nbits) { st = ZMI_BUF_ERROR; break; }
            inf_drop(B, eb);
            if (op == INF_OP_LIT) {
                if (opos >= cap) { st = ZMI_NEED_OUTPUT; break; }
                litbuf |= (uint64_t)(e >> 16) << (8u * nlit);
                ++nlit;
                ++opos;
                if (nlit == 8u) INF_FLUSH_LITS();
                continue;
            }
            if (op == INF_OP_EOB) break;
            // length
            uint32_t xb = op & 0x0Fu;
            if (xb > B.nbits) { st = ZMI_BUF_ERROR; break; }
            uint32_t mlen = (e >> 16) + inf_peek(B, xb);
            inf_drop(B, xb);
            inf_refill(B);
            e = inf_lookup(S->dtab, INF_DROOT, B);
            eb = e & 0xFFu; op = (e >> 8) & 0xFFu;
            if (op == INF_OP_BAD || eb == 0u || !(op & INF_OP_BASE)) { st = (B.nbits < 15u && B.ipos >= B.n) ? ZMI_BUF_ERROR : ZMI_DATA_ERROR; break; }  // "invalid distance code"
            xb = op & 0x0Fu;
            if (eb + xb > B.nbits) { st = ZMI_BUF_ERROR; break; }
            inf_drop(B, eb);
            uint32_t dist = (e >> 16) + inf_peek(B, xb);
            inf_drop(B, xb);
            if (dist > opos) { st = ZMI_DATA_ERROR; break; }  // "invalid distance too far back"
            if (opos + mlen > cap) { st = ZMI_NEED_OUTPUT; break; }
            INF_FLUSH_LITS();
            if (dist >= mlen || dist >= 64u) {
                for (uint32_t base = 0; base < mlen; base += 64u) {
                    uint32_t i = base + lane;
                    if (i < mlen) dst[opos + i] = dst[opos + i - dist];
                    zmi_wave_sync();
                }
            } else {
                // overlapping run shorter than a wave: replicate the dist-byte period
                for (uint32_t base = 0; base < mlen; base += 64u) {
                    uint32_t i = base + lane;
                    if (i < mlen) dst[opos + i] = dst[opos - dist + (i % dist)];
                }
                zmi_wave_sync();
            }
            opos += mlen;
        }
        INF_FLUSH_LITS();
#undef INF_FLUSH_LITS
    }

    // ---- trailer ----
    uint32_t chk = 0;
    if (st == ZMI_OK) {
        // give back whole unread bytes
        B.ipos -= B.nbits >> 3;
        if (kind_found == 1u) {
            if (B.ipos + 4u > B.n) st = ZMI_BUF_ERROR;
            else {
                chk = ((uint32_t)B.src[B.ipos] << 24) | ((uint32_t)B.src[B.ipos + 1u] << 16) |
                      ((uint32_t)B.src[B.ipos + 2u] << 8) | B.src[B.ipos + 3u];
                B.ipos += 4u;
            }
        } else if (kind_found == 2u) {
            // CRC first (4 bytes), ISIZE after 4 more -- a stream cut between the two still reports a
            // CRC mismatch as a data error, as the reference's streaming state machine does
            if (B.ipos + 4u > B.n) st = ZMI_BUF_ERROR;
            else {
                chk = B.src[B.ipos] | ((uint32_t)B.src[B.ipos + 1u] << 8) | ((uint32_t)B.src[B.ipos + 2u] << 16) |
                      ((uint32_t)B.src[B.ipos + 3u] << 24);
                B.ipos += 4u;
                if (B.ipos + 4u > B.n) st = ZMI_TRAILER_SHORT;
                else {
                    uint32_t isize = B.src[B.ipos] | ((uint32_t)B.src[B.ipos + 1u] << 8) |
                                     ((uint32_t)B.src[B.ipos + 2u] << 16) | ((uint32_t)B.src[B.ipos + 3u] << 24);
                    B.ipos += 4u;
                    if (isize != opos) st = ZMI_LENGTH_MISMATCH;  // "incorrect length check" (after the CRC check)
                }
            }
        }
    }
    if (lane == 0) {
        out_len[s] = opos;
        in_used[s] = B.ipos;
        check[s] = chk;
        status[s] = st;
    }
}

// after the checksum kernel: compare trailer values with the checksums of the produced bytes and
// resolve the internal status codes to zlib's numbering.  detail (optional): 0 none, 1 more input
// needed, 2 more output space needed.
__global__ void __launch_bounds__(256) zmi_inflate_verify_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                                 const uint32_t* __restrict__ in_len, uint32_t wrap,
                                                                 const uint32_t* __restrict__ check,
                                                                 const uint32_t* __restrict__ adler,
                                                                 const uint32_t* __restrict__ crc, uint32_t n,
                                                                 int32_t* __restrict__ status, int32_t* __restrict__ detail) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    int32_t st = status[s];
    int32_t det = 0;
    if (st == ZMI_NEED_OUTPUT) { st = ZMI_BUF_ERROR; det = 2; }
    else if (st == ZMI_BUF_ERROR) det = 1;
    else if (st == ZMI_OK || st == ZMI_TRAILER_SHORT || st == ZMI_LENGTH_MISMATCH) {
        uint32_t kind = wrap;
        if (wrap == 3u) {
            const uint8_t* p = in + in_off[s];
            kind = (in_len[s] >= 2u && p[0] == 0x1Fu && p[1] == 0x8Bu) ? 2u : 1u;
        }
        bool bad = (kind == 1u && check[s] != adler[s]) || (kind == 2u && check[s] != crc[s]);  // "incorrect data check"
        if (bad || st == ZMI_LENGTH_MISMATCH) st = ZMI_DATA_ERROR;
        else if (st == ZMI_TRAILER_SHORT) { st = ZMI_BUF_ERROR; det = 1; }
    }
    status[s] = st;
    if (detail) detail[s] = det;
}

extern "C" int zmi_launch_inflate(const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n_streams,
                                  uint32_t wrap, uint8_t* d_out, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                                  uint32_t* d_out_len, uint32_t* d_in_used, uint32_t* d_check, int32_t* d_status,
                                  hipStream_t stream) {
    if (n_streams == 0) return 0;
    ZMI_LAUNCH(zmi_inflate_kernel, dim3(n_streams), dim3(64), 0, stream, d_in, d_in_off, d_in_len, wrap, d_out, d_out_off,
               d_out_cap, d_out_len, d_in_used, d_check, d_status);
    return 0;
}

extern "C" int zmi_launch_inflate_verify(const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                         uint32_t n_streams, uint32_t wrap, const uint32_t* d_check, const uint32_t* d_adler,
                                         const uint32_t* d_crc, int32_t* d_status, int32_t* d_detail, hipStream_t stream) {
    if (n_streams == 0) return 0;
    ZMI_LAUNCH(zmi_inflate_verify_kernel, dim3((n_streams + 255u) / 256u), dim3(256), 0, stream, d_in, d_in_off, d_in_len,
               wrap, d_check, d_adler, d_crc, n_streams, d_status, d_detail);
    return 0;
}
