// inflate.hip -- batch inflate: one wave per DEFLATE stream (raw / zlib / gzip wrapped).
//
// Reference semantics being reproduced:
//   header / trailer state machine   zlib-rs/src/inflate.rs:927-1275,1398-1430,1814-1831
//   block type, stored, dynamic      zlib-rs/src/inflate.rs:1287-1349,1604-1777
//   code tables                      zlib-rs/src/inflate/inftrees.rs:42-245 (root 9 / 8 / 7 here; ENOUGH 852 + 592 there,
//                                    zlib-rs/src/lib.rs:88-102)
//   hot loop                         zlib-rs/src/inflate.rs:1918-2158 (inflate_fast_help_impl)
//   match copy                       zlib-rs/src/inflate/writer.rs:266-300
// Error behaviour mirrors the reference's Z_DATA_ERROR cases (invalid block type, stored length
// mismatch, too many symbols, invalid code lengths set, missing end-of-block, invalid
// literal/length or distance code, distance too far back) and Z_BUF_ERROR for truncated input
// or a too-small output buffer.
//
// MI355X design: the symbol decode of one stream is serial in its bit position, so the batch supplies the
// parallelism -- one 64-lane wave per stream -- and the work of a stream is split so that no HBM round
// trip sits in a dependent chain:
//   decode  (zmi_inflate_kernel)          Huffman decoding only, two ways.  Fast pass (while >= 4 KiB of input lie ahead):
//           3.75 KiB of the block are staged in LDS and cut into 64 sub-sequences, one per lane, decoded serially by that
//           lane; a prefix code resynchronises, so lanes started at a guess fall into step, lanes whose start is not the
//           exit of the lane below walk again, and the consistent prefix is written (inf_fast_pass).  Token rounds
//           (ends of streams, anything unusual): 64 lanes decode the tokens starting at 64 consecutive bit positions
//           speculatively, the real chain is walked with scalar lane reads, a wave scan places the outputs.  Either way
//           literals go straight to their final place in HBM; a back-reference leaves a 3-byte record (length,
//           distance) in the first bytes of the hole it will fill and sets a bit in a per-stream bitmap (1 bit per
//           output byte).  No window is needed, so a wave holds only its lookup tables and the staged input in LDS
//           (9 KiB, 17 streams per CU).
//   resolve (zmi_inflate_resolve_kernel)  streams the output once through a 6 KiB LDS ring and fills the holes in
//           order; a source up to RES_NEAR bytes back lies in the ring (LDS -> LDS), one further back is read from
//           HBM, where everything in front of the batch is final already (16 streams per CU).
// Algorithmic HBM traffic: (1/ratio) B read + 1 B written per output byte; the two-pass split adds one more
// read and write of the output plus the bitmap (1/8 B per byte).
// The decode kernel has a second instantiation for streams that arrive in pieces (zmi_inflate_resume_dev): it can
// start at a bit offset and reports the last block boundary it reached, see the comment at the kernel.
#include "zmi_device.h"
#include "zmi_kernels.h"

// internal status codes resolved by zmi_inflate_verify_kernel (the CRC of the output is only known there)
#define ZMI_TRAILER_SHORT (-1005)     // gzip: CRC present, ISIZE cut off   -> data error if CRC wrong, else buf error
#define ZMI_LENGTH_MISMATCH (-1003)   // gzip: ISIZE wrong                  -> data error either way
#define ZMI_NEED_OUTPUT (-1006)       // output capacity exhausted          -> Z_BUF_ERROR (detail 2)
#define ZMI_BLOCK_STOP (-1007)        // resumable decode: stopped where the caller asked (in_bit flags) -> Z_BUF_ERROR (detail 3)
// data errors by cause: -3000 - k, reported as Z_DATA_ERROR with detail 16 + k (k indexes the reference's messages,
// zlib-rs/src/inflate.rs: the strings passed to State::bad); plain ZMI_DATA_ERROR stays "cause not recorded"
#define ZMI_DERR(k) (-3000 - (int32_t)(k))
enum { DE_STORED_LEN = 1, DE_BLOCK_TYPE, DE_TOO_MANY_SYMS, DE_CODE_LENGTHS_SET, DE_BIT_LENGTH_REPEAT, DE_MISSING_EOB,
       DE_LITLEN_SET, DE_DIST_SET, DE_TOO_FAR_BACK, DE_HEADER_CHECK, DE_CODE };
#define INF_CHUNK 1024u
#define RES_RING 5120u               // resolve pass: output history kept in LDS: RES_NEAR + RES_SPAN + 258 + RES_BLK and slack; a
                                     // multiple of RES_BLK; with the chunk tables 6.75 KiB per stream, 23 streams per CU.  (Round 4:
                                     // what this kernel lacks is waves to hide its round trips behind, not ring -- 6144 / 2560 ran at
                                     // 35.0 ms per 16 Ki streams, 8192 / 4608 at 37.2, 10240 / 6656 at 45.9, this at 31.7; 4096 / 1152
                                     // with batches of 512 bytes at 33.4: profiles/r04_inflate_experiments.txt)
#define RES_NEAR 1664u               // resolve pass: a back-reference further than this reads its source from HBM (final there:
                                     // everything in front of the batch has been written back), a nearer one from the ring
#define RES_BLK 1024u                // resolve pass: bytes staged per load step
#define RES_SPAN 1024u               // resolve pass: output bytes one batch of holes may span
static_assert(RES_NEAR >= RES_SPAN + 258u + 255u, "a far source must end in front of the write-back frontier (first hole of the batch, rounded down to 256)");
static_assert(1023u + RES_NEAR + RES_SPAN + 258u + RES_BLK <= RES_RING && RES_RING % RES_BLK == 0u, "ring budget");
#define ZMI_NO_SCRATCH (-4)          // Z_MEM_ERROR: the bitmap scratch of the context does not cover this stream
// roots 9 / 8: worst-case table sizes 852 (zlib's ENOUGH_LENS, inftrees.h) and 400 (exhaustive search over all
// complete 30-symbol codes with the exact-fit sub-tables inf_build makes; root 6 gives zlib's 592)
#define INF_LROOT 9u
#define INF_DROOT 8u
#define INF_LSIZE 852u
#define INF_DSIZE 400u

// table entry: val << 16 | op << 8 | bits
#define INF_OP_LIT 0x00u
#define INF_OP_BASE 0x10u   // | extra-bit count: length or distance base
#define INF_OP_EOB 0x20u
#define INF_OP_BAD 0x40u
#define INF_OP_LINK 0x80u   // | sub-table index bits; val = sub-table offset
#define INF_ENTRY(val, op, bits) (((uint32_t)(val) << 16) | ((uint32_t)(op) << 8) | (uint32_t)(bits))

// the compressed bytes of one fast pass (inf_fast_pass): 64 sub-sequences of 60 bytes
#define INF_SUB_BITS 480u   // 15 dwords: an ODD stride between the lanes' read positions (the walks stay within a dword or two of lock step,
                            // so with 14 the three input reads of every token step were 4-way bank conflicts for the whole pass)
#ifndef INF_WARM_BITS
#define INF_WARM_BITS 192u
#endif
#define INF_FAST_BYTES 4096u     // 16 (alignment) + 64 * 60 + 6 (a token may end 48 bits behind the last boundary) + read slack
struct InfShared {
    uint32_t ltab[INF_LSIZE];
    uint32_t dtab[INF_DSIZE];
    uint32_t misc[8];
    uint32_t rs[4];       // resumable decode: where the block being decoded starts (byte, bit, output position, complete)
    union {
        struct {          // block headers, table construction, the token rounds' input chunk
            uint16_t sorted[320];
            uint8_t lens[320];
            uint8_t stage[320];
            uint32_t cnt[16];
            uint32_t offs[16];
            uint32_t fcode[16];   // first canonical code of every length
            uint16_t gid[288];    // second-level tables (inf_build): the run a long code belongs to (bit 15: it starts the run),
            uint16_t goff[288];   // ... a run's table offset
            uint8_t gsb[288];     // ... and index bits
            __attribute__((aligned(16))) uint8_t inbuf[INF_CHUNK + 32];  // staged compressed input (one coalesced load per KiB)
        };
        // a fast pass needs none of those (only the tables) and reloads the chunk behind itself: its staged input lies over
        // them -- 9.0 KiB per stream instead of 11.4, 17 streams per CU instead of 13
        __attribute__((aligned(16))) uint8_t fb[INF_FAST_BYTES];
    };
};

static_assert(sizeof(InfShared) <= 9152u, "the decode kernel's LDS per stream (17 streams per CU)");

// One per workgroup, at file scope: a function that is not inlined reaches it by name, as LDS (through a pointer argument
// it would be a generic pointer -- a 64-bit add and a null check in front of every access)
__shared__ InfShared g_inf_lds;

struct InfBits {
    const uint8_t* src;
    uint32_t n;
    uint32_t ipos;   // next unread input byte
    uint64_t hold;
    uint32_t nbits;
    uint8_t* inbuf;  // LDS chunk holding input bytes [cbase, cbase + INF_CHUNK)
    int32_t cbase;   // may be negative for the first chunk of a misaligned stream
};

// (re)load the LDS input chunk so that it starts at the 16-byte aligned address at or below ipos;
// returns the stream offset of inbuf[0] (negative for the first chunk of a misaligned stream)
static __device__ __noinline__ int32_t inf_load_chunk(const uint8_t* src, uint32_t n, uint32_t ipos, uint8_t* inbuf_) {
    uint8_t* const inbuf = g_inf_lds.inbuf;   // (= inbuf_)
    (void)inbuf_;
    const uint32_t lane = zmi_lane();
    const uint32_t mis = (uint32_t)((uintptr_t)(src + ipos) & 15u);
    const int32_t cbase = (int32_t)ipos - (int32_t)mis;
    zmi_wave_order();
    const int32_t so = cbase + (int32_t)(16u * lane);  // stream offset of this lane's 16 bytes
    uint4 q;
    q.x = q.y = q.z = q.w = 0u;
    if (so >= 0 && (uint32_t)so + 16u <= n) {
        q = *(const uint4*)(src + so);  // 16-byte aligned by construction
    } else if (so + 16 > 0 && so < (int32_t)n) {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        for (int j = 0; j < 16; ++j) {
            int32_t o = so + j;
            if (o >= 0 && (uint32_t)o < n) w[j >> 2] |= (uint32_t)src[o] << (8 * (j & 3));
        }
        q.x = w[0]; q.y = w[1]; q.z = w[2]; q.w = w[3];
    }
    *(uint4*)(inbuf + 16u * lane) = q;
    if (lane < 2u) *(uint4*)(inbuf + INF_CHUNK + 16u * lane) = uint4{0u, 0u, 0u, 0u};
    zmi_wave_order();
    return cbase;
}
static __device__ __forceinline__ uint32_t inf_byte(const InfBits& B, uint32_t i) { return zmi_uniform(B.src[i]); }
// keep at least 33 valid bits in the buffer while input remains; reads come from the LDS chunk
static __device__ __forceinline__ void inf_refill(InfBits& B) {
    if (B.nbits <= 32u) {
        if (B.ipos + 4u <= B.n) {
            uint32_t off = (uint32_t)((int32_t)B.ipos - B.cbase);
            if (off + 4u > INF_CHUNK) {
                B.cbase = (int32_t)zmi_uniform((uint32_t)inf_load_chunk(B.src, B.n, B.ipos, B.inbuf));
                off = (uint32_t)((int32_t)B.ipos - B.cbase);
            }
            B.hold |= (uint64_t)zmi_uniform(zmi_load32u(B.inbuf, off)) << B.nbits;
            B.ipos += 4u;
            B.nbits += 32u;
        } else {
            while (B.nbits <= 56u && B.ipos < B.n) {
                B.hold |= (uint64_t)inf_byte(B, B.ipos++) << B.nbits;
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
static __device__ __forceinline__ uint32_t inf_entry(uint32_t kind, uint32_t sym, uint32_t l) {
    if (kind == 0u) return INF_ENTRY(sym, INF_OP_LIT, l);
    if (kind == 1u) {
        if (sym < 256u) return INF_ENTRY(sym, INF_OP_LIT, l);
        if (sym == 256u) return INF_ENTRY(0, INF_OP_EOB, l);
        if (sym - 257u < 29u) return INF_ENTRY(inf_lbase[sym - 257u], INF_OP_BASE | inf_lext[sym - 257u], l);
        return INF_ENTRY(0, INF_OP_BAD, l);
    }
    if (sym < 30u) return INF_ENTRY(inf_dbase[sym], INF_OP_BASE | inf_dext[sym], l);
    return INF_ENTRY(0, INF_OP_BAD, l);
}

// Lane-parallel where the work is: the length histogram (LDS atomics), every symbol's canonical code (first code of
// its length + its rank among the lower-indexed symbols of that length: ballot + mbcnt, running counts in scalar
// registers) and the root-table entries of the codes that fit the root.  Only the codes longer than the root (few)
// are walked serially, because the sub-tables are sized by looking ahead in canonical order.
static __device__ __noinline__ uint32_t inf_build(InfShared* S_, uint32_t kind, uint32_t nsym, uint32_t* tab_, uint32_t root,
                                     uint32_t cap) {
    InfShared* const S = &g_inf_lds;                           // (= S_)
    uint32_t* const tab = kind == 1u ? S->ltab : S->dtab;      // (= tab_: the literal/length table, or the distance table --
    (void)S_; (void)tab_;                                      //  which the code-length code borrows)
    const uint32_t lane = zmi_lane();
    if (lane < 16u) S->cnt[lane] = 0;
    zmi_wave_sync();
    for (uint32_t i = lane; i < nsym; i += 64u) atomicAdd(&S->cnt[S->lens[i]], 1u);
    zmi_wave_sync();
    if (lane == 0) {
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
        uint32_t o = 0, code = 0;
        S->offs[0] = 0;
        S->fcode[0] = 0;
        for (uint32_t l = 1; l <= 15u; ++l) {
            S->offs[l] = o;
            S->fcode[l] = code;
            o += S->cnt[l];
            code = (code + S->cnt[l]) << 1;
        }
        S->misc[0] = bad;
        S->misc[1] = maxl;
        S->misc[2] = o;  // number of coded symbols
    }
    zmi_wave_sync();
    if (zmi_uniform(S->misc[0])) return 1u;
    const uint32_t maxl = zmi_uniform(S->misc[1]);
    const uint32_t ncoded = zmi_uniform(S->misc[2]);
    const uint32_t rsize = 1u << root;
    for (uint32_t i = lane; i < rsize; i += 64u) tab[i] = INF_ENTRY(0, INF_OP_BAD, 0);
    zmi_wave_sync();
    if (lane == 0u) S->misc[3] = rsize;   // entries this table occupies (the root alone so far); inflateCodesUsed sums them
    if (maxl == 0u) return 0u;  // no codes at all: every lookup reports an invalid code

    // every symbol: its place in the (length, index) order and its code; short codes fill their root entries
    uint32_t run[16];
#pragma unroll
    for (uint32_t d = 0; d < 16u; ++d) run[d] = 0;
    for (uint32_t base = 0; base < nsym; base += 64u) {
        const uint32_t i = base + lane;
        const uint32_t l = i < nsym ? S->lens[i] : 0u;
        uint32_t r = 0;
#pragma unroll
        for (uint32_t d = 1; d < 16u; ++d) {
            const uint64_t m = __ballot(l == d);
            r = l == d ? run[d] + zmi_mbcnt(m) : r;
            run[d] += (uint32_t)__popcll(m);
        }
        if (l != 0u) {
            S->sorted[S->offs[l] + r] = (uint16_t)i;
            if (l <= root) {
                const uint32_t code = S->fcode[l] + r;
                const uint32_t rev = __brev(code) >> (32u - l);   // LSB-first bit pattern
                const uint32_t ent = inf_entry(kind, i, l);
                for (uint32_t j = 0; j < (1u << (root - l)); ++j) tab[rev + (j << l)] = ent;
            }
        }
    }
    zmi_wave_sync();
    if (maxl <= root) return 0u;

    // Codes longer than the root: second-level tables, lane-parallel.  In canonical order (length, then symbol -- the order
    // of S->sorted) the codes that share their first `root` bits are neighbours and their lengths do not decrease, so a
    // sub-table is a run of that order, sized by the run's LAST code.  Three sweeps over the at most 286 long codes:
    // runs (a code starts one when its prefix differs from its neighbour's; the code in front of it closes the run before
    // and sizes it), offsets (a scan over the runs' sizes), entries.  (Until round 3 this was a serial walk with a
    // look-ahead per run: 75-200 K cycles per block on data with many long codes -- the record and random-walk classes of
    // the benchmark spent a quarter to a half of their decode time here.)
    const uint32_t k0 = zmi_uniform(S->offs[root + 1u]);
    const uint32_t nlong = ncoded - k0;
    uint32_t ngroups = 0, carry_p = 0xFFFFFFFFu, carry_l = 0;
    for (uint32_t base = 0; base < nlong; base += 64u) {
        const bool in = base + lane < nlong;
        const uint32_t k = k0 + base + lane;
        const uint32_t sym = in ? S->sorted[k] : 0u;
        const uint32_t l = in ? S->lens[sym] : 16u;
        const uint32_t code = in ? S->fcode[l] + (k - S->offs[l]) : 0u;          // canonical, MSB first
        const uint32_t pfx = in ? code >> (l - root) : 0xFFFFFFFEu;
        const uint32_t up_p = zmi_lane_up1(pfx), up_l = zmi_lane_up1(l);
        const uint32_t before_p = lane == 0u ? carry_p : up_p, before_l = lane == 0u ? carry_l : up_l;
        const bool first = in && pfx != before_p;
        const uint64_t fm = __ballot(first);
        const uint32_t gid = ngroups + zmi_mbcnt(fm) + (first ? 1u : 0u) - 1u;     // the run this code belongs to
        if (in) S->gid[base + lane] = (uint16_t)(gid | (first ? 0x8000u : 0u));
        if (first && gid != 0u) S->gsb[gid - 1u] = (uint8_t)(before_l - root);     // the code in front closed the run before
        ngroups += (uint32_t)__popcll((unsigned long long)fm);
        carry_p = zmi_readlane(pfx, 63u);
        carry_l = zmi_readlane(l, 63u);
    }
    if (lane == 0) S->gsb[ngroups - 1u] = (uint8_t)(maxl - root);                  // the last run ends with the longest code
    zmi_wave_sync();
    uint32_t used = rsize;
    for (uint32_t base = 0; base < ngroups; base += 64u) {
        const uint32_t g = base + lane;
        const uint32_t size = g < ngroups ? 1u << S->gsb[g] : 0u;
        const uint32_t incl = zmi_wave_incl_scan(size);
        if (g < ngroups) S->goff[g] = (uint16_t)(used + incl - size);
        used += zmi_readlane(incl, 63u);
    }
    if (used > cap) return 2u;
    if (lane == 0u) S->misc[3] = used;
    for (uint32_t j = rsize + lane; j < used; j += 64u) tab[j] = INF_ENTRY(0, INF_OP_BAD, 0);
    zmi_wave_sync();
    for (uint32_t base = 0; base < nlong; base += 64u) {
        if (base + lane < nlong) {
            const uint32_t k = k0 + base + lane;
            const uint32_t sym = S->sorted[k];
            const uint32_t l = S->lens[sym];
            const uint32_t code = S->fcode[l] + (k - S->offs[l]);
            const uint32_t rev = __brev(code) >> (32u - l);                        // LSB-first bit pattern
            const uint32_t gw = S->gid[base + lane];
            const uint32_t g = gw & 0x7FFFu;
            const uint32_t sb = S->gsb[g], off = S->goff[g];
            if (gw & 0x8000u) tab[rev & (rsize - 1u)] = INF_ENTRY(off, INF_OP_LINK | sb, root);
            const uint32_t ent = inf_entry(kind, sym, l);
            const uint32_t sl = l - root, srev = rev >> root;
            for (uint32_t j = 0; j < (1u << (sb - sl)); ++j) tab[off + srev + (j << sl)] = ent;
        }
    }
    zmi_wave_sync();
    return 0u;
}

// One speculative token: the literal / length code at the low end of (hi:lo) and, if it is a length, the
// distance code behind it.  rem = input bits left from this token's first bit.  Branch-free apart from the
// second-level lookups: selects cost two VALU instructions, a divergent branch costs exec-mask bookkeeping on
// the scalar unit, which is the busier one.
struct InfTok {
    uint32_t tw;     // bits 0-5 token length in bits; above 63 it stops the chain walk: 64 = end of block, 128 / 256 = error
    uint32_t kind;   // 0 literal, 1 back-reference, 2 end of block
    uint32_t val;    // literal byte | match length
    uint32_t dist;
};
static __device__ __forceinline__ InfTok inf_tok(const InfShared* S, uint32_t lo, uint32_t hi, int32_t rem) {
    uint32_t e = S->ltab[lo & ((1u << INF_LROOT) - 1u)];
    if ((e >> 8) & INF_OP_LINK) {
        uint32_t sb = (e >> 8) & 0x0Fu;
        e = S->ltab[(e >> 16) + ((lo >> INF_LROOT) & ((1u << sb) - 1u))];
    }
    const uint32_t bits = e & 0xFFu, op = (e >> 8) & 0xFFu;
    const bool bad = op == INF_OP_BAD || bits == 0u;                  // "invalid literal/length code"
    const bool is_eob = op == INF_OP_EOB;
    const bool is_len = (op & INF_OP_BASE) != 0u;
    const uint32_t xb = is_len ? (op & 0x0Fu) : 0u;
    const uint32_t used = bits + xb;                                    // <= 20
    InfTok T;
    T.val = (e >> 16) + ((lo >> bits) & ((1u << xb) - 1u));
    // distance code: skipped by a scalar branch when no lane holds a length code (literal-only data)
    uint32_t dlen = 0;
    bool dbad = false;
    T.dist = 0;
    if (__ballot(is_len)) {
        const uint32_t rest = __builtin_amdgcn_alignbit(hi, lo, used);
        uint32_t d = S->dtab[rest & ((1u << INF_DROOT) - 1u)];
        if (is_len && ((d >> 8) & INF_OP_LINK)) {
            uint32_t sb = (d >> 8) & 0x0Fu;
            d = S->dtab[(d >> 16) + ((rest >> INF_DROOT) & ((1u << sb) - 1u))];
        }
        const uint32_t dbits = d & 0xFFu, dop = (d >> 8) & 0xFFu;
        dbad = dop == INF_OP_BAD || dbits == 0u || !(dop & INF_OP_BASE);   // "invalid distance code"
        const uint32_t dxb = dop & 0x0Fu;
        T.dist = (d >> 16) + ((rest >> dbits) & ((1u << dxb) - 1u));
        dlen = dbits + dxb;
    }
    const uint32_t t = is_len ? used + dlen : bits;                     // <= 48
    T.kind = is_len ? 1u : (is_eob ? 2u : 0u);
    // 1: more input needed, 2: invalid data (the reference decides by the bits that are left)
    const uint32_t err_len = (int32_t)used > rem ? 1u : (dbad ? ((rem - (int32_t)used) < 15 ? 1u : 2u) : ((int32_t)t > rem ? 1u : 0u));
    const uint32_t err = bad ? (rem < 15 ? 1u : 2u) : ((int32_t)bits > rem ? 1u : (is_len ? err_len : 0u));
    T.tw = err ? (err << 7) : (t | (is_eob ? 64u : 0u));
    return T;
}
// walk the token chain through one 64-position window: sets the bit of every token start in M, leaves in w the
// word that stopped it (> 63) or the last hop, in pos the position reached (>= 64: ran off the window)
static __device__ __forceinline__ void inf_walk(uint32_t tw, uint32_t& pos, uint64_t& M, uint32_t& w) {
#ifdef ZMI_EMU
    do {
        w = zmi_readlane(tw, pos);
        if (w > 63u) break;
        M |= 1ull << pos;
        pos += w;
    } while (pos < 64u);
#else
    // six scalar instructions and one lane read per token (the compiler's version of the loop above needs twelve
    // and a third branch); `pos` is written by SALU only, so the lane select of v_readlane has no VALU->SGPR hazard
    asm volatile(
        "1:\n\t"
        "v_readlane_b32 %[w], %[tw], %[pos]\n\t"
        "s_cmp_gt_u32 %[w], 63\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_bitset1_b64 %[M], %[pos]\n\t"
        "s_add_u32 %[pos], %[pos], %[w]\n\t"
        "s_cmp_lt_u32 %[pos], 64\n\t"
        "s_cbranch_scc1 1b\n\t"
        "2:"
        : [w] "=&s"(w), [pos] "+s"(pos), [M] "+s"(M)
        : [tw] "v"(tw)
        : "scc");
#endif
}
// every lane on the chain stores its own token: the literal byte, or the back-reference record + bitmap bit
static __device__ __forceinline__ void inf_emit(uint8_t* dst, uint32_t* bm32, bool on, const InfTok& T, uint32_t off) {
    if (on && T.kind == 0u) dst[off] = (uint8_t)T.val;
    if (on && T.kind == 1u) {
        const uint32_t rec = (T.dist - 1u) | ((T.val - 3u) << 15);   // 15 + 8 bits, fits the smallest hole
        dst[off] = (uint8_t)rec;
        dst[off + 1u] = (uint8_t)(rec >> 8);
        dst[off + 2u] = (uint8_t)(rec >> 16);
        atomicOr(&bm32[off >> 5], 1u << (off & 31u));
    }
}

// ---- lane-serial fast path (batch decode only) ----------------------------------------------------------------------
// The token rounds above spend 64 lanes on the ~8-15 tokens that really start inside a 128-bit window: ~230 VALU + ~250
// SALU instructions per round.  A prefix code resynchronises: a decoder started at a wrong bit falls into step with the
// true token sequence after a few dozen bits (the property massively parallel Huffman decoders are built on).  So the
// next 3.75 KiB of the block are cut into 64 sub-sequences of 480 bits and every LANE decodes its own one serially --
// 64 tokens per ~45 instructions instead of ~10 per ~480:
//   1. sync:  lane 0 starts at the true position, lane i at "bit i * 480"; each decodes until it crosses into the next
//             sub-sequence and reports where (its exit).  Lanes whose start is not the exit of the lane below restart
//             there; after a few iterations a prefix 0..m of the lanes is consistent -- lane 0 is right by construction,
//             so the whole prefix is the true token chain.  The same pass counts every lane's output bytes and how far its
//             matches reach back.
//   2. scan:  output offsets; the prefix is cut in front of a lane that saw an invalid code, would pass the output
//             capacity or reaches behind the history -- those cases are left to the token rounds, which report them
//             with the reference's codes and byte counts.
//   3. write: the lanes of the prefix decode once more and store literals / back-reference records where they belong.
// Nothing is committed unless it is certain; whenever the pass cannot commit a single lane the caller runs one token
// round instead.  Takes 4 KiB of input behind the current position; near the end of a stream the single-wave kernel stages what
// is left (zeros behind it) and shortens the sub-sequences so that all 64 lie inside the input (down to 96 bits a lane: the last
// ~900 bytes of a stream go through the token rounds).
struct InfLane {
    uint32_t exit;    // bit position (relative to fb) behind the last token decoded
    uint32_t nout;    // output bytes of those tokens
    uint32_t need;    // bytes of history in front of this lane's first output byte that its matches reach
    uint32_t flags;   // 1 invalid code, 2 end of block (exit = first bit behind it)
};
template <bool WRITE>
static __device__ __forceinline__ InfLane inf_lane_decode(const InfShared* S, const uint8_t* fb, uint32_t start, uint32_t boundary,
                                                          bool active, uint8_t* dst, uint32_t* bm32, uint32_t obase,
                                                          uint32_t nout_total) {
    InfLane R;
    const uint32_t* fw = (const uint32_t*)fb;
    uint32_t pos = start, nout = 0, need = 0, flags = 0;
    bool go = active && pos < boundary;
    // WRITE: a lane's output is one contiguous range [obase, obase + nout_total) (known from the sync walk), written front to
    // back.  Byte stores from 64 lanes are 64 different cache lines per instruction -- the texture path takes them one line per
    // cycle, and a match was three of them plus an atomic: the write walks kept that path busy for half of the kernel's time.
    // So the bytes are collected in a 64-bit accumulator over the aligned dword the lane stands in (and the one behind it: a
    // 3-byte record may straddle), a dword is stored when the lane leaves it -- the part of it that lies in a hole is
    // whatever the accumulator holds, zeros: the resolve pass overwrites every hole -- and the bitmap bits are collected
    // per 32-bit word.  Only the two dwords at the ends of the range are shared with the neighbours: their bytes are stored
    // one by one behind the loop.
    const uint32_t A = WRITE ? (uint32_t)((uintptr_t)dst & 3u) : 0u;
    uint32_t* const dw = (uint32_t*)(dst - A);
    const uint32_t o_first = obase + A, o_end = o_first + (active ? nout_total : 0u);   // (in bytes from dw)
    uint32_t o = o_first, first_acc = 0, bmw = 0, bmacc = 0;
    uint64_t acc = 0;
    // One token per lane and iteration; the loop is wave-uniform (lanes that are done ride along with their state frozen) and
    // instruction-issue bound (17 streams per CU: the LDS round trips hide behind the other waves), so it is written for
    // instruction count: every quantity comes straight out of the entry's fields (the low nibble of `op` is the number of
    // extra bits for a length / distance code and ZERO for a literal, an end of block and an invalid code; holes of the
    // tables are INF_OP_BAD entries, inf_build), nothing is gated by `go` except the four state updates at the end, the
    // second-level reads and the distance code sit behind wave-uniform branches.
    if (__ballot(go)) do {
        const uint32_t wi = pos >> 5;
        const uint32_t d0 = fw[wi], d1 = fw[wi + 1u], d2 = fw[wi + 2u];
        const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, pos & 31u);
        uint32_t e = S->ltab[lo & ((1u << INF_LROOT) - 1u)];
        if (__ballot((e & (INF_OP_LINK << 8)) != 0u)) {
            const bool link = (e & (INF_OP_LINK << 8)) != 0u;
            const uint32_t e2 = S->ltab[link ? (e >> 16) + __builtin_amdgcn_ubfe(lo, INF_LROOT, (e >> 8) & 0x0Fu) : 0u];
            e = link ? e2 : e;
        }
        const uint32_t bits = e & 0xFFu, xb = (e >> 8) & 0x0Fu;
        const uint32_t val = (e >> 16) + __builtin_amdgcn_ubfe(lo, bits, xb);   // literal byte | match length
        const uint32_t used = bits + xb;                                        // <= 20
        const bool is_len = (e & (INF_OP_BASE << 8)) != 0u, is_eob = (e & (INF_OP_EOB << 8)) != 0u;
        bool err = (e & (INF_OP_BAD << 8)) != 0u;                               // "invalid literal/length code"
        uint32_t dist = 0, adv = used;
        // ONE second table read serves both kinds of lane: the distance code behind a length, the next literal behind a literal (the
        // distance table stands behind the literal / length table in LDS, so it is one gather with a per-lane index; both start at
        // the bit behind the token's `used` bits).  As two reads behind two wave-uniform branches they were two LDS round trips in
        // every iteration that has both kinds of lane -- most iterations of anything but pure literals.
        const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, pos & 31u);
        const uint32_t rest = __builtin_amdgcn_alignbit(hi, lo, used);
        static_assert(offsetof(InfShared, dtab) == offsetof(InfShared, ltab) + 4u * INF_LSIZE, "dtab directly behind ltab");
        const uint32_t x = S->ltab[is_len ? INF_LSIZE + (rest & ((1u << INF_DROOT) - 1u)) : (rest & ((1u << INF_LROOT) - 1u))];
        if (__ballot(is_len)) {
            uint32_t d = x;
            const bool dlink = is_len && (d & (INF_OP_LINK << 8)) != 0u;
            if (__ballot(dlink)) {
                const uint32_t d2x = S->dtab[dlink ? (d >> 16) + __builtin_amdgcn_ubfe(rest, INF_DROOT, (d >> 8) & 0x0Fu) : 0u];
                d = dlink ? d2x : d;
            }
            const uint32_t dbits = d & 0xFFu, dxb = (d >> 8) & 0x0Fu;
            const uint32_t dv = (d >> 16) + __builtin_amdgcn_ubfe(rest, dbits, dxb);
            err = err || (is_len && (d & (INF_OP_BASE << 8)) == 0u);            // "invalid distance code" (a hole or a non-distance entry)
            dist = is_len ? dv : 0u;
            adv += is_len ? dbits + dxb : 0u;
        }
        const bool live = go && !err;                  // the token counts
        const bool mat = live && is_len, lit = live && !is_len && !is_eob;
        // A literal takes the literals behind it along while they are first-level entries too and still start inside the lane's
        // sub-sequence (a token that starts behind the boundary belongs to the next lane: the exits must not depend on this):
        // one more table read and ~10 instructions each against a whole iteration -- literal-dense streams (half of the
        // benchmark mix's decode time) need a third fewer iterations (round 4).  Up to four, out of the 64-bit window (the codes
        // of first-level entries are <= 9 bits).
        uint32_t lit2 = 0, n_lit = 1u;
        if (__ballot(lit)) {
            const bool two = lit && (x & 0xFF00u) == 0u && pos + bits < boundary;   // op byte 0: a plain literal (holes are INF_OP_BAD)
            lit2 = two ? (x >> 16) << 8 : 0u;
            adv += two ? (x & 0xFFu) : 0u;
            n_lit = two ? 2u : 1u;
            if (__ballot(two)) {
                const uint32_t e3 = S->ltab[__builtin_amdgcn_alignbit(hi, lo, adv) & ((1u << INF_LROOT) - 1u)];   // (adv <= 24)
                const bool three = two && (e3 & 0xFF00u) == 0u && pos + adv < boundary;
                lit2 |= three ? (e3 >> 16) << 16 : 0u;
                adv += three ? (e3 & 0xFFu) : 0u;
                n_lit = three ? 3u : n_lit;
                if (__ballot(three)) {
                    const uint32_t e4 = S->ltab[__builtin_amdgcn_alignbit(hi, lo, adv & 31u) & ((1u << INF_LROOT) - 1u)];
                    const bool four = three && adv < 32u && (e4 & 0xFF00u) == 0u && pos + adv < boundary;
                    lit2 |= four ? (e4 >> 16) << 24 : 0u;
                    adv += four ? (e4 & 0xFFu) : 0u;
                    n_lit = four ? 4u : n_lit;
                }
            }
        }
        if (WRITE) {
            const uint32_t rec = (dist - 1u) | ((val - 3u) << 15);         // the record the resolve pass reads (inf_emit)
            const uint32_t n = mat ? val : (lit ? n_lit : 0u);
            const uint32_t v = mat ? rec : (lit ? (val | lit2) : 0u);
            acc |= (uint64_t)v << ((o & 3u) * 8u);
            const uint32_t no = o + n, w = o >> 2, cross = (no >> 2) - w;
            if (mat) {
                const uint32_t off = o - A, bi = off >> 5;
                if (bi != bmw && bmacc != 0u) { atomicOr(&bm32[bmw], bmacc); bmacc = 0u; }
                bmw = bi;
                bmacc |= 1u << (off & 31u);
            }
            if (cross != 0u) {
                const uint32_t lo32 = (uint32_t)acc, hi32 = (uint32_t)(acc >> 32);
                if (w == (o_first >> 2) && (o_first & 3u) != 0u) first_acc = lo32; else dw[w] = lo32;
                if (cross > 1u && hi32 != 0u) dw[w + 1u] = hi32;           // (the tail of a record; dwords that are all hole stay unwritten)
                acc = cross > 1u ? 0ull : (uint64_t)hi32;
            }
            o = no;
        }
        const uint32_t reach = dist > nout ? dist - nout : 0u;             // history in front of this lane's output
        need = (mat && reach > need) ? reach : need;
        nout += mat ? val : (lit ? n_lit : 0u);
        pos += live ? adv : 0u;
        flags |= go ? (err ? 1u : (is_eob ? 2u : 0u)) : 0u;
        go = live && !is_eob && pos < boundary;
    } while (__ballot(go));
    if (WRITE) {
        if (bmacc != 0u) atomicOr(&bm32[bmw], bmacc);
        uint8_t* const db = dst - A;
        const bool first_pending = (o_first & 3u) != 0u && (o_first >> 2) < (o_end >> 2);
#pragma unroll
        for (uint32_t j = 1; j < 4u; ++j) {
            const uint32_t q = (o_first & ~3u) + j;
            if (first_pending && q >= o_first) db[q] = (uint8_t)(first_acc >> (8u * j));
        }
#pragma unroll
        for (uint32_t j = 0; j < 3u; ++j) {
            const uint32_t q = (o_end & ~3u) + j;
            if (q >= o_first && q < o_end) db[q] = (uint8_t)((uint32_t)acc >> (8u * j));
        }
    }
    R.nout = nout; R.need = need; R.flags = flags;
    R.exit = pos;
    return R;
}


// ---- blocks coded with the FIXED Huffman code (BTYPE 01; inflate/inffixed_tbl.rs:7): where do the lanes of a fast pass start?
// The warm-up walk of inf_fast_pass lives on the self-synchronisation of a prefix code, and the fixed code over literal-dense
// data has next to none: its literals are 8 (or 9) bits long, so a walk started off the true token chain reads 8-bit codes of
// a shifted alphabet and keeps its phase error for thousands of bits (measured on Z_FIXED streams of the benchmark classes:
// median > 20 000 bits; round 3: few lanes committed per pass, 6-8x the time of the same data with dynamic codes -- and the
// reference's own level 1, deflate/algorithm/quick.rs:12-158, emits nothing but such blocks).  But the fixed code needs no
// tables to be DELIMITED: a token's length follows from its first bits by a handful of compares.  So every lane walks its
// sub-sequence from every bit offset at which the chain could enter it, with a table-free walker that only counts bits, and
// notes (a) which of its walks (tracks) passes through each of the first 32 bit offsets and (b) where each track leaves the
// sub-sequence.  The true chain is then a walk over the lanes with two small lookups per lane -- "the token chain enters
// lane k at offset o: track map_k[o] is on it, and leaves at exit_k[track]" -- after which every lane knows its true first
// token and the pass continues as for a dynamic block, with nothing to fix.
// Cost: 8-10 bit-counting walks (~12 instructions a token against ~62 of the table walk) instead of a warm-up and the restart rounds.
struct InfFixedStep { uint32_t adv; uint32_t stop; };   // bits of the token at the low end of `lo` (>= 32 significant bits); stop: 1 invalid, 2 end of block
static __device__ __forceinline__ InfFixedStep inf_fixed_step(uint32_t lo, uint32_t hi) {
    InfFixedStep R;
    const uint32_t r = __brev(lo);                 // the code's bits, first bit on top
    const uint32_t c7 = r >> 25, c8 = r >> 24;
    // 7 bits 0000000 .. 0010111: symbols 256 .. 279; 8 bits 00110000 .. 10111111: literals 0 .. 143; 11000000 .. 11000111: 280 .. 287;
    // 9 bits 110010000 .. 111111111: literals 144 .. 255
    const bool s7 = c7 < 24u, l8 = c8 >= 0x30u && c8 < 0xC0u, s8 = c8 >= 0xC0u && c8 < 0xC8u;
    const uint32_t bits = s7 ? 7u : ((l8 || s8) ? 8u : 9u);
    const uint32_t sym = s7 ? 256u + c7 : (s8 ? 280u + (c8 - 0xC0u) : 0u);   // 0: a literal
    R.stop = sym == 256u ? 2u : (sym >= 286u ? 1u : 0u);
    uint32_t adv = bits;
    if (sym > 256u && sym < 286u) {
        const uint32_t idx = sym - 257u;
        const uint32_t xb = (idx < 8u || idx == 28u) ? 0u : (idx >> 2) - 1u;
        adv += xb;
        // the distance code: 5 bits, first bit on top, then its extra bits
        const uint64_t w = ((uint64_t)hi << 32) | lo;
        const uint32_t d = __brev((uint32_t)(w >> adv)) >> 27;
        if (d >= 30u) R.stop = 1u;
        adv += 5u + (d < 4u ? 0u : (d >> 1) - 1u);
    }
    R.adv = adv;
    return R;
}
// all lanes: the true first token of every lane's sub-sequence [p_rel + lane * sub, p_rel + (lane + 1) * sub) of a fixed-code
// block staged in fb, given that a token starts at p_rel.  Returns the number of leading lanes whose start is known (the
// chain ends early where the walkers met an invalid code / an end of block, or where a lane has more than 15 distinct tracks:
// the lanes behind it start at a guess, as in a dynamic block); *start receives the lane's start.
// TRACKS: a token of the fixed code has at most 31 bits, so the chain enters a sub-sequence at one of its first 31 bit offsets.
// A lane walks from every offset 0 .. 31 that no earlier walk of its own has passed through (a walk that steps on a visited
// offset has merged into that track: same exit) -- in the 8-bit regime of literal-dense data that is eight walks (offsets
// j, j + 8, j + 16, j + 24 are one track), around matches a few more that merge within a token or two.
// (four named registers and selects: an array indexed with a lane-varying offset would be placed in scratch memory)
struct InfQuad { uint32_t a, b, c, d; };
static __device__ __forceinline__ uint32_t quad_pick(const InfQuad& q, uint32_t r) { return r == 0u ? q.a : (r == 1u ? q.b : (r == 2u ? q.c : q.d)); }
static __device__ __forceinline__ void quad_or(InfQuad& q, uint32_t r, uint32_t v) {
    q.a |= r == 0u ? v : 0u; q.b |= r == 1u ? v : 0u; q.c |= r == 2u ? v : 0u; q.d |= r == 3u ? v : 0u;
}
struct InfFixedMaps {
    InfQuad map;   // 32 offsets x 4 bits: the track that passes through the offset (1 .. 15; 0: none)
    InfQuad ex;    // 16 tracks x 8 bits: the offset behind the boundary at which the track leaves (0xFF: it stopped inside)
};
static __device__ __forceinline__ InfFixedMaps inf_fixed_tracks(const uint8_t* fb, uint32_t p_rel, uint32_t sub, bool active) {
    const uint32_t lane = zmi_lane();
    const uint32_t* fw = (const uint32_t*)fb;
    const uint32_t nominal = p_rel + lane * sub, boundary = nominal + sub;
    InfFixedMaps F;
    F.map.a = F.map.b = F.map.c = F.map.d = 0u;
    F.ex.a = F.ex.b = F.ex.c = F.ex.d = 0u;
    bool busy = active;
    for (uint32_t t = 1; t < 16u; ++t) {
        // the lowest offset that no track has passed through yet (a zero nibble of the map)
        uint32_t o = 32u;
        if (busy) {
#pragma unroll
            for (int r = 3; r >= 0; --r) {
                const uint32_t x = r == 3 ? F.map.d : (r == 2 ? F.map.c : (r == 1 ? F.map.b : F.map.a));
                const uint32_t z = ~(x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x11111111u;   // bit 4 n set: nibble n is zero
                o = z ? 8u * (uint32_t)r + ((uint32_t)(__ffs((int)z) - 1) >> 2) : o;
            }
            // (offset 31 is never entered: the longest token has 31 bits, a token that started inside the lane below ends at offset 30 at the latest)
            busy = o < 31u;
        }
        if (__ballot(busy) == 0ull) break;
        if (busy) {
            uint32_t pos = nominal + o, code = 0xFFu;
            bool open = true;
            // the first 32 bits: token by token, every offset passed is claimed for this track (or found claimed: merged)
            while (pos - nominal < 32u) {
                const uint32_t oo = pos - nominal;
                const uint32_t u = (quad_pick(F.map, oo >> 3) >> (4u * (oo & 7u))) & 15u;
                if (u != 0u) { code = (quad_pick(F.ex, u >> 2) >> (8u * (u & 3u))) & 0xFFu; open = false; break; }   // merged into track u: its exit
                quad_or(F.map, oo >> 3, t << (4u * (oo & 7u)));
                if (pos >= boundary) { code = pos - boundary; open = false; break; }
                const uint32_t wi = pos >> 5, sh = pos & 31u;
                const uint32_t d0 = fw[wi], d1 = fw[wi + 1u], d2 = fw[wi + 2u];
                const InfFixedStep T = inf_fixed_step(__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh));
                if (T.stop != 0u) { open = false; break; }
                pos += T.adv;
            }
            // the rest of the sub-sequence: nothing to note but where the walk leaves it.  Literals -- the bulk of the tokens where
            // this code is used -- are taken up to three from one 32-bit window (8 or 9 bits each, told apart by one compare)
            while (open) {
                if (pos >= boundary) { code = pos - boundary; break; }
                const uint32_t wi = pos >> 5, sh = pos & 31u;
                const uint32_t d0 = fw[wi], d1 = fw[wi + 1u];
                const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh);
                const uint32_t r = __brev(lo);
                uint32_t c8 = r >> 24;
                if (c8 >= 0x30u && (c8 < 0xC0u || c8 >= 0xC8u)) {
                    uint32_t used = c8 >= 0xC8u ? 9u : 8u;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        c8 = (r << used) >> 24;
                        const bool lit = c8 >= 0x30u && (c8 < 0xC0u || c8 >= 0xC8u);
                        if (!lit || pos + used >= boundary) break;
                        used += c8 >= 0xC8u ? 9u : 8u;
                    }
                    pos += used;
                    continue;
                }
                const uint32_t d2 = fw[wi + 2u];
                const InfFixedStep T = inf_fixed_step(lo, __builtin_amdgcn_alignbit(d2, d1, sh));
                if (T.stop != 0u) break;
                pos += T.adv;
            }
            quad_or(F.ex, t >> 2, code << (8u * (t & 3u)));
        }
    }
    return F;
}
// the chain over the 64 lanes of a wave, entered at bit offset `off` of lane 0 (wave-uniform): *mine = the offset at which it
// enters this lane, *out = the offset at which it leaves lane 63 (0xFF: it ended inside the wave); returns the number of leading
// lanes whose entry is known
static __device__ __forceinline__ uint32_t inf_fixed_chain(const InfFixedMaps& F, uint32_t off, uint32_t* mine_out, uint32_t* out) {
    const uint32_t lane = zmi_lane();
    uint32_t known = 0u, mine = 0u;
    *out = 0xFFu;
    for (uint32_t k = 0; k < 64u; ++k) {
        if (lane == k) mine = off;
        known = k + 1u;
        if (off >= 31u) break;
        const uint32_t mw = off < 8u ? zmi_readlane(F.map.a, k) : (off < 16u ? zmi_readlane(F.map.b, k) : (off < 24u ? zmi_readlane(F.map.c, k) : zmi_readlane(F.map.d, k)));
        const uint32_t u = (mw >> (4u * (off & 7u))) & 15u;
        if (u == 0u) break;                                   // (more than 15 tracks in lane k: the lanes behind it start at a guess)
        const uint32_t ew = u < 4u ? zmi_readlane(F.ex.a, k) : (u < 8u ? zmi_readlane(F.ex.b, k) : (u < 12u ? zmi_readlane(F.ex.c, k) : zmi_readlane(F.ex.d, k)));
        const uint32_t e = (ew >> (8u * (u & 3u))) & 0xFFu;
        if (e == 0xFFu) break;                                // an end of block / invalid code inside lane k: the chain ends there
        off = e;
        if (k == 63u) *out = e;
    }
    *mine_out = mine;
    return known;
}
static __device__ __forceinline__ uint32_t inf_fixed_sync(const uint8_t* fb, uint32_t p_rel, uint32_t sub, uint32_t* start) {
    const InfFixedMaps F = inf_fixed_tracks(fb, p_rel, sub, true);
    uint32_t mine = 0u, out = 0u;
    const uint32_t known = inf_fixed_chain(F, 0u, &mine, &out);
    *start = p_rel + zmi_lane() * sub + mine;
    return known;
}

// the staged input of a pass near the end of the stream: the `lim` (< INF_FAST_BYTES) bytes that exist, zeros behind them.  Out of
// line and rolled: inlined into the pass, its guarded loads took the decode kernel from 74 to 140 VGPRs (3 waves per SIMD instead of
// 6: 43 -> 50 ms per 16 Ki streams); it runs once or twice per stream.
static __device__ __noinline__ void inf_stage_tail(const uint8_t* line, uint32_t lim) {
    uint8_t* fb = g_inf_lds.fb;
    const uint32_t lane = zmi_lane();
#pragma unroll 1
    for (uint32_t o = 4u * lane; o < INF_FAST_BYTES; o += 256u) {
        uint32_t v = 0u;
        if (o + 4u <= lim) v = *(const uint32_t*)(line + o);   // (`line` is 16-byte aligned)
        else
            for (uint32_t j = 0; j < 4u; ++j)
                if (o + j < lim) v |= (uint32_t)line[o + j] << (8u * j);
        *(uint32_t*)(fb + o) = v;
    }
}

// One fast pass from bit P of the stream.  Returns the number of lanes committed (0: nothing done); *bits_used / *out_made
// / *hit_eob describe what was committed.  All lanes call.
static __device__ __forceinline__ uint32_t inf_fast_pass(InfShared* S, const uint8_t* src, uint64_t P, uint8_t* dst,
                                                      uint32_t* bm32, uint32_t opos, uint32_t cap, uint32_t hist, uint32_t sub,
                                                      uint32_t* bits_used, uint32_t* out_made, uint32_t* hit_eob, bool fixed_code, uint32_t n_in) {
    // sub: bits per lane, 96 .. INF_SUB_BITS (wave-uniform; see the caller for how it is chosen)
    const uint32_t lane = zmi_lane();
    // stage the input: from the 16-byte line holding bit P, 4 KiB, four coalesced loads
    const uint32_t ib = (uint32_t)(P >> 3);
    const uint32_t mis = (uint32_t)((uintptr_t)(src + ib) & 15u);
    const uint8_t* line = src + ib - mis;
    zmi_wave_order();
    const uint32_t lim = n_in - (ib - mis);   // bytes of the stream from `line` on; what lies behind them is staged as zeros (a pass
                                              // near the end of the stream: its lanes' sub-sequences end in front of that point)
    if (lim >= INF_FAST_BYTES) {
#pragma unroll
        for (uint32_t k = 0; k < INF_FAST_BYTES / 1024u; ++k) *(uint4*)(S->fb + 16u * (lane + 64u * k)) = *(const uint4*)(line + 16u * (lane + 64u * k));
    } else inf_stage_tail(line, lim);
    zmi_wave_order();
    const uint32_t p_rel = (mis << 3) | ((uint32_t)P & 7u);
    const uint32_t boundary = p_rel + (lane + 1u) * sub;
    // 1. sync.  A lane's first guess is not "my sub-sequence starts with a token" but the result of a short warm-up walk
    // through the tail of the sub-sequence below: after INF_WARM_BITS most walks are in step already, so the first full
    // walk is usually the right one and the restart round below has few (often no) lanes to fix.
    uint32_t start = p_rel + lane * sub;
    if (fixed_code) {
        // the fixed code does not re-synchronise, but it can be delimited without tables: the lanes' true starts (inf_fixed_sync)
        uint32_t fs = start;
        const uint32_t known = inf_fixed_sync(S->fb, p_rel, sub, &fs);
        if (lane < known) start = fs;
    } else {
        // (lane 0 rides along switched off: it must still hold a position inside the staged bytes, its reads happen; with
        // short sub-sequences the lowest lanes warm up from the pass's own first bit, which is a true token start)
        const uint32_t wfrom = start - p_rel > INF_WARM_BITS ? start - INF_WARM_BITS : p_rel;
        const InfLane Wm = inf_lane_decode<false>(S, S->fb, wfrom, start, lane != 0u, nullptr, nullptr, 0u, 0u);
        if (lane != 0u && Wm.flags == 0u) start = Wm.exit;   // (a warm-up that ran into an invalid code or an end of block: keep the guess)
    }
    InfLane R = inf_lane_decode<false>(S, S->fb, start, boundary, true, nullptr, nullptr, 0u, 0u);
    uint32_t good = 1u;   // lanes 0 .. good-1 are known to sit on the true token chain
    for (uint32_t it = 0;; ++it) {
        // the lane below tells where this lane has to start
        const uint32_t below_exit = (uint32_t)__shfl_up((int)R.exit, 1u);
        const bool wrong = lane != 0u && below_exit != start;
        const uint64_t stopm = __ballot(R.flags != 0u);      // lanes that ended their walk early (end of block / invalid code)
        const uint64_t wrongm = __ballot(wrong);
        // Lane 0 is right by construction, so every lane below the first wrong one is on the true chain -- up to and
        // including the first of them that stopped: behind an end of block (or an invalid code) the chain does not go on
        const uint32_t first_wrong = wrongm ? (uint32_t)__ffsll((unsigned long long)wrongm) - 1u : 64u;
        const uint32_t first_stop = stopm ? (uint32_t)__ffsll((unsigned long long)stopm) : 64u;   // index + 1
        good = first_wrong < first_stop ? first_wrong : first_stop;
        if (first_stop <= first_wrong || first_wrong >= 64u || it == 5u) break;
        // restart the wrong lanes where their neighbours ended (most fall into step inside their own sub-sequence, so
        // the next check usually finds everything consistent)
        if (wrong) start = below_exit;
        const InfLane N = inf_lane_decode<false>(S, S->fb, start, boundary, wrong, nullptr, nullptr, 0u, 0u);
        if (wrong) R = N;
    }
    // 2. scan: offsets, and what can be committed
    const bool in = lane < good;
    const uint32_t nout = in ? R.nout : 0u;
    const uint32_t incl = zmi_wave_incl_scan(nout);
    const uint32_t base = opos + incl - nout;
    const bool bad = in && ((R.flags & 1u) != 0u || base + nout > cap || R.need > base + hist);
    const uint64_t badm = __ballot(bad);
    uint32_t commit = good;
    if (badm) { const uint32_t fb1 = (uint32_t)__ffsll((unsigned long long)badm) - 1u; commit = fb1 < commit ? fb1 : commit; }
    if (commit == 0u) return 0u;
    // 3. write
    (void)inf_lane_decode<true>(S, S->fb, start, boundary, lane < commit, dst, bm32, base, R.nout);
    const uint32_t last = commit - 1u;
    *bits_used = zmi_readlane(R.exit, last) - p_rel;
    *out_made = zmi_readlane(incl, last);
    *hit_eob = (zmi_readlane(R.flags, last) >> 1) & 1u;
    return commit;
}


// ---- one stream on several waves (the NW = INF_MW instantiations: launches of a few streams) --------------------------
// A stream alone on the chip is one wave alone on a CU: every LDS round trip of its token steps is exposed, a 1 MiB text
// stream took 9.5 ms to decode (round 2).  The fast pass above is already a set of independent sub-sequence walks that
// are stitched together afterwards, so it extends across waves as it stands: wave w of the workgroup takes the 64
// sub-sequences behind those of wave w - 1 (its own 4 KiB of staged input, the block's tables shared), the waves walk
// concurrently, and the stitch runs over 64 x nact lanes -- inside a wave as before, between waves through LDS: the lane 0
// of wave w has to start where lane 63 of wave w - 1 ended, and a wave commits only if every wave below it committed all
// its lanes.  Wave 0 is the stream's master (headers, table construction, token rounds, everything the single-wave kernel
// does); the other waves sleep at a workgroup barrier until it posts a pass.  A pass covers at most the rest of the block
// (the tables change behind an end-of-block), so the gain is bounded by the block size: 16 waves x up to 3.75 KiB of
// compressed data, more than the ~20 KiB of a 16 383-symbol zlib block.
#define INF_MW 16u
#define INF_MW_SUB 192u   // a pass is spread over as many waves as give every lane about this many bits (below that the 192-bit warm-up dominates)
#define INF_MW_ROUNDS 4u   // cross-wave fix-up rounds (a wave whose lane 0 started at a wrong guess restarts it and re-stitches)
struct InfMulti {
    __attribute__((aligned(16))) uint8_t fb[INF_MW - 1u][INF_FAST_BYTES];
    uint32_t cmd;            // 1: a pass is posted, 2: the stream is done (helpers leave)
    uint32_t nact, sub;      // waves taking part, bits per lane
    uint32_t Plo, Phi;       // bit position of the pass in the stream
    uint32_t opos, cap, hist;
    uint32_t start0[INF_MW];   // per wave, bits relative to P: where its lane 0 started ...
    uint32_t exit63[INF_MW];   // ... where its lane 63 ended,
    uint32_t good[INF_MW];     // how many of its leading lanes are consistent with its lane 0,
    uint32_t stopped[INF_MW];  // whether one of them hit an end of block / an invalid code,
    uint32_t nsum[INF_MW];     // output bytes of the lanes it would commit,
    uint32_t cut[INF_MW];      // first of those lanes that must not be committed (64: none)
    uint32_t res_lanes, res_bits, res_out, res_eob;
    uint32_t fixed;            // the block is coded with the fixed code: the lanes' starts come from inf_fixed_tracks / inf_fixed_chain
    uint32_t fx_off[INF_MW + 1u];   // ... the offset at which the token chain enters wave w (0xFF: not known)
};
__shared__ InfMulti g_inf_mw;

// one pass, executed by ALL waves of the workgroup between the barrier that posts it and the one that ends it
static __device__ __noinline__ void inf_pass_mw(const uint8_t* src, uint8_t* dst, uint32_t* bm32) {
    InfShared* S = &g_inf_lds;
    InfMulti* M = &g_inf_mw;
    const uint32_t lane = zmi_lane(), wave = zmi_uniform(zmi_wave());
    const uint32_t nact = zmi_uniform(M->nact), sub = zmi_uniform(M->sub);
    const uint64_t P = ((uint64_t)zmi_uniform(M->Phi) << 32) | zmi_uniform(M->Plo);
    const uint32_t opos = zmi_uniform(M->opos), cap = zmi_uniform(M->cap), hist = zmi_uniform(M->hist);
    const bool act = wave < nact;
    uint8_t* fb = wave == 0u ? S->fb : M->fb[wave - 1u];
    const uint32_t span = 64u * sub;                 // bits per wave
    const uint32_t org = wave * span;                // this wave's first bit, relative to P
    uint32_t p_rel = 0, start = 0, boundary = 0, good = 0;
    uint64_t stopm = 0;
    InfLane R;
    R.exit = 0; R.nout = 0; R.need = 0; R.flags = 0;
    if (act) {
        // stage 4 KiB: wave 0 from the 16-byte line holding bit P, the others 32 bytes earlier -- their lane 0 is a guess
        // like any other lane's and warms up through the bits below the wave's first one
        const uint64_t Pw = P + org;
        const uint32_t back = wave ? 32u : 0u;
        const uint32_t ib = (uint32_t)(Pw >> 3) - back;
        const uint32_t mis = (uint32_t)((uintptr_t)(src + ib) & 15u);
        const uint8_t* line = src + ib - mis;
        zmi_wave_order();
#pragma unroll
        for (uint32_t k = 0; k < INF_FAST_BYTES / 1024u; ++k) *(uint4*)(fb + 16u * (lane + 64u * k)) = *(const uint4*)(line + 16u * (lane + 64u * k));
        zmi_wave_order();
        p_rel = ((mis + back) << 3) | ((uint32_t)Pw & 7u);
        boundary = p_rel + (lane + 1u) * sub;
        start = p_rel + lane * sub;
    }
    const bool fixed_code = zmi_uniform(M->fixed) != 0u;
    if (fixed_code) {
        // fixed code: every wave finds its tracks at once; the chain then runs through the waves in order (a wave's turn is a few
        // hundred scalar reads: the walks above it are what costs), each wave handing the next the offset it enters at
        InfFixedMaps F = inf_fixed_tracks(fb, p_rel, sub, act);
        if (wave == 0u && lane == 0u) M->fx_off[0] = 0u;
        for (uint32_t w = 0; w < nact; ++w) {
            __syncthreads();
            if (wave == w) {
                const uint32_t entry = zmi_uniform(M->fx_off[w]);
                uint32_t mine = 0u, out = 0xFFu, known = 0u;
                if (entry != 0xFFu) known = inf_fixed_chain(F, entry, &mine, &out);
                if (lane < known) start += mine;
                if (lane == 0u) M->fx_off[w + 1u] = known == 64u ? out : 0xFFu;
            }
        }
        __syncthreads();
        if (act) R = inf_lane_decode<false>(S, fb, start, boundary, true, nullptr, nullptr, 0u, 0u);
    } else if (act) {
        {
            const bool warm = lane != 0u || wave != 0u;
            const uint32_t wfrom = (wave != 0u || start - p_rel > INF_WARM_BITS) ? start - INF_WARM_BITS : p_rel;
            const InfLane Wm = inf_lane_decode<false>(S, fb, wfrom, start, warm, nullptr, nullptr, 0u, 0u);
            if (warm && Wm.flags == 0u) start = Wm.exit;
        }
        R = inf_lane_decode<false>(S, fb, start, boundary, true, nullptr, nullptr, 0u, 0u);
    }
    bool l0_redo = false;   // this wave's lane 0 has been moved to where the wave below ended and has to walk again
    for (uint32_t g = 0;; ++g) {
        if (act) {
            // inside the wave, as inf_fast_pass: a lane restarts where the lane below ended until a prefix is consistent
            // (lane 0's restart, decided between the waves below, rides in the same walk as the lanes it may upset)
            for (uint32_t it = 0;; ++it) {
                const uint32_t below_exit = (uint32_t)__shfl_up((int)R.exit, 1u);
                const bool wrong = lane != 0u ? below_exit != start : l0_redo;
                stopm = l0_redo ? 0ull : __ballot(R.flags != 0u);   // (lane 0's old walk says nothing any more)
                const uint64_t wrongm = __ballot(wrong);
                const uint32_t first_wrong = wrongm ? (uint32_t)__ffsll((unsigned long long)wrongm) - 1u : 64u;
                const uint32_t first_stop = stopm ? (uint32_t)__ffsll((unsigned long long)stopm) : 64u;   // index + 1
                good = first_wrong < first_stop ? first_wrong : first_stop;
                if (first_stop <= first_wrong || first_wrong >= 64u || it == 5u) break;
                if (wrong && lane != 0u) start = below_exit;
                const InfLane N = inf_lane_decode<false>(S, fb, start, boundary, wrong, nullptr, nullptr, 0u, 0u);
                if (wrong) R = N;
                l0_redo = false;
            }
            const uint32_t s0 = zmi_readlane(start, 0u), e63 = zmi_readlane(R.exit, 63u);
            if (lane == 0u) {
                M->start0[wave] = s0 - p_rel + org;
                M->exit63[wave] = e63 - p_rel + org;
                M->good[wave] = good;
                M->stopped[wave] = (stopm & (good >= 64u ? ~0ull : ((1ull << good) - 1ull))) != 0ull ? 1u : 0u;
            }
        }
        __syncthreads();
        if (g + 1u == INF_MW_ROUNDS) break;
        // between the waves: lane 0 of wave w belongs where lane 63 of wave w - 1 ended
        if (act && wave != 0u) {
            const uint32_t want = zmi_uniform(M->exit63[wave - 1u]), have = zmi_uniform(M->start0[wave]);
            // (a wave below that stopped early -- end of block, invalid code -- ended in front of this wave's bits: nothing to link to)
            if (want != have && want >= org && want < org + 64u) {
                if (lane == 0u) start = want - org + p_rel;
                l0_redo = true;
            }
        }
        __syncthreads();   // (this round's values have been read before the next round overwrites them)
    }
    // what every wave commits: the chain runs through the waves as long as each one is complete and linked to the one below
    uint32_t commit_of[INF_MW];
    {
        bool alive = true;
#pragma unroll
        for (uint32_t w = 0; w < INF_MW; ++w) {
            commit_of[w] = 0u;
            if (w < nact) {
                const bool linked = w == 0u || zmi_uniform(M->start0[w]) == zmi_uniform(M->exit63[w - 1u]);
                const uint32_t gw = zmi_uniform(M->good[w]);
                commit_of[w] = (alive && linked) ? gw : 0u;
                alive = alive && linked && gw == 64u && zmi_uniform(M->stopped[w]) == 0u;
            }
        }
    }
    uint32_t commit = 0;
#pragma unroll
    for (uint32_t w = 0; w < INF_MW; ++w) commit = w == wave ? commit_of[w] : commit;
    // output offsets, and the first lane that cannot be committed (invalid code, output full, reaches behind the history:
    // those are left to the token rounds, which report them with the reference's codes)
    const uint32_t nout = lane < commit ? R.nout : 0u;
    const uint32_t incl = zmi_wave_incl_scan(nout);
    if (lane == 63u) M->nsum[wave] = incl;
    __syncthreads();
    uint32_t wbase = opos;
#pragma unroll
    for (uint32_t w = 0; w < INF_MW; ++w) wbase += (w < wave && w < nact) ? zmi_uniform(M->nsum[w]) : 0u;
    const uint32_t base = wbase + incl - nout;
    const bool bad = lane < commit && ((R.flags & 1u) != 0u || base + nout > cap || R.need > base + hist);
    const uint64_t badm = __ballot(bad);
    if (lane == 0u) M->cut[wave] = badm ? (uint32_t)__ffsll((unsigned long long)badm) - 1u : 64u;
    __syncthreads();
    uint32_t fc = 0, total_lanes = 0, last_wave = 0;
    {
        bool open = true;
#pragma unroll
        for (uint32_t w = 0; w < INF_MW; ++w) {
            uint32_t c = 0u;
            if (w < nact && open) {
                const uint32_t cu = zmi_uniform(M->cut[w]);
                c = cu < commit_of[w] ? cu : commit_of[w];
                if (c < 64u || cu < 64u) open = false;   // a wave that does not commit all 64 lanes ends the pass
            }
            if (c != 0u) last_wave = w;
            total_lanes += c;
            fc = w == wave ? c : fc;
        }
    }
    if (fc != 0u) (void)inf_lane_decode<true>(S, fb, start, boundary, lane < fc, dst, bm32, base, R.nout);
    if (total_lanes != 0u && wave == last_wave) {
        const uint32_t k = fc - 1u;
        const uint32_t bits = zmi_readlane(R.exit, k) - p_rel + org, outm = wbase - opos + zmi_readlane(incl, k),
                       eobf = (zmi_readlane(R.flags, k) >> 1) & 1u;
        if (lane == 0u) { M->res_bits = bits; M->res_out = outm; M->res_eob = eobf; }
    }
    if (wave == 0u && lane == 0u) M->res_lanes = total_lanes;
    __syncthreads();
}

// wrap: 0 raw, 1 zlib, 2 gzip, 3 auto (zlib or gzip by magic)
// RESUME (raw streams only; the streaming ABI's resumable inflate, zlib-rs/src/inflate.rs:288-320 keeps the same
// facts in its Mode / BitReader / Window): stream s starts at bit in_bit[s] (0..7) of its first byte, and
// resume[4s..4s+3] receives {byte, bit, output position, complete} of the start of the block the decode stopped
// in -- or of the end of the final block.  Decoding the same input again from there, with the output in front of
// that point as history, continues the stream.  A separate instantiation: the batch kernel's registers are full.
template <bool RESUME, uint32_t NW>   // NW: waves per stream, 1 (the batch kernels) or INF_MW (launches of a few streams)
__global__ void __launch_bounds__(64 * NW) zmi_inflate_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                                                         const uint32_t* __restrict__ in_len, uint32_t wrap,
                                                         uint8_t* out, const uint64_t* __restrict__ out_off,
                                                         const uint32_t* __restrict__ out_cap,
                                                         uint32_t* __restrict__ out_len, uint32_t* __restrict__ in_used,
                                                         uint32_t* __restrict__ check, int32_t* __restrict__ status,
                                                         uint64_t* __restrict__ bitmap, const uint64_t* __restrict__ bm_off,
                                                         const uint32_t* __restrict__ out_hist,
                                                         const uint32_t* __restrict__ in_bit, uint32_t* __restrict__ resume,
                                                         const uint32_t* __restrict__ order) {
    InfShared* S = &g_inf_lds;
    const uint32_t lane = zmi_lane();
    const uint32_t s = order[blockIdx.x];   // longest streams first (zmi_inflate_order_kernel)
    InfBits B;
    B.src = in + in_off[s];
    B.n = in_len[s];
    B.ipos = 0;
    B.hold = 0;
    B.nbits = 0;
    B.inbuf = S->inbuf;
    B.cbase = -(int32_t)(2u * INF_CHUNK);  // nothing staged yet: the first refill loads a chunk
    uint8_t* dst = out + out_off[s];
    const uint32_t cap = out_cap[s];
    uint32_t opos = 0;
    int32_t st = ZMI_OK;
    // preset dictionary (inflateSetDictionary, zlib-rs/src/inflate.rs:2492-2536): `hist` bytes in front of the output
    // region are history the stream may refer to
    const uint32_t hist = out_hist ? out_hist[s] : 0u;
    const uint64_t bmo = bm_off[s];
    uint32_t* bm32 = (uint32_t*)(bitmap + (bmo == ~0ull ? 0ull : bmo));   // bit p set: a back-reference starts at output byte p
    if (bmo == ~0ull) {
        if (lane == 0) {
            out_len[s] = 0; in_used[s] = 0; check[s] = 0; status[s] = ZMI_NO_SCRATCH;
            if (RESUME) { resume[4u * s] = 0; resume[4u * s + 1u] = in_bit ? (in_bit[s] & 7u) : 0u; resume[4u * s + 2u] = 0; resume[4u * s + 3u] = 0; }
        }
        return;   // (every wave of the workgroup: nothing has been posted yet)
    }
    if (NW > 1u && zmi_wave() != 0u) {
        // helper waves: they take part in the fast passes the master (wave 0) posts and leave when it is done
        for (;;) {
            __syncthreads();
            if (zmi_uniform(g_inf_mw.cmd) == 2u) return;
            inf_pass_mw(B.src, dst, bm32);
        }
    }
    uint32_t kind_found = wrap;  // resolved wrapper: 0 raw, 1 zlib, 2 gzip
    uint32_t fixed_ready = 0;
    uint32_t last_blk_bits = 0;   // token bits of the block decoded last (0: none yet): sizes the fast passes of the next one
    uint32_t fskip = 0, fskip_len = 0;   // token rounds left before the next fast pass is tried, and how many it was last time

    // ---- wrapper header ----
    if (wrap == 3u) kind_found = (B.n >= 2u && inf_byte(B, 0) == 0x1Fu && inf_byte(B, 1) == 0x8Bu) ? 2u : 1u;
    if (kind_found == 1u) {
        if (B.n < 2u) st = ZMI_BUF_ERROR;
        else {
            uint32_t cmf = inf_byte(B, 0), flg = inf_byte(B, 1);
            if ((cmf & 0x0Fu) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u) st = ZMI_DATA_ERROR;
            B.ipos = 2;
            if (st == ZMI_OK && (flg & 0x20u)) {          // FDICT: a DICTID follows (checked by the host against the dictionary)
                if (hist == 0u) st = 2;                    // Z_NEED_DICT
                else if (B.n < 6u) st = ZMI_BUF_ERROR;
                else B.ipos = 6;
            }
        }
    } else if (kind_found == 2u) {
        if (B.n < 10u) st = ZMI_BUF_ERROR;
        else if (inf_byte(B, 0) != 0x1Fu || inf_byte(B, 1) != 0x8Bu || inf_byte(B, 2) != 8u || (inf_byte(B, 3) & 0xE0u)) st = ZMI_DATA_ERROR;
        else {
            uint32_t flg = inf_byte(B, 3);
            uint32_t p = 10;
            if (flg & 4u) {  // FEXTRA
                if (p + 2u > B.n) st = ZMI_BUF_ERROR;
                else { uint32_t xl = inf_byte(B, p) | ((uint32_t)inf_byte(B, p + 1u) << 8); p += 2u + xl; }
            }
            if (st == ZMI_OK && (flg & 8u)) {  // FNAME
                while (p < B.n && inf_byte(B, p) != 0) ++p;
                ++p;
            }
            if (st == ZMI_OK && (flg & 16u)) {  // FCOMMENT
                while (p < B.n && inf_byte(B, p) != 0) ++p;
                ++p;
            }
            if (st == ZMI_OK && (flg & 2u)) p += 2u;  // FHCRC (not verified)
            if (st == ZMI_OK && p > B.n) st = ZMI_BUF_ERROR;
            B.ipos = p;
        }
    }

    // RESUME, what inflate(Z_BLOCK) / inflate(Z_TREES) need (zlib-rs/src/inflate.rs:1276-1284,1323,1369,1772): in_bit[s] bits
    // 8..23 = stop at the boundary behind that many complete blocks (0: no limit), bit 24 = stop behind the header of the
    // first block (its tables are built, nothing of its data is decoded): resume = {byte, bit of the first bit behind the
    // header, output position, 2 | last-block flag << 2}, status ZMI_BLOCK_STOP either way
    uint32_t max_blocks = 0, blocks_done = 0;
    bool hdr_stop = false;
    if (RESUME) {
        const uint32_t ibw = in_bit ? in_bit[s] : 0u;
        const uint32_t sb = ibw & 7u;
        max_blocks = (ibw >> 8) & 0xFFFFu;
        hdr_stop = ((ibw >> 24) & 1u) != 0u;
        if (lane == 0) { S->rs[0] = 0; S->rs[1] = sb; S->rs[2] = 0; S->rs[3] = 0; S->misc[5] = 0; }
        if (sb) {
            inf_refill(B);
            if (B.nbits < sb) st = ZMI_BUF_ERROR;
            else inf_drop(B, sb);
        }
    }

    // ---- blocks ----
    uint32_t last = 0;
    while (st == ZMI_OK && !last) {
        inf_refill(B);
        if (RESUME && lane == 0) {
            const uint64_t at = 8ull * B.ipos - B.nbits;
            S->rs[0] = (uint32_t)(at >> 3); S->rs[1] = (uint32_t)at & 7u; S->rs[2] = opos;
        }
        if (RESUME && max_blocks != 0u && blocks_done >= max_blocks) { st = ZMI_BLOCK_STOP; break; }
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
            uint32_t l = inf_byte(B, B.ipos) | ((uint32_t)inf_byte(B, B.ipos + 1u) << 8);
            uint32_t nl = inf_byte(B, B.ipos + 2u) | ((uint32_t)inf_byte(B, B.ipos + 3u) << 8);
            B.ipos += 4u;
            if ((l ^ 0xFFFFu) != nl) { st = ZMI_DERR(DE_STORED_LEN); break; }   // "invalid stored block lengths"
            if (RESUME && hdr_stop) {
                if (lane == 0) { S->rs[0] = B.ipos; S->rs[1] = 0u; S->rs[2] = opos; S->rs[3] = 2u | (last << 2); }
                st = ZMI_BLOCK_STOP;
                break;
            }
            {
                // a stored block that is not complete yet, or has no room: what there is is copied (Mode::CopyBlock,
                // inflate.rs:1374-1394: min(length, room, input)) -- the bytes count as output of an unfinished block, the
                // checkpoint stays in front of the block's header
                const uint32_t have = B.n - B.ipos, room = cap > opos ? cap - opos : 0u;
                uint32_t copy = l < have ? l : have;
                copy = copy < room ? copy : room;
                for (uint32_t i = lane; i < copy; i += 64u) dst[opos + i] = B.src[B.ipos + i];
                opos += copy;
                B.ipos += copy;
                if (copy < l) { st = (room < l && room <= have) ? ZMI_NEED_OUTPUT : ZMI_BUF_ERROR; break; }
            }
            ++blocks_done;
            continue;
        }
        if (type == 3u) { st = ZMI_DERR(DE_BLOCK_TYPE); break; }  // "invalid block type"
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
            if (nlen > 286u || ndist > 30u) { st = ZMI_DERR(DE_TOO_MANY_SYMS); break; }  // "too many length or distance symbols"
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
            if (zmi_uniform(inf_build(S, 0u, 19u, S->dtab, 7u, INF_DSIZE))) { st = ZMI_DERR(DE_CODE_LENGTHS_SET); break; }  // "invalid code lengths set"
            // ---- the nlen + ndist code lengths, run-length coded with the code-length code ----
            // Same scheme as the symbol rounds below: lane i decodes the code-length token (<= 7 + 7 bits) that would
            // start at bit P + i, the real chain is walked with scalar lane reads, a wave scan places the runs.  Zero
            // runs (symbols 17 / 18) need no stores at all -- the array is cleared up front -- and a "repeat previous"
            // (16) fetches its value from the nearest earlier token that is not a 16.
            uint32_t have = 0, prevl = 0;
            const uint32_t total = nlen + ndist;
            for (uint32_t i = lane; i < 320u; i += 64u) S->stage[i] = 0;
            zmi_wave_sync();
            {
                uint64_t P = 8ull * B.ipos - B.nbits;
                const uint64_t Pend = 8ull * B.n;
                const uint32_t* iw = (const uint32_t*)B.inbuf;
                while (have < total && st == ZMI_OK) {
                    uint32_t ib = (uint32_t)(P >> 3);
                    uint32_t relbyte = (uint32_t)((int32_t)ib - B.cbase);
                    if ((int32_t)relbyte < 0 || relbyte + 20u > INF_CHUNK) {
                        B.cbase = (int32_t)zmi_uniform((uint32_t)inf_load_chunk(B.src, B.n, ib, B.inbuf));
                        relbyte = (uint32_t)((int32_t)ib - B.cbase);
                    }
                    const uint32_t bo = ((relbyte << 3) | ((uint32_t)P & 7u)) + lane;
                    const uint32_t wi = bo >> 5, sh = bo & 31u;
                    const uint32_t lo = __builtin_amdgcn_alignbit(iw[wi + 1u], iw[wi], sh);
                    const uint64_t left = Pend - P;
                    const int32_t rem = (int32_t)(left > 0x40000000ull ? 0x40000000u : (uint32_t)left) - (int32_t)lane;
                    const uint32_t e = S->dtab[lo & 127u];
                    const uint32_t bits = e & 0xFFu, sym = e >> 16;
                    const bool bad = ((e >> 8) & 0xFFu) != INF_OP_LIT || bits == 0u;
                    const uint32_t xb = sym == 16u ? 2u : (sym == 17u ? 3u : (sym == 18u ? 7u : 0u));
                    const uint32_t t = bits + xb;
                    const uint32_t x = (lo >> bits) & ((1u << xb) - 1u);
                    const uint32_t tw = bad ? 256u : ((int32_t)t > rem ? 128u : t);   // 128: more input needed, 256: invalid code
                    uint32_t pos = 0, w = 0;
                    uint64_t M = 0;
                    inf_walk(tw, pos, M, w);
                    int32_t rst = ZMI_OK;
                    if (pos < 64u) rst = (w & 128u) ? ZMI_BUF_ERROR : ZMI_DERR(DE_CODE_LENGTHS_SET);
                    bool on = (M >> lane) & 1ull;
                    const uint32_t rep = on ? (sym < 16u ? 1u : (sym == 18u ? 11u + x : 3u + x)) : 0u;
                    const uint32_t incl = zmi_wave_incl_scan(rep), excl = incl - rep;
                    // "invalid bit length repeat": a repeat with nothing in front of it, or a run past the last length
                    const uint64_t badrep = __ballot(on && ((sym == 16u && have + excl == 0u) || have + incl > total));
                    const uint64_t ends = __ballot(on && have + incl == total);
                    const uint32_t kb = badrep ? (uint32_t)__ffsll((unsigned long long)badrep) - 1u : 64u;
                    const uint32_t ke = ends ? (uint32_t)__ffsll((unsigned long long)ends) - 1u : 64u;
                    if (kb < 64u && kb <= ke) { st = ZMI_DERR(DE_BIT_LENGTH_REPEAT); break; }
                    if (ke < 64u) {   // the token in lane ke completes the header; nothing behind it belongs to it
                        M &= (2ull << ke) - 1ull;
                        on = (M >> lane) & 1ull;
                        pos = ke + zmi_readlane(t, ke);
                        rst = ZMI_OK;
                    }
                    // value of every run: its own symbol, zero, or (16) that of the nearest earlier token that is not a 16
                    const uint32_t own = sym < 16u ? sym : 0u;
                    const uint64_t N = __ballot(on && sym != 16u);
                    const uint64_t below = N & ((1ull << lane) - 1ull);
                    const uint32_t src_lane = below ? 63u - (uint32_t)__clzll((unsigned long long)below) : lane;
                    const uint32_t fetched = (uint32_t)__shfl((int)own, (int)src_lane);
                    const uint32_t val = sym == 16u ? (below ? fetched : prevl) : own;
                    if (on && val != 0u)
                        for (uint32_t j = 0; j < rep; ++j) S->stage[have + excl + j] = (uint8_t)val;
                    if (M) {
                        const uint32_t kl = 63u - (uint32_t)__clzll((unsigned long long)M);   // last token of this round
                        prevl = zmi_readlane(val, kl);
                        have += zmi_readlane(incl, kl);
                    }
                    P += pos;
                    if (rst != ZMI_OK) st = rst;
                }
                B.ipos = (uint32_t)(P >> 3);
                B.hold = 0;
                B.nbits = 0;
                if (st == ZMI_OK) {
                    inf_refill(B);
                    inf_drop(B, (uint32_t)P & 7u);
                }
            }
            if (st != ZMI_OK) break;
            zmi_wave_sync();
            for (uint32_t i = lane; i < nlen; i += 64u) S->lens[i] = S->stage[i];
            zmi_wave_sync();
            if (zmi_uniform(S->lens[256]) == 0) { st = ZMI_DERR(DE_MISSING_EOB); break; }  // "invalid code -- missing end-of-block"
            if (zmi_uniform(inf_build(S, 1u, nlen, S->ltab, INF_LROOT, INF_LSIZE))) { st = ZMI_DERR(DE_LITLEN_SET); break; }  // "invalid literal/lengths set"
            if (RESUME && lane == 0u) S->misc[5] = S->misc[3];
            if (lane < ndist) S->lens[lane] = S->stage[nlen + lane];
            zmi_wave_sync();
            if (zmi_uniform(inf_build(S, 2u, ndist, S->dtab, INF_DROOT, INF_DSIZE))) { st = ZMI_DERR(DE_DIST_SET); break; }  // "invalid distances set"
            // table entries in use for the most recent dynamic block: what inflateCodesUsed reports (the reference's state.next,
            // zlib-rs/src/inflate.rs:1753-1768,2372 -- there with roots 10 / 9, here with this kernel's 9 / 8 and exact-fit sub-tables)
            if (RESUME && lane == 0u) S->misc[5] += S->misc[3];
        }

        if (RESUME && hdr_stop) {   // behind the block type bits (fixed codes) or the transmitted code lengths (dynamic)
            if (lane == 0) {
                const uint64_t at = 8ull * B.ipos - B.nbits;
                S->rs[0] = (uint32_t)(at >> 3); S->rs[1] = (uint32_t)at & 7u; S->rs[2] = opos; S->rs[3] = 2u | (last << 2);
            }
            st = ZMI_BLOCK_STOP;
            break;
        }
        // ---- symbol rounds ----
        // Lane i decodes, speculatively, the complete tokens (literal, end-of-block, or length + distance with
        // their extra bits: at most 48 bits) that would start at bits P + i and P + 64 + i: two windows per round,
        // so that their LDS round trips overlap and the per-round scalar work is spread over twice the tokens.
        // The real token chain is then walked from bit P with scalar lane reads, a wave scan places the outputs,
        // and every lane on the chain stores its own token(s).
        {
            uint64_t P = 8ull * B.ipos - B.nbits;   // bit position of the next token
            const uint64_t Pblk = P;                // ... and of the block's first one
            const uint64_t Pend = 8ull * B.n;
            const uint32_t* iw = (const uint32_t*)B.inbuf;
            bool eob = false;
            while (!eob && st == ZMI_OK) {
                {
                    // lane-serial fast pass while there are >= 4 KiB of input behind P (see inf_fast_pass); it commits only
                    // what is certain, everything unusual falls through to a token round below
                    // (a block whose code does not re-synchronise -- the fixed code over literal-dense data is nearly fixed-length --
                    // commits a lane or two per pass at the price of 64: after such a pass the token rounds take over for a
                    // while, twice as long every time it happens again)
                    if (fskip != 0u) --fskip;
                    // (the single-wave kernel also takes the last kilobytes of a stream through a pass, with shorter sub-sequences: the
                    // token rounds are ~25x the instructions per token, and the last 4 KiB of every 440 KB stream were 8 % of its decode)
                    else if (Pend - P >= 8ull * (INF_FAST_BYTES + 32u) || (NW == 1u && Pend - P >= 64u * 96u + 128u)) {
                        // Bits per lane.  A pass ends at the end of the block, and the lanes behind that point have worked
                        // for nothing: streams of small blocks (drifting data: 4 KiB a block) spent every second pass
                        // on a block's last few hundred bytes at the price of 3.75 KiB.  The block before is the estimate
                        // of how much is left of this one; what is left is spread over all 64 lanes.
                        uint32_t sub = INF_SUB_BITS;
                        if (last_blk_bits != 0u) {
                            const uint64_t done = P - Pblk;
                            const uint32_t rest = done < (uint64_t)last_blk_bits ? last_blk_bits - (uint32_t)done : 0u;
                            if (rest == 0u) sub = 288u;   // longer than the block before: feel the way forward
                            else if (rest < 56u * INF_SUB_BITS) { sub = (rest + (rest >> 3) + 63u) >> 6; sub = (((sub + 31u) >> 5) | 1u) << 5; sub = sub < 96u ? 96u : (sub > INF_SUB_BITS ? INF_SUB_BITS : sub); }   // (whole dwords, an odd number of them)
                        }
                        if (NW == 1u && Pend - P < 8ull * (INF_FAST_BYTES + 32u)) {
                            // near the end of the stream: all 64 sub-sequences (and the 64 bits a lane reads past its own) inside the input
                            uint32_t room = ((uint32_t)(Pend - P) - 128u) >> 6;
                            room &= ~31u;                                   // whole dwords,
                            if ((room & 32u) == 0u) room -= 32u;            // an odd number of them (>= 96 - 32: the entry condition)
                            if (room < sub) sub = room;
                            if (sub < 64u) sub = 64u;
                        }
                        uint32_t fbits = 0, fout = 0, feob = 0;
                        uint32_t lanes;
                        if (NW > 1u) {
                            // how many waves: what the block before suggests is left of this one, spread over 64 lanes each; every
                            // wave needs its 4 KiB (and the 32 bytes in front) inside the input
                            uint32_t nact = NW;
                            if (last_blk_bits != 0u) {
                                const uint64_t done = P - Pblk;
                                const uint32_t rest = done < (uint64_t)last_blk_bits ? last_blk_bits - (uint32_t)done : 0u;
                                if (rest == 0u) { nact = 2u; sub = 288u; }
                                else {
                                    const uint32_t want = rest + (rest >> 3);
                                    // (a pass takes as long as one lane's sub-sequence: more waves with shorter ones, not fewer with full ones)
                                    nact = (want + 64u * INF_MW_SUB - 1u) / (64u * INF_MW_SUB);
                                    nact = nact < 1u ? 1u : (nact > NW ? NW : nact);
                                    sub = (want + 64u * nact - 1u) / (64u * nact);
                                    sub = (((sub + 31u) >> 5) | 1u) << 5;
                                    sub = sub < 96u ? 96u : (sub > INF_SUB_BITS ? INF_SUB_BITS : sub);
                                }
                            } else sub = INF_SUB_BITS;
                            while (nact > 1u && Pend - P < (uint64_t)(nact - 1u) * 64u * sub + 8ull * (INF_FAST_BYTES + 32u)) --nact;
                            if (lane == 0) {
                                InfMulti* M = &g_inf_mw;
                                M->cmd = 1u; M->nact = nact; M->sub = sub; M->Plo = (uint32_t)P; M->Phi = (uint32_t)(P >> 32);
                                M->opos = opos; M->cap = cap; M->hist = hist; M->fixed = fixed_ready;
                            }
                            __syncthreads();
                            inf_pass_mw(B.src, dst, bm32);
                            lanes = zmi_uniform(g_inf_mw.res_lanes);
                            if (lanes != 0u) { fbits = zmi_uniform(g_inf_mw.res_bits); fout = zmi_uniform(g_inf_mw.res_out); feob = zmi_uniform(g_inf_mw.res_eob); }
                        } else
                        lanes = zmi_uniform(inf_fast_pass(S, B.src, P, dst, bm32, opos, cap, hist, sub, &fbits, &fout, &feob, fixed_ready != 0u, B.n));
                        B.cbase = -(int32_t)(2u * INF_CHUNK);   // the pass staged its input over the token rounds' chunk
                        if (lanes < 8u && sub >= 288u) { fskip_len = fskip_len == 0u ? 4u : (fskip_len < 64u ? fskip_len * 2u : 64u); fskip = fskip_len; }
                        else if (lanes >= 32u) fskip_len = 0u;
                        if (lanes != 0u) {
                            P += zmi_uniform(fbits);
                            opos += zmi_uniform(fout);
                            eob = zmi_uniform(feob) != 0u;
                            continue;
                        }
                    }
                }
                uint32_t ib = (uint32_t)(P >> 3);
                uint32_t relbyte = (uint32_t)((int32_t)ib - B.cbase);
                // the two windows read up to 27 bytes past their first byte; the serial reader may also have
                // moved the chunk past bits it still held when it handed over
                if ((int32_t)relbyte < 0 || relbyte + 28u > INF_CHUNK) {
                    B.cbase = (int32_t)zmi_uniform((uint32_t)inf_load_chunk(B.src, B.n, ib, B.inbuf));
                    relbyte = (uint32_t)((int32_t)ib - B.cbase);
                }
                const uint32_t bo = ((relbyte << 3) | ((uint32_t)P & 7u)) + lane;   // my bit offset inside inbuf
                const uint32_t wi = bo >> 5, sh = bo & 31u;
                const uint32_t d0 = iw[wi], d1 = iw[wi + 1u], d2 = iw[wi + 2u], d3 = iw[wi + 3u], d4 = iw[wi + 4u];
                const uint64_t left = Pend - P;   // input bits from P to the end of the stream
                const int32_t rem = (int32_t)(left > 0x40000000ull ? 0x40000000u : (uint32_t)left) - (int32_t)lane;
                const InfTok TA = inf_tok(S, __builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh), rem);
                const InfTok TB = inf_tok(S, __builtin_amdgcn_alignbit(d3, d2, sh), __builtin_amdgcn_alignbit(d4, d3, sh), rem - 64);

                // walk the chain of real tokens through both windows
                uint32_t pos = 0, w = 0;
                uint64_t MA = 0, MB = 0;
                int32_t rst = ZMI_OK;
                inf_walk(TA.tw, pos, MA, w);
                bool inB = false;
                if (pos >= 64u) {
                    pos -= 64u;
                    inB = true;
                    inf_walk(TB.tw, pos, MB, w);
                }
                if (pos < 64u) {   // stopped by an end-of-block or an error token
                    if (w & 64u) {
                        if (inB) MB |= 1ull << pos; else MA |= 1ull << pos;
                        pos += w & 63u;
                        eob = true;
                    } else rst = (w & 128u) ? ZMI_BUF_ERROR : ZMI_DERR(DE_CODE);
                }
                if (inB) pos += 64u;   // bits consumed

                // place the outputs
                bool onA = (MA >> lane) & 1ull, onB = (MB >> lane) & 1ull;
                const uint32_t lenA = onA ? (TA.kind == 0u ? 1u : (TA.kind == 1u ? TA.val : 0u)) : 0u;
                const uint32_t lenB = onB ? (TB.kind == 0u ? 1u : (TB.kind == 1u ? TB.val : 0u)) : 0u;
                const uint32_t inclA = zmi_wave_incl_scan(lenA);
                const uint32_t totA = zmi_readlane(inclA, 63u);
                const uint32_t inclB = zmi_wave_incl_scan(lenB) + totA;
                const uint32_t exclA = inclA - lenA, exclB = inclB - lenB;
                const bool farA = onA && TA.kind == 1u && TA.dist > opos + exclA + hist;   // "invalid distance too far back"
                const bool farB = onB && TB.kind == 1u && TB.dist > opos + exclB + hist;
                const bool fullA = onA && lenA != 0u && opos + inclA > cap;
                const bool fullB = onB && lenB != 0u && opos + inclB > cap;
                const uint64_t cutA = __ballot(farA || fullA), cutB = __ballot(farB || fullB);
                uint32_t tot;
                if (cutA | cutB) {
                    // the round ends in front of token k; whatever the walk found behind it is not reached
                    uint32_t fb;
                    if (cutA) {
                        const uint32_t k = (uint32_t)__ffsll((unsigned long long)cutA) - 1u;
                        fb = (uint32_t)((__ballot(farA) >> k) & 1ull);
                        MA &= (1ull << k) - 1ull;
                        MB = 0;
                        tot = zmi_readlane(exclA, k);
                        pos = k;
                    } else {
                        const uint32_t k = (uint32_t)__ffsll((unsigned long long)cutB) - 1u;
                        fb = (uint32_t)((__ballot(farB) >> k) & 1ull);
                        MB &= (1ull << k) - 1ull;
                        tot = zmi_readlane(exclB, k);
                        pos = 64u + k;
                    }
                    onA = (MA >> lane) & 1ull;
                    onB = (MB >> lane) & 1ull;
                    eob = false;
                    rst = fb ? ZMI_DERR(DE_TOO_FAR_BACK) : ZMI_NEED_OUTPUT;
                } else {
                    tot = zmi_readlane(inclB, 63u);
                }
                inf_emit(dst, bm32, onA, TA, opos + exclA);
                inf_emit(dst, bm32, onB, TB, opos + exclB);
                opos += tot;
                P += pos;
                if (rst != ZMI_OK) st = rst;
            }
            if (eob) { last_blk_bits = P - Pblk > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(P - Pblk); ++blocks_done; }
            // hand the bit position back to the serial reader (block headers, stored blocks, trailer)
            B.ipos = (uint32_t)(P >> 3);
            B.hold = 0;
            B.nbits = 0;
            if (st == ZMI_OK) {
                inf_refill(B);
                inf_drop(B, (uint32_t)P & 7u);
            }
        }
    }
    // ---- trailer ----
    uint32_t chk = 0;
    if (st == ZMI_OK) {
        if (RESUME && lane == 0) {   // complete: the position is that of the first bit behind the final block
            const uint64_t at = 8ull * B.ipos - B.nbits;
            S->rs[0] = (uint32_t)(at >> 3); S->rs[1] = (uint32_t)at & 7u; S->rs[2] = opos; S->rs[3] = 1u;
        }
        // give back whole unread bytes
        B.ipos -= B.nbits >> 3;
        if (kind_found == 1u) {
            if (B.ipos + 4u > B.n) st = ZMI_BUF_ERROR;
            else {
                chk = ((uint32_t)inf_byte(B, B.ipos) << 24) | ((uint32_t)inf_byte(B, B.ipos + 1u) << 16) |
                      ((uint32_t)inf_byte(B, B.ipos + 2u) << 8) | inf_byte(B, B.ipos + 3u);
                B.ipos += 4u;
            }
        } else if (kind_found == 2u) {
            // CRC first (4 bytes), ISIZE after 4 more -- a stream cut between the two still reports a
            // CRC mismatch as a data error, as the reference's streaming state machine does
            if (B.ipos + 4u > B.n) st = ZMI_BUF_ERROR;
            else {
                chk = inf_byte(B, B.ipos) | ((uint32_t)inf_byte(B, B.ipos + 1u) << 8) | ((uint32_t)inf_byte(B, B.ipos + 2u) << 16) |
                      ((uint32_t)inf_byte(B, B.ipos + 3u) << 24);
                B.ipos += 4u;
                if (B.ipos + 4u > B.n) st = ZMI_TRAILER_SHORT;
                else {
                    uint32_t isize = inf_byte(B, B.ipos) | ((uint32_t)inf_byte(B, B.ipos + 1u) << 8) |
                                     ((uint32_t)inf_byte(B, B.ipos + 2u) << 16) | ((uint32_t)inf_byte(B, B.ipos + 3u) << 24);
                    B.ipos += 4u;
                    if (isize != opos) st = ZMI_LENGTH_MISMATCH;  // "incorrect length check" (after the CRC check)
                }
            }
        }
    }
    if (lane == 0) {
        out_len[s] = opos;
        in_used[s] = B.ipos;
        check[s] = RESUME ? S->misc[5] : chk;   // (a resumable decode is a raw stream: no trailer value; the word carries the tables' size)
        status[s] = st;
        if (RESUME) {
            resume[4u * s] = S->rs[0]; resume[4u * s + 1u] = S->rs[1]; resume[4u * s + 2u] = S->rs[2]; resume[4u * s + 3u] = S->rs[3];
        }
    }
    if (NW > 1u) {   // the helper waves leave with the master
        if (lane == 0) g_inf_mw.cmd = 2u;
        __syncthreads();
    }
}

// ---- bitmap planning: stream s owns the 64-bit words [bm_off[s], bm_off[s] + cap/64 + 2) of the scratch ----
// Exclusive scan over the word counts by one workgroup; a stream that does not fit gets ~0 (its decode
// reports Z_MEM_ERROR) -- the host cannot know the capacities without a synchronisation.
__global__ void __launch_bounds__(1024) zmi_inflate_plan_kernel(const uint32_t* __restrict__ out_cap, uint32_t n,
                                                                 uint64_t cap_words, uint64_t* __restrict__ bm_off) {
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = t * per, hi = lo + per < n ? lo + per : n;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += (uint64_t)(out_cap[i] >> 6) + 2ull;
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        uint64_t v = t >= d ? part[t - d] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t o = part[t] - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint64_t w = (uint64_t)(out_cap[i] >> 6) + 2ull;
        bm_off[i] = o + w <= cap_words ? o : ~0ull;
        o += w;
    }
}

__global__ void __launch_bounds__(256) zmi_inflate_clear_kernel(const uint32_t* __restrict__ out_cap, const uint64_t* __restrict__ bm_off,
                                                                 uint64_t* __restrict__ bitmap) {
    const uint32_t s = blockIdx.x;
    const uint64_t o = bm_off[s];
    if (o == ~0ull) return;
    const uint64_t w = (uint64_t)(out_cap[s] >> 6) + 2ull;
    for (uint64_t i = threadIdx.x; i < w; i += 256u) bitmap[o + i] = 0ull;
}

// Workgroup -> stream, largest compressed size first.  A launch is a few rounds of streams per CU (16 384 streams on 256
// CUs x 17), and a stream's decode time goes with its token count: a literal-dense 1 MiB stream alone on the chip takes
// 50 ms, a launch of 16 384 mixed ones 110 -- dealt in arrival order, the launch ended with a few long streams running
// on an empty chip.  Longest first is the classic remedy; the compressed size is the estimate that is free.  One
// workgroup: bucket sort on the top 10 bits below the largest size (the order inside a bucket does not matter).
__global__ void __launch_bounds__(1024) zmi_inflate_order_kernel(const uint32_t* __restrict__ in_len, uint32_t n, uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[1024];
    __shared__ uint32_t top;
    const uint32_t t = threadIdx.x;
    hist[t] = 0;
    if (t == 0) top = 0;
    __syncthreads();
    uint32_t m = 0;
    for (uint32_t i = t; i < n; i += 1024u) m = in_len[i] > m ? in_len[i] : m;
    atomicMax(&top, m);
    __syncthreads();
    const uint32_t hi = top;
    const uint32_t shift = hi >= 1024u ? 32u - (uint32_t)__clz(hi) - 10u : 0u;
    for (uint32_t i = t; i < n; i += 1024u) atomicAdd(&hist[in_len[i] >> shift], 1u);
    __syncthreads();
    // start of bucket b in descending order = streams in the buckets above it: suffix sum by doubling
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t v = t + d < 1024u ? hist[t + d] : 0u;
        __syncthreads();
        hist[t] += v;
        __syncthreads();
    }
    // hist[b] = streams in buckets >= b: bucket b ends there; fill it from its end downwards
    for (uint32_t i = t; i < n; i += 1024u) {
        const uint32_t b = in_len[i] >> shift;
        order[atomicSub(&hist[b], 1u) - 1u] = i;
    }
}

// ---- resolve pass: fill the back-reference holes of one stream, in order, inside an LDS ring ----
// ring[x & RES_MASK] holds output byte x for x in [loaded - RES_RING, loaded).  Lines are staged from HBM
// strictly in order (they carry the literals and the 3-byte records the decode pass left in the holes),
// a hole is filled from the ring, and lines that can no longer change are streamed back.
// ring index of output byte x; rb = start of the ring lap the load frontier is in (x within one lap of it)
static __device__ __forceinline__ uint32_t res_ri(uint32_t x, uint32_t rb) {
    int32_t i = (int32_t)(x - rb);
    i -= i >= (int32_t)RES_RING ? (int32_t)RES_RING : 0;
    i += i < 0 ? (int32_t)RES_RING : 0;
    return (uint32_t)i;
}
static __device__ __forceinline__ uint32_t res_inc(uint32_t i, uint32_t d) {
    i += d;
    return i >= RES_RING ? i - RES_RING : i;
}
// bytes [lo, n_out) of dst exist (lo > 0 only with a preset dictionary shorter than its 1 KiB-aligned slot)
static __device__ __forceinline__ uint4 res_fetch(const uint8_t* dst, uint32_t n_out, uint32_t at, bool aligned16, uint32_t lo) {
    const uint32_t so = at + 16u * zmi_lane();
    uint4 q;
    q.x = q.y = q.z = q.w = 0u;
    if (aligned16 && so >= lo && so + 16u <= n_out) q = *(const uint4*)(dst + so);
    else if (so < n_out && so + 16u > lo) {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        for (uint32_t j = 0; j < 16u; ++j)
            if (so + j >= lo && so + j < n_out) w[j >> 2] |= (uint32_t)dst[so + j] << (8u * (j & 3u));
        q.x = w[0]; q.y = w[1]; q.z = w[2]; q.w = w[3];
    }
    return q;
}
// `pre` holds block [loaded, loaded + RES_BLK) when `have` is set: the load of the next block is always in
// flight while the holes of the current one are filled
// `keep` = lowest output byte the batch being staged for may still read from the ring (its first hole - RES_NEAR)
static __device__ __forceinline__ void res_stage(uint8_t* ring, const uint8_t* dst, uint32_t n_out, uint32_t& loaded, uint32_t upto,
                                                 bool aligned16, uint4& pre, bool& have, uint32_t& rb, uint32_t lo, uint32_t keep) {
    const uint32_t lane = zmi_lane();
    zmi_wave_order();   // ring reads issued so far (write-back of final lines) stay in front of the stores below
    if (upto > loaded + RES_RING - 2048u) {   // a long stretch without holes: only the near window matters
        // restart at the block holding the oldest byte the batch can read from the ring.  Ring budget: keep's block
        // start .. upto + RES_BLK <= 1023 + RES_NEAR + RES_SPAN + 258 + RES_BLK <= RES_RING.
        loaded = keep & ~(RES_BLK - 1u);
        have = false;
    }
    while (loaded < upto) {
        if (!have) pre = res_fetch(dst, n_out, loaded, aligned16, lo);
        rb = (loaded / RES_RING) * RES_RING;
        *(uint4*)(ring + (loaded - rb) + 16u * lane) = pre;   // RES_RING is a multiple of RES_BLK: a block never wraps
        loaded += RES_BLK;
        pre = res_fetch(dst, n_out, loaded, aligned16, lo);
        have = true;
    }
    rb = (loaded / RES_RING) * RES_RING;
    zmi_wave_order();
}
static __device__ __forceinline__ void res_writeback(const uint8_t* ring, uint8_t* dst, uint32_t n_out, uint32_t from, uint32_t upto,
                                                     bool aligned4, uint32_t rb) {
    const uint32_t lane = zmi_lane();
    zmi_wave_order();
    for (uint32_t c = from; c < upto; c += 256u) {
        const uint32_t o = c + 4u * lane;
        const uint32_t w = *(const uint32_t*)(ring + res_ri(o, rb));
        if (aligned4 && o + 4u <= n_out) *(uint32_t*)(dst + o) = w;
        else
            for (uint32_t j = 0; j < 4u; ++j)
                if (o + j < n_out) dst[o + j] = (uint8_t)(w >> (8u * j));
    }
}

// One chunk = 64 bitmap words = 4096 output bytes.  Its holes are listed in LDS and taken 64 at a time, one
// hole per lane.  A hole may be filled once every earlier hole that starts below the end of its source is
// done; `need` is that count, found by ranking the source end in the chunk's bitmap.  Short, non-overlapping
// holes that are ready are filled together (lane-per-hole, 16 bytes per step); self-overlapping ones are
// filled by the whole wave when they are the first unfinished hole.  Text-like data finishes a batch in 3-5 steps.
// the first `left` (<= 16) bytes of w[] -> ring[ip...].  Neighbouring holes may share a dword, so only the hole's own bytes may be
// written: each of the (up to five) aligned dwords the bytes fall into gets ONE masked store, ds_mskor_b32 (LDS[a] = (LDS[a] & ~mask) |
// value, atomic against the other lanes' masked stores to the same dword), where sixteen predicated byte stores stood -- 10 of the
// kernel's 32 ms per 16 Ki streams, measured by doubling them (round 4).  With 23 streams per CU the kernel runs at the LDS pipe's
// throughput: what counts is the number of LDS operations (unaligned wide stores, which gfx950 takes, were slower than the bytes).
static __device__ __forceinline__ void res_mskor(uint8_t* p4, uint32_t mask, uint32_t val) {
#ifdef ZMI_EMU
    uint32_t* q = (uint32_t*)p4;
    *q = (*q & ~mask) | val;
#else
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)p4;
    asm volatile("ds_mskor_b32 %0, %1, %2" : : "v"(a), "v"(mask), "v"(val) : "memory");
#endif
}
static __device__ __forceinline__ void res_put16(uint8_t* ring, uint32_t ip, const uint32_t* w, uint32_t left) {
    if (ip + 20u <= RES_RING) {
        const uint32_t sh = ip & 3u, back = (4u - sh) & 3u;
        uint8_t* const p4 = ring + (ip & ~3u);
        // the 16 bytes shifted to the ring's dword grid: e[k] = bytes 4k - sh .. 4k - sh + 3 of w
        uint32_t e[5];
        e[0] = w[0] << (8u * sh);
        e[1] = sh ? __builtin_amdgcn_alignbyte(w[1], w[0], back) : w[1];
        e[2] = sh ? __builtin_amdgcn_alignbyte(w[2], w[1], back) : w[2];
        e[3] = sh ? __builtin_amdgcn_alignbyte(w[3], w[2], back) : w[3];
        e[4] = sh ? w[3] >> (8u * back) : 0u;
        const uint32_t end = sh + left;   // bytes [sh, end) of the 20
#pragma unroll
        for (uint32_t k = 0; k < 5u; ++k) {
            const uint32_t lo = sh > 4u * k ? sh - 4u * k : 0u;                 // first byte of dword k that is the hole's
            const uint32_t hi = end > 4u * k ? (end - 4u * k < 4u ? end - 4u * k : 4u) : 0u;   // one past the last
            if (hi > lo) {
                const uint32_t m = (0xFFFFFFFFu >> (32u - 8u * hi)) & (0xFFFFFFFFu << (8u * lo));
                res_mskor(p4 + 4u * k, m, e[k] & m);
            }
        }
    } else {
#pragma unroll
        for (uint32_t j = 0; j < 16u; ++j)
            if (j < left) ring[res_inc(ip, j)] = (uint8_t)(w[j >> 2] >> (8u * (j & 3u)));
    }
}
// one word of output that this kernel (this wave) may have written back earlier: read past the CU's L1
static __device__ __forceinline__ uint32_t res_ld_final(const uint32_t* p) {
#ifdef ZMI_EMU
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
#define RES_SHORT 16u
#define RES_CW 32u          // bitmap words (of 64 output bytes) per chunk: 2 KiB of output; the chunk tables are 1.75 KiB
struct ResChunk {
    uint64_t cw[RES_CW];         // bitmap words of the chunk
    uint32_t cex[RES_CW];        // holes in the chunk before word w
    uint16_t list[22u * RES_CW]; // hole positions relative to the chunk (a hole is >= 3 bytes: <= 22 per word)
};

__global__ void __launch_bounds__(64) zmi_inflate_resolve_kernel(uint8_t* out, const uint64_t* __restrict__ out_off,
                                                                 const uint32_t* __restrict__ out_len,
                                                                 const uint64_t* __restrict__ bitmap,
                                                                 const uint64_t* __restrict__ bm_off,
                                                                 const uint32_t* __restrict__ out_hist,
                                                                 const uint32_t* __restrict__ order) {
    ZMI_DYN_SMEM(smem);
    uint8_t* ring = smem;
    ResChunk* C = (ResChunk*)(smem + RES_RING);
    const uint32_t lane = zmi_lane();
    // the decode pass's order backwards: the work here goes with the number of back-references, and among streams of one
    // size the best-compressed ones have the most (41.0 -> 35.2 ms per 16 Ki streams against the decode order)
    const uint32_t s = order[gridDim.x - 1u - blockIdx.x];
    const uint64_t bmo = bm_off[s];
    const uint32_t n_real = out_len[s];
    if (bmo == ~0ull || n_real == 0u) return;
    const uint64_t* bm = bitmap + bmo;
    const uint32_t nwords = (n_real + 63u) >> 6;
    // With a preset dictionary of `hist` bytes in front of the output, every position in this kernel is shifted by
    // `shift` (hist rounded up to the staging block, so that lines stay aligned): the dictionary occupies
    // [lo, shift), the output [shift, n_out); nothing below `shift` is ever written back.
    const uint32_t hist = out_hist ? out_hist[s] : 0u;
    const uint32_t shift = (hist + RES_BLK - 1u) & ~(RES_BLK - 1u), lo = shift - hist;
    const uint32_t n_out = n_real + shift;
    uint8_t* dst = out + out_off[s] - shift;
    const bool aligned16 = ((uintptr_t)dst & 15u) == 0u, aligned4 = ((uintptr_t)dst & 3u) == 0u;

    uint32_t loaded = 0, wb = 0, rb = 0;
    bool any = false, have = false;
    uint4 pre;
    pre.x = pre.y = pre.z = pre.w = 0u;
    for (uint32_t cbase = 0; cbase < nwords; cbase += RES_CW) {
        const uint32_t idx = cbase + lane;
        const uint64_t v = (lane < RES_CW && idx < nwords) ? bm[idx] : 0ull;
        const uint32_t pc = (uint32_t)__popcll(v);
        const uint32_t incl = zmi_wave_incl_scan(pc);
        const uint32_t total = zmi_readlane(incl, 63u);
        if (total == 0u) continue;
        const uint32_t chunk0 = (cbase << 6) + shift;
        zmi_wave_order();
        if (lane < RES_CW) { C->cw[lane] = v; C->cex[lane] = incl - pc; }
        {
            uint64_t t = v;
            uint32_t k = incl - pc;
            while (t) {
                C->list[k++] = (uint16_t)(lane * 64u + (uint32_t)__ffsll((unsigned long long)t) - 1u);
                t &= t - 1ull;
            }
        }
        zmi_wave_order();
        for (uint32_t b0 = 0, nb = 0; b0 < total; b0 += nb) {
            const uint32_t cand = total - b0 < 64u ? total - b0 : 64u;
            const uint32_t rel = lane < cand ? C->list[b0 + lane] : 0u;
            // the batch ends where the holes (sorted) get further than RES_SPAN from its first one
            const uint32_t rel0 = zmi_readlane(rel, 0u);
            nb = (uint32_t)__popcll(__ballot(lane < cand && rel - rel0 <= RES_SPAN));
            const bool active = lane < nb;
            const uint32_t p = chunk0 + rel;
            const uint32_t p_first = chunk0 + rel0, p_last = chunk0 + zmi_readlane(rel, nb - 1u);
            if (!any) {
                // first hole of the stream: everything in front of it is final already; start one window back
                any = true;
                loaded = p_first > RES_NEAR ? (p_first - RES_NEAR) & ~(RES_BLK - 1u) : 0u;
                wb = p_first & ~255u;
            }
            {   // lines in front of this batch can no longer change
                const uint32_t fin = p_first & ~255u, upto = loaded < fin ? loaded : fin;
                if (upto > wb) res_writeback(ring, dst, n_out, wb, upto, aligned4, rb);
                if (fin > wb) wb = fin;
            }
            const uint32_t keep = p_first > RES_NEAR ? p_first - RES_NEAR : 0u;
            if (p_last + 3u > loaded) res_stage(ring, dst, n_out, loaded, p_last + 3u, aligned16, pre, have, rb, lo, keep);
            uint32_t rec = 0;
            if (active) {
                const uint32_t a = res_ri(p, rb);
                const uint32_t r0 = *(const uint32_t*)(ring + (a & ~3u)), r1 = *(const uint32_t*)(ring + res_inc(a & ~3u, 4u));
                rec = __builtin_amdgcn_alignbyte(r1, r0, a & 3u) & 0xFFFFFFu;
            }
            const uint32_t mlen = (rec >> 15) + 3u, md = (rec & 0x7FFFu) + 1u;
            const uint32_t last_end = p_last + zmi_readlane(mlen, nb - 1u);
            if (last_end > loaded) res_stage(ring, dst, n_out, loaded, last_end, aligned16, pre, have, rb, lo, keep);
            const uint32_t s0 = p - md;
            const uint32_t e = s0 + (mlen < md ? mlen : md);   // end of the bytes this hole reads
            uint32_t need = 0;                                  // holes of this batch that must be finished first
            if (active && e > p_first) {
                const uint32_t er = e - chunk0;                 // chunk0 <= p_first < e <= p
                const uint64_t below = C->cw[er >> 6] & ((1ull << (er & 63u)) - 1ull);
                need = C->cex[er >> 6] + (uint32_t)__popcll(below) - b0;
            }
            const bool coop = md < mlen;   // a hole that reads its own output goes through the whole-wave path
            const uint32_t mpack = mlen | (md << 16);
            uint64_t done = nb == 64u ? 0ull : ~0ull << nb;
            zmi_wave_order();
            // holes whose source lies further back than the ring reaches: the source is final in HBM (it ends in front of
            // this batch's first hole, and everything up to there has been resolved and written back), so they are
            // filled first, all at once, whatever order the holes of the batch depend on each other in
            const bool farh = active && md > RES_NEAR;
            if (__ballot(farh)) {
#ifndef ZMI_EMU
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's write-backs have reached L2
#endif
                const uint32_t lim = farh ? mlen : 0u;
                for (uint32_t base = 0; __ballot(base < lim); base += RES_SHORT) {
                    uint32_t w[4] = {0u, 0u, 0u, 0u};
                    const uint32_t left = base < lim ? lim - base : 0u;
                    if (left) {
                        const uintptr_t P = (uintptr_t)(dst + s0 + base);
                        const uint32_t* q = (const uint32_t*)(P & ~(uintptr_t)3);
                        const uint32_t sh = (uint32_t)(P & 3u), span = sh + (left < RES_SHORT ? left : RES_SHORT);
                        // the 20 bytes from the dword the source starts in: always inside this stream's output below the hole (a far
                        // source ends more than RES_NEAR - 258 bytes in front of it).  Two loads, 16 + 4 bytes, past the CU's L1
                        // (sc0 sc1: this wave may have written these lines back earlier) -- five dword atomics cost 3.7 of the
                        // kernel's 28.5 ms (measured by doubling them)
                        uint32_t q0, q1, q2, q3, q4;
                        (void)span;
#ifdef ZMI_EMU
                        q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3]; q4 = q[4];
#else
                        {
                            uint4 v4;
                            asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dword %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                                         : "=&v"(v4), "=&v"(q4) : "v"(q) : "memory");
                            q0 = v4.x; q1 = v4.y; q2 = v4.z; q3 = v4.w;
                        }
#endif
                        w[0] = __builtin_amdgcn_alignbyte(q1, q0, sh);
                        w[1] = __builtin_amdgcn_alignbyte(q2, q1, sh);
                        w[2] = __builtin_amdgcn_alignbyte(q3, q2, sh);
                        w[3] = __builtin_amdgcn_alignbyte(q4, q3, sh);
                    }
                    res_put16(ring, res_ri(p + base, rb), w, left);
                }
                done |= __ballot(farh);
                zmi_wave_order();
            }
            while (~done) {
                const uint32_t D = (uint32_t)__ffsll((unsigned long long)~done) - 1u;   // first unfinished hole
                const uint64_t R = __ballot(active && !((done >> lane) & 1ull) && !coop && need <= D);
                if ((R >> D) & 1ull) {
                    // lane-per-hole, RES_SHORT bytes per step: sources are final and disjoint from every destination
                    const bool mine = (R >> lane) & 1ull;
                    const uint32_t lim = mine ? mlen : 0u;
                    for (uint32_t base = 0; __ballot(base < lim); base += RES_SHORT) {
                        uint32_t w[4] = {0u, 0u, 0u, 0u};
                        if (base < lim) {
                            const uint32_t a = res_ri(s0 + base, rb), a4 = a & ~3u, sh = a & 3u;
                            uint32_t q0, q1, q2, q3, q4;
                            if (a4 + 20u <= RES_RING) {
                                const uint32_t* q = (const uint32_t*)(ring + a4);
                                q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3]; q4 = q[4];
                            } else {   // the 20 bytes wrap around the end of the ring
                                q0 = *(const uint32_t*)(ring + a4);
                                q1 = *(const uint32_t*)(ring + res_inc(a4, 4u));
                                q2 = *(const uint32_t*)(ring + res_inc(a4, 8u));
                                q3 = *(const uint32_t*)(ring + res_inc(a4, 12u));
                                q4 = *(const uint32_t*)(ring + res_inc(a4, 16u));
                            }
                            w[0] = __builtin_amdgcn_alignbyte(q1, q0, sh);
                            w[1] = __builtin_amdgcn_alignbyte(q2, q1, sh);
                            w[2] = __builtin_amdgcn_alignbyte(q3, q2, sh);
                            w[3] = __builtin_amdgcn_alignbyte(q4, q3, sh);
                        }
                        const uint32_t left = base < lim ? lim - base : 0u;
                        res_put16(ring, res_ri(p + base, rb), w, left);
                    }
                    done |= R;
                } else {
                    // the first unfinished hole is long or overlaps itself: the whole wave fills it
                    const uint32_t mp = zmi_readlane(mpack, D);
                    const uint32_t cl = mp & 0xFFFFu, cd = mp >> 16;
                    const uint32_t cp = chunk0 + zmi_readlane(rel, D);
                    const uint32_t cs = cp - cd;
                    // byte i of the copy is source byte (i mod cd), all of them original
                    if (cd >= cl) {
                        uint8_t v5[5];
#pragma unroll
                        for (uint32_t k = 0; k < 5u; ++k) {
                            uint32_t i = lane + 64u * k;
                            v5[k] = i < cl ? ring[res_ri(cs + i, rb)] : (uint8_t)0;
                        }
#pragma unroll
                        for (uint32_t k = 0; k < 5u; ++k) {
                            uint32_t i = lane + 64u * k;
                            if (i < cl) ring[res_ri(cp + i, rb)] = v5[k];
                        }
                    } else {
                        for (uint32_t i = lane; i < cl; i += 64u) ring[res_ri(cp + i, rb)] = ring[res_ri(cs + i % cd, rb)];
                    }
                    done |= 1ull << D;
                }
                zmi_wave_order();
            }
        }
    }
    if (any) {
        const uint32_t end = (n_out + 255u) & ~255u, upto = loaded < end ? loaded : end;
        if (upto > wb) res_writeback(ring, dst, n_out, wb, upto, aligned4, rb);
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
    else if (st == ZMI_BLOCK_STOP) { st = ZMI_BUF_ERROR; det = 3; }
    else if (st <= ZMI_DERR(1)) { det = 16 + (-3000 - st); st = ZMI_DATA_ERROR; }   // the cause travels in detail
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

// mw_max: launches of up to this many streams take the multi-wave kernel (the context's value, 512 in the product; the tests run
// every inflate family under 16 and under 512 -- ZMI_INF_MW_MAX with ZMI_TUNING, read per call in zmi_api.hip, no process state)
extern "C" int zmi_launch_inflate(const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n_streams,
                                  uint32_t wrap, uint8_t* d_out, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                                  uint32_t* d_out_len, uint32_t* d_in_used, uint32_t* d_check, int32_t* d_status,
                                  uint64_t* d_bitmap, uint64_t bitmap_words, uint64_t* d_bm_off, const uint32_t* d_out_hist,
                                  const uint32_t* d_in_bit, uint32_t* d_resume, uint32_t* d_order, uint32_t mw_max, hipStream_t stream) {
    if (n_streams == 0) return 0;
    ZMI_LAUNCH(zmi_inflate_order_kernel, dim3(1), dim3(1024), 0, stream, d_in_len, n_streams, d_order);
    ZMI_LAUNCH(zmi_inflate_plan_kernel, dim3(1), dim3(1024), 0, stream, d_out_cap, n_streams, bitmap_words, d_bm_off);
    ZMI_LAUNCH(zmi_inflate_clear_kernel, dim3(n_streams), dim3(256), 0, stream, d_out_cap, (const uint64_t*)d_bm_off, d_bitmap);
    // a launch of up to a few hundred streams gives every stream a workgroup of INF_MW waves (inf_pass_mw): such a launch takes as
    // long as its slowest stream, and a stream alone on a CU is latency-bound (32 ... 512 streams of 1 MiB, every data class: decode
    // 17.7 ... 18.1 -> 13.7 ... 13.9 ms; at 1024 streams one wave each is ahead, 18.2 against 19.3 ms); thousands of streams fill the
    // chip with one wave each
    const bool mw = n_streams <= mw_max;
#define INF_GO(R, W, IB, RS) ZMI_LAUNCH((zmi_inflate_kernel<R, W>), dim3(n_streams), dim3(64u * W), 0, stream, d_in, d_in_off, d_in_len, wrap, d_out, \
                                d_out_off, d_out_cap, d_out_len, d_in_used, d_check, d_status, d_bitmap, (const uint64_t*)d_bm_off, d_out_hist, IB, RS,  \
                                (const uint32_t*)d_order)
    if (d_resume) { if (mw) INF_GO(true, INF_MW, d_in_bit, d_resume); else INF_GO(true, 1u, d_in_bit, d_resume); }
    else { if (mw) INF_GO(false, INF_MW, (const uint32_t*)nullptr, (uint32_t*)nullptr); else INF_GO(false, 1u, (const uint32_t*)nullptr, (uint32_t*)nullptr); }
#undef INF_GO
    return 0;
}

extern "C" int zmi_launch_inflate_resolve(uint8_t* d_out, const uint64_t* d_out_off, const uint32_t* d_out_len, uint32_t n_streams,
                                          const uint64_t* d_bitmap, const uint64_t* d_bm_off, const uint32_t* d_out_hist,
                                          const uint32_t* d_order, hipStream_t stream) {
    if (n_streams == 0) return 0;
#ifndef ZMI_EMU
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)zmi_inflate_resolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RES_RING + sizeof(ResChunk)));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
#endif
    ZMI_LAUNCH(zmi_inflate_resolve_kernel, dim3(n_streams), dim3(64), RES_RING + sizeof(ResChunk), stream, d_out, d_out_off, d_out_len, d_bitmap, d_bm_off, d_out_hist, d_order);
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
