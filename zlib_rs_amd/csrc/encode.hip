// encode.hip -- deflate stage 2: parse selection, Huffman code construction and bit packing.
//
// Reference semantics being reproduced (format, not structure):
//   lazy match selection        zlib-rs/src/deflate/algorithm/{medium,slow}.rs
//   tally_lit / tally_dist      zlib-rs/src/deflate.rs:1464-1521
//   build_tree / gen_bitlen /
//   gen_codes                   zlib-rs/src/deflate.rs:1945-2160   (any valid <=15-bit prefix code is acceptable)
//   scan_tree / send_tree /
//   send_all_trees              zlib-rs/src/deflate.rs:2171-2314,1177-1260
//   zng_tr_flush_block          zlib-rs/src/deflate.rs:2316-2434   (stored / static / dynamic by cost)
//   BitWriter / compress_block  zlib-rs/src/deflate.rs:907-1175
//   header / trailer            zlib-rs/src/deflate.rs:1572-1601,2574-2627,2772-2789
//
// MI355X design: ONE WAVE PER 64 KiB PIECE of a shard (64-thread workgroups, 6.7 KiB LDS, 23 waves resident per
// CU), no workgroup barriers at all.  The batch supplies the parallelism (16 pieces x thousands of shards); the
// pieces of a shard end byte aligned and are concatenated by zmi_compact_kernel.  Inside a piece the wave walks
// two segments of 64 positions per trip (independent until the last step: twice the shuffles in flight):
//   * parse: every lane owns one position and knows its best match (from lz77.hip).  The greedy/lazy
//     rule (three positions deep) gives each position a local "next token" pointer; five rounds of pointer
//     doubling per half wave (ds_bpermute) and two scalar joins turn the pointers into the 64-bit set of
//     positions reachable from the segment's entry point -- the serial token walk of the CPU parse becomes
//     log2(32) shuffles.
//   * tokens are compacted with mbcnt and overwrite the match scratch in place (token index <=
//     position index), histogrammed with LDS atomics.
//   * blocks follow the data: after every sub-block of 4096 tokens an entropy test decides whether it joins the
//     open block or starts a new one (enc_split_pays).
//   * per block: lane-parallel rank sort of the symbol frequencies (scalar lane reads), two-queue Huffman merge on
//     wave-uniform state (leaf queue in registers, node queue a FIFO in LDS scratch), depths by parallel relaxation, Kraft-exact length limiting,
//     canonical codes by ballot ranks, the RFC 1951 header's run-length coding for all runs at once (ballots, a closed-form symbol
//     count per run, a prefix sum) and emitted through the bit packer.
//   * bit packing: a wave prefix-sum over the code lengths gives every token its output bit
//     position; codes are OR-ed into an LDS staging window with ds_or and whole dwords are stored
//     coalesced.  No serial bit writer on the token path.
// Bound: instruction issue (integer VALU/SALU/LDS), not HBM: 4 B/position scratch read + <=4 B written
// back + 1/ratio B of output per input byte.
#include "zmi_device.h"
#include "zmi_kernels.h"

#define ENC_NL 288u
#define ENC_ND 32u
#define ENC_NBL 19u
#define ENC_STG 128u

// 6.7 KiB per wave: 23 single-wave workgroups fit a CU's 160 KiB (the 9.2 KiB of round 1 held it at 17; the kernel waits on
// memory half of its cycles, so resident waves are what hides that).  The code tables of the emission pass share their
// space with the tree builder's scratch: the builder is done (and the header written) before the codes are generated.
struct EncShared {
    uint32_t lfreq[ENC_NL];   // symbol counts of the open block
    uint32_t dfreq[ENC_ND];
    uint32_t lfreq2[ENC_NL];  // ... and of the sub-block being tokenised (joins the block or starts the next one)
    uint32_t dfreq2[ENC_ND];
    union {
        struct {              // tree construction (enc_rank_sort, enc_huff_lengths_w)
            uint16_t order[ENC_NL];
            uint16_t lpar[ENC_NL];
            uint16_t ipar[ENC_NL];
            uint16_t idep[ENC_NL];
        };
        struct {              // emission (after the header): bit-reversed code | len << 16
            uint32_t lcode[ENC_NL];
            uint32_t dcode[ENC_ND];
        };
    };
    uint8_t hsym[ENC_NL + ENC_ND];   // (directly behind idep: the tree builder's node queue runs on from idep into these two)
    uint8_t hext[ENC_NL + ENC_ND];
    uint32_t stg[ENC_STG];
    uint8_t llen[ENC_NL];
    uint8_t dlen[ENC_ND];
    uint8_t bllen[32];
    uint32_t blfreq[32];
    uint32_t blcode[32];
    uint32_t cnt[16];
    uint32_t next[16];
    uint32_t misc[16];
};
#define M_NNZ 0
#define M_REL 1
#define M_CHOICE 2
#define M_HC 3
#define M_HLIT 4
#define M_HDIST 5
#define M_HCLEN 6
#define M_DYNHDR 7
#define M_FAR4 8    // farthest distance at which a match of 4 / 5 / 6 bytes is still cheaper than its literals, judged by the
#define M_FAR5 9    // code lengths of the block emitted last (enc_far_limits)
#define M_FAR6 10

// ---- symbol mapping (RFC 1951 3.2.5), computed instead of table-driven ----
// length 3..258 -> (code index 0..28, extra bit count, extra value)
static __device__ __forceinline__ void enc_len_sym(uint32_t len, uint32_t& idx, uint32_t& eb, uint32_t& ev) {
    uint32_t l = len - 3u;
    if (l < 8u) { idx = l; eb = 0; ev = 0; }
    else if (l == 255u) { idx = 28u; eb = 0; ev = 0; }
    else {
        uint32_t k = 31u - (uint32_t)__clz(l);  // >= 3
        eb = k - 2u;
        idx = 4u * (k - 1u) + ((l >> eb) & 3u);
        ev = l & ((1u << eb) - 1u);
    }
}
// distance 1..32768 -> (code 0..29, extra bit count, extra value)
// the symbol indices alone, without branches (the tokeniser counts symbols; the extra bits matter at emission only)
static __device__ __forceinline__ uint32_t enc_len_idx(uint32_t len) {
    const uint32_t l = len - 3u;
    const uint32_t k = 31u - (uint32_t)__clz(l | 4u);            // >= 2
    const uint32_t f = 4u * (k - 1u) + ((l >> (k - 2u)) & 3u);   // l in 4..7: k = 2, f = l
    return l == 255u ? 28u : (l < 4u ? l : f);
}
static __device__ __forceinline__ uint32_t enc_dist_idx(uint32_t dist) {
    const uint32_t d = dist - 1u;
    const uint32_t k = 31u - (uint32_t)__clz(d | 2u);            // >= 1
    const uint32_t f = 2u * k + ((d >> (k - 1u)) & 1u);          // d in 2..3: k = 1, f = d
    return d < 2u ? d : f;
}
static __device__ __forceinline__ void enc_dist_sym(uint32_t dist, uint32_t& idx, uint32_t& eb, uint32_t& ev) {
    // (branch-free: distances 1..4 are their own codes, then two codes per power of two)
    const uint32_t d = dist - 1u;
    const uint32_t k = 31u - (uint32_t)__clz(d | 2u);            // >= 1
    eb = k - 1u;                                                 // d < 4: k = 1, no extra bits
    const uint32_t f = 2u * k + ((d >> eb) & 1u);                // d in 2..3: f = d
    idx = d < 2u ? d : f;
    ev = d & ((1u << eb) - 1u);
}
static __device__ __forceinline__ uint32_t enc_lext(uint32_t idx) { return (idx < 8u || idx == 28u) ? 0u : (idx >> 2) - 1u; }
static __device__ __forceinline__ uint32_t enc_dext(uint32_t idx) { return idx < 4u ? 0u : (idx >> 1) - 1u; }
static __device__ __forceinline__ uint32_t enc_static_llen(uint32_t s) {
    return s < 144u ? 8u : (s < 256u ? 9u : (s < 280u ? 7u : 8u));
}

// ---- bit writer over the LDS staging window ----
struct EncWriter {
    uint32_t* outw;   // shard output, 4-byte aligned
    uint32_t cap_w;   // capacity in dwords
    uint32_t wbase;   // dwords already stored
    uint32_t rel;     // valid bits in stg[]
    uint32_t err;
};

// lane 0 only: append nbits (<= 32) of val
static __device__ __forceinline__ void enc_put0(EncShared* S, uint32_t& rel, uint32_t val, uint32_t nbits) {
    uint32_t w = rel >> 5, sh = rel & 31u;
    uint64_t v = (uint64_t)val << sh;
    S->stg[w] |= (uint32_t)v;
    if (sh + nbits > 32u) S->stg[w + 1u] |= (uint32_t)(v >> 32);
    rel += nbits;
}

// all lanes: store the complete dwords of the staging window, keep the partial one
static __device__ __forceinline__ void enc_flush(EncShared* S, EncWriter& W) {
    const uint32_t lane = zmi_lane();
    uint32_t nfull = W.rel >> 5;
    if (nfull == 0) return;
    if (W.wbase + nfull > W.cap_w) W.err = 1u;
    if (!W.err)
        for (uint32_t i = lane; i < nfull; i += 64u) W.outw[W.wbase + i] = S->stg[i];
    uint32_t carry = S->stg[nfull];
    zmi_wave_sync();
    for (uint32_t i = lane; i <= nfull; i += 64u) S->stg[i] = (i == 0u) ? carry : 0u;
    zmi_wave_sync();
    W.wbase += nfull;
    W.rel &= 31u;
}

// all lanes: lane i appends nbits (<= 48) of bits; lanes are concatenated in lane order.
// The staging window (ENC_STG dwords) is written back when the group would not fit behind what it holds, not after every
// group: a group of 64 tokens of text is ~800 bits, so a flush (store loop, carry, clearing: ~25 instructions) serves four
// or five groups.  Callers that append with enc_put0 flush first (enc_flush_block, the end of a piece).
#define ENC_STG_BITS (ENC_STG * 32u - 64u)   // (a group's last lane may touch the dword behind its last bit twice over)
static __device__ __forceinline__ void enc_emit_group(EncShared* S, EncWriter& W, uint64_t bits, uint32_t nbits) {
    const uint32_t incl = zmi_wave_incl_scan(nbits);
    const uint32_t total = zmi_readlane(incl, 63u);
    if (W.rel + total > ENC_STG_BITS) enc_flush(S, W);   // (wave-uniform)
    if (nbits) {
        uint32_t o = W.rel + incl - nbits;
        uint32_t w = o >> 5, sh = o & 31u;
        uint64_t v = bits << sh;
        atomicOr(&S->stg[w], (uint32_t)v);
        uint32_t w1 = (uint32_t)(v >> 32);
        if (w1) atomicOr(&S->stg[w + 1u], w1);
        if (sh) {
            uint32_t w2 = (uint32_t)(bits >> (64u - sh));
            if (w2) atomicOr(&S->stg[w + 2u], w2);
        }
    }
    W.rel += total;
    zmi_wave_sync();
}
// the same for at most 32 bits per lane (a group of literals, the dynamic header): 32-bit arithmetic, two dwords at most
static __device__ __forceinline__ void enc_emit_group16(EncShared* S, EncWriter& W, uint32_t bits, uint32_t nbits) {
    const uint32_t incl = zmi_wave_incl_scan(nbits);
    const uint32_t total = zmi_readlane(incl, 63u);
    if (W.rel + total > ENC_STG_BITS) enc_flush(S, W);
    if (nbits) {
        const uint32_t o = W.rel + incl - nbits;
        const uint32_t w = o >> 5, sh = o & 31u;
        atomicOr(&S->stg[w], bits << sh);
        if (sh + nbits > 32u) atomicOr(&S->stg[w + 1u], bits >> (32u - sh));
    }
    W.rel += total;
    zmi_wave_sync();
}

// entry `idx` (wave-uniform, < 320) of a table held as five registers per lane (entry = lane + 64 * register)
static __device__ __forceinline__ uint32_t enc_rd5(const uint32_t (&r)[5], uint32_t idx) {
    const uint32_t l = idx & 63u;
    switch (idx >> 6) {
        case 0: return zmi_readlane(r[0], l);
        case 1: return zmi_readlane(r[1], l);
        case 2: return zmi_readlane(r[2], l);
        case 3: return zmi_readlane(r[3], l);
        default: return zmi_readlane(r[4], l);
    }
}
static __device__ __forceinline__ void enc_wr5(uint32_t (&r)[5], uint32_t idx, uint32_t v) {
    const uint32_t l = idx & 63u;
    switch (idx >> 6) {
        case 0: r[0] = zmi_writelane(r[0], v, l); break;
        case 1: r[1] = zmi_writelane(r[1], v, l); break;
        case 2: r[2] = zmi_writelane(r[2], v, l); break;
        case 3: r[3] = zmi_writelane(r[3], v, l); break;
        default: r[4] = zmi_writelane(r[4], v, l); break;
    }
}

// ---- lane-parallel rank sort of the symbols with non-zero frequency (ascending freq, then index) ----
static __device__ void enc_rank_sort(EncShared* S, const uint32_t* freq, uint32_t nsym) {
    const uint32_t lane = zmi_lane();
    // one comparison per pair: key = count << 9 | symbol (a block has fewer than 2^23 tokens)
    uint32_t key[5], rk[5];
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < 5u; ++k) {
        const uint32_t i = lane + 64u * k;
        const uint32_t f = i < nsym ? freq[i] : 0u;
        key[k] = f ? ((f << 9) | i) : 0u;
        rk[k] = 0;
        mine += f != 0u;
    }
    // every symbol with a non-zero count is broadcast once (a scalar lane read, no LDS round trip) and counted against
#pragma unroll
    for (uint32_t c = 0; c < 5u; ++c) {
        uint64_t nz = __ballot(key[c] != 0u);
        while (nz) {
            const uint32_t jj = (uint32_t)__ffsll((unsigned long long)nz) - 1u;
            nz &= nz - 1ull;
            const uint32_t kj = zmi_readlane(key[c], jj);
#pragma unroll
            for (uint32_t k = 0; k < 5u; ++k) rk[k] += kj < key[k] ? 1u : 0u;
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < 5u; ++k)
        if (key[k] != 0u) S->order[rk[k]] = (uint16_t)(key[k] & 0x1FFu);
    uint32_t nnz = zmi_wave_sum(mine);
    if (lane == 0) S->misc[M_NNZ] = nnz;
    zmi_wave_sync();
}

// ---- all lanes: optimal code lengths (<= maxbits) for the symbols listed in S->order[0..nnz), wave-cooperative ----
// The two-queue merge itself is serial, but it runs on wave-uniform (scalar) state with the heads of both queues
// held in registers, so a merge costs one LDS round trip, not six; everything around it (sorted weights, node
// depths, depth histogram, length assignment) is lane-parallel.
static __device__ void enc_huff_lengths_w(EncShared* S, const uint32_t* freq, uint32_t nsym, uint32_t nnz, uint32_t maxbits,
                                          uint8_t* lens) {
    const uint32_t lane = zmi_lane();
    for (uint32_t i = lane; i < nsym; i += 64u) lens[i] = 0;
    zmi_wave_sync();
    if (nnz <= 1u) {
        if (lane == 0) {
            if (nnz == 0u) { lens[0] = 1; lens[1] = 1; }
            else {
                uint32_t s = S->order[0];
                lens[s] = 1;
                lens[s == 0u ? 1u : 0u] = 1;  // a second, unused code keeps the set complete (cf. deflate.rs:1957-1977)
            }
        }
        zmi_wave_sync();
        return;
    }
    // The sorted leaf weights live in five registers per lane (the head of the leaf queue is a scalar lane read); the weights
    // of the nodes made so far are a FIFO in LDS -- the scratch of the steps that come later: idep and, behind it in the
    // struct, the header's symbol list (1216 bytes for up to 285 nodes).  Every store of the loop is wave-uniform (all lanes
    // write the same value to the same address: no exec masking around it) and plain LDS traffic.  (Round 2 kept the node
    // queue in five registers too: indexing a register array with a wave-uniform index is a switch, ~60 instructions per merge.
    // Rounds 2-4 had the queue in two separate arrays behind a volatile pointer: the compiler made FLAT accesses with
    // system scope and a full wait of them, a memory round trip per merge -- 10 of the kernel's 88 ms.)
    uint32_t swr[5];
#pragma unroll
    for (uint32_t c = 0; c < 5u; ++c) {
        const uint32_t k = lane + 64u * c;
        swr[c] = k < nnz ? freq[S->order[k]] : 0xFFFFFFFFu;
    }
    zmi_wave_sync();   // (the weights have been read through S->order: the node queue may now reuse what lies behind it)
    static_assert(offsetof(EncShared, hsym) == offsetof(EncShared, idep) + sizeof(uint16_t) * ENC_NL, "node queue: idep, hsym, hext in a row");
    static_assert(offsetof(EncShared, hext) == offsetof(EncShared, hsym) + ENC_NL + ENC_ND, "node queue: idep, hsym, hext in a row");
    uint32_t* const nq = (uint32_t*)S->idep;
    {
        uint32_t li = 0, ii = 0;
        uint32_t cur = swr[0];              // the register holding the leaf queue's current 64 entries
        uint32_t lw = zmi_readlane(cur, 0); // head of the leaf queue (all ones: exhausted)
        uint32_t nw = 0xFFFFFFFFu;          // head of the node queue (all ones: nothing made yet)
        for (uint32_t ni = 0; ni + 1u < nnz; ++ni) {
            uint32_t w = 0;
#pragma unroll
            for (int pick = 0; pick < 2; ++pick) {
                if (lw <= nw) {
                    w += lw;
                    S->lpar[li] = (uint16_t)ni;
                    ++li;
                    if ((li & 63u) == 0u) cur = li == 64u ? swr[1] : (li == 128u ? swr[2] : (li == 192u ? swr[3] : swr[4]));
                    lw = zmi_readlane(cur, li & 63u);   // entries past nnz hold all ones
                } else {
                    w += nw;
                    S->ipar[ii] = (uint16_t)ni;
                    ++ii;
                    nw = ii < ni ? zmi_uniform(nq[ii]) : 0xFFFFFFFFu;
                }
            }
            nq[ni] = w;
            if (ii == ni) nw = w;   // the node just made is the head of its queue
        }
    }
    zmi_wave_sync();
    // node depths: a parent always has a larger index; relax until nothing changes (tree height rounds)
    const uint32_t root = nnz - 2u;
    for (uint32_t k = lane; k <= root; k += 64u) S->idep[k] = k == root ? (uint16_t)0 : (uint16_t)0xFFFFu;
    zmi_wave_sync();
    for (;;) {
        bool changed = false;
        for (uint32_t k = lane; k < root; k += 64u) {
            if (S->idep[k] == 0xFFFFu) {
                const uint32_t dp = S->idep[S->ipar[k]];
                if (dp != 0xFFFFu) { S->idep[k] = (uint16_t)(dp + 1u); changed = true; }
            }
        }
        zmi_wave_sync();
        if (!__ballot(changed)) break;
    }
    // histogram of leaf depths, capped at maxbits
    if (lane <= maxbits) S->cnt[lane] = 0;
    zmi_wave_sync();
    bool capped = false;
    for (uint32_t k = lane; k < nnz; k += 64u) {
        uint32_t d = (uint32_t)S->idep[S->lpar[k]] + 1u;
        if (d > maxbits) { d = maxbits; capped = true; }
        atomicAdd(&S->cnt[d], 1u);
    }
    const bool over = __ballot(capped) != 0ull;
    zmi_wave_sync();
    if (over && lane == 0) {
        // Kraft sum in units of 2^-maxbits; every step below lowers it by exactly one unit:
        // a leaf at the deepest non-full level d becomes an internal node whose children are that
        // leaf and one leaf lifted from level maxbits.
        uint32_t K = 0;
        for (uint32_t d = 1; d <= maxbits; ++d) K += S->cnt[d] << (maxbits - d);
        const uint32_t full = 1u << maxbits;
        while (K > full) {
            uint32_t d = maxbits - 1u;
            while (S->cnt[d] == 0u) --d;
            S->cnt[d]--;
            S->cnt[d + 1u] += 2u;
            S->cnt[maxbits]--;
            --K;
        }
    }
    zmi_wave_sync();
    // rarest symbols get the longest codes: sorted symbol k takes the depth whose slice of the order it falls in
    for (uint32_t k = lane; k < nnz; k += 64u) {
        uint32_t acc = 0, dk = 1;
        for (uint32_t d = maxbits; d >= 1u; --d) {
            const uint32_t c = S->cnt[d];
            if (k >= acc && k < acc + c) dk = d;
            acc += c;
        }
        lens[S->order[k]] = (uint8_t)dk;
    }
    zmi_wave_sync();
}

// ---- all lanes: canonical codes (bit-reversed for LSB-first emission), as gen_codes (trees.rs) ----
// A symbol's code is next[len] + its rank among the lower-indexed symbols of the same length.  Everything stays in registers
// and wave-uniform masks: per length one ballot per chunk of 64 symbols gives the rank (mbcnt) and the count (popcount) that
// moves next[] on -- no histogram in LDS, no serial pass of lane 0 (round 4).
template <uint32_t NCH>
static __device__ void enc_gen_codes_w(const uint8_t* lens, uint32_t nsym, uint32_t maxbits, uint32_t* table) {
    const uint32_t lane = zmi_lane();
    uint32_t l[NCH], code[NCH];
#pragma unroll
    for (uint32_t c = 0; c < NCH; ++c) {
        const uint32_t i = lane + 64u * c;
        l[c] = i < nsym ? lens[i] : 0u;
        code[c] = 0;
    }
    uint32_t next = 0, count = 0;
    for (uint32_t d = 1; d <= maxbits; ++d) {
        next = (next + count) << 1;
        uint32_t at = next;
#pragma unroll
        for (uint32_t c = 0; c < NCH; ++c) {
            const uint64_t m = __ballot(l[c] == d);
            if (l[c] == d) code[c] = at + zmi_mbcnt(m);
            at += (uint32_t)__popcll(m);
        }
        count = at - next;
    }
#pragma unroll
    for (uint32_t c = 0; c < NCH; ++c) {
        const uint32_t i = lane + 64u * c;
        if (i < nsym) table[i] = l[c] ? ((__brev(code[c]) >> (32u - l[c])) | (l[c] << 16)) : 0u;
    }
    zmi_wave_sync();
}

// ---- all lanes: plan the dynamic header -- run-length code the code lengths, build the code-length code ----
// leaves the symbol list in hsym/hext (count in misc[M_HC]), HLIT/HDIST/HCLEN and the header's size in bits in misc.
// The run-length scan is serial but reads the lengths from registers with scalar lane reads; the code-length code
// (19 symbols) goes through the same wave-cooperative builders as the two big codes.
static __device__ void enc_header_plan_w(EncShared* S) {
    const uint32_t lane = zmi_lane();
    // HLIT / HDIST: trailing zero lengths are not transmitted
    uint32_t hlit = 257u, hdist = 1u;
    for (uint32_t base = 0; base < 320u; base += 64u) {
        const uint32_t i = base + lane;
        const uint64_t ml = __ballot(i < 286u && S->llen[i] != 0);
        if (ml) { const uint32_t top = base + 64u - (uint32_t)__clzll((unsigned long long)ml); hlit = top > hlit ? top : hlit; }
    }
    {
        const uint64_t md = __ballot(lane < 30u && S->dlen[lane] != 0);
        if (md) { const uint32_t top = 64u - (uint32_t)__clzll((unsigned long long)md); hdist = top > hdist ? top : hdist; }
    }
    const uint32_t total = hlit + hdist;
    uint32_t lr[5];
#pragma unroll
    for (uint32_t c = 0; c < 5u; ++c) {
        const uint32_t i = lane + 64u * c;
        lr[c] = i < hlit ? S->llen[i] : (i < total ? S->dlen[i - hlit] : 0xFFu);
    }
    // Run-length coding, all runs at once (round 4; a scalar walk over the ~300 lengths was a tenth of the kernel): an entry
    // opens a run if it differs from the one in front of it (ballots: 5 x 64 wave-uniform bits, entry `total` closes the last
    // run), the run's length is the distance to the next set bit, the number of header symbols a run becomes has a closed form,
    // a prefix sum places it, and each run's lane writes its symbols.  Same symbols as trees.rs `scan_tree` would produce for the
    // two codes written as one sequence (RFC 1951 3.2.7 allows runs across the boundary).
    uint64_t sm[5];
    {
        uint32_t carry = 0x1FFu;   // "the length in front of entry 0": equal to none
#pragma unroll
        for (uint32_t c = 0; c < 5u; ++c) {
            uint32_t pv = zmi_lane_up1(lr[c]);
            if (lane == 0u) pv = carry;
            sm[c] = __ballot(lr[c] != pv);   // (entries past `total` all read 0xFF: no run starts behind the closing one)
            carry = zmi_readlane(lr[c], 63u);
        }
    }
    uint32_t fa[5];   // first run start in the chunks behind chunk c
    fa[4] = 320u;
#pragma unroll
    for (int c = 3; c >= 0; --c) fa[c] = sm[c + 1] ? 64u * (uint32_t)(c + 1) + (uint32_t)__ffsll((unsigned long long)sm[c + 1]) - 1u : fa[c + 1];
    uint32_t run[5], cnt[5], off[5];
    uint32_t hc = 0;
#pragma unroll
    for (uint32_t c = 0; c < 5u; ++c) {
        const uint32_t i = lane + 64u * c;
        const bool start = ((sm[c] >> lane) & 1ull) != 0ull && i < total;
        const uint64_t above = lane < 63u ? sm[c] >> (lane + 1u) : 0ull;
        const uint32_t nxt = above ? i + (uint32_t)__ffsll((unsigned long long)above) : fa[c];
        const uint32_t r = start ? nxt - i : 0u;
        uint32_t n = 0;
        if (start) {
            if (lr[c] == 0u) { const uint32_t q = r / 138u, m = r - 138u * q; n = q + (m >= 3u ? 1u : m); }
            else { const uint32_t q = (r - 1u) / 6u, m = (r - 1u) - 6u * q; n = 1u + q + (m >= 3u ? 1u : m); }
        }
        run[c] = r;
        cnt[c] = n;
        const uint32_t incl = zmi_wave_incl_scan(n);
        off[c] = hc + incl - n;
        hc += zmi_readlane(incl, 63u);
    }
#pragma unroll
    for (uint32_t c = 0; c < 5u; ++c) {
        if (cnt[c] == 0u) continue;
        const uint32_t v = lr[c];
        uint32_t r = run[c], o = off[c];
        if (v == 0u) {
            while (r >= 11u) { const uint32_t t = r > 138u ? 138u : r; S->hsym[o] = 18; S->hext[o] = (uint8_t)(t - 11u); ++o; r -= t; }
            if (r >= 3u) { S->hsym[o] = 17; S->hext[o] = (uint8_t)(r - 3u); ++o; r = 0; }
            while (r > 0u) { S->hsym[o] = 0; S->hext[o] = 0; ++o; --r; }
        } else {
            S->hsym[o] = (uint8_t)v; S->hext[o] = 0; ++o; --r;
            while (r >= 3u) { const uint32_t t = r > 6u ? 6u : r; S->hsym[o] = 16; S->hext[o] = (uint8_t)(t - 3u); ++o; r -= t; }
            while (r > 0u) { S->hsym[o] = (uint8_t)v; S->hext[o] = 0; ++o; --r; }
        }
    }
    if (lane < 32u) S->blfreq[lane] = 0;
    zmi_wave_sync();
    for (uint32_t k = lane; k < hc; k += 64u) atomicAdd(&S->blfreq[S->hsym[k]], 1u);
    zmi_wave_sync();
    enc_rank_sort(S, S->blfreq, ENC_NBL);
    enc_huff_lengths_w(S, S->blfreq, ENC_NBL, zmi_uniform(S->misc[M_NNZ]), 7u, S->bllen);
    enc_gen_codes_w<1>(S->bllen, ENC_NBL, 7u, S->blcode);
    const uint8_t blorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const uint64_t used = __ballot(lane < 19u && S->bllen[blorder[lane < 19u ? lane : 0u]] != 0);
    uint32_t hclen = used ? 64u - (uint32_t)__clzll((unsigned long long)used) : 0u;
    if (hclen < 4u) hclen = 4u;
    uint32_t part = 0;
    if (lane < ENC_NBL) part = S->blfreq[lane] * (S->bllen[lane] + (lane == 16u ? 2u : (lane == 17u ? 3u : (lane == 18u ? 7u : 0u))));
    const uint32_t bits = 3u + 14u + 3u * hclen + zmi_wave_sum(part);
    if (lane == 0) {
        S->misc[M_HC] = hc;
        S->misc[M_HLIT] = hlit;
        S->misc[M_HDIST] = hdist;
        S->misc[M_HCLEN] = hclen;
        S->misc[M_DYNHDR] = bits;
    }
    zmi_wave_sync();
}

// all lanes: how far back may a 4 / 5 / 6 byte match lie before its length + distance codes cost more bits than the
// literals it replaces?  Classic zlib has one fixed answer (TOO_FAR = 4096 for 3-byte matches); the reference has none
// outside Z_FILTERED (zlib-rs/src/deflate/algorithm/slow.rs:69-74).  Here the answer follows the data: after every
// sub-block the symbol counts gathered so far (lf / df: the open block) give an entropy estimate of the average literal
// cost and of the price of every length / distance code, and the parse of the next sub-block uses it.  Text (literals
// ~5 bits) keeps 4-byte matches out to a few KiB, record data with cheap literals drops them beyond a few hundred
// bytes.  (Estimated from counts, not from the Huffman code lengths of the last block: text grows one block per piece,
// so a limit that waited for a finished block would never be used.)
static __device__ __noinline__ void enc_far_limits(EncShared* S, const uint32_t* lf, const uint32_t* df) {
    const uint32_t lane = zmi_lane();
    uint32_t nl = 0, nlit = 0;
    float hl = 0.f;   // sum f log2 f over the literals
    for (uint32_t i = lane; i < 286u; i += 64u) {
        const uint32_t f = lf[i];
        nl += f;
        if (i < 256u && f) { nlit += f; hl += (float)f * __log2f((float)f); }
    }
    nl = zmi_wave_sum(nl);
    nlit = zmi_wave_sum(nlit);
    const uint32_t hl16 = zmi_wave_sum((uint32_t)(hl * 16.f));
    uint32_t dfl = lane < 30u ? df[lane] : 0u;
    const uint32_t nd = zmi_wave_sum(dfl);
    if (nlit < 256u || nd < 64u) return;   // too little to judge by: keep the limits in force
    const float lgN = __log2f((float)nl);
    // average literal cost in bits: log2 N - (sum f log2 f) / nlit
    const float lit = lgN - (float)hl16 * (1.f / 16.f) / (float)nlit;
    // price of every distance code: -log2 of its share (a code not seen yet: one occurrence) + its extra bits
    const float dcost = __log2f((float)nd) - __log2f((float)(dfl ? dfl : 1u)) + (float)enc_dext(lane < 30u ? lane : 0u);
#pragma unroll
    for (uint32_t L = 4u; L <= 6u; ++L) {
        const uint32_t fl = lf[257u + (L - 3u)];          // lengths 4, 5, 6 are codes 258, 259, 260: no extra bits
        const float lcost = lgN - __log2f((float)(fl ? fl : 1u));
        const bool ok = lane < 30u && lcost + dcost <= (float)L * lit;
        const uint64_t m = __ballot(ok);
        // the farthest affordable code: its range ends where the next code's base starts
        uint32_t far = 1u;
        if (m) {
            const uint32_t top = 63u - (uint32_t)__clzll((unsigned long long)m);
            far = top >= 29u ? 32768u : (top < 3u ? top + 1u : ((2u + ((top + 1u) & 1u)) << (((top + 1u) >> 1) - 1u)));
        }
        if (lane == 0) S->misc[M_FAR4 + (L - 4u)] = far;
    }
    zmi_wave_sync();
}

// all lanes: emit one deflate block for tokens [tok, tok+ntok) / raw bytes [bstart, bend)
// (out of line: called from two places; the writer state and the parameters travel by value so that they stay in
// registers on both sides of the call)
static __device__ __noinline__ EncWriter enc_flush_block(EncShared* S, EncWriter W, const uint32_t* tok, uint32_t ntok,
                                                         const uint8_t* src, uint32_t bstart, uint32_t bend, uint32_t is_final,
                                                         uint32_t strategy) {
    const uint32_t lane = zmi_lane();
    enc_flush(S, W);   // the header is appended by lane 0 (enc_put0): the window must be near empty
    if (lane == 0) S->lfreq[256] += 1u;  // end-of-block
    zmi_wave_sync();
    enc_rank_sort(S, S->lfreq, 286u);
    enc_huff_lengths_w(S, S->lfreq, ENC_NL, zmi_uniform(S->misc[M_NNZ]), 15u, S->llen);
    enc_rank_sort(S, S->dfreq, 30u);
    enc_huff_lengths_w(S, S->dfreq, ENC_ND, zmi_uniform(S->misc[M_NNZ]), 15u, S->dlen);
    const uint32_t blen = bend - bstart;
    const uint32_t nsub = blen ? (blen + 32767u) / 32768u : 1u;
    const uint32_t stored_bits = 8u * blen + 42u * nsub;
    enc_header_plan_w(S);
    // cost of the block with the dynamic and with the static code (lane-parallel over the symbols)
    uint32_t dpart = 0, spart = 0;
    for (uint32_t i = lane; i < 286u + 30u; i += 64u) {
        if (i < 286u) {
            const uint32_t f = S->lfreq[i];
            const uint32_t eb = i > 256u ? enc_lext(i - 257u) : 0u;
            dpart += f * (S->llen[i] + eb);
            spart += f * (enc_static_llen(i) + eb);
        } else {
            const uint32_t d = i - 286u;
            const uint32_t f = S->dfreq[d];
            const uint32_t eb = enc_dext(d);
            dpart += f * (S->dlen[d] + eb);
            spart += f * (5u + eb);
        }
    }
    const uint32_t dyn = zmi_uniform(S->misc[M_DYNHDR]) + zmi_wave_sum(dpart);
    const uint32_t stat = 3u + zmi_wave_sum(spart);
    uint32_t choice;   // 0 stored, 1 static, 2 dynamic
    if (strategy == 100u) choice = 0u;  // level 0: stored blocks only
    else if (strategy == 4u) choice = (stored_bits < stat) ? 0u : 1u;
    else if (stored_bits <= dyn && stored_bits <= stat) choice = 0u;
    else choice = (stat <= dyn) ? 1u : 2u;
    if (choice == 2u) {
        const uint32_t hc = zmi_uniform(S->misc[M_HC]);
        {
            // BFINAL, BTYPE, HLIT, HDIST, HCLEN (17 bits, lane 0) and the HCLEN code-length-code lengths (3 bits each, lanes 1 ..)
            // in one pass of the wave bit packer (a serial enc_put0 per field was ~400 instructions per block until round 4)
            const uint8_t blorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            const uint32_t hclen = zmi_uniform(S->misc[M_HCLEN]);
            uint32_t hb = 0, hn = 0;
            if (lane == 0u) {
                hb = is_final | (2u << 1) | ((S->misc[M_HLIT] - 257u) << 3) | ((S->misc[M_HDIST] - 1u) << 8) | ((hclen - 4u) << 13);
                hn = 17u;
            } else if (lane <= hclen) {
                hb = S->bllen[blorder[lane - 1u]];
                hn = 3u;
            }
            enc_emit_group16(S, W, hb, hn);
        }
        // the run-length coded code lengths, 64 symbols per step through the wave bit packer
        for (uint32_t base = 0; base < hc; base += 64u) {
            const uint32_t k = base + lane;
            uint32_t bits = 0;
            uint32_t nb = 0;
            if (k < hc) {
                const uint32_t hs = S->hsym[k];
                const uint32_t c = S->blcode[hs];
                bits = c & 0xFFFFu;
                nb = c >> 16;
                const uint32_t xb = hs == 16u ? 2u : (hs == 17u ? 3u : (hs == 18u ? 7u : 0u));
                bits |= (uint32_t)S->hext[k] << nb;
                nb += xb;   // <= 7 + 7
            }
            enc_emit_group16(S, W, bits, nb);
        }
        enc_gen_codes_w<5>(S->llen, 286u, 15u, S->lcode);
        enc_gen_codes_w<1>(S->dlen, 30u, 15u, S->dcode);
    } else if (choice == 1u) {
        if (lane == 0) {
            uint32_t rel = W.rel;
            enc_put0(S, rel, is_final | (1u << 1), 3u);
            S->misc[M_REL] = rel;
        }
        for (uint32_t i = lane; i < ENC_NL; i += 64u) S->llen[i] = (uint8_t)enc_static_llen(i);
        if (lane < ENC_ND) S->dlen[lane] = 5;
        zmi_wave_sync();
        W.rel = zmi_uniform(S->misc[M_REL]);
        enc_flush(S, W);
        enc_gen_codes_w<5>(S->llen, ENC_NL, 15u, S->lcode);
        enc_gen_codes_w<1>(S->dlen, 30u, 15u, S->dcode);
    }
    if (choice != 0u) {
        // tokens + the end-of-block symbol as virtual token index ntok; the next group's tokens are
        // fetched while the current group is encoded (the HBM/L2 latency is the longest stall here)
        uint32_t tk_next = lane < ntok ? tok[lane] : 0u;
        for (uint32_t base = 0; base <= ntok; base += 64u) {
            uint32_t i = base + lane;
            const uint32_t tk = tk_next;
            tk_next = (i + 64u < ntok) ? tok[i + 64u] : 0u;
            // a group of literals only (the common case on literal-dense data, a third of the groups of text): codes of <= 15 bits.
            // (The end-of-block symbol rides as the literal 256 of the virtual token behind the last one; table reads are done by
            // every lane -- a guarded read is a divergent branch -- and the symbol mapping is branch-free: the nest of
            // literal / match / end-of-block / short-length cases was seven exec-mask regions per group.)
            const uint32_t len = (tk >> 8) & 0x1FFu;
            const bool on = i <= ntok, mt = i < ntok && len != 0u;
            const uint32_t sym = i < ntok ? (tk & 0xFFu) : 256u;
            if (__ballot(mt) == 0ull) {
                const uint32_t c = S->lcode[sym];
                enc_emit_group16(S, W, c & 0xFFFFu, on ? c >> 16 : 0u);
                continue;
            }
            // length 3..258 -> code index, extra bits (as enc_len_sym, without its branches)
            const uint32_t l = (mt ? len : 3u) - 3u;
            const uint32_t k = 31u - (uint32_t)__clz(l | 4u);
            const bool top = l == 255u;
            const uint32_t leb = top ? 0u : k - 2u;
            const uint32_t li = top ? 28u : (l < 4u ? l : 4u * (k - 1u) + ((l >> (k - 2u)) & 3u));
            const uint32_t lev = l & ((1u << leb) - 1u);
            const uint32_t lc = S->lcode[mt ? 257u + li : sym];
            uint64_t bits = lc & 0xFFFFu;
            uint32_t nb = on ? lc >> 16 : 0u;
            if (mt) {
                uint32_t di, deb, dev;
                enc_dist_sym((tk >> 17) + 1u, di, deb, dev);
                const uint32_t dc = S->dcode[di];
                bits |= (uint64_t)lev << nb;
                nb += leb;
                bits |= (uint64_t)(dc & 0xFFFFu) << nb;
                nb += dc >> 16;
                bits |= (uint64_t)dev << nb;
                nb += deb;
            }
            enc_emit_group(S, W, bits, nb);
        }
    } else {
        // stored: sub-blocks of <= 32768 bytes, raw bytes go through the same staging window
        uint32_t p = bstart;
        for (uint32_t sb = 0; sb < nsub; ++sb) {
            uint32_t l = bend - p;
            if (l > 32768u) l = 32768u;
            uint32_t fin = (is_final && sb + 1u == nsub) ? 1u : 0u;
            if (lane == 0) {
                uint32_t rel = W.rel;
                enc_put0(S, rel, fin, 3u);
                rel = (rel + 7u) & ~7u;
                enc_put0(S, rel, l & 0xFFFFu, 16u);
                enc_put0(S, rel, (~l) & 0xFFFFu, 16u);
                S->misc[M_REL] = rel;
            }
            zmi_wave_sync();
            W.rel = S->misc[M_REL];
            enc_flush(S, W);
            for (uint32_t base = 0; base < l; base += 256u) {
                uint32_t o = base + lane * 4u;
                uint32_t v = 0, nb = 0;
                for (uint32_t j = 0; j < 4u; ++j)
                    if (o + j < l) { v |= (uint32_t)src[p + o + j] << (8u * j); nb += 8u; }
                enc_emit_group(S, W, v, nb);
            }
            p += l;
        }
    }
    return W;
}

// Should the sub-block (lfreq2/dfreq2: nh tokens) start a new deflate block instead of joining the open one
// (lfreq/dfreq: nH tokens)?  Zeroth-order entropy of the two symbol sets apart and together, lane-parallel over the
// alphabets; a block of its own must earn the bits of another dynamic header.  This is what makes blocks follow the
// data: stationary input (text) grows them to the span limit, drifting statistics get short blocks with their own
// codes.  (The reference cuts blocks by symbol count only, lit_bufsize = 16383 symbols, deflate.rs:321.)
static __device__ __noinline__ bool enc_split_pays(const EncShared* S, uint32_t nH, uint32_t nh, uint32_t hdr_bits) {
    const uint32_t lane = zmi_lane();
    float sH = 0.f, sh = 0.f, sm = 0.f;
    uint32_t dH = 0, dh = 0;
    for (uint32_t i = lane; i < ENC_NL + ENC_ND; i += 64u) {
        const bool isd = i >= ENC_NL;
        const uint32_t a = isd ? S->dfreq[i - ENC_NL] : S->lfreq[i];
        const uint32_t b = isd ? S->dfreq2[i - ENC_NL] : S->lfreq2[i];
        if (isd) { dH += a; dh += b; }
        if (a) sH += (float)a * __log2f((float)a);
        if (b) sh += (float)b * __log2f((float)b);
        if (a + b) sm += (float)(a + b) * __log2f((float)(a + b));
    }
    // fixed point (1/16 bit) so that the integer wave reduction can be used
    const uint32_t qH = zmi_wave_sum((uint32_t)(sH * 16.f)), qh = zmi_wave_sum((uint32_t)(sh * 16.f)), qm = zmi_wave_sum((uint32_t)(sm * 16.f));
    dH = zmi_wave_sum(dH);
    dh = zmi_wave_sum(dh);
    auto nlgn = [](uint32_t n) { return n ? (float)n * __log2f((float)n) : 0.f; };
    const float apart = nlgn(nH) + nlgn(dH) - (float)qH * (1.f / 16.f) + nlgn(nh) + nlgn(dh) - (float)qh * (1.f / 16.f);
    const float joined = nlgn(nH + nh) + nlgn(dH + dh) - (float)qm * (1.f / 16.f);
    return apart + (float)hdr_bits < joined;
}

// One segment of 64 positions: drop the short far matches (enc_far_limits), apply the lazy rule, return the lane's step (1:
// a literal).  m: the lane's match word (filtered in place); nxt: the words of the segment behind it (its first three
// positions are the look-ahead of this segment's last three lanes).
// (the look-ahead of the last three lanes comes from `nxt`, which is filtered only a trip later, so a position
// can be deferred in favour of a match at +1..+3 that is then dropped.  Filtering the words when they are loaded was built
// in round 3: ratio identical to four decimals on lcet10.txt / paper-100k.pdf / the benchmark shards -- three of 64 lanes,
// and only when the filter bites -- but 82 VGPRs instead of 80, which is a wave per SIMD: encode 99.3 -> 102.1 ms.  Kept as is.)
static __device__ __forceinline__ uint32_t enc_seg_step(uint32_t& m, uint32_t nxt, uint32_t pos, uint32_t pend, const zmi_enc_params& prm,
                                                        uint32_t far4, uint32_t far5, uint32_t far6) {
    {
        const uint32_t l0 = (m >> 8) & 0x1FFu, d0 = (m >> 17) + 1u;
        const uint32_t lim = l0 == 4u ? far4 : (l0 == 5u ? far5 : far6);
        if (l0 >= 4u && l0 <= 6u && d0 > lim) m &= 0xFFu;
    }
    // the matches one, two and three positions on: the lane above (one DPP move each), the top lanes from the next segment
    const uint32_t m1 = zmi_lane_down1(m, zmi_readlane(nxt, 0u));
    const uint32_t m2 = zmi_lane_down1(m1, zmi_readlane(nxt, 1u));
    const uint32_t m3 = zmi_lane_down1(m2, zmi_readlane(nxt, 2u));
    const bool valid = pos < pend;
    const uint32_t mlen = (m >> 8) & 0x1FFu, mlen1 = (m1 >> 8) & 0x1FFu, mlen2 = (m2 >> 8) & 0x1FFu, mlen3 = (m3 >> 8) & 0x1FFu;
    // lazy evaluation, three positions deep: a short match steps aside for a longer one right behind it, or for one
    // two / three positions on that is longer by more than the literals in between cost (all matches are known
    // here, so looking further than the reference's one-position lazy rule, algorithm/medium.rs / slow.rs, is a
    // shuffle, not a search: lcet10.txt +0.8 %, benchmark shards +0.6 % for ~12 instructions per 64 positions)
    // (bitwise on purpose: as && / || the compiler builds two nested divergent branches per segment, eight scalar instructions each)
    const bool defer = (mlen < prm.max_lazy) & ((mlen1 > mlen) | (mlen2 > mlen + prm.lazy2) | (mlen3 > mlen + prm.lazy3));
    uint32_t step = 1u;
    if (valid & (mlen >= 4u) & !defer) step = mlen;
    // a token may not cross the end of the piece (the next piece starts a fresh parse there)
    if (valid && pos + step > pend) { step = pend - pos; if (step < 3u) step = 1u; }
    return step;
}

// Levels 3-9: which positions become tokens was decided by the cost parse (csrc/parse.hip), two bits per position: 0 = literal,
// 1 = the match at its full length, 2 = one byte shorter.  The token chain follows these steps; nothing is judged here.
static __device__ __forceinline__ uint32_t enc_seg_step_cp(uint32_t m, uint32_t decw, uint32_t pos, uint32_t pend) {
    const uint32_t dec = (decw >> (2u * (zmi_lane() & 15u))) & 3u;
    const bool valid = pos < pend;
    const uint32_t room = valid ? pend - pos : 0u;       // (the parse clipped its lengths the same way: a token ends with its piece)
    uint32_t mlen = (m >> 8) & 0x1FFu;
    mlen = mlen < room ? mlen : room;
    uint32_t step = mlen + 1u - dec;
    step = ((dec != 0u) & (step >= 3u) & (step <= mlen)) ? step : 1u;
    return step;
}

#if defined(ZMI_EMU) || defined(ENC_NO_OCC)
#define ENC_OCC
#else
// six waves per SIMD: at 82+ VGPRs the register file holds five, and this kernel lives on the waves that hide its LDS
// round trips (measured: a wave per SIMD less costs ~3 %)
#define ENC_OCC __attribute__((amdgpu_waves_per_eu(6, 8)))
#endif
// CP: the token choice comes from the cost parse's decisions (`dec`, 2 bits per position, 16 bytes per segment of 64; levels 3-9);
// otherwise from the lazy rule below (levels 1 and 2, Z_HUFFMAN_ONLY).
// (A template kernel and not two kernels around one inlined body: in that form both came out at 116 VGPRs instead of 80.)
template <bool CP>
__global__ void __launch_bounds__(64) ENC_OCC zmi_encode_kernel_t(const uint8_t* __restrict__ data, const uint64_t* __restrict__ off,
                                                        const uint32_t* __restrict__ len, uint32_t first_shard,
                                                        uint32_t* match, uint64_t match_stride, const uint32_t* __restrict__ dec, uint64_t dec_stride,
                                                        const uint32_t* __restrict__ adler, const uint32_t* __restrict__ crc,
                                                        uint8_t* __restrict__ out, uint64_t out_stride, uint32_t pieces,
                                                        uint32_t region_stride, uint32_t* __restrict__ piece_len,
                                                        zmi_enc_params prm) {
    __shared__ EncShared Sh;
    EncShared* S = &Sh;
    const uint32_t lane = zmi_lane();
    // Workgroup -> (shard, piece): consecutive workgroups take consecutive SHARDS of one piece index, in the rotated
    // order of zmi_xcd_spread.  The dispatcher deals workgroups round-robin over 8 XCDs x 4 shader engines; with
    // "pieces of a shard are neighbours" and 8 pieces per shard every shader engine saw two of the benchmark's eight data
    // classes only, and the launch ran at the pace of the engine that had drawn the literal-heavy class (its pieces take
    // three times as long as text): 230 ms per 16 Ki shards against 146 ms for the classes one by one.
    const uint32_t n_local = gridDim.x / pieces;
    const uint32_t local = zmi_xcd_spread(blockIdx.x % n_local, n_local);   // shard index inside this launch group
    const uint32_t piece = blockIdx.x / n_local;
    const uint32_t s = first_shard + local;
    const uint8_t* src = data + off[s];
    const uint32_t n = len[s];
    uint32_t* tokbuf = match + (uint64_t)local * match_stride;
    const uint32_t* decw = CP ? dec + (uint64_t)local * dec_stride : nullptr;   // dword 4 * segment + lane / 16 holds this lane's two bits

    // piece geometry: `pieces` equal ranges (multiple of 64 positions); ranges past the end are empty
    uint32_t psize = ((n + pieces - 1u) / pieces + 63u) & ~63u;
    if (psize == 0u) psize = 64u;
    const uint32_t npieces = n ? (n + psize - 1u) / psize : 1u;  // pieces that hold data
    const uint32_t pstart = piece * psize;
    uint32_t pend = pstart + psize;
    if (pend > n) pend = n;
    const bool is_first = piece == 0u;
    // the piece that ends the deflate stream (BFINAL + trailer); in chained mode only the last shard's
    const bool ends_stream = prm.chain_mode == 0u || (prm.chain_mode == 1u && s == prm.last_shard);
    const bool is_last = piece + 1u == npieces && ends_stream;
    if (piece >= npieces) {
        if (lane == 0) piece_len[(uint64_t)s * pieces + piece] = 0u;
        return;
    }

    EncWriter W;
    W.outw = (uint32_t*)(out + (uint64_t)s * out_stride + (uint64_t)piece * region_stride);
    W.cap_w = region_stride >> 2;
    W.wbase = 0;
    W.rel = 0;
    W.err = 0;

    for (uint32_t i = lane; i < ENC_STG; i += 64u) S->stg[i] = 0;
    for (uint32_t i = lane; i < ENC_NL; i += 64u) { S->lfreq[i] = 0; S->lfreq2[i] = 0; }
    if (lane < ENC_ND) { S->dfreq[lane] = 0; S->dfreq2[lane] = 0; }
    zmi_wave_sync();

    // stream header (first piece only)
    if (lane == 0) {
        uint32_t rel = 0;
        if (!is_first) {
        } else if (prm.wrap == 1u) {
            uint32_t lf = prm.level < 2u ? 0u : (prm.level < 6u ? 1u : (prm.level == 6u ? 2u : 3u));
            uint32_t h = (0x78u << 8) | (lf << 6);
            h += 31u - (h % 31u);
            enc_put0(S, rel, h >> 8, 8u);
            enc_put0(S, rel, h & 0xFFu, 8u);
        } else if (prm.wrap == 2u) {
            uint32_t xfl = prm.level == 9u ? 2u : (prm.level < 2u ? 4u : 0u);
            enc_put0(S, rel, 0x00088B1Fu, 32u);
            enc_put0(S, rel, 0u, 32u);
            enc_put0(S, rel, xfl | (3u << 8), 16u);
        }
        S->misc[M_REL] = rel;
    }
    zmi_wave_sync();
    W.rel = S->misc[M_REL];
    enc_flush(S, W);

    if (n == 0u) {
        // empty input: a final static block holding only end-of-block (as the reference emits: 03 00)
        if (lane == 0) {
            uint32_t rel = W.rel;
            enc_put0(S, rel, 1u | (1u << 1), 3u);
            enc_put0(S, rel, 0u, 7u);
            S->misc[M_REL] = rel;
        }
        zmi_wave_sync();
        W.rel = S->misc[M_REL];
    }

    const uint32_t seg0 = pstart / 64u;
    const uint32_t nseg = (pend + 63u) / 64u;  // one past the last segment of this piece
    uint32_t e = 0;        // entry offset into the current segment (>= 64: segment fully covered by a match)
    uint32_t ntok = 0;     // tokens of the open block + the sub-block behind it
    uint32_t nH = 0;       // ... of which the open block holds this many (the sub-block: ntok - nH)
    uint32_t tok0 = pstart;  // scratch index of the open block's first token (<= the position it stands for)
    uint32_t bstart = pstart;  // first input byte covered by the open block
    uint32_t bendH = pstart;   // end of the open block's input
    // the match words are fetched two segments ahead: the word of the NEXT segment is needed right away (its first
    // position decides the lazy rule of this segment's last one), so a load issued in the same iteration would put
    // one HBM round trip on every segment
    // short matches that cost more than their literals are dropped before the parse looks at them (enc_far_limits)
    if (lane == 0) { S->misc[M_FAR4] = prm.far4; S->misc[M_FAR5] = prm.far5; S->misc[M_FAR6] = 32768u; }
    zmi_wave_sync();
    uint32_t far4 = prm.far4, far5 = prm.far5, far6 = 32768u;
    // Two segments (128 positions) per trip: their lazy rules and pointer-doubling chains do not depend on each other (only
    // the last step, picking the chain that starts where the token before ended, is serial), so the two sets of
    // shuffles are in flight together -- the kernel sits between latency and issue bound, and this is occupancy
    // that costs no LDS.  Match words are fetched two trips ahead.
    uint32_t m_a = (pstart + lane < pend) ? tokbuf[pstart + lane] : 0u;
    uint32_t m_b = (pstart + 64u + lane < pend) ? tokbuf[pstart + 64u + lane] : 0u;
    uint32_t m_c = (pstart + 128u + lane < pend) ? tokbuf[pstart + 128u + lane] : 0u;
    // (the decisions of a trip's two segments, fetched a trip ahead like the match words: 4 distinct dwords per segment)
    const uint32_t dlast = CP ? 4u * (nseg - 1u) + 3u : 0u;
    uint32_t d_a = 0u, d_b = 0u;
    if constexpr (CP) {
        d_a = decw[4u * seg0 + (lane >> 4)];
        d_b = decw[4u * (seg0 + 1u) + (lane >> 4) < dlast ? 4u * (seg0 + 1u) + (lane >> 4) : dlast];
    }
    for (uint32_t seg = seg0; seg < nseg; seg += 2u) {
        const uint32_t posA = seg * 64u + lane, posB = posA + 64u;
        // (loads at a clamped index + a select: a guarded load is a divergent branch, six scalar instructions around one load)
        const uint32_t plast = pend - 1u;   // (pend > 0 here: an empty shard has no segments)
        const uint32_t ld_d = tokbuf[posA + 192u < plast ? posA + 192u : plast], ld_e = tokbuf[posA + 256u < plast ? posA + 256u : plast];
        const uint32_t m_d = (posA + 192u < pend) ? ld_d : 0u;
        const uint32_t m_e = (posA + 256u < pend) ? ld_e : 0u;
        uint32_t stepA, stepB;
        if constexpr (CP) {
            const uint32_t ia = 4u * (seg + 2u) + (lane >> 4), ib = 4u * (seg + 3u) + (lane >> 4);
            const uint32_t n_a = decw[ia < dlast ? ia : dlast], n_b = decw[ib < dlast ? ib : dlast];
            stepA = enc_seg_step_cp(m_a, d_a, posA, pend);
            stepB = enc_seg_step_cp(m_b, d_b, posB, pend);
            d_a = n_a;
            d_b = n_b;
        } else {
            stepA = enc_seg_step(m_a, m_b, posA, pend, prm, far4, far5, far6);
            stepB = enc_seg_step(m_b, m_c, posB, pend, prm, far4, far5, far6);
        }
        const uint64_t validA = __ballot(posA < pend), validB = __ballot(posB < pend);
        uint32_t JA = lane + stepA, RA = zmi_lane_bit32_here(lane), JB = lane + stepB, RB = RA;
        const bool any_match = __ballot(stepA > 1u || stepB > 1u) != 0ull;
        if (any_match) {
            // Which positions start a token: follow lane -> lane + step.  Pointer doubling, the two halves of the wave on
            // their own: five rounds on (next position, 32-bit set of visited lanes of the half) instead of six on a
            // 64-bit set; the halves are joined through scalar reads below (the entry lane is wave-uniform).
            const uint32_t hb = (lane & 32u) + 32u;          // end of this lane's half
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const bool inA = JA < hb, inB = JB < hb;
                const uint32_t sA = inA ? JA : lane, sB = inB ? JB : lane;
                const uint32_t JAn = (uint32_t)__shfl((int)JA, (int)sA), RAn = (uint32_t)__shfl((int)RA, (int)sA);
                const uint32_t JBn = (uint32_t)__shfl((int)JB, (int)sB), RBn = (uint32_t)__shfl((int)RB, (int)sB);
                if (inA) { JA = JAn; RA |= RAn; }
                if (inB) { JB = JBn; RB |= RBn; }
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint32_t J = half ? JB : JA, R = half ? RB : RA, step = half ? stepB : stepA, mw = half ? m_b : m_a;
            const uint64_t validm = half ? validB : validA;
            if (e < 64u) {
                uint64_t mask;
                if (!any_match) {
                    mask = ~0ull << e;
                    e = 0u;
                } else {
                    uint32_t mlo = 0, mhi = 0, ex = e;
                    if (e < 32u) { mlo = zmi_readlane(R, e); ex = zmi_readlane(J, e); }
                    if (ex < 64u) { mhi = zmi_readlane(R, ex); ex = zmi_readlane(J, ex); }
                    mask = ((uint64_t)mhi << 32) | mlo;
                    e = ex - 64u;
                }
                mask &= validm;
                const bool in = (mask >> lane) & 1ull;
                if (in) {
                    uint32_t tk = step > 1u ? ((mw & ~(0x1FFu << 8)) | (step << 8)) : (mw & 0xFFu);
                    tokbuf[tok0 + ntok + zmi_mbcnt(mask)] = tk;
                    // (one count for every token, a second one for the matches: as if / else the literal and the length count were
                    // two divergent regions.  Counting in a pass of its own over the stored tokens -- 64 lanes, 64 tokens, symbols
                    // worked out once for the emission too -- took these sixty instructions out of the walk and was 5 % SLOWER:
                    // they ride in the shadow of the walk's shuffles, profiles/r05_parse_tables.txt)
                    const bool mt = step > 1u;
                    atomicAdd(&S->lfreq2[mt ? 257u + enc_len_idx(step) : (mw & 0xFFu)], 1u);
                    if (mt) atomicAdd(&S->dfreq2[enc_dist_idx((mw >> 17) + 1u)], 1u);
                }
                ntok += (uint32_t)__popcll(mask);
            } else {
                e -= 64u;
            }
        }
        m_a = m_c;
        m_b = m_d;
        m_c = m_e;
        const uint32_t done = (seg + 2u) * 64u;   // (may lie one segment behind the piece: everything below clamps to pend)
        const bool last_seg = seg + 2u >= nseg;
        // a sub-block closes after block_tokens tokens, optionally not before it spans min_sub_span bytes of input (or holds
        // twice the tokens): literal-dense data cuts a block, with a full tree construction, every 4 KiB -- the random-walk
        // class spends half of its encode time building trees, but those small blocks are also where its ratio comes from
        const bool sub_full = ntok - nH >= prm.block_tokens && (done - bendH >= prm.min_sub_span || ntok - nH >= 2u * prm.block_tokens);
        if (sub_full || done - bstart >= prm.block_span || last_seg) {
            // ---- sub-block boundary: join the open block, or close it and start the next one here ----
            uint32_t bmid = done + e;
            if (bmid > pend) bmid = pend;
            const uint32_t nh = ntok - nH;
            zmi_wave_sync();
            if (nH != 0u && enc_split_pays(S, nH, nh, prm.split_hdr_bits)) {
                __threadfence_block();  // token stores of other lanes must be visible to the encode pass
                zmi_wave_sync();
                W = enc_flush_block(S, W, tokbuf + tok0, nH, src, bstart, bendH, 0u, prm.strategy);
                tok0 += nH;             // the sub-block's tokens stay where they are and become the open block
                ntok -= nH;
                nH = 0;
                bstart = bendH;
                for (uint32_t i = lane; i < ENC_NL; i += 64u) S->lfreq[i] = 0;
                if (lane < ENC_ND) S->dfreq[lane] = 0;
                zmi_wave_sync();
            }
            for (uint32_t i = lane; i < ENC_NL; i += 64u) { S->lfreq[i] += S->lfreq2[i]; S->lfreq2[i] = 0; }
            if (lane < ENC_ND) { S->dfreq[lane] += S->dfreq2[lane]; S->dfreq2[lane] = 0; }
            nH = ntok;
            bendH = bmid;
            zmi_wave_sync();
            if (!CP && !last_seg) {
                enc_far_limits(S, S->lfreq, S->dfreq);   // the open block's statistics price the next sub-block's short matches
                far4 = zmi_uniform(S->misc[M_FAR4]); far5 = zmi_uniform(S->misc[M_FAR5]); far6 = zmi_uniform(S->misc[M_FAR6]);
            }
            if (last_seg || bendH - bstart >= prm.block_span) {
                const uint32_t is_final = (last_seg && is_last) ? 1u : 0u;  // is_last implies last piece
                __threadfence_block();
                zmi_wave_sync();
                W = enc_flush_block(S, W, tokbuf + tok0, nH, src, bstart, bendH, is_final, prm.strategy);
                bstart = bendH;
                tok0 = done;
                ntok = 0;
                nH = 0;
                for (uint32_t i = lane; i < ENC_NL; i += 64u) S->lfreq[i] = 0;
                if (lane < ENC_ND) S->dfreq[lane] = 0;
                zmi_wave_sync();
            }
        }
    }

    // end of piece: the last piece appends the wrapper trailer, every other piece an empty stored
    // block (00 00 FF FF after padding -- the Z_SYNC_FLUSH marker, zlib-rs/src/deflate.rs:2733-2738)
    // so that the next piece starts on a byte boundary and the pieces can simply be concatenated
    enc_flush(S, W);
    if (lane == 0) {
        uint32_t rel = W.rel;
        if (!is_last) {
            enc_put0(S, rel, 0u, 3u);
            rel = (rel + 7u) & ~7u;
            enc_put0(S, rel, 0xFFFF0000u, 32u);
        } else {
            rel = (rel + 7u) & ~7u;
            if (prm.wrap == 1u) {
                uint32_t a = adler[s];
                uint32_t be = (a >> 24) | ((a >> 8) & 0xFF00u) | ((a << 8) & 0xFF0000u) | (a << 24);
                enc_put0(S, rel, be, 32u);
            } else if (prm.wrap == 2u) {
                enc_put0(S, rel, crc[s], 32u);
                enc_put0(S, rel, n, 32u);
            }
        }
        S->misc[M_REL] = rel;
    }
    zmi_wave_sync();
    W.rel = S->misc[M_REL];
    enc_flush(S, W);
    uint32_t tail = W.rel >> 3;  // 0..3 bytes left in stg[0]
    uint32_t total = W.wbase * 4u + tail;
    if (total > region_stride) W.err = 1u;
    if (lane == 0) {
        if (!W.err) {
            uint8_t* ob = (uint8_t*)(W.outw + W.wbase);
            uint32_t v = S->stg[0];
            for (uint32_t j = 0; j < tail; ++j) ob[j] = (uint8_t)(v >> (8u * j));
        }
        piece_len[(uint64_t)s * pieces + piece] = W.err ? 0xFFFFFFFFu : total;
    }
}

// The cost-parse instantiation is compiled in a translation unit of its own (encode_cp.hip includes this file with ENC_CP_UNIT):
// the two kernels share their out-of-line helpers (enc_flush_block, enc_split_pays ...), and with both instantiations in one unit
// the lazy kernel's register allocation changed with them (4 VGPR spills instead of 1, level 1: 78.7 -> 83.4 ms per 16 Ki shards).
extern "C" int zmi_launch_encode_cp(const uint8_t* d_data, const uint64_t* d_off, const uint32_t* d_len, uint32_t first_shard,
                                    uint32_t n_shards, uint32_t* d_match, uint64_t match_stride, const uint32_t* d_adler,
                                    const uint32_t* d_crc, uint8_t* d_out, uint64_t out_stride, uint32_t pieces, uint32_t region,
                                    uint32_t* d_piece_len, const uint32_t* d_dec, uint64_t dec_stride, zmi_enc_params prm, hipStream_t stream);
#ifdef ENC_CP_UNIT
extern "C" int zmi_launch_encode_cp(const uint8_t* d_data, const uint64_t* d_off, const uint32_t* d_len, uint32_t first_shard,
                                    uint32_t n_shards, uint32_t* d_match, uint64_t match_stride, const uint32_t* d_adler,
                                    const uint32_t* d_crc, uint8_t* d_out, uint64_t out_stride, uint32_t pieces, uint32_t region,
                                    uint32_t* d_piece_len, const uint32_t* d_dec, uint64_t dec_stride, zmi_enc_params prm, hipStream_t stream) {
    ZMI_LAUNCH(zmi_encode_kernel_t<true>, dim3(n_shards * pieces), dim3(64), 0, stream, d_data, d_off, d_len, first_shard, d_match,
               match_stride, d_dec, dec_stride, d_adler, d_crc, d_out, out_stride, pieces, region, d_piece_len, prm);
    return 0;
}
#else
// Concatenate the pieces of every shard inside its output slot (piece r lives at r*region_stride and
// only ever moves towards lower addresses), then publish length and status.  One 256-thread
// workgroup per shard; 4 KiB blocks are loaded with aligned 16-byte reads, staged in LDS and
// written back as aligned dwords (the destination is byte-misaligned in general), so every block
// is one coalesced load + one coalesced store with all threads busy.
#define CMP_BLK 4096u
__global__ void __launch_bounds__(256) zmi_compact_kernel(uint8_t* __restrict__ out, uint64_t out_stride, uint32_t first_shard,
                                                          uint32_t pieces, uint32_t region_stride,
                                                          const uint32_t* __restrict__ piece_len,
                                                          uint32_t* __restrict__ out_len, int32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[CMP_BLK + 16];
    const uint32_t t = threadIdx.x;
    const uint32_t s = first_shard + blockIdx.x;
    uint8_t* slot = out + (uint64_t)s * out_stride;
    const uint32_t* pl = piece_len + (uint64_t)s * pieces;
    uint32_t cur = pl[0];
    bool err = cur == 0xFFFFFFFFu;
    for (uint32_t r = 1; r < pieces && !err; ++r) {
        const uint32_t l = pl[r];
        if (l == 0xFFFFFFFFu) { err = true; break; }
        const uint8_t* srcp = slot + (uint64_t)r * region_stride;  // 16-byte aligned
        uint8_t* dstp = slot + cur;
        for (uint32_t base = 0; base < l; base += CMP_BLK) {
            const uint32_t nblk = l - base < CMP_BLK ? l - base : CMP_BLK;
            if (t * 16u < nblk) *(uint4*)(stage + t * 16u) = *(const uint4*)(srcp + base + t * 16u);
            __syncthreads();
            uint8_t* A = dstp + base;
            uint32_t head = (4u - (uint32_t)((uintptr_t)A & 3u)) & 3u;
            if (head > nblk) head = nblk;
            const uint32_t ndw = (nblk - head) >> 2;
            uint32_t* A4 = (uint32_t*)(A + head);
            for (uint32_t k = t; k < ndw; k += 256u) A4[k] = zmi_load32u(stage, head + 4u * k);
            if (t == 0) {
                for (uint32_t j = 0; j < head; ++j) A[j] = stage[j];
                for (uint32_t j = head + 4u * ndw; j < nblk; ++j) A[j] = stage[j];
            }
            __syncthreads();
        }
        cur += l;
    }
    if (t == 0) {
        out_len[s] = err ? 0u : cur;
        status[s] = err ? ZMI_BUF_ERROR : ZMI_OK;
    }
}

extern "C" int zmi_launch_encode(const uint8_t* d_data, const uint64_t* d_off, const uint32_t* d_len, uint32_t first_shard,
                                 uint32_t n_shards, uint32_t* d_match, uint64_t match_stride, const uint32_t* d_adler,
                                 const uint32_t* d_crc, uint8_t* d_out, uint64_t out_stride, uint32_t out_cap,
                                 uint32_t* d_out_len, int32_t* d_status, uint32_t pieces, uint32_t* d_piece_len,
                                 const uint32_t* d_dec, uint64_t dec_stride, zmi_enc_params prm, hipStream_t stream) {
    if (n_shards == 0) return 0;
    if (prm.block_span < 64u) prm.block_span = 64u;
    prm.block_span &= ~63u;
    if (pieces < 1u) pieces = 1u;
    uint32_t region = (uint32_t)((out_cap / pieces) & ~15u);
    if (prm.cost_parse && d_dec)
        zmi_launch_encode_cp(d_data, d_off, d_len, first_shard, n_shards, d_match, match_stride, d_adler, d_crc, d_out, out_stride, pieces, region,
                             d_piece_len, d_dec, dec_stride, prm, stream);
    else
        ZMI_LAUNCH(zmi_encode_kernel_t<false>, dim3(n_shards * pieces), dim3(64), 0, stream, d_data, d_off, d_len, first_shard, d_match,
                   match_stride, d_dec, dec_stride, d_adler, d_crc, d_out, out_stride, pieces, region, d_piece_len, prm);
    ZMI_LAUNCH(zmi_compact_kernel, dim3(n_shards), dim3(256), 0, stream, d_out, out_stride, first_shard, pieces, region,
               d_piece_len, d_out_len, d_status);
    return 0;
}
#endif
