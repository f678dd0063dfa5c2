// parse.hip -- the cost parse (levels 3-9): which of the matches found by lz77.hip become tokens is decided by PRICE.
//
// What the reference does here: deflate_medium finds a match, looks at the next position and lets the longer one push the other
// back (`fizzle_matches`, zlib-rs/src/deflate/algorithm/medium.rs:264-331); deflate_slow defers by one position
// (slow.rs:12-161).  Both are local rules over one or two candidates, because the reference only searches where its parse
// stands.  Here every position's best match is known before the parse starts, so the parse can be the shortest path it really is:
//     cost[i] = min(price(literal i) + cost[i + 1],  price(l, dist_i) + cost[i + l]  for l in {len_i, len_i - 1})
// evaluated backwards, with prices from the symbol statistics of the parse itself.  Measured on the matches of lz77.hip at
// budget 4 against the three-deep lazy rule this engine used until round 4 (tools/parse_lab.py): lcet10.txt +2.5 %, XML-like
// +2.1 %, exe-like +1.4 %, record data -0.8 % (a feedback effect of pricing from one's own statistics; anchored prices would
// give +0.2 % there, and nothing cheap anchors them); on the MI355X lcet10.txt level 6 2.849 -> 2.920 (the reference: 2.929),
// level 9 2.901 -> 2.962 (2.956), the benchmark mix 2.253 -> 2.266.
//
// MI355X form (not the reference's loop): a CHUNK is 4096 positions = 64 STRIPS of 64; lane j owns strip j and runs the
// recurrence over it serially, all 64 lanes at once -- no shuffles, no divergence, one 16-bit LDS cell (the cost, quarter bits,
// relative to the end of the strip) per position, and the strip's 64 decisions in four registers (2 bits each: 0 literal, 1 the
// match, 2 the match one byte shorter).  A target behind the end of the strip is priced by extrapolation (the running average
// cost of a byte): strips are decided independently and the token chain later runs through them as it falls -- the error stays
// in the last tokens of a strip (strips of 64 / 128 / 256 / one per piece: +2.61 / +2.84 / +2.95 / +3.0 % on lcet10.txt).
// The gathers of a position (two table reads, two cells, two prices) are issued two positions ahead of its arithmetic
// (stage A / stage B): a target lies at least three positions behind, so its cell has been stored by then.
// Prices: static code for a wave's first chunk, afterwards -log2 of the symbols' shares in the tokens of the wave's own parse
// so far -- counted by walking every strip from its first position (the true chain enters a strip a few bytes in; for the
// statistics that makes no difference: tools/parse_lab.py LAB_STAT0).  A wave covers up to 64 chunks (256 KiB) of one shard.
// Why a kernel of its own: inside the encoder the chunk's cells cost 16 KiB of LDS and took it from 23 to 7 waves per CU -- its
// emission and token passes live on resident waves and ran 3x slower (326 against 102 ms per 16 Ki shards,
// profiles/r05_cost_parse_phase_profile.txt).  Here: 12.8 KiB of LDS, and the encoder only reads two bits per position.
// Bound: first VALU issue (55 VALU per position and lane: 34.9 ms per 16 Ki shards, every SIMD cycle taken -- profiles/
// r05_issue.json of the first form), now, at 22 VALU, the LDS pipe and its latency (5 gathers + 1 store per position; with 7
// instead of 12 waves per CU +22 %, profiles/r05_parse_tables.txt): ~24 ms.  HBM: 4 B read + 0.25 B written per position.
#include "zmi_device.h"
#include "zmi_kernels.h"

#define PAR_NL 288u
#define PAR_ND 32u
#define PAR_BIAS 2048
#define PAR_SPAN 64u      // chunks per wave in a batch of thousands of shards (256 KiB); a small launch takes shorter spans (zmi_launch_parse)

#define PAR_NOMATCH 0xFFFFu   // the price of a match that does not exist: above anything a literal can cost
#define PAR_EXT 260u          // cells behind a strip: a token of 258 from its last position ends 257 behind it
struct ParShared {
    uint32_t lfreq[256u + 260u];   // the tokens chosen so far (this wave's span): literals by value, and behind them (256 + l) matches by
                              // LENGTH (folded into the 29 length symbols when prices are made: counting costs no arithmetic),
    uint32_t dfreq[2u * PAR_ND];   // and by the slot of their distance (par_dq).  A literal counts into slot 32 + its lane modulo 32, which
                              // nothing reads: the scan has one region and two atomics per token, and 64 literals at one
                              // address would be 64 atomics one after the other (random data: 6.2 against 3.7 ms per 2048 shards)
    uint32_t lsc[32];         // the length symbols' counts (par_prices)
    uint16_t psym[PAR_NL];    // prices in quarter bits: a literal / length symbol,
    uint32_t plen2[260];      // a match of the length l the word holds (index l): its symbol + its extra bits in the low half, the same
                              // for l - 1 in the high half; PAR_NOMATCH where the parse takes no match (l < 4, l - 1 < 4)
    uint8_t pdq[PAR_ND];      // a distance code + its extra bits (<= 60 + 4 * 13), at the slot par_dq() finds from the match word
    uint16_t cost[64u * 66u]; // the chunk's cells: row = strip, pitch 33 dwords (the lanes of a step hit 32 different banks)
    uint16_t ext[PAR_EXT];    // the cells BEHIND a strip, the same for every strip: j positions behind it cost j * aq less (par_ext)
};

#define PAR_PITCH 66u
static __device__ __forceinline__ uint32_t par_len_idx(uint32_t len) {
    const uint32_t l = len - 3u;
    const uint32_t k = 31u - (uint32_t)__clz(l | 4u);
    const uint32_t f = 4u * (k - 1u) + ((l >> (k - 2u)) & 3u);
    return l == 255u ? 28u : (l < 4u ? l : f);
}
static __device__ __forceinline__ uint32_t par_dist_idx(uint32_t dist) {
    const uint32_t d = dist - 1u;
    const uint32_t k = 31u - (uint32_t)__clz(d | 2u);
    const uint32_t f = 2u * k + ((d >> (k - 1u)) & 1u);
    return d < 2u ? d : f;
}
// the slot of a match word's distance in pdq[], in four instructions: 2 d + 1 (d = distance - 1, bits 17 ... 31 of the word) as a
// float has floor(log2 d) + 1 in its exponent and the bit of d below the leading one on top of its mantissa -- the two numbers a
// distance code is made of (deflate.rs d_code: 2 k + that bit).  Slot = code for codes 1 ... 29, slot 30 for code 0 (exponent 127).
static __device__ __forceinline__ uint32_t par_dq(uint32_t w) {
    return (__builtin_bit_cast(uint32_t, (float)((w >> 16) | 1u)) >> 22) & 31u;
}
static __device__ __forceinline__ uint32_t par_lext(uint32_t idx) { return (idx < 8u || idx == 28u) ? 0u : (idx >> 2) - 1u; }
static __device__ __forceinline__ uint32_t par_dext(uint32_t idx) { return idx < 4u ? 0u : (idx >> 1) - 1u; }
static __device__ __forceinline__ uint32_t par_static_llen(uint32_t s) { return s < 144u ? 8u : (s < 256u ? 9u : (s < 280u ? 7u : 8u)); }

// all lanes: the price tables from the counts (`allow` and 256 tokens counted), or from the static code (first chunk, Z_FIXED,
// next to nothing counted)
static __device__ __noinline__ void par_prices(ParShared* S, bool allow) {
    const uint32_t lane = zmi_lane();
    uint32_t nl = 0;
    if (allow) {
        if (lane < 32u) S->lsc[lane] = 0u;
        zmi_wave_sync();
        for (uint32_t i = lane; i < 256u; i += 64u) nl += S->lfreq[i];
        for (uint32_t l = lane; l < 259u; l += 64u) {
            const uint32_t c = S->lfreq[256u + l];
            nl += c;
            if (c) atomicAdd(&S->lsc[par_len_idx(l >= 3u ? l : 3u)], c);
        }
        nl = zmi_wave_sum(nl);
        zmi_wave_sync();
    }
    if (nl >= 256u) {
        const uint32_t nd = zmi_wave_sum(lane >= 1u && lane <= 30u ? S->dfreq[lane] : 0u);
        const float lgl = __log2f((float)nl + 72.f), lgd = __log2f((float)nd + 8.f);   // (a quarter count for every symbol: unseen is dear, not impossible)
        for (uint32_t i = lane; i < PAR_NL; i += 64u) {
            const uint32_t cnt = i < 256u ? S->lfreq[i] : (i >= 257u && i < 286u ? S->lsc[i - 257u] : 0u);
            const float b = 4.f * (lgl - __log2f((float)cnt + 0.25f)) + 0.5f;
            S->psym[i] = (uint16_t)(b < 4.f ? 4u : (b > 60.f ? 60u : (uint32_t)b));
        }
        if (lane < 30u) {
            const uint32_t slot = lane ? lane : 30u;
            const float b = 4.f * (lgd - __log2f((float)S->dfreq[slot] + 0.25f)) + 0.5f;
            S->pdq[slot] = (uint8_t)((b < 4.f ? 4u : (b > 60.f ? 60u : (uint32_t)b)) + 4u * par_dext(lane));
        }
    } else {
        for (uint32_t i = lane; i < PAR_NL; i += 64u) S->psym[i] = (uint16_t)(4u * par_static_llen(i));
        if (lane < 30u) S->pdq[lane ? lane : 30u] = (uint8_t)(4u * (5u + par_dext(lane)));
    }
    zmi_wave_sync();
    for (uint32_t l = lane; l < 259u; l += 64u) {
        const uint32_t li = par_len_idx(l >= 4u ? l : 4u), lj = par_len_idx(l >= 5u ? l - 1u : 4u);
        const uint32_t lo = l >= 4u ? (uint32_t)(S->psym[257u + li] + 4u * par_lext(li)) : PAR_NOMATCH;
        const uint32_t hi = l >= 5u ? (uint32_t)(S->psym[257u + lj] + 4u * par_lext(lj)) : PAR_NOMATCH;
        S->plen2[l] = lo | (hi << 16);
    }
    zmi_wave_sync();
}
// all lanes: the cells behind a strip for the running average `aq` (quarter bits per byte): costs are relative to the end of the
// strip and biased by PAR_BIAS, so j positions behind it stand at PAR_BIAS - j * aq (not below zero)
static __device__ __forceinline__ void par_ext(ParShared* S, uint32_t aq) {
    for (uint32_t j = zmi_lane(); j < PAR_EXT; j += 64u) {
        const uint32_t over = j * aq;
        S->ext[j] = (uint16_t)((uint32_t)PAR_BIAS - (over < (uint32_t)PAR_BIAS ? over : (uint32_t)PAR_BIAS));
    }
    zmi_wave_sync();
}

// Stage A of a position: everything that does not depend on the positions behind it -- the word's fields and the five gathers.
// Straight-line on purpose, and short: the parse is bound by VALU issue (profiles/r05_issue.json: 21.7 G instructions x 4 cycles =
// every SIMD cycle of its 34.9 ms), so what can be a table is a table --
//   * "is there a match of this length" is the PRICE (PAR_NOMATCH in plen2): no compare, no select in stage B;
//   * the cost of a target is ONE gather whether the target lies in the strip (the lane's row) or behind it (ext[], shared by all
//     lanes): the two byte offsets differ by a per-batch constant, the select picks the base (until round 5 the extrapolation was
//     arithmetic per candidate: mask, multiply, two subtractions, a minimum, two selects -- 22 of a position's 55 instructions).
// `rowb` / `extb`: byte offsets (inside ParShared) of the lane's cell and of the ext[] entry that stand for position 16 b of the
// strip, both less PAR_AOFF so that every position of a batch (p = -2 ... 15) reads at a non-negative immediate; `lim` = 64 - 16 b.
template <bool V> struct ParTag { static constexpr bool value = V; };
struct ParA {
    uint32_t plit, pd;
    uint32_t pl2, cw[2];
};
#define PAR_AOFF 6u
// (The piece-end rule -- a token ends with its piece, positions behind the shard's end are nothing -- is applied to the WORDS when a
// batch is loaded, and only in a batch that comes within a token's reach of such an end: par_clip_batch.  Stage A itself has no
// check.)
static __device__ __forceinline__ ParA par_stage_a(const ParShared* S, uint32_t rowb, uint32_t extb, uint32_t lim, uint32_t w, int p) {
    ParA a;
    const uint32_t len = (w >> 8) & 0x1FFu;                 // (below 4: no match -- any cell will do, the price says no)
#ifdef ZMI_EMU
    if (len > 258u) abort();   // (plen2[] has 260 entries and lfreq[] 516 -- lz77.hip never writes a longer match; the CPU build checks)
#endif
    a.plit = S->psym[w & 0xFFu];
    a.pd = S->pdq[par_dq(w)];
    a.pl2 = S->plen2[len];
    // targets k + len <= 64 are read from the lane's row -- cell 64 of every row holds what ext[0] holds, the bias -- so the one
    // compare serves both candidates: the shorter one's target is the cell in front (k + len = 64: cell 63, the strip's last)
    const uint32_t off = (len <= lim - (uint32_t)p ? rowb : extb) + 2u * len + (uint32_t)(2 * p + (int)PAR_AOFF);
    const char* const base = (const char*)S;
    a.cw[0] = *(const uint16_t*)(base + off);
    a.cw[1] = *(const uint16_t*)(base + off - 2u);
    return a;
}
// words of positions first ... first + 15 of a strip whose piece ends at pe: lengths clipped to the room left, positions at or
// behind pe become literals of symbol 0 (their cost is a constant per position: it shifts every cost in front of them alike)
static __device__ __forceinline__ void par_clip_batch(uint32_t (&w)[16], uint32_t first, uint32_t pe) {
#pragma unroll
    for (uint32_t q = 0; q < 16u; ++q) {
        const uint32_t pos = first + q;
        const uint32_t room = pos < pe ? pe - pos : 0u;
        const uint32_t len = (w[q] >> 8) & 0x1FFu;
        const uint32_t l = len < room ? len : room;
        w[q] = pos < pe ? ((w[q] & ~(0x1FFu << 8)) | (l << 8)) : 0u;
    }
}

__global__ void __launch_bounds__(64) zmi_parse_kernel(const uint32_t* __restrict__ len, uint32_t first_shard, uint32_t n_shards,
                                                       const uint32_t* __restrict__ match, uint64_t match_stride,
                                                       uint32_t* __restrict__ dec, uint64_t dec_stride, uint32_t pieces, uint32_t strategy,
                                                       uint32_t span_chunks) {
    __shared__ ParShared Sh;
    ParShared* S = &Sh;
    const uint32_t lane = zmi_lane();
    const uint32_t local = zmi_xcd_spread(blockIdx.x % n_shards, n_shards);   // shard inside this launch group
    const uint32_t span = blockIdx.x / n_shards;
    const uint32_t n = len[first_shard + local];
    const uint32_t nchunks = (n + 4095u) >> 12;
    if (span * span_chunks >= nchunks) return;
    const uint32_t* words = match + (uint64_t)local * match_stride;
    uint32_t* dout = dec + (uint64_t)local * dec_stride;
    // the encoder's pieces: a token ends with its piece
    uint32_t psize = ((n + pieces - 1u) / pieces + 63u) & ~63u;
    if (psize == 0u) psize = 64u;
    const uint32_t last4 = (n - 1u) & ~3u;               // (n > 0: a shard without positions has no chunks)

    for (uint32_t i = lane; i < 256u + 260u; i += 64u) S->lfreq[i] = 0u;
    S->dfreq[lane] = 0u;
    zmi_wave_sync();
    uint32_t aq = 12u;        // running average cost of a byte, quarter bits (3 bits per byte before anything is known)
    S->cost[PAR_PITCH * lane + 64u] = (uint16_t)PAR_BIAS;   // (cell 64 of a row = ext[0], see par_stage_a; the recurrence never writes it)
    par_ext(S, aq);
    bool counted = true;      // the counts have changed since the prices were made (the first chunk: nothing is priced yet)
    const uint32_t c_end = (span + 1u) * span_chunks < nchunks ? (span + 1u) * span_chunks : nchunks;
    const uint32_t row0 = (uint32_t)offsetof(ParShared, cost) + 2u * PAR_PITCH * lane - PAR_AOFF;   // byte offsets, see par_stage_a
    const uint32_t ext0 = (uint32_t)offsetof(ParShared, ext) - 128u - PAR_AOFF;
    for (uint32_t ch = span * span_chunks; ch < c_end; ++ch) {
        if (counted) par_prices(S, strategy != 4u);
        counted = false;
        const uint32_t sb = (ch << 12) + 64u * lane;     // this lane's strip
        uint32_t pe = (sb / psize + 1u) * psize;         // the end of the piece the strip lies in (psize is a multiple of 64)
        pe = pe < n ? pe : n;
        // The strip's match words, 16 positions (64 bytes) per lane and batch, two batches ahead of the arithmetic: the 64 lanes
        // read 64 different lines of 128 bytes, and the two halves of a line are two consecutive batches
        uint32_t wa[16], wb[16], wc[16];                 // batch b, b - 1, b - 2
        auto load16 = [&](uint32_t (&dst)[16], uint32_t first) {
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) {
                const uint32_t a = first + 4u * q;
                const uint4 v = *(const uint4*)(words + (a < last4 ? a : last4));
                dst[4u * q] = v.x; dst[4u * q + 1u] = v.y; dst[4u * q + 2u] = v.z; dst[4u * q + 3u] = v.w;
            }
        };
        const bool near_end = __ballot(pe < sb + 64u + 258u) != 0ull;   // (wave-uniform: some strip of the chunk is within reach of an end)
        load16(wa, sb + 48u);
        load16(wb, sb + 32u);
        if (near_end) { par_clip_batch(wa, sb + 48u, pe); par_clip_batch(wb, sb + 32u, pe); }
        int32_t cnext = 0;                                // cost of position k + 1, relative to the end of the strip
        uint32_t d0 = 0u, d1 = 0u, d2 = 0u, d3 = 0u;      // the strip's decisions, 16 positions per register
        uint16_t* const row = S->cost + PAR_PITCH * lane;
        auto strip = [&](auto clip_tag) {
            ParA s0 = par_stage_a(S, row0 + 96u, ext0 + 96u, 16u, wa[15], 15);   // (targets behind the strip: no cell is read before its store)
            ParA s1 = par_stage_a(S, row0 + 96u, ext0 + 96u, 16u, wa[14], 14);
            // (the batch loop stays rolled and every position's code stays together: unrolled and left to the scheduler, the strip's
            // 64 positions were one region of 332 VGPRs)
#pragma unroll 1
            for (int b = 3; b >= 0; --b) {
                if (b >= 2) {
                    load16(wc, sb + 16u * (uint32_t)(b - 2));
                    if (near_end) par_clip_batch(wc, sb + 16u * (uint32_t)(b - 2), pe);
                }
                uint32_t dcur = 0u;
                const uint32_t rowb = row0 + 32u * (uint32_t)b, extb = ext0 + 32u * (uint32_t)b, lim = 64u - 16u * (uint32_t)b;
#pragma unroll
                for (int t = 15; t >= 0; --t) {
                    const uint32_t k = 16u * (uint32_t)b + (uint32_t)t;
                    // stage A of position k - 2 (the last two steps of a strip: positions in front of it, harmless and unused)
                    const ParA s2 = par_stage_a(S, rowb, extb, lim, t >= 2 ? wa[t >= 2 ? t - 2 : 0] : wb[t + 14], t - 2);
                    // stage B of position k: the three alternatives as (biased cost << 2 | code), one three-way minimum (costs are biased
                    // by PAR_BIAS in the cells, so everything is unsigned; an alternative that does not exist carries PAR_NOMATCH)
                    const uint32_t st_next = (uint32_t)(cnext + PAR_BIAS);
                    const uint32_t alt0 = (((s0.pl2 & 0xFFFFu) + s0.pd + s0.cw[0]) << 2) | 1u;
                    const uint32_t alt1 = (((s0.pl2 >> 16) + s0.pd + s0.cw[1]) << 2) | 2u;
                    uint32_t m = (s0.plit + st_next) << 2;
                    m = m < alt0 ? m : alt0;
                    m = m < alt1 ? m : alt1;
                    const uint32_t code = m & 3u;
                    const uint32_t st = m >> 2;           // (at most PAR_BIAS + 64 literals of 15 bits + PAR_BIAS: the literal is always there)
                    row[k] = (uint16_t)st;
                    zmi_wave_order();   // (the gathers of the stage A below this point may read this cell: the store stays in front of them)
                    cnext = (int32_t)st - PAR_BIAS;
                    dcur |= code << (2u * (uint32_t)t);
                    s0 = s1;
                    s1 = s2;
                    zmi_sched_fence();
                }
                if (b == 3) d3 = dcur; else if (b == 2) d2 = dcur; else if (b == 1) d1 = dcur; else d0 = dcur;
#pragma unroll
                for (uint32_t q = 0; q < 16u; ++q) { wa[q] = wb[q]; wb[q] = wc[q]; }
            }
        };
        strip(ParTag<true>{});
        // the decisions leave: 16 bytes per strip, 1 KiB per chunk, in position order (segment g of the shard: bytes 16 g ...)
        if ((uint64_t)(sb >> 4) + 4u <= dec_stride) {
            uint4 v;
            v.x = d0; v.y = d1; v.z = d2; v.w = d3;
            *(uint4*)(dout + (sb >> 4)) = v;
        }
        // the running average cost of a byte, for the next chunk's extrapolations: what this chunk's strips cost from their starts
        {
            const uint32_t c0 = ch << 12;
            const uint32_t npos = n - c0 < 4096u ? n - c0 : 4096u;
            const int32_t total = (int32_t)zmi_wave_sum((uint32_t)cnext);
            const int32_t a = total > 0 ? total / (int32_t)npos : 0;
            aq = a < 1 ? 1u : (uint32_t)a;
            par_ext(S, aq);
        }
        // statistics for the next chunk's prices: the tokens of every strip, walked from its first position (a masked scan over the
        // 64 positions: `nxt` is where the next token starts; no pointer chase, the words are read again in ascending order).
        // Counted as they are -- a literal by its value, a match by its length and by the slot of its distance -- so that a position
        // costs some twenty instructions and no branch; par_prices folds the lengths into symbols.
        // (until round 5 every second chunk: half the scans for -0.04 ... -0.14 % of ratio, tools/parse_lab.py LAB_EVERY)
        // (round 6: the first four chunks of a span, then every fourth -- the counts are cumulative over the span, so a later chunk
        // moves the prices less: 19 scans per 64 chunks instead of 32, benchmark mix 2.2685 -> 2.2681, lcet10.txt 2.9208 -> 2.9205 on the
        // CPU emulator; every fourth alone: 2.2674; the first 8 every second, then every eighth: 2.2676 -- profiles/r06_parse_scan_rule.txt)
#ifndef PAR_SCAN_RULE
#define PAR_SCAN_RULE(rel) ((rel) < 4u || ((rel) & 3u) == 0u)
#endif
        if (ch + 1u < c_end && PAR_SCAN_RULE(ch - span * span_chunks)) {
            uint32_t nxt = 0u;
            const uint32_t kend = pe > sb ? (pe - sb < 64u ? pe - sb : 64u) : 0u;   // positions of the strip inside its piece
#pragma unroll 1
            for (uint32_t b = 0; b < 4u; ++b) {
                load16(wa, sb + 16u * b);
                if (near_end) par_clip_batch(wa, sb + 16u * b, pe);   // (the words the recurrence saw)
                const uint32_t db = b == 0u ? d0 : (b == 1u ? d1 : (b == 2u ? d2 : d3));
#pragma unroll
                for (uint32_t t = 0; t < 16u; ++t) {
                    const uint32_t k = 16u * b + t;
                    const uint32_t w = wa[t];
                    const uint32_t code = (db >> (2u * t)) & 3u;
                    const uint32_t more = code ? ((w >> 8) & 0x1FFu) - code : 0u;   // the token's length - 1
                    const uint32_t li = code ? 256u + 1u + more : (w & 0xFFu);
                    const uint32_t di = code ? par_dq(w) : 32u + (lane & 31u);
                    if ((k == nxt) & (k < kend)) {
                        nxt = k + 1u + more;
                        atomicAdd(&S->lfreq[li], 1u);
                        atomicAdd(&S->dfreq[di], 1u);
                    }
                }
            }
            counted = true;
        }
        zmi_wave_sync();
    }
}

#ifdef ZMI_EMU
// test hook (CPU build only): the slot par_dq() gives the distance in a match word, and the slot its distance code's price is kept in
extern "C" uint32_t zmi_emu_par_dq(uint32_t word) { return par_dq(word); }
extern "C" uint32_t zmi_emu_par_slot_of_code(uint32_t code) { return code ? code : 30u; }
extern "C" uint32_t zmi_emu_par_dist_code(uint32_t dist) { return par_dist_idx(dist); }
#endif
extern "C" int zmi_launch_parse(const uint32_t* d_len, uint32_t first_shard, uint32_t n_shards, uint32_t max_len, const uint32_t* d_match,
                                uint64_t match_stride, uint32_t* d_dec, uint64_t dec_stride, uint32_t pieces, uint32_t strategy,
                                uint32_t span_chunks, hipStream_t stream) {
    if (n_shards == 0 || max_len == 0) return 0;
    const uint32_t nchunks = (max_len + 4095u) >> 12;
    // a wave's span: 64 chunks in a batch (its prices adapt over 256 KiB); the segments of one deflate() call take 4 so that the
    // chip sees a few hundred waves (a 4 MiB call: 275 -> 70 us; the first chunk of every span is priced with the static code).
    // The caller decides (zmi_api.hip): what a shard compresses to must not depend on how many shards its launch holds.
    if (span_chunks < 1u) span_chunks = 1u;
    if (span_chunks > PAR_SPAN) span_chunks = PAR_SPAN;
    const uint32_t spans = (nchunks + span_chunks - 1u) / span_chunks;
    ZMI_LAUNCH(zmi_parse_kernel, dim3(n_shards * spans), dim3(64), 0, stream, d_len, first_shard, n_shards, d_match, match_stride, d_dec,
               dec_stride, pieces < 1u ? 1u : pieces, strategy, span_chunks);
    return 0;
}
