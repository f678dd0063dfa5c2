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
// profiles/r05_cost_parse_phase_profile.txt).  Here: 10.4 KiB of LDS, and the encoder only reads two bits per position.
// Bound: instruction issue (~70 VALU + 6 LDS gathers per 64 positions); HBM: 4 B read + 0.25 B written per position.
#include "zmi_device.h"
#include "zmi_kernels.h"

#define PAR_NL 288u
#define PAR_ND 32u
#define PAR_BIAS 2048
#define PAR_SPAN 64u      // chunks per wave in a batch of thousands of shards (256 KiB); a small launch takes shorter spans (zmi_launch_parse)

struct ParShared {
    uint32_t lfreq[PAR_NL];   // literal / length symbols of the tokens chosen so far (this wave's span)
    uint32_t dfreq[PAR_ND];
    uint16_t psym[PAR_NL];    // prices in quarter bits: a literal / length symbol,
    uint32_t plen2[256];      // a match length l (index l - 3): its symbol + its extra bits, low half; the same for l - 1 in the high half
    uint16_t pdist[PAR_ND];   // a distance code + its extra bits
    uint16_t cost[64u * 66u]; // the chunk's cells: row = strip, pitch 33 dwords (the lanes of a step hit 32 different banks)
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
static __device__ __forceinline__ uint32_t par_lext(uint32_t idx) { return (idx < 8u || idx == 28u) ? 0u : (idx >> 2) - 1u; }
static __device__ __forceinline__ uint32_t par_dext(uint32_t idx) { return idx < 4u ? 0u : (idx >> 1) - 1u; }
static __device__ __forceinline__ uint32_t par_static_llen(uint32_t s) { return s < 144u ? 8u : (s < 256u ? 9u : (s < 280u ? 7u : 8u)); }

// all lanes: the price tables from the counts (`dynamic`), or from the static code (first chunk, Z_FIXED, next to nothing counted)
static __device__ __noinline__ void par_prices(ParShared* S, bool dynamic) {
    const uint32_t lane = zmi_lane();
    if (dynamic) {
        uint32_t nl = 0;
        for (uint32_t i = lane; i < PAR_NL; i += 64u) nl += S->lfreq[i];
        nl = zmi_wave_sum(nl);
        const uint32_t nd = zmi_wave_sum(lane < 30u ? S->dfreq[lane] : 0u);
        const float lgl = __log2f((float)nl + 72.f), lgd = __log2f((float)nd + 8.f);   // (a quarter count for every symbol: unseen is dear, not impossible)
        for (uint32_t i = lane; i < PAR_NL; i += 64u) {
            const float b = 4.f * (lgl - __log2f((float)S->lfreq[i] + 0.25f)) + 0.5f;
            S->psym[i] = (uint16_t)(b < 4.f ? 4u : (b > 60.f ? 60u : (uint32_t)b));
        }
        if (lane < PAR_ND) {
            const float b = 4.f * (lgd - __log2f((float)(lane < 30u ? S->dfreq[lane] : 0u) + 0.25f)) + 0.5f;
            S->pdist[lane] = (uint16_t)((b < 4.f ? 4u : (b > 60.f ? 60u : (uint32_t)b)) + 4u * par_dext(lane < 30u ? lane : 0u));
        }
    } else {
        for (uint32_t i = lane; i < PAR_NL; i += 64u) S->psym[i] = (uint16_t)(4u * par_static_llen(i));
        if (lane < PAR_ND) S->pdist[lane] = (uint16_t)(4u * (5u + par_dext(lane < 30u ? lane : 0u)));
    }
    zmi_wave_sync();
    for (uint32_t i = lane; i < 256u; i += 64u) {
        const uint32_t li = par_len_idx(i + 3u), lj = par_len_idx(i > 0u ? i + 2u : 3u);
        S->plen2[i] = (uint32_t)(S->psym[257u + li] + 4u * par_lext(li)) | ((uint32_t)(S->psym[257u + lj] + 4u * par_lext(lj)) << 16);
    }
    zmi_wave_sync();
}

// Stage A of a position: everything that does not depend on the positions behind it -- the word's fields and the six gathers.
// `row` = this lane's row of cells.  Straight-line on purpose: selects, a 24-bit multiply, loads at clamped indices (written with
// `? :` on guarded loads and a 32-bit multiply the compiler made three divergent branches per position of it).
template <bool V> struct ParTag { static constexpr bool value = V; };
struct ParA {
    uint32_t plit, pd, len;
    uint32_t pl2, cw[2];
};
// (The piece-end rule -- a token ends with its piece, positions behind the shard's end are nothing -- is applied to the WORDS when a
// batch is loaded, and only in a batch that comes within a token's reach of such an end: par_clip_batch.  Stage A itself has no
// check left: five instructions per position less in fifteen chunks of sixteen.)
static __device__ __forceinline__ ParA par_stage_a(const ParShared* S, const uint16_t* row, uint32_t w, uint32_t k) {
    ParA a;
    const uint32_t len = (w >> 8) & 0x1FFu;
    a.plit = S->psym[w & 0xFFu];
    a.len = len;
    a.pd = S->pdist[par_dist_idx((w >> 17) + 1u)];
    a.pl2 = S->plen2[(len >= 4u ? len : 4u) - 3u];
#pragma unroll
    for (uint32_t c = 0; c < 2u; ++c) {
        const uint32_t l = len >= 4u + c ? len - c : 4u;
        const uint32_t tg = (k + l) & 0x3FFu;                // (k wraps for the two positions "in front of" a strip: any cell will do)
        a.cw[c] = row[tg < 63u ? tg : 63u];
    }
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

    for (uint32_t i = lane; i < PAR_NL; i += 64u) S->lfreq[i] = 0u;
    if (lane < PAR_ND) S->dfreq[lane] = 0u;
    zmi_wave_sync();
    uint32_t ntok = 0u;       // tokens counted so far
    uint32_t aq = 12u;        // running average cost of a byte, quarter bits (3 bits per byte before anything is known)
    const uint32_t c_end = (span + 1u) * span_chunks < nchunks ? (span + 1u) * span_chunks : nchunks;
    for (uint32_t ch = span * span_chunks; ch < c_end; ++ch) {
        par_prices(S, strategy != 4u && ntok >= 256u);
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
            ParA s0 = par_stage_a(S, row, wa[15], 63u);   // (targets behind the strip: no cell is read before its store)
            ParA s1 = par_stage_a(S, row, wa[14], 62u);
            // (the batch loop stays rolled and every position's code stays together: unrolled and left to the scheduler, the strip's
            // 64 positions were one region of 332 VGPRs)
#pragma unroll 1
            for (int b = 3; b >= 0; --b) {
                if (b >= 2) {
                    load16(wc, sb + 16u * (uint32_t)(b - 2));
                    if (near_end) par_clip_batch(wc, sb + 16u * (uint32_t)(b - 2), pe);
                }
                uint32_t dcur = 0u;
#pragma unroll
                for (int t = 15; t >= 0; --t) {
                    const uint32_t k = 16u * (uint32_t)b + (uint32_t)t;
                    // stage A of position k - 2 (the last two steps of a strip: positions in front of it, harmless and unused)
                    const ParA s2 = par_stage_a(S, row, t >= 2 ? wa[t >= 2 ? t - 2 : 0] : wb[t + 14], k - 2u);
                    // stage B of position k: the three alternatives as (biased cost << 2 | code), one three-way minimum (costs are biased
                    // by PAR_BIAS in the cells, so everything is unsigned; an alternative that does not exist is all ones)
                    const uint32_t st_next = (uint32_t)(cnext + PAR_BIAS);
                    uint32_t alt[2];
#pragma unroll
                    for (uint32_t c = 0; c < 2u; ++c) {
                        const bool ok = s0.len >= 4u + c;
                        const uint32_t l = ok ? s0.len - c : 4u;
                        const uint32_t tg = k + l;
                        const bool inside = tg < 64u;
                        const uint32_t over = __umul24(tg & 0x1FFu, aq) - 64u * aq;   // (tg - 64) * aq, full-rate; only used behind the strip
                        uint32_t behind = (uint32_t)PAR_BIAS - (over < (uint32_t)PAR_BIAS ? over : (uint32_t)PAR_BIAS);
#ifndef ZMI_EMU
                        asm volatile("" : "+v"(behind));   // (computed for every lane and selected: left alone the compiler branches around these instructions)
#endif
                        const uint32_t cc = inside ? s0.cw[c] : behind;
                        const uint32_t tot = (c ? s0.pl2 >> 16 : s0.pl2 & 0xFFFFu) + s0.pd + cc;
                        alt[c] = ok ? ((tot << 2) | (c + 1u)) : 0xFFFFFFFFu;
                    }
                    uint32_t m = (s0.plit + st_next) << 2;
                    m = m < alt[0] ? m : alt[0];
                    m = m < alt[1] ? m : alt[1];
                    const uint32_t code = m & 3u;
                    uint32_t st = m >> 2;
                    st = st > 16383u ? 16383u : st;
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
        }
        // statistics for the next chunk's prices: the tokens of every strip, walked from its first position (a masked scan over the
        // 64 positions: `nxt` is where the next token starts; no pointer chase, the words are read again in ascending order)
        // (every second chunk: the prices of chunks 2 i + 1 and 2 i + 2 come from the counts up to chunk 2 i -- half the scans for
        // -0.04 ... -0.14 % of ratio, tools/parse_lab.py LAB_EVERY)
        if (ch + 1u < c_end && ((ch - span * span_chunks) & 1u) == 0u) {
            uint32_t nxt = 0u, cnt = 0u;
#pragma unroll 1
            for (uint32_t b = 0; b < 4u; ++b) {
                load16(wa, sb + 16u * b);
                const uint32_t db = b == 0u ? d0 : (b == 1u ? d1 : (b == 2u ? d2 : d3));
#pragma unroll
                for (uint32_t t = 0; t < 16u; ++t) {
                    const uint32_t k = 16u * b + t, pos = sb + k;
                    const uint32_t w = wa[t];
                    const uint32_t code = (db >> (2u * t)) & 3u;
                    const uint32_t room = pos < pe ? pe - pos : 0u;
                    uint32_t l = (w >> 8) & 0x1FFu;
                    l = l < room ? l : room;
                    const uint32_t step = code ? l + 1u - code : 1u;
                    if ((k == nxt) & (pos < pe)) {   // (one region, two atomics: a literal counts its distance in slot 31, which no price reads)
                        nxt = k + step;
                        ++cnt;
                        atomicAdd(&S->lfreq[code ? 257u + par_len_idx(step) : (w & 0xFFu)], 1u);
                        atomicAdd(&S->dfreq[code ? par_dist_idx((w >> 17) + 1u) : 31u], 1u);
                    }
                }
            }
            ntok += zmi_wave_sum(cnt);
        }
        zmi_wave_sync();
    }
}

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
