// lz77.hip -- LZ77 longest-match search for every byte position of every shard (deflate stage 1).
//
// What the reference does here (CPU, one thread): per position, insert a 4-byte multiplicative hash
// into head[]/prev[] (zlib-rs/src/deflate/hash_calc.rs:30-59) and walk up to max_chain links of the
// hash chain comparing against the 32 KiB sliding window (zlib-rs/src/deflate/longest_match.rs:15-346,
// compare256.rs), driven by the lazy parse loop of deflate/algorithm/medium.rs:12-178.
//
// MI355X design (not a translation):
//   * one 1024-thread workgroup per shard, the whole search state in LDS (152 KiB, 1 workgroup/CU):
//       win   32 KiB ring of input bytes (+32 mirrored bytes so an unaligned 16-byte read may straddle the wrap)
//       prev  32 Ki x u16 ring: DISTANCE to the previous position with the same hash (0 = end of chain) --
//             deltas instead of positions remove the reference's slide_hash pass (deflate/slide_hash.rs)
//       head  16 Ki x u16: last position+1 (mod 2^16) per bucket of the 6-byte hash (the chain)
//       head4 8 Ki x u16: last position+1 (mod 2^16) per bucket of the 4-byte hash (chain-less: one probe per position),
//       c4    ring with the answer of that probe for the positions not searched yet
//   * NO workgroup barrier in the steady state.  Waves 0 .. P-1 (P = 1-3 by level, tile k built by wave k mod P) are the
//     PRODUCERS: they stream the shard from HBM (one coalesced 16 B/lane load per 1 KiB, issued a tile ahead), hash 64
//     positions per step, insert them into the 16-bit heads in position order (plain read + write: the read gives the
//     chain predecessor; a token passed from tile to tile serialises the inserts of different producers) and publish a
//     `ready` frontier.  They are throttled only by the ring: they may not overwrite bytes that the oldest in-flight
//     search can still reference.
//   * The other waves are SEARCHERS.  A searcher claims 64 consecutive positions (one per lane) from a shared
//     LDS counter and walks the chains as a tight per-lane loop.  Every chain step issues ONE round of LDS reads
//     (prev link of the candidate + its first 16 window bytes as five aligned dwords), so it costs one LDS
//     latency.  A candidate equal in all 16 bytes ENDS the walk and is extended after it, 16 bytes per round, by
//     the wave together -- one comparing lane per run of neighbouring positions with the same distance, the others take the
//     leader's length minus their offset (round 4; round 5: at every level).
//     A claim is a bounded amount of work (<= max_chain steps), so a slow wave delays the ring by far less than
//     its slack; claims are dynamic, so no wave waits for another (the first version had a barrier per 1 KiB
//     tile and spent 61 % of its wave-cycles waiting).
//   * window carry-over (prm.carry): the bytes in front of a segment of a longer stream are hashed but not
//     searched, so the segment matches into them -- also how a preset dictionary works.
//   * output: one u32 per position  lit | len<<8 | (dist-1)<<17  (len = 0: no match >= 4).
//     The parse (greedy/lazy selection) happens in encode.hip, which sees the best match of every
//     position, not just the visited ones.
// Bound: the LDS pipe first (gathers at random addresses: a gathered dword per chain step costs what a dozen VALU instructions
// cost; the producers' pace alone is 85 of the kernel's 115 ms at level 6), instruction issue second -- not HBM and not LDS
// latency (DESIGN.md sections 3.0 / 3.0a, profiles/r04_deflate_experiments.txt; two claims per wave in one loop -- more loads
// in flight -- made it slower).  Round 5 counters: VALU instructions take 83 % of all SIMD cycles (3.4 per byte: ~50 per 64 positions
// in the producers, ~50 claim set-up and probe, 34 per chain step at 1.9 steps per position); fetching a candidate's five dwords
// only after ONE byte of it agreed (67 % of the candidates of text fail that) was slower at budgets 2 and 4 and faster at 16 --
// a second dependent round trip costs more than the dwords saved (profiles/r05_parse_tables.txt).  HBM traffic is 1 B read +
// 4 B scratch written per input byte (measured: exactly that, profiles/r05_traffic.json).
#include "zmi_device.h"
#include "zmi_kernels.h"

#define LZ_T 1024u
#define LZ_SUB (LZ_T / 64u)
#define LZ_NW 16u
#define LZ_WSIZE 32768u
#define LZ_WMASK 32767u
// Hash heads are 16-bit (position + 1 modulo 2^16; a distance beyond max_dist means "empty or stale"): twice the buckets
// in the same LDS.  Buckets are what the search quality hangs on: a 32 KiB text window has ~20 K distinct 6-grams, with
// 8 Ki buckets more than half of all chain links led to a different string (measured: 41 % of the chain candidates of
// English text were real) and every such candidate costs a searcher a full step.  16 Ki + 8 Ki buckets are worth about
// 1.5 chain steps (lcet10.txt, budget 4: 2.791 -> 2.827; budget 3 then equals the old budget 4).  The price: no 16-bit
// LDS atomics, so an insert is a plain read + write; the 64 positions of one step that fall into the same bucket all
// see the bucket's previous occupant (they lose each other as predecessors -- only distances below 64 are affected, a
// few bits per match) and the HIGHEST lane -- the most recent position -- becomes the new head: gfx950 applies the lanes
// of one ds_write_b16 to the same address in lane order (tools/lds_winner.hip, profiles/r03_lds_winner.txt: contiguous and
// strided groups of 2..64 lanes, always the highest lane's value stays), which is also what the CPU emulator of the tests
// does, so the compressed bytes are defined by the algorithm, not by the schedule.
#define LZ_HBITS 14
#define LZ_HSIZE (1u << LZ_HBITS)
#define LZ_MIRROR 32u
#define LZ_CTL 128u
#define LZ_H4BITS 13
#define LZ_H4SIZE (1u << LZ_H4BITS)
#define LZ_C4RING 4096u   // positions: the tile being built ends at most this far past the oldest search (see `hold`)
#define LZ_MAX_DIST (LZ_WSIZE - LZ_C4RING - LZ_T - 16u)   // 27 632: farthest back-reference; a producer's round keeps bytes
                          // [0, (k + 2) * LZ_T + 16) resident while it builds tile k, which ends LZ_T + 16 below that
#define LZ_SMEM (LZ_WSIZE + LZ_MIRROR + 2u * LZ_WSIZE + 2u * LZ_HSIZE + 2u * LZ_H4SIZE + 2u * LZ_C4RING + LZ_CTL)

struct LzCtl {
    uint32_t ready;  // positions < ready are searchable (chain built, look-ahead bytes loaded)
    uint32_t next;   // next unclaimed position
    uint32_t atok;   // two producers: the tile whose hash-head atomics may be issued next (keeps them in position order)
    uint32_t pad;
    uint32_t wmin[LZ_NW];  // per wave: lower bound of its oldest in-flight position
    uint32_t stored[4];    // per producer wave: input bytes [.., stored) of its last chunk are in the ring
};

#ifdef ZMI_EMU
static inline uint32_t lz_ld_acq(uint32_t* p) { return *p; }
static inline void lz_st_rel(uint32_t* p, uint32_t v) { *p = v; }
static inline void lz_pause() { emu::spin_yield(); }
static inline void lz_pause_short() { emu::spin_yield(); }
static inline void lz_st_relaxed(uint32_t* p, uint32_t v) { *p = v; }
#else
static __device__ __forceinline__ uint32_t lz_ld_acq(uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
static __device__ __forceinline__ void lz_st_rel(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#ifndef LZ_PAUSE_N
#define LZ_PAUSE_N 4
#endif
static __device__ __forceinline__ void lz_pause() { __builtin_amdgcn_s_sleep(LZ_PAUSE_N); }
static __device__ __forceinline__ void lz_pause_short() { __builtin_amdgcn_s_sleep(1); }
static __device__ __forceinline__ void lz_st_relaxed(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#endif

static __device__ __forceinline__ uint32_t lz_ring32(const uint8_t* win, uint32_t pos) {
    return zmi_load32u(win, pos & LZ_WMASK);
}
// 8 bytes at ring position pos (unaligned): three aligned dwords + two v_alignbyte
static __device__ __forceinline__ void lz_ring64(const uint8_t* win, uint32_t pos, uint32_t& lo, uint32_t& hi) {
    uint32_t r = pos & LZ_WMASK;
    const uint32_t* w = (const uint32_t*)(win + (r & ~3u));
    uint32_t a = w[0], b = w[1], c = w[2];
    lo = __builtin_amdgcn_alignbyte(b, a, r & 3u);
    hi = __builtin_amdgcn_alignbyte(c, b, r & 3u);
}
// 16 bytes at ring position pos (unaligned): FIVE aligned dwords + four v_alignbyte -- what a gather costs follows the dwords it
// fetches (profiles/r04_deflate_experiments.txt), and two lz_ring64 fetch six
static __device__ __forceinline__ void lz_ring128(const uint8_t* win, uint32_t pos, uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
    const uint32_t r = pos & LZ_WMASK;
    const uint32_t* w = (const uint32_t*)(win + (r & ~3u));
    const uint32_t a = w[0], b = w[1], c = w[2], d = w[3], e = w[4];
    const uint32_t sh = r & 3u;
    o0 = __builtin_amdgcn_alignbyte(b, a, sh);
    o1 = __builtin_amdgcn_alignbyte(c, b, sh);
    o2 = __builtin_amdgcn_alignbyte(d, c, sh);
    o3 = __builtin_amdgcn_alignbyte(e, d, sh);
}
// number of equal leading bytes (0..8) given the XOR of two 8-byte strings
static __device__ __forceinline__ uint32_t lz_match8(uint32_t xlo, uint32_t xhi) {
    if (xlo) return (uint32_t)(__ffs(xlo) - 1) >> 3;
    if (xhi) return 4u + ((uint32_t)(__ffs(xhi) - 1) >> 3);
    return 8u;
}

static __device__ __forceinline__ void lz_store_chunk(uint8_t* win, uint32_t pos, const zmi_b16& v) {
    uint32_t r = pos & LZ_WMASK;  // pos is a multiple of 16
    uint4 q;
    q.x = v.w[0]; q.y = v.w[1]; q.z = v.w[2]; q.w = v.w[3];
    *(uint4*)(win + r) = q;
    if (r < LZ_MIRROR) *(uint4*)(win + LZ_WSIZE + r) = q;  // first 32 bytes are mirrored past the end
}

// insert the 1024 positions of one tile into the hash structures in position order (one wave, 16
// steps of 64; the three phases let the LDS reads, the table updates and the stores of all steps pipeline).
// H6 = false: one chain keyed by a 4-byte hash (the reference's structure, hash_calc.rs:30-59).
// H6 = true : the chain is keyed by a 6-byte hash -- far sparser, every link is a >= 6-byte match --
//             and a second, chain-less table remembers the most recent position of every 4-byte
//             hash; its answer for position p is parked in a small ring (c4) until p is searched.
// Registers: per step ONE word (6-byte bucket | 4-byte bucket << 14, later the two old head values) plus one word with
// the sibling distances of all 16 steps, 2 bits each -- the kernel sits at the 128-VGPR limit of a 1024-thread
// workgroup, and the producer's instructions come out of the same VALU budget as the searchers' (a third of it before
// this was slimmed down).
template <bool H6, bool full>   // full: every position of the tile has its 6 bytes (all tiles but the last): no bounds checks
static __device__ __forceinline__ void lz_build_tile_t(const uint8_t* win, uint16_t* prev, uint16_t* head, uint16_t* head4,
                                                       uint16_t* c4, uint32_t tile, uint32_t n, uint32_t max_dist, LzCtl* ctl,
                                                       uint32_t producers) {
    const uint32_t lane = zmi_lane();
    // A tile is 1024-aligned and the rings are multiples of 1024: inside a tile nothing wraps, so every address below is
    // "tile base + lane + 64 s" with the 64 s folding into the instruction's offset field, and the byte alignment of a
    // lane's window reads (lane & 3) is the same in all 16 steps
    const uint32_t p0 = tile * LZ_T + lane;
    const uint8_t* wt = win + ((tile * LZ_T) & LZ_WMASK) + (lane & ~3u);
    uint16_t* pt = prev + ((tile * LZ_T) & LZ_WMASK) + lane;
    uint16_t* ct = c4 + ((tile * LZ_T) & (LZ_C4RING - 1u)) + lane;
    const uint32_t sh = lane & 3u;
    uint32_t st[LZ_SUB];
    uint32_t sibs = 0;   // step s, bits 2s..2s+1: 1 / 2 = the lane 1 / 2 below holds the same 6-byte bucket in this step
#pragma unroll
    for (uint32_t s = 0; s < LZ_SUB; ++s) {
        const uint32_t p = p0 + s * 64u;
        const uint32_t* w = (const uint32_t*)(wt + s * 64u);
        uint32_t h2, h42 = 0;   // byte offsets into the tables: bucket * 2
        if (H6) {
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh), hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
            // full-rate 24 x 24 bit multiplies (v_mul_u32_u24) instead of the quarter-rate 32-bit ones a 64-bit
            // multiplicative hash needs: bytes 0-2, 3-5 (and byte 3 alone for the 4-byte hash) are scrambled
            // separately and summed; the top bits of the sum depend on every input bit
            const uint32_t a = lo & 0xFFFFFFu, b = (lo >> 24) | ((hi & 0xFFFFu) << 8);
            const uint32_t ma = __umul24(a, 0x9E3779u);
            h2 = ((ma + __umul24(b, 0x85EBCBu)) >> (31 - LZ_HBITS)) & ~1u;
            h42 = ((ma + __umul24(lo >> 24, 0xC2B2AFu)) >> (31 - LZ_H4BITS)) & ~1u;
        } else {
            const uint32_t v = __builtin_amdgcn_alignbyte(w[1], w[0], sh);
            h2 = ((v * 2654435761u) >> (31 - LZ_HBITS)) & ~1u;  // multiplier as hash_calc.rs:30-33
        }
        // The positions of one step that share a bucket do not see each other through the table (plain read, then write):
        // the nearest one within two lanes below is found with DPP lane shifts instead -- runs and short periods (zeros,
        // "abab") keep their distance-1 / -2 links, which is where such data gets its cheapest matches.  (+1: bucket 0 must
        // not look like the zero that lanes 0 and 1 read from "below the wave"; positions without their bytes -- last
        // tile only -- carry a key no bucket has.)
        const uint32_t key = (full || p + (H6 ? 6u : 4u) <= n) ? h2 + 1u : 2u * LZ_HSIZE + 1u + 2u * (lane & 3u);
        const uint32_t u1 = zmi_lane_up1(key), u2 = zmi_lane_up1(u1);
        sibs |= (u1 == key ? 1u : (u2 == key ? 2u : 0u)) << (2u * s);
        st[s] = h2 | (h42 << 16);
        if ((s & 7u) == 7u) zmi_sched_fence();   // eight steps' loads in flight at a time, not sixteen (registers)
    }
    // several producers hash different tiles concurrently; the inserts themselves must happen in position order.
    // Everything between taking the token and passing it on is serial for the whole workgroup (1024 times per MiB): the
    // hashes must be finished BEFORE the wait (left alone, the compiler sinks half of the arithmetic of the loop above
    // below the spin loop to save registers: ~130 VALU instructions inside the critical section instead of ~50)
#ifndef ZMI_EMU
#pragma unroll
    for (uint32_t s = 0; s < LZ_SUB; ++s) asm volatile("" : "+v"(st[s]));
    asm volatile("" : "+v"(sibs));
#endif
    if (producers > 1u) {
        while (lz_ld_acq(&ctl->atok) != tile) lz_pause_short();
    }
    // table update: read the bucket's occupant, write this position.  The LDS runs a wave's instructions in order, so
    // the 64 reads of a step see the writes of the step before without any wait in between
#pragma unroll
    for (uint32_t s = 0; s < LZ_SUB; ++s) {
        const uint32_t p = p0 + s * 64u;
        uint16_t* hp = (uint16_t*)((uint8_t*)head + (st[s] & 0xFFFFu));
        uint16_t* h4p = (uint16_t*)((uint8_t*)head4 + (st[s] >> 16));
        const bool in6 = full || p + (H6 ? 6u : 4u) <= n, in4 = H6 && (full || p + 4u <= n);
        // (a position without its bytes -- last tile only -- gets distance 0: no link)
        const uint32_t old = in6 ? *hp : ((p + 1u) & 0xFFFFu), old4 = in4 ? *h4p : ((p + 1u) & 0xFFFFu);
        zmi_wave_order();   // all 64 reads of the step, then its writes (one instruction each on the hardware)
        if (in6) *hp = (uint16_t)(p + 1u);
        if (in4) *h4p = (uint16_t)(p + 1u);
        st[s] = old | (old4 << 16);
        zmi_wave_order();  // steps are position-ordered
    }
    // once this store is visible the table updates above have been applied (in-order LDS)
    // (relaxed: the LDS runs a wave's instructions in order, so the token need not wait for the data of the reads above)
    if (producers > 1u && lane == 0) lz_st_relaxed(&ctl->atok, tile + 1u);
    // The links are stored RAW: the distance to the bucket's previous occupant modulo 2^16, whatever it is -- a bucket never
    // written gives p + 1, a stale one anything.  Whether a link is alive (1 <= distance <= min(max_dist, position)) is the
    // searchers' question, asked once per position and folded into the walk's distance test: here it was three instructions
    // per position and table, and the producers' instruction stream is what the low levels run at.
    (void)max_dist;
#pragma unroll
    for (uint32_t s = 0; s < LZ_SUB; ++s) {
        const uint32_t p = p0 + s * 64u;
        const uint32_t d = (p + 1u - (st[s] & 0xFFFFu)) & 0xFFFFu;
        const uint32_t sb = (sibs >> (2u * s)) & 3u;
        pt[s * 64u] = (uint16_t)(sb != 0u ? sb : d);   // a sibling of the same step is the nearer predecessor
        if (H6) ct[s * 64u] = (uint16_t)((p + 1u - (st[s] >> 16)) & 0xFFFFu);
    }
}

template <bool H6>
static __device__ __forceinline__ void lz_build_tile(const uint8_t* win, uint16_t* prev, uint16_t* head, uint16_t* head4,
                                                     uint16_t* c4, uint32_t tile, uint32_t n, uint32_t max_dist, LzCtl* ctl,
                                                     uint32_t producers) {
    if ((tile + 1u) * LZ_T + 6u <= n) lz_build_tile_t<H6, true>(win, prev, head, head4, c4, tile, n, max_dist, ctl, producers);
    else lz_build_tile_t<H6, false>(win, prev, head, head4, c4, tile, n, max_dist, ctl, producers);
}

template <bool H6>
__global__ void __launch_bounds__(1024) zmi_lz77_kernel_t(const uint8_t* __restrict__ data, const uint64_t* __restrict__ off,
                                                        const uint32_t* __restrict__ len, uint32_t first_shard,
                                                        uint32_t* __restrict__ match, uint64_t match_stride,
                                                        zmi_lz_params prm) {
#ifdef ZMI_EMU
    ZMI_DYN_SMEM(smem);
#else
    __shared__ __attribute__((aligned(16))) uint8_t smem[LZ_SMEM];   // static: LDS addresses fold into the instructions' offsets
#endif
    uint8_t* win = smem;
    uint16_t* prev = (uint16_t*)(smem + LZ_WSIZE + LZ_MIRROR);
    uint16_t* head = (uint16_t*)(smem + LZ_WSIZE + LZ_MIRROR + 2u * LZ_WSIZE);
    uint16_t* head4 = head + LZ_HSIZE;
    uint16_t* c4 = (uint16_t*)(head4 + LZ_H4SIZE);
    LzCtl* ctl = (LzCtl*)(c4 + LZ_C4RING);

    const uint32_t t = threadIdx.x;
    const uint32_t lane = zmi_lane();
    const uint32_t wave = zmi_wave();
    const uint32_t local = zmi_xcd_spread(blockIdx.x, gridDim.x);
    const uint32_t s = first_shard + local;
    // window carry-over: the bytes in front of a segment that belong to the same stream become `hist` extra
    // positions at the start of the shard; they are hashed (producer) but not searched, so every position index
    // in this kernel is "virtual" = hist + position in the segment, and results are stored at index - hist
    uint32_t hist = 0;
    if (prm.carry) {
        const uint64_t before = off[s] - off[0] + prm.dict_len;   // history bytes of this stream in front of the segment
        // whole tiles of history: max_dist rounded up (a small window -- windowBits 9..12 -- would otherwise see no history
        // or dictionary at all; positions farther back than max_dist are hashed but never reached), at most the 27 KiB
        // that the full window gets
        uint32_t reach = (prm.max_dist + LZ_T - 1u) & ~(LZ_T - 1u);
        if (reach > LZ_WSIZE - 5u * LZ_T) reach = LZ_WSIZE - 5u * LZ_T;
        hist = before < reach ? (uint32_t)before & ~15u : reach;  // multiples of 16 keep the 16-byte load path
    }
    const uint8_t* src = data + off[s] - hist;
    const uint32_t n = len[s] + hist;
    uint32_t* mout = match + (uint64_t)local * match_stride - hist;
    const bool aligned = (((uintptr_t)src) & 15u) == 0;
    const uint32_t ntiles = (n + LZ_T - 1u) / LZ_T;
    if (ntiles == 0) return;

    for (uint32_t i = t; i < (LZ_HSIZE + LZ_H4SIZE) / 2u; i += 1024u) ((uint32_t*)head)[i] = 0u;   // head4 follows head
    if (t == 0) { ctl->ready = 0u; ctl->next = hist; ctl->atok = 0u; ctl->stored[0] = 0u; ctl->stored[1] = 0u; ctl->stored[2] = 0u; ctl->stored[3] = 0u; }
    if (t < LZ_NW) ctl->wmin[t] = 0xFFFFFFFFu;
    __syncthreads();

    const uint32_t P = prm.producers < 1u ? 1u : (prm.producers > 4u ? 4u : prm.producers);
    if (wave < P) {
        // ---------------- producer(s) ----------------
        // Tile k is built by producer wave k mod P.  With P = 2 (the low levels, where the searchers outrun a single
        // producer) the two waves overlap their loads, hashing and link stores; only the hash-head atomics are
        // serialised in position order (ctl->atok) and the ready frontier is published in tile order.
        // The HBM load of the chunk a wave needs for its NEXT tile is issued at the top of the round and consumed at
        // the top of the following one, so its latency hides behind the throttle wait and the hash inserts.
        if (wave == 0) {
            for (uint32_t c = lane * 16u; c < LZ_T + 16u; c += 1024u) {
                zmi_b16 v0 = zmi_ld16(src + c, c < n ? n - c : 0u, aligned);
                lz_store_chunk(win, c, v0);
            }
        }
        uint32_t cpos = (wave + 1u) * LZ_T + 16u + lane * 16u;   // chunk that this wave's first round must make resident
        zmi_b16 cur = zmi_ld16(src + cpos, cpos < n ? n - cpos : 0u, aligned);
        uint32_t cached_min = 0u;
        // How far the producers may run ahead of the oldest search: the window ring allows 32 KiB - max_dist, but the ring of
        // probe answers (c4: LZ_C4RING positions) must not be overwritten before it is read either -- with a short max_dist
        // (windowBits < 15, Z_RLE) the window alone would let tile q + 4096 be built while position q is still waiting.
        // LZ_MAX_DIST is the distance at which both limits coincide: the tile being built ends at most LZ_C4RING positions
        // past the oldest search.
        const uint32_t hold = prm.max_dist > LZ_MAX_DIST ? prm.max_dist : LZ_MAX_DIST;
        for (uint32_t k = wave; k < ntiles; k += P) {
            const uint32_t npos = cpos + P * LZ_T;
            zmi_b16 nxt = zmi_ld16(src + npos, npos < n ? n - npos : 0u, aligned);  // for this wave's next tile
            const uint32_t E = (k + 2u) * LZ_T + 16u;  // bytes [0, E) must be resident after this round
            // ring throttle: byte E-1 lands on the slot of byte E-1-32768, which the oldest in-flight
            // search (position q) may still read while q - max_dist <= E-1-32768
            while ((uint64_t)E + hold > (uint64_t)cached_min + LZ_WSIZE) {
                uint32_t v = 0xFFFFFFFFu;
                if (lane < LZ_NW) v = lz_ld_acq(&ctl->wmin[lane]);
                else if (lane == LZ_NW) v = lz_ld_acq(&ctl->next);
                uint32_t m = v;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    uint32_t o = __shfl_xor(m, d);
                    m = o < m ? o : m;
                }
                cached_min = m;
                if ((uint64_t)E + hold <= (uint64_t)m + LZ_WSIZE) break;
                lz_pause();
            }
            lz_store_chunk(win, cpos, cur);
            zmi_wave_sync();
            if (P > 1u) {
                if (lane == 0) lz_st_rel(&ctl->stored[wave], E);
                // hashing tile k reads bytes up to (k+1)*T + 7: the chunk the other wave stored in its round k-1
                if (k > 0u) while (lz_ld_acq(&ctl->stored[(wave + P - 1u) % P]) < (k + 1u) * LZ_T + 16u) lz_pause();
            }
            lz_build_tile<H6>(win, prev, head, head4, c4, k, n, prm.max_dist, ctl, P);
            zmi_wave_sync();
            uint32_t r = (k + 1u) * LZ_T;
            if (r > n) r = n;
            if (P > 1u) {   // publish in tile order
                const uint32_t before = k * LZ_T < n ? k * LZ_T : n;
                while (lz_ld_acq(&ctl->ready) < before) lz_pause();
            }
            if (lane == 0) lz_st_rel(&ctl->ready, r);
            cur = nxt;
            cpos = npos;
        }
        return;
    }

    // ---------------- searchers ----------------
    // Each searcher wave claims 64 consecutive positions at a time (one per lane) and runs the chain
    // walk as a tight per-lane loop: ONE round of LDS reads per candidate (the prev link of the
    // candidate, its first 8 window bytes and, once the best match is >= 8, the 4 bytes ending at
    // the best length), so a chain step costs one LDS latency, not two.
    for (;;) {
        // claim prm.claim positions (64 per round, one position per lane); one LDS atomic per claim
        uint32_t base0 = 0;
        if (lane == 0) {
            uint32_t tnext = lz_ld_acq(&ctl->next);
            lz_st_rel(&ctl->wmin[wave], tnext);  // lower bound of everything this wave still needs, BEFORE the claim
            base0 = atomicAdd(&ctl->next, prm.claim);
        }
        base0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)base0);
        if (base0 >= n) break;
#ifdef ZMI_LZ_MEASURE   // measurement builds only (variants/): the product's searcher loop carries no branch for it
        if (prm.dbg == 1u) {   // ZMI_LZ_DBG=1 under ZMI_TUNING: the producers' pace alone -- the searchers claim, read
                               // one byte and store a literal (profiles/r03_lz77_experiments.txt)
            const uint32_t need = base0 + 64u < n ? base0 + 64u : n;
            while (lz_ld_acq(&ctl->ready) < need) lz_pause();
            if (base0 + lane < n) mout[base0 + lane] = win[(base0 + lane) & LZ_WMASK];
            continue;
        }
#endif
      for (uint32_t half = 0; half * 64u < prm.claim; ++half) {
        const uint32_t base = base0 + half * 64u;
        if (base >= n) break;
        const uint32_t need = base + 64u < n ? base + 64u : n;
        while (lz_ld_acq(&ctl->ready) < need) lz_pause();
        // Every lane runs the same straight-line code, also the lanes behind the end of the shard in its last claim (their ring
        // addresses are valid whatever they hold; maxlen = 0 keeps them out of the probe, out of the walk and out of the store):
        // a validity branch around the set-up cost a dozen instructions per claim for registers that had to be defined on both sides.
        const uint32_t p = base + lane;
        uint32_t mylo, myhi, my2, my3;
        lz_ring128(win, p, mylo, myhi, my2, my3);
        uint32_t maxlen = p < n ? n - p : 0u;
        maxlen = maxlen > 258u ? 258u : maxlen;
        // links are raw (lz_build_tile): alive if 1 <= distance <= lim
        const uint32_t lim = p < prm.max_dist ? p : prm.max_dist;
        uint32_t delta = prev[p & LZ_WMASK];
        delta = delta - 1u < lim ? delta : 0u;
        uint32_t blen = 3u, bdist = 0u;
        // H6: the most recent 4-byte match (no chain of its own) is looked at first, outside the chain loop and with an
        // 8-byte compare only: what it is for are the 4- and 5-byte matches the 6-byte chain cannot see; a longer match
        // is in the chain as well.  (Peeling a full-size step out of the loop was slower -- lanes without a probe idle
        // through it -- but this one is a third of a chain step and takes the probe bookkeeping out of every step.)
        if (H6) {
            const uint32_t d4 = c4[p & (LZ_C4RING - 1u)];
            if (d4 - 1u < lim && d4 != delta && maxlen >= 4u) {
                uint32_t a, b;
                lz_ring64(win, p - d4, a, b);
                const uint32_t c0 = zmi_ffbl(a ^ mylo), c1 = zmi_ffbl(b ^ myhi) | 32u;
                uint32_t l = (c0 < c1 ? c0 : c1) >> 3;
                l = l > 8u ? 8u : l;
                l = l > maxlen ? maxlen : l;
                if (l >= 4u) { blen = l; bdist = d4; }
            }
        }
        // candidates per position: max_chain counts the probe.  In a claim where not one of the 64 probes hit (a stretch of
        // incompressible data: every chain candidate there is a hash collision, and each costs a full step) the positions walk
        // prm.barren_chain links only.  (Decided from the claim's own data: the output does not depend on which wave ran
        // which claim.)
        const bool barren = H6 && __ballot(blen >= 4u) == 0ull;
        uint32_t chain = H6 ? (prm.max_chain > 1u ? prm.max_chain - 1u : prm.max_chain) : prm.max_chain;
        if (H6 && barren && chain > prm.barren_chain) chain = prm.barren_chain;
        // a match this long ends the walk: nice_len, the end of the input, good_len (the reference quarters the remaining chain
        // there, longest_match.rs:60-66: of these budgets nothing is left) -- and at most 16: the walk ends at the first candidate
        // equal in all 16 bytes it looks at, how long that match really is the wave finds out afterwards (below).
        // (Until round 5 a second instantiation for the budgets above 8 went on behind such a match and extended it inside the loop:
        // 22 ms per candidate against 7, and with the cost parse no level's choice -- profiles/r05_level9_budget_sweep.txt.)
        uint32_t stoplen = prm.nice_len < maxlen ? prm.nice_len : maxlen;
        if (prm.good_len < stoplen) stoplen = prm.good_len;
        if (stoplen > 16u) stoplen = 16u;
        if (maxlen >= 4u && delta != 0u && chain != 0u && prm.max_chain != 0u) {
            uint32_t cand = p - delta;
            // The loop body is straight-line: a candidate is decided by its first 16 bytes (five aligned dwords, one LDS round trip
            // together with the prev link); a candidate equal in all 16 ends the walk (stoplen <= 16), so nothing in here concerns
            // matches of 16+ bytes.
            for (;;) {
                const uint32_t r = cand & LZ_WMASK;
                const uint32_t* w = (const uint32_t*)(win + (r & ~3u));
                const uint32_t dn = prev[r];
                const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
                const uint32_t sh = r & 3u;
                // first differing byte of the 16: v_ffbl gives -1 for an all-equal dword, so OR-ing in the dword's
                // bit offset keeps that "infinite" and a three-way minimum picks the first mismatch
                const uint32_t x0 = __builtin_amdgcn_alignbyte(w1, w0, sh) ^ mylo, x1 = __builtin_amdgcn_alignbyte(w2, w1, sh) ^ myhi;
                const uint32_t x2 = __builtin_amdgcn_alignbyte(w3, w2, sh) ^ my2, x3 = __builtin_amdgcn_alignbyte(w4, w3, sh) ^ my3;
                const uint32_t c0 = zmi_ffbl(x0), c1 = zmi_ffbl(x1) | 32u;
                const uint32_t c2 = zmi_ffbl(x2) | 64u, c3 = zmi_ffbl(x3) | 96u;
                uint32_t m3 = c0 < c1 ? c0 : c1;
                m3 = m3 < c2 ? m3 : c2;
                m3 = m3 < c3 ? m3 : c3;
                uint32_t l = m3 >> 3;        // 0..15, or 0x1FFFFFFF when all 16 bytes are equal
                l = l > 16u ? 16u : l;
                l = l > maxlen ? maxlen : l;
                // (a candidate that differs inside its first 16 bytes cannot beat a best match of 16+: nothing to do)
                const bool better = l > blen;
                // the walk ends at a match of stoplen or more that is also the best so far: one compare against the larger bound
                const uint32_t bar = blen + 1u > stoplen ? blen + 1u : stoplen;
                const uint32_t dist = p - cand;
                blen = better ? l : blen;
                bdist = better ? dist : bdist;
                cand -= dn;
                chain -= 1u;
                // ... and at the end of the chain (dn = 0: dn - 1 wraps to the largest value) or where the raw link leads out of the
                // window, in front of the shard or nowhere (dist + dn > lim; dist <= lim holds on entry and after every step): one
                // compare for both.  (Bitwise, not short-circuit: as `||` the compiler built a branch per term.)
                const bool stop = (l >= bar) | (dn - 1u >= lim - dist) | ((int32_t)chain <= 0);
                if (stop) break;
            }
        }
        {
            // The walk ended at the first candidate equal in 16 bytes; its real length is found here, the wave
            // together: NEIGHBOURING positions mostly hold the same match one byte further on -- the same distance, one byte
            // shorter -- so of a run of lanes with one distance only the lowest (the leader) compares bytes, and lane leader + k
            // takes the leader's length - k (exact: both stop at the same mismatching byte).  A leader that ran into its cap
            // (258 or the end of the shard) says nothing about its followers: those compare for themselves, as does a follower
            // whose inherited length would fall below the 16 bytes it has seen.  On XML- and record-like data this is most of
            // the bytes the extension used to compare (every lane of a 60-byte match walked the same 60 bytes).
            bool pend = blen >= 16u && maxlen > 16u;
            if (__ballot(pend) != 0ull) {
                const uint32_t dprev = zmi_lane_up1(pend ? bdist : 0u);
                bool work = pend && dprev != bdist;   // leaders first
                const uint64_t lm = __ballot(work);
                uint32_t L = 16u;
                for (int round = 0; round < 2; ++round) {
                    if (work) {
                        const uint32_t cand = p - bdist;
                        uint32_t l = 16u;
                        for (;;) {
                            uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
                            lz_ring128(win, p + l, a0, a1, a2, a3);
                            lz_ring128(win, cand + l, b0, b1, b2, b3);
                            uint32_t m = lz_match8(a0 ^ b0, a1 ^ b1);
                            if (m == 8u) m += lz_match8(a2 ^ b2, a3 ^ b3);
                            l += m;
                            if (m < 16u || l >= maxlen) break;
                        }
                        L = l;
                    }
                    if (round == 1) break;
                    // followers: the leader is the highest leader lane at or below this one
                    const uint64_t below = lm & ((2ull << lane) - 1ull);
                    const uint32_t j = below ? 63u - (uint32_t)__clzll((unsigned long long)below) : lane;
                    const uint32_t capped = (work && L >= maxlen) ? 0x10000u : 0u;
                    const uint32_t Lj = (uint32_t)__shfl((int)(L | capped), (int)j);
                    const uint32_t inh = (Lj & 0xFFFFu) - (lane - j);
                    const bool follower = pend && !work;
                    const bool take = follower && (Lj & 0x10000u) == 0u && (Lj & 0xFFFFu) >= 16u + (lane - j);
                    if (take) L = inh;
                    work = follower && !take;
                    if (__ballot(work) == 0ull) break;
                }
                if (pend) blen = L > maxlen ? maxlen : L;
            }
        }
        // (whether a short match far back is worth its codes is the encoder's call: it knows the prices, enc_far_limits)
        if (p < n) {
            uint32_t res = mylo & 0xFFu;
            if (blen >= 4u) res |= (blen << 8) | ((bdist - 1u) << 17);
            mout[p] = res;
        }
      }
    }
    if (lane == 0) lz_st_rel(&ctl->wmin[wave], 0xFFFFFFFFu);
}

extern "C" int zmi_launch_lz77(const uint8_t* d_data, const uint64_t* d_off, const uint32_t* d_len, uint32_t first_shard,
                               uint32_t n_shards, uint32_t* d_match, uint64_t match_stride, zmi_lz_params prm,
                               hipStream_t stream) {
    if (n_shards == 0) return 0;
    // ring budget: 32 KiB = max_dist + the tile being built + a tile of read-ahead + 3 tiles of slack for the searches in flight
    // (one tile more than round 1: searchers spent a quarter of their time waiting for the producers, the producers 39 % of
    // theirs waiting for the slowest search to release the ring -- 158.4 -> 148.9 ms for 1 KiB of reach, ratio -0.08 %)
    if (prm.max_dist > LZ_MAX_DIST) prm.max_dist = LZ_MAX_DIST;
    if (prm.claim != 128u && prm.claim != 192u && prm.claim != 256u) prm.claim = 64u;
    // 152 KiB of LDS per workgroup, allocated statically in the product build (LZ_DYN: the emulator hands it out at launch)
#ifdef ZMI_EMU
    const uint32_t LZ_DYN = LZ_SMEM;
#else
    const uint32_t LZ_DYN = 0u;
#endif
#define LZ_GO(H) ZMI_LAUNCH((zmi_lz77_kernel_t<H>), dim3(n_shards), dim3(1024), LZ_DYN, stream, d_data, d_off, d_len, first_shard, \
                            d_match, match_stride, prm)
    if (prm.hash6) LZ_GO(true); else LZ_GO(false);
#undef LZ_GO
    return 0;
}
