// lz77.hip -- LZ77 longest-match search for every byte position of every shard (deflate stage 1).
//
// What the reference does here (CPU, one thread): per position, insert a 4-byte multiplicative hash
// into head[]/prev[] (zlib-rs/src/deflate/hash_calc.rs:30-59) and walk up to max_chain links of the
// hash chain comparing against the 32 KiB sliding window (zlib-rs/src/deflate/longest_match.rs:15-346,
// compare256.rs), driven by the lazy parse loop of deflate/algorithm/medium.rs:12-178.
//
// MI355X design (not a translation):
//   * one 1024-thread workgroup per shard, the whole search state in LDS (131 KiB, 1 workgroup/CU):
//       win  32 KiB ring of input bytes (+16 mirrored bytes so a dword read may straddle the wrap)
//       prev 32 Ki x u16 ring: DISTANCE to the previous position with the same hash (0 = end of
//            chain) -- storing deltas instead of positions removes the reference's slide_hash pass
//            (deflate/slide_hash.rs) entirely
//       head 8 Ki x u32: last position+1 per hash bucket
//   * the shard is consumed in tiles of 1024 positions, ONE barrier per tile, software-pipelined:
//       during phase k  wave 0  loads tile k+2 from HBM (one coalesced 16 B/lane load),
//                               inserts tile k+1 into head/prev in position order
//                               (64 positions per LDS atomic-max; the returned old value IS the
//                               chain predecessor),
//                       all waves pull 64-position sub-tiles of tile k from an LDS ticket counter
//                               and search EVERY position in parallel (one lane = one position;
//                               the chain walk and the match extension are per-lane loops).
//   * output: one u32 per position  lit | len<<8 | (dist-1)<<17  (len = 0: no match >= 4),
//     written coalesced (256 B per wave store).  The parse (greedy/lazy selection) happens in
//     encode.hip, which sees the best match of every position, not just the visited ones.
// Bound: LDS bandwidth/latency (random 4-byte window reads), not HBM: algorithmic HBM traffic is
// 1 B read + 4 B scratch written per input byte.
#include "zmi_device.h"
#include "zmi_kernels.h"

#define LZ_T 1024u
#define LZ_SUB (LZ_T / 64u)
#define LZ_WSIZE 32768u
#define LZ_WMASK 32767u
#define LZ_HBITS 13
#define LZ_HSIZE (1u << LZ_HBITS)
#define LZ_MIRROR 16u
#define LZ_SMEM (LZ_WSIZE + LZ_MIRROR + 2u * LZ_WSIZE + 4u * LZ_HSIZE + 16u)

static __device__ __forceinline__ uint32_t lz_ring32(const uint8_t* win, uint32_t pos) {
    return zmi_load32u(win, pos & LZ_WMASK);
}

static __device__ __forceinline__ void lz_store_chunk(uint8_t* win, uint32_t pos, const zmi_b16& v) {
    uint32_t r = pos & LZ_WMASK;  // pos is a multiple of 16
    uint4 q;
    q.x = v.w[0]; q.y = v.w[1]; q.z = v.w[2]; q.w = v.w[3];
    *(uint4*)(win + r) = q;
    if (r == 0) *(uint4*)(win + LZ_WSIZE) = q;
}

// insert the 1024 positions of one tile into head/prev, in position order (one wave)
static __device__ __forceinline__ void lz_build_tile(const uint8_t* win, uint16_t* prev, uint32_t* head, uint32_t tile,
                                                     uint32_t n, uint32_t max_dist) {
    const uint32_t lane = zmi_lane();
#pragma unroll 4
    for (uint32_t s = 0; s < LZ_SUB; ++s) {
        uint32_t p = tile * LZ_T + s * 64u + lane;
        uint32_t delta = 0;
        if (p + 4u <= n) {
            uint32_t v = lz_ring32(win, p);
            uint32_t h = (v * 2654435761u) >> (32 - LZ_HBITS);  // multiplier as hash_calc.rs:30-33
            uint32_t old = atomicMax(&head[h], p + 1u);
            if (old != 0u && old <= p) {
                uint32_t d = p + 1u - old;
                if (d <= max_dist) delta = d;
            }
        }
        prev[p & LZ_WMASK] = (uint16_t)delta;
        zmi_wave_sync();  // steps are position-ordered (no-op on hardware: the wave runs in lockstep)
    }
}

static __device__ __forceinline__ uint32_t lz_search(const uint8_t* win, const uint16_t* prev, uint32_t p, uint32_t n,
                                                     const zmi_lz_params& prm) {
    uint32_t res = win[p & LZ_WMASK];
    uint32_t maxlen = n - p;
    if (maxlen > 258u) maxlen = 258u;
    if (maxlen >= 4u) {
        const uint32_t my4 = lz_ring32(win, p);
        uint32_t delta = prev[p & LZ_WMASK];
        uint32_t cand = p - delta;
        uint32_t blen = 3u, bdist = 0u;
        uint32_t tail = my4;
        uint32_t chain = prm.max_chain;
        while (delta != 0u && chain != 0u) {
            --chain;
            uint32_t dist = p - cand;
            if (dist > prm.max_dist) break;
            if (lz_ring32(win, cand + blen - 3u) == tail && (blen == 3u || lz_ring32(win, cand) == my4)) {
                uint32_t l = 4u;
                for (;;) {
                    if (l + 4u > maxlen) {
                        while (l < maxlen && win[(p + l) & LZ_WMASK] == win[(cand + l) & LZ_WMASK]) ++l;
                        break;
                    }
                    uint32_t x = lz_ring32(win, p + l) ^ lz_ring32(win, cand + l);
                    if (x) {
                        l += (uint32_t)(__ffs(x) - 1) >> 3;
                        break;
                    }
                    l += 4u;
                }
                if (l > blen) {
                    blen = l;
                    bdist = dist;
                    if (l >= prm.nice_len || l >= maxlen) break;
                    tail = lz_ring32(win, p + blen - 3u);
                    if (l >= prm.good_len) chain >>= 1;
                }
            }
            delta = prev[cand & LZ_WMASK];
            cand -= delta;
        }
        if (blen >= 4u) res |= (blen << 8) | ((bdist - 1u) << 17);
    }
    return res;
}

__global__ void __launch_bounds__(1024) zmi_lz77_kernel(const uint8_t* __restrict__ data, const uint64_t* __restrict__ off,
                                                        const uint32_t* __restrict__ len, uint32_t first_shard,
                                                        uint32_t* __restrict__ match, uint64_t match_stride,
                                                        zmi_lz_params prm) {
    ZMI_DYN_SMEM(smem);
    uint8_t* win = smem;
    uint16_t* prev = (uint16_t*)(smem + LZ_WSIZE + LZ_MIRROR);
    uint32_t* head = (uint32_t*)(smem + LZ_WSIZE + LZ_MIRROR + 2u * LZ_WSIZE);
    uint32_t* ctr = head + LZ_HSIZE;

    const uint32_t t = threadIdx.x;
    const uint32_t lane = zmi_lane();
    const uint32_t wave = zmi_wave();
    const uint32_t s = first_shard + blockIdx.x;
    const uint8_t* src = data + off[s];
    const uint32_t n = len[s];
    uint32_t* mout = match + (uint64_t)blockIdx.x * match_stride;
    const bool aligned = (((uintptr_t)src) & 15u) == 0;
    const uint32_t ntiles = (n + LZ_T - 1u) / LZ_T;
    if (ntiles == 0) return;

    // prologue: clear head, load bytes [0, 2T+16), build tile 0
    for (uint32_t i = t; i < LZ_HSIZE; i += 1024u) head[i] = 0u;
    if (t < 2u) ctr[t] = 0u;
    if (t < (2u * LZ_T + 16u) / 16u) {
        uint32_t c = t * 16u;
        zmi_b16 v = zmi_ld16(src + c, c < n ? n - c : 0u, aligned);
        lz_store_chunk(win, c, v);
    }
    __syncthreads();
    if (wave == 0) lz_build_tile(win, prev, head, 0u, n, prm.max_dist);
    __syncthreads();

    for (uint32_t k = 0; k < ntiles; ++k) {
        if (wave == 0) {
            // bytes available at phase start: [0, (k+2)T+16); fetch the next T
            uint32_t c = (k + 2u) * LZ_T + 16u + lane * 16u;
            zmi_b16 v = zmi_ld16(src + c, c < n ? n - c : 0u, aligned);
            if (k + 1u < ntiles) lz_build_tile(win, prev, head, k + 1u, n, prm.max_dist);
            lz_store_chunk(win, c, v);
            if (lane == 0) ctr[(k + 1u) & 1u] = 0u;
        }
        for (;;) {
            uint32_t sub = 0;
            if (lane == 0) sub = atomicAdd(&ctr[k & 1u], 1u);
            sub = (uint32_t)__builtin_amdgcn_readfirstlane((int)sub);
            if (sub >= LZ_SUB) break;
            uint32_t p = k * LZ_T + sub * 64u + lane;
            if (p < n) mout[p] = lz_search(win, prev, p, n, prm);
        }
        __syncthreads();
    }
}

extern "C" int zmi_launch_lz77(const uint8_t* d_data, const uint64_t* d_off, const uint32_t* d_len, uint32_t first_shard,
                               uint32_t n_shards, uint32_t* d_match, uint64_t match_stride, zmi_lz_params prm,
                               hipStream_t stream) {
    if (n_shards == 0) return 0;
    if (prm.max_dist > LZ_WSIZE - 3u * LZ_T - 16u) prm.max_dist = LZ_WSIZE - 3u * LZ_T - 16u;
#ifndef ZMI_EMU
    // 131 KiB of dynamic LDS: above the 64 KiB default, must be requested explicitly
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)zmi_lz77_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LZ_SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
#endif
    ZMI_LAUNCH(zmi_lz77_kernel, dim3(n_shards), dim3(1024), LZ_SMEM, stream, d_data, d_off, d_len, first_shard, d_match,
               match_stride, prm);
    return 0;
}
