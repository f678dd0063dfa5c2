// resolve_jump.hip -- the resolve pass for FEW streams: back-references resolved by pointer jumping, on the whole chip.
//
// What it replaces: zmi_inflate_resolve_kernel (inflate.hip) gives every stream one wave, which fills the stream's holes
// in order -- right for thousands of streams, but one stream alone is one wave alone on the chip: a 1 MiB text stream took
// 8.1 ms (round 2: uncompress() of 1 MiB 17.4 ms, a CPU core does it in 3).  Reference semantics: the match copy of
// zlib-rs/src/inflate/writer.rs:266-300 (a back-reference repeats bytes produced earlier; with dist < len the copy feeds
// on itself), done for all bytes at once instead of in stream order:
//   init    every output byte gets a pointer: itself if it is a literal, "byte - dist" if it lies in a hole (the decode
//           pass left a 3-byte record {dist, len} at the start of every hole and a bit in the per-stream bitmap).  A
//           pointer below zero leads into the history in front of the stream (dictionary / earlier output): final bytes.
//   jump    ptr[b] = ptr[ptr[ptr[b]]] for all bytes, round after round.  Every round cuts every chain to a third, so
//           ceil(log3(output bytes)) rounds are enough for any stream (a 1 MiB run of one byte is the worst case: 13);
//           a round in which nothing changed ends it -- the later launches return at once.
//   gather  out[b] = out[ptr[b]]: a chain's root is a literal (or history) byte, which nobody writes.
// Work is O(bytes x rounds) instead of O(bytes), 4 B of scratch per output byte: for a handful of streams that is a few
// microseconds per round on 256 CUs (the 4 MiB pointer array of a 1 MiB stream stays in the L2s), for thousands of streams
// the one-wave-per-stream kernel wins by far.  zmi_api.hip picks (n_streams <= 16 and <= 1 GiB of output capacity; measured, 1 MiB streams of the benchmark mix:
// 1 stream 8.1 -> 0.6 ms, 8 streams 11.6 -> 4.0 ms, 64 streams 11.7 serial against 27.7 ms).
#include "zmi_device.h"
#include "zmi_kernels.h"

#define JUMP_ITEMS 8u   // bytes per thread
#define JUMP_DONE 0x40000000   // bit 30 of a pointer: it leads to a root (capacities stay below 2^30), nothing left to do for this byte

// global index space: stream s owns indices [64 * bm_off[s], 64 * bm_off[s] + out_len[s]) -- the bit positions of its bitmap
struct JumpWhere {
    uint32_t s;      // stream, 0xFFFFFFFF: the index belongs to nobody (slack, a stream without scratch, or behind its output)
    uint32_t b;      // byte of that stream's output
    uint64_t base;   // global index of the stream's byte 0
};
static __device__ __forceinline__ JumpWhere jump_where(uint64_t idx, const uint64_t* __restrict__ bm_off, const uint32_t* __restrict__ out_len,
                                                        uint32_t n) {
    JumpWhere W;
    W.s = 0xFFFFFFFFu; W.b = 0; W.base = 0;
    // the plan kernel lays the streams out in index order: the last stream whose range starts at or below idx (n is small)
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        uint64_t o = bm_off[mid];
        if (o == ~0ull) {   // (streams without scratch sit behind the ones that fit: treat as "above")
            hi = mid;
            continue;
        }
        if (o * 64ull <= idx) lo = mid + 1u; else hi = mid;
    }
    if (lo == 0u) return W;
    const uint32_t s = lo - 1u;
    const uint64_t base = bm_off[s] * 64ull;
    const uint64_t b = idx - base;
    if (bm_off[s] == ~0ull || b >= (uint64_t)out_len[s]) return W;
    W.s = s; W.b = (uint32_t)b; W.base = base;
    return W;
}

__global__ void __launch_bounds__(256) zmi_jump_init_kernel(const uint8_t* __restrict__ out, const uint64_t* __restrict__ out_off,
                                                            const uint32_t* __restrict__ out_len, uint32_t n,
                                                            const uint64_t* __restrict__ bitmap, const uint64_t* __restrict__ bm_off,
                                                            int32_t* __restrict__ ptr, uint64_t n_idx, uint32_t* __restrict__ flags) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (tid == 0) { for (uint32_t r = 0; r < 40u; ++r) flags[r] = 0u; }
  for (uint32_t it = 0; it < JUMP_ITEMS; ++it) {
    const uint64_t idx = tid + (uint64_t)it * gridDim.x * 256u;
    if (idx >= n_idx) return;
    const JumpWhere W = jump_where(idx, bm_off, out_len, n);
    if (W.s == 0xFFFFFFFFu) continue;
    const uint64_t* bm = bitmap + bm_off[W.s];
    const uint8_t* dst = out + out_off[W.s];
    // the nearest hole start at or below b, at most 257 bytes back (a hole is at most 258 bytes long)
    int32_t p = (int32_t)W.b;
    uint32_t w = W.b >> 6;
    uint64_t m = bm[w] & ((2ull << (W.b & 63u)) - 1ull);
    for (uint32_t k = 0; k < 5u && m == 0ull && w > 0u; ++k) m = bm[--w];
    if (m != 0ull) {
        const uint32_t q = (w << 6) + 63u - (uint32_t)__clzll((unsigned long long)m);
        if (W.b - q < 258u) {
            const uint32_t rec = (uint32_t)dst[q] | ((uint32_t)dst[q + 1u] << 8) | ((uint32_t)dst[q + 2u] << 16);
            const uint32_t dist = (rec & 0x7FFFu) + 1u, len = (rec >> 15) + 3u;
            if (W.b - q < len) p = (int32_t)W.b - (int32_t)dist;
        }
    }
    ptr[idx] = p;
  }
}

__global__ void __launch_bounds__(256) zmi_jump_round_kernel(const uint32_t* __restrict__ out_len, uint32_t n,
                                                             const uint64_t* __restrict__ bm_off, int32_t* ptr, uint64_t n_idx,
                                                             uint32_t* flags, uint32_t round) {
#ifdef ZMI_EMU
    if (round > 0u && flags[round - 1u] == 0u) return;
#else
    if (round > 0u && __hip_atomic_load(&flags[round - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;   // settled
#endif
    const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    bool changed = false;
    // eight bytes per thread, a grid apart (coalesced); unrolled: the eight pointer chases are in flight together
    int32_t p[JUMP_ITEMS], pp[JUMP_ITEMS];
    uint64_t base[JUMP_ITEMS];
    int32_t self[JUMP_ITEMS];
#pragma unroll
    for (uint32_t it = 0; it < JUMP_ITEMS; ++it) {
        const uint64_t idx = tid + (uint64_t)it * gridDim.x * 256u;
        p[it] = -1; self[it] = -1; base[it] = 0;
        if (idx < n_idx) {
            const JumpWhere W = jump_where(idx, bm_off, out_len, n);
            // (plain accesses: a pointer read while its owner updates it is the old or the new value -- both are ancestors)
            if (W.s != 0xFFFFFFFFu) { p[it] = ptr[idx]; self[it] = (int32_t)W.b; base[it] = W.base; }
        }
    }
#pragma unroll
    for (uint32_t it = 0; it < JUMP_ITEMS; ++it) {
        // a byte that is its own root, one whose chain ends in the history (p < 0) and one marked done are finished: most
        // bytes after two or three rounds, and they cost no second (scattered) read any more
        const bool open = p[it] >= 0 && p[it] != self[it] && (p[it] & JUMP_DONE) == 0;
        pp[it] = open ? ptr[base[it] + (uint32_t)p[it]] : p[it];
        if (open && pp[it] == p[it]) pp[it] = p[it] | JUMP_DONE;   // p is a root
        else if (open && pp[it] >= 0 && (pp[it] & JUMP_DONE) == 0) {
            // a second hop in the same round: a chain shrinks to a third per round instead of a half -- a third fewer passes
            // over the pointer array and a third fewer launches for the same number of scattered reads (a 15 MiB stream spent
            // 5.5 of its 11 ms in 25 rounds)
            const int32_t p3 = ptr[base[it] + (uint32_t)pp[it]];
            pp[it] = p3 == pp[it] ? (pp[it] | JUMP_DONE) : p3;
        }
    }
#pragma unroll
    for (uint32_t it = 0; it < JUMP_ITEMS; ++it)
        if (pp[it] != p[it]) {
            ptr[tid + (uint64_t)it * gridDim.x * 256u] = pp[it];
            changed = changed || (pp[it] >= 0 && (pp[it] & JUMP_DONE) == 0);   // still on its way
        }
    // "this round changed something", one word for the whole launch: raised with a plain store once it has been seen clear (a quarter of
    // a million waves each doing an atomic OR on this ONE address was what a busy round took its time for: ~0.87 ms of atomics in a
    // row at the L2 for 15.7 M bytes, three such rounds per uncompress() of bench.py's stream -- profiles/r05_uncompress_trace.txt)
    if (__ballot(changed) != 0ull && zmi_lane() == 0u) {
#ifdef ZMI_EMU
        flags[round] = 1u;
#else
        if (__hip_atomic_load(&flags[round], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
            __hip_atomic_store(&flags[round], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
}

__global__ void __launch_bounds__(256) zmi_jump_gather_kernel(uint8_t* out, const uint64_t* __restrict__ out_off,
                                                              const uint32_t* __restrict__ out_len, uint32_t n,
                                                              const uint64_t* __restrict__ bm_off, const int32_t* __restrict__ ptr,
                                                              uint64_t n_idx) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    for (uint32_t it = 0; it < JUMP_ITEMS; ++it) {
        const uint64_t idx = tid + (uint64_t)it * gridDim.x * 256u;
        if (idx >= n_idx) return;
        const JumpWhere W = jump_where(idx, bm_off, out_len, n);
        if (W.s == 0xFFFFFFFFu) continue;
        int32_t p = ptr[idx];
        if (p == (int32_t)W.b) continue;
        if (p >= 0) p &= ~JUMP_DONE;
        uint8_t* dst = out + out_off[W.s];
        dst[W.b] = dst[(int64_t)p];   // a root: a literal of this stream, or (p < 0) a byte of the history in front of it
    }
}

// ---- one stream decoded as SEGMENTS (zmi_inflate_split, zmi_api.hip) ----------------------------------------------------
// The segments of a stream that was cut at its flush points are decoded side by side, each into a region of its own with a
// bitmap of its own; their outputs are then copied back to back into the final buffer (records included) and resolved as
// ONE stream: byte g of the final output belongs to segment j = the last one with soff[j] <= g, its hole (if any) is found in
// that segment's bitmap at g - soff[j], its pointer is g - distance in the stream's coordinates.  A segment behind the first
// was decoded with "32 KiB of history are there" taken on trust: a pointer that reaches in front of the history that really
// exists raises *err (the caller then falls back to the serial decode, which reports the reference's error).
__global__ void __launch_bounds__(256) zmi_jump_init_seg_kernel(const uint8_t* __restrict__ fin, const uint64_t* __restrict__ soff,
                                                                uint32_t nseg, const uint64_t* __restrict__ bitmap,
                                                                const uint64_t* __restrict__ bm_off, int32_t* __restrict__ ptr,
                                                                uint64_t total, uint32_t hist_len, uint32_t* __restrict__ flags,
                                                                uint32_t* __restrict__ err) {
    const uint64_t tid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (tid == 0) { for (uint32_t r = 0; r < 40u; ++r) flags[r] = 0u; }
    for (uint32_t it = 0; it < JUMP_ITEMS; ++it) {
        const uint64_t g = tid + (uint64_t)it * gridDim.x * 256u;
        if (g >= total) return;
        uint32_t lo = 0, hi = nseg;   // the last segment that starts at or below g
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (soff[mid] <= g) lo = mid; else hi = mid;
        }
        const uint32_t b = (uint32_t)(g - soff[lo]);
        const uint64_t* bm = bitmap + bm_off[lo];
        const uint8_t* dst = fin + soff[lo];
        int32_t p = (int32_t)g;
        uint32_t w = b >> 6;
        uint64_t m = bm[w] & ((2ull << (b & 63u)) - 1ull);
        for (uint32_t k = 0; k < 5u && m == 0ull && w > 0u; ++k) m = bm[--w];
        if (m != 0ull) {
            const uint32_t q = (w << 6) + 63u - (uint32_t)__clzll((unsigned long long)m);
            if (b - q < 258u) {
                const uint32_t rec = (uint32_t)dst[q] | ((uint32_t)dst[q + 1u] << 8) | ((uint32_t)dst[q + 2u] << 16);
                const uint32_t dist = (rec & 0x7FFFu) + 1u, len = (rec >> 15) + 3u;
                if (b - q < len) {
                    p = (int32_t)g - (int32_t)dist;
                    if ((int64_t)g - (int64_t)dist < -(int64_t)hist_len) { atomicOr(err, 1u); p = (int32_t)g; }
                }
            }
        }
        ptr[g] = p;
    }
}

// the final buffer as one stream: d_one_off[0] = 0 (its bitmap offset is not used by the rounds: only the index base),
// d_one_len[0] = total
extern "C" int zmi_launch_resolve_jump_segments(uint8_t* d_fin, const uint64_t* d_soff, uint32_t nseg, const uint64_t* d_bitmap,
                                                const uint64_t* d_bm_off, int32_t* d_ptr, uint64_t total, uint32_t hist_len,
                                                uint32_t rounds, uint32_t* d_flags, uint32_t* d_err, const uint64_t* d_one_off,
                                                const uint32_t* d_one_len, hipStream_t stream) {
    if (nseg == 0 || total == 0) return 0;
    if (rounds > 40u) rounds = 40u;
    const uint32_t grid = (uint32_t)((total + 256u * JUMP_ITEMS - 1u) / (256u * JUMP_ITEMS));
    ZMI_LAUNCH(zmi_jump_init_seg_kernel, dim3(grid), dim3(256), 0, stream, (const uint8_t*)d_fin, d_soff, nseg, d_bitmap, d_bm_off, d_ptr,
               total, hist_len, d_flags, d_err);
    for (uint32_t r = 0; r < rounds; ++r)
        ZMI_LAUNCH(zmi_jump_round_kernel, dim3(grid), dim3(256), 0, stream, d_one_len, 1u, d_one_off, d_ptr, total, d_flags, r);
    ZMI_LAUNCH(zmi_jump_gather_kernel, dim3(grid), dim3(256), 0, stream, d_fin, d_one_off, d_one_len, 1u, d_one_off, (const int32_t*)d_ptr, total);
    return 0;
}

// rounds: ceil(log2(largest possible output of one stream)) + 1, from the host's bound on the capacities
extern "C" int zmi_launch_resolve_jump(uint8_t* d_out, const uint64_t* d_out_off, const uint32_t* d_out_len, uint32_t n_streams,
                                       const uint64_t* d_bitmap, const uint64_t* d_bm_off, int32_t* d_ptr, uint64_t n_idx,
                                       uint32_t rounds, uint32_t* d_flags, hipStream_t stream) {
    if (n_streams == 0 || n_idx == 0) return 0;
    if (rounds > 40u) rounds = 40u;
    const uint32_t grid = (uint32_t)((n_idx + 256u * JUMP_ITEMS - 1u) / (256u * JUMP_ITEMS));
    ZMI_LAUNCH(zmi_jump_init_kernel, dim3(grid), dim3(256), 0, stream, (const uint8_t*)d_out, d_out_off, d_out_len, n_streams, d_bitmap,
               d_bm_off, d_ptr, n_idx, d_flags);
    for (uint32_t r = 0; r < rounds; ++r)
        ZMI_LAUNCH(zmi_jump_round_kernel, dim3(grid), dim3(256), 0, stream, d_out_len, n_streams, d_bm_off, d_ptr, n_idx, d_flags, r);
    ZMI_LAUNCH(zmi_jump_gather_kernel, dim3(grid), dim3(256), 0, stream, d_out, d_out_off, d_out_len, n_streams, d_bm_off,
               (const int32_t*)d_ptr, n_idx);
    return 0;
}
