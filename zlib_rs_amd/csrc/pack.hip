// pack.hip -- turning the batch's strided output slots into dense, ordered byte ranges (the "stitch").
//
// What the reference does here: its parallel-deflate recipe compresses pieces independently and then simply appends
// the finished byte strings in order (zlib-rs/src/deflate.rs:4145-4221 `split_deflate`; multi-member gzip is read
// back the same way, libz-rs-sys/src/gz.rs:1464-1506).  On the device the compressed shards sit in compress_bound-
// strided slots (72 GiB of slots for ~29 GiB of data at the headline size), so the stitch is two small kernels:
//   zmi_scan_sizes_kernel   exclusive prefix sum of the u32 sizes -> u64 byte offsets (n + 1 entries)
//   zmi_copy_ranges_kernel  range i: len[i] bytes from src + src_off[i] to dst + dst_off[i], any alignment
// One call packs a rank's slots into its slab (src_off = i * stride, dst_off = scan); after the slab exchange the
// same kernel scatters a peer's slab into the globally ordered output (src_off = the peer's scan, dst_off = the
// global offsets of its shards).  HBM-bound: one read + one write of the compressed bytes; 4 KiB tiles are loaded
// with aligned 16-byte reads into LDS and stored as aligned dwords (source and destination are misaligned against
// each other in general), 256-thread workgroups, several workgroups per range so that a few large ranges still fill
// the chip.
#include "zmi_device.h"
#include "zmi_kernels.h"

#define PK_T 256u
#define PK_TILE 4096u

__global__ void __launch_bounds__(1024) zmi_scan_sizes_kernel(const uint32_t* __restrict__ len, uint32_t n,
                                                               uint64_t* __restrict__ off) {
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += len[i];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint64_t v = t >= d ? part[t - d] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t o = part[t] - sum;
    for (uint32_t i = lo; i < hi; ++i) { off[i] = o; o += len[i]; }
    if (t == 1023u) off[n] = part[1023];
}

// n * split jobs: job b handles tiles b % split, b % split + split, ... of range b / split; the grid is the number of jobs, or
// fewer workgroups that take the jobs in turn (a destination in host memory: see zmi_launch_copy_ranges_few)
__global__ void __launch_bounds__(PK_T) zmi_copy_ranges_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ src_off,
                                                              uint64_t src_stride, const uint32_t* __restrict__ len,
                                                              uint8_t* __restrict__ dst, const uint64_t* __restrict__ dst_off,
                                                              uint64_t dst_cap, uint32_t split, uint32_t jobs) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[PK_TILE + 32];
    const uint32_t t = threadIdx.x;
  for (uint32_t job = blockIdx.x; job < jobs; job += gridDim.x) {
    const uint32_t r = job / split, part = job % split;
    const uint32_t l = len[r];
    const uint64_t so = src_off ? src_off[r] : (uint64_t)r * src_stride;
    const uint64_t d0 = dst_off[r];
    if (d0 + l > dst_cap) continue;   // does not fit: the caller sees that from the offsets
    const uint8_t* s = src + so;
    const uint32_t mis = (uint32_t)((uintptr_t)s & 15u);   // the tile loads start at the 16-byte line at or below s
    for (uint32_t base = part * PK_TILE; base < l; base += split * PK_TILE) {
        const uint32_t nb = l - base < PK_TILE ? l - base : PK_TILE;
        // stage[mis + k] = s[base + k]; 16-byte aligned loads, the first / last line of a range byte-wise
        const uint8_t* line0 = s + base - mis;
        const uint32_t span = mis + nb;
        for (uint32_t c = t * 16u; c < span; c += PK_T * 16u) {
            if (base + c >= mis && c + 16u <= span && (base != 0 || c >= 16u || mis == 0u)) {
                *(uint4*)(stage + c) = *(const uint4*)(line0 + c);
            } else {
                for (uint32_t j = 0; j < 16u; ++j)
                    if (c + j >= mis && c + j < span) stage[c + j] = line0[c + j];
            }
        }
        __syncthreads();
        uint8_t* A = dst + d0 + base;
        uint32_t head = (4u - (uint32_t)((uintptr_t)A & 3u)) & 3u;
        if (head > nb) head = nb;
        const uint32_t ndw = (nb - head) >> 2;
        uint32_t* A4 = (uint32_t*)(A + head);
        for (uint32_t k = t; k < ndw; k += PK_T) {
            const uint32_t o = mis + head + 4u * k;
            const uint32_t* w = (const uint32_t*)(stage + (o & ~3u));
            A4[k] = __builtin_amdgcn_alignbyte(w[1], w[0], o & 3u);
        }
        if (t == 0) {
            for (uint32_t j = 0; j < head; ++j) A[j] = stage[mis + j];
            for (uint32_t j = head + 4u * ndw; j < nb; ++j) A[j] = stage[mis + j];
        }
        __syncthreads();
    }
  }
}

// out[i] = min(len[i], cap[i]): what a stream's output region really holds (inflate counts past a full region)
__global__ void __launch_bounds__(256) zmi_clamp_lens_kernel(const uint32_t* __restrict__ len, const uint32_t* __restrict__ cap, uint32_t n,
                                                             uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = len[i] < cap[i] ? len[i] : cap[i];
}
extern "C" int zmi_launch_clamp_lens(const uint32_t* d_len, const uint32_t* d_cap, uint32_t n, uint32_t* d_out, hipStream_t stream) {
    if (n == 0) return 0;
    ZMI_LAUNCH(zmi_clamp_lens_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, d_len, d_cap, n, d_out);
    return 0;
}

extern "C" int zmi_launch_scan_sizes(const uint32_t* d_len, uint32_t n, uint64_t* d_off, hipStream_t stream) {
    ZMI_LAUNCH(zmi_scan_sizes_kernel, dim3(1), dim3(1024), 0, stream, d_len, n, d_off);
    return 0;
}

extern "C" int zmi_launch_copy_ranges(const uint8_t* d_src, const uint64_t* d_src_off, uint64_t src_stride, const uint32_t* d_len,
                                      uint32_t n, uint8_t* d_dst, const uint64_t* d_dst_off, uint64_t dst_cap, uint32_t max_len,
                                      hipStream_t stream) {
    if (n == 0) return 0;
    // a few thousand workgroups fill the chip; small batches of large ranges are split across workgroups
    uint32_t split = 1;
    const uint32_t tiles = (max_len + PK_TILE - 1u) / PK_TILE;
    while ((uint64_t)n * split < 4096u && split < tiles) split <<= 1;
    ZMI_LAUNCH(zmi_copy_ranges_kernel, dim3(n * split), dim3(PK_T), 0, stream, d_src, d_src_off, src_stride, d_len, d_dst, d_dst_off,
               dst_cap, split, n * split);
    return 0;
}

// The same copy with `groups` workgroups only, for a destination in pinned HOST memory (the host-buffer pipeline's slab): the
// stores leave over PCIe at ~50 GB/s whatever the launch looks like, and a launch of thousands of workgroups sits on every CU
// for the 5 ms that takes -- the next chunk's match search (one 1024-thread workgroup per CU, 152 KiB of LDS) could not
// start beside it.  A few dozen workgroups keep the link busy and leave the CUs to the kernels.
extern "C" int zmi_launch_copy_ranges_few(const uint8_t* d_src, const uint64_t* d_src_off, uint64_t src_stride, const uint32_t* d_len,
                                          uint32_t n, uint8_t* d_dst, const uint64_t* d_dst_off, uint64_t dst_cap, uint32_t max_len,
                                          uint32_t groups, hipStream_t stream) {
    if (n == 0) return 0;
    uint32_t split = 1;
    const uint32_t tiles = (max_len + PK_TILE - 1u) / PK_TILE;
    while ((uint64_t)n * split < 4096u && split < tiles) split <<= 1;
    // the `groups` workgroups that run at a time should work on ONE range, tile beside tile (job = range * split + part, the
    // workgroups take consecutive jobs): sixteen workgroups writing sixteen different megabytes of host memory ran at a
    // quarter of the link's rate (the inflate pipeline's per-stream ranges: 19 -> 5 GiB/s until this was fixed)
    while (split < groups && split < tiles) split <<= 1;
    const uint64_t jobs64 = (uint64_t)n * split;
    const uint32_t jobs = jobs64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)jobs64;
    ZMI_LAUNCH(zmi_copy_ranges_kernel, dim3(jobs < groups ? jobs : groups), dim3(PK_T), 0, stream, d_src, d_src_off, src_stride, d_len, d_dst,
               d_dst_off, dst_cap, split, jobs);
    return 0;
}
