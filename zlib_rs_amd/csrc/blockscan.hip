// blockscan.hip -- where do the blocks of a raw deflate stream start?  (zmi_inflate_blocks, zmi_api.hip)
//
// A deflate stream is decoded serially because nothing in it says where its blocks begin: the reference's inflate walks
// one block after the other (zlib-rs/src/inflate.rs:1276 ff., Mode::Type -> Table -> Len -> ...), and a stream written without
// flush points (what every ordinary compressor emits: blocks of 16 383 symbols, zlib-rs/src/deflate.rs:321) offers no byte
// aligned restart either.  But a DYNAMIC block announces itself: behind its three header bits stand HLIT / HDIST / HCLEN, a
// complete code-length code and the run-length coded lengths of two more complete codes (inflate.rs:1604-1777), and random
// bits almost never look like that.  This kernel tries EVERY bit position of the stream:
//   stage 1 (every position, a few instructions): BTYPE = 10, HLIT <= 29, HDIST <= 29, and the HCLEN code-length-code
//            lengths fill their code space exactly (Kraft sum = 1; one position in ~250 survives),
//   stage 2 (the survivors): the code lengths are decoded with that code exactly as the decoder does -- no repeat without a
//            predecessor, no run past the end, an end-of-block code, both codes complete (or the single-code forms zlib
//            accepts, inflate/inftrees.rs:78-90) -- which leaves the true block starts and, per gigabyte, a false one or none.
// The positions found are PROPOSALS: zmi_inflate_blocks decodes the stretches between them side by side and keeps a cut only
// if the decode in front of it ended exactly there (the discipline of zmi_inflate_split).  Fixed and stored blocks are not
// looked for (three header bits prove nothing); they ride along in the segment of the dynamic block in front of them.
// Cost: HBM-trivial (the stream is read once, 24 bytes per thread and 64 positions); ~40 instructions per bit position.
#include "zmi_device.h"
#include "zmi_kernels.h"

#define BS_T 256u

// bits [pos, pos + 32) of the stream (pos in bits); bytes past n read as zero
static __device__ __forceinline__ uint32_t bs_bits(const uint8_t* __restrict__ in, uint32_t n, uint64_t pos) {
    const uint32_t b = (uint32_t)(pos >> 3), s = (uint32_t)pos & 7u;
    uint64_t v = 0;
    const uint32_t mis = (uint32_t)((uintptr_t)(in + b) & 3u);   // (the stream need not start on a dword)
    if (b >= mis && b - mis + 8u <= n) {
        // two aligned dwords and a funnel shift instead of five byte loads (a survivor of stage 1 reads some 300 symbols one after
        // the other, each behind this load: the walk was 1.8 of an uncompress() call's 7.2 ms on the device)
        const uint32_t* w = (const uint32_t*)(in + b - mis);
        const uint32_t w0 = w[0], w1 = w[1];
        const uint32_t sh = 8u * mis + s;   // < 32
        return sh ? ((w0 >> sh) | (w1 << (32u - sh))) : w0;
    }
#pragma unroll
    for (uint32_t k = 0; k < 5u; ++k) v |= (uint64_t)(b + k < n ? in[b + k] : 0u) << (8u * k);
    return (uint32_t)(v >> s);
}

// stage 2: the code lengths behind a plausible header at bit `at` (the first of the three header bits).  Serial, one lane.
static __device__ bool bs_validate(const uint8_t* __restrict__ in, uint32_t n, uint64_t at) {
    const uint64_t end = 8ull * n;
    uint32_t w = bs_bits(in, n, at);
    const uint32_t nlen = ((w >> 3) & 31u) + 257u, ndist = ((w >> 8) & 31u) + 1u, ncode = ((w >> 13) & 15u) + 4u;
    uint64_t p = at + 17u;
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19];
    for (uint32_t i = 0; i < 19u; ++i) cl[i] = 0;
    for (uint32_t i = 0; i < ncode; ++i) { cl[order[i]] = (uint8_t)(bs_bits(in, n, p) & 7u); p += 3u; }
    // canonical decoding of the code-length code, count / first-code form (lengths 1..7)
    uint32_t cnt[8];
    for (uint32_t l = 0; l < 8u; ++l) cnt[l] = 0;
    for (uint32_t i = 0; i < 19u; ++i) cnt[cl[i]]++;
    uint8_t sorted[19];
    {
        uint32_t offs[8];
        offs[1] = 0;
        for (uint32_t l = 1; l < 7u; ++l) offs[l + 1u] = offs[l] + cnt[l];
        for (uint32_t i = 0; i < 19u; ++i)
            if (cl[i]) sorted[offs[cl[i]]++] = (uint8_t)i;
    }
    const uint32_t total = nlen + ndist;
    uint32_t have = 0, prev = 0;
    uint32_t lcnt[16], dcnt[16];   // lengths seen, literal / length code and distance code
    for (uint32_t l = 0; l < 16u; ++l) { lcnt[l] = 0; dcnt[l] = 0; }
    bool eob = false;
    while (have < total) {
        if (p + 7u > end + 32u) return false;   // (runs out of the stream: the real decoder would ask for more input, not a block to cut at)
        const uint32_t bits = bs_bits(in, n, p);
        // decode one symbol, bit by bit (MSB of the code first: the stream holds it bit-reversed)
        uint32_t code = 0, first = 0, index = 0, sym = 0xFFu, len = 0;
        for (uint32_t l = 1; l <= 7u; ++l) {
            code |= (bits >> (l - 1u)) & 1u;
            const uint32_t c = cnt[l];
            if (code < first + c) { sym = sorted[index + (code - first)]; len = l; break; }
            index += c;
            first = (first + c) << 1;
            code <<= 1;
        }
        if (sym == 0xFFu) return false;
        p += len;
        uint32_t rep = 1, val = sym;
        if (sym == 16u) {
            if (have == 0u) return false;                           // "invalid bit length repeat"
            rep = 3u + ((bits >> len) & 3u); p += 2u; val = prev;
        } else if (sym == 17u) { rep = 3u + ((bits >> len) & 7u); p += 3u; val = 0u; }
        else if (sym == 18u) { rep = 11u + ((bits >> len) & 127u); p += 7u; val = 0u; }
        if (have + rep > total) return false;                       // a run past the last length
        {   // the run [have, have + rep): what lies below nlen counts for the literal / length code, the rest for the distance code
            const uint32_t inl = have >= nlen ? 0u : (have + rep <= nlen ? rep : nlen - have);
            lcnt[val] += inl;
            dcnt[val] += rep - inl;
            if (have <= 256u && 256u < have + inl && val != 0u) eob = true;
        }
        have += rep;
        prev = val;
    }
    if (p > end) return false;
    if (!eob) return false;                                         // "invalid code -- missing end-of-block"
    // both codes: not over-subscribed; incomplete only in the single-code form (inflate/inftrees.rs:78-90)
    for (int which = 0; which < 2; ++which) {
        const uint32_t* c = which ? dcnt : lcnt;
        int32_t left = 1;
        uint32_t maxl = 0;
        for (uint32_t l = 1; l <= 15u; ++l) {
            left <<= 1;
            left -= (int32_t)c[l];
            if (left < 0) return false;
            if (c[l]) maxl = l;
        }
        if (left > 0 && maxl != 0u && maxl != 1u) return false;
    }
    return true;
}

// Stage 1: thread g looks at the 64 bit positions [64 g, 64 g + 64) of in[0 .. n); positions below first_bit are skipped.
// pre[] receives the positions that pass (any order), *pre_count their number (may exceed pre_cap: the list is then incomplete
// and the caller falls back to the serial decode).  Stage 2 runs as a launch of its own, one lane per survivor: inside this
// loop a survivor's ~300-symbol walk would hold the other 63 lanes of its wave (measured: 3.4 ms of a 6.6 MB stream's scan).
__global__ void __launch_bounds__(BS_T) zmi_block_scan_kernel(const uint8_t* __restrict__ in, uint32_t n, uint64_t first_bit,
                                                               uint32_t* __restrict__ pre, uint32_t pre_cap, uint32_t* __restrict__ pre_count) {
    const uint64_t g = (uint64_t)blockIdx.x * BS_T + threadIdx.x;
    const uint64_t p0 = g * 64u;
    const bool live = p0 < 8ull * n;   // (no early return: the wave hands its survivors in together, below)
    // 24 bytes behind the thread's first position: 64 positions + 74 header bits
    const uint32_t b0 = live ? (uint32_t)(p0 >> 3) : 0u;
    uint64_t q[3];
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) {
        uint64_t v = 0;
        const uint32_t o = b0 + 8u * k;
        if (o + 8u <= n) {
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j) v |= (uint64_t)in[o + j] << (8u * j);
        } else {
            for (uint32_t j = 0; j < 8u; ++j) v |= (uint64_t)(o + j < n ? in[o + j] : 0u) << (8u * j);
        }
        q[k] = v;
    }
    uint64_t mine = 0ull;   // bit i: position p0 + i passed
    for (uint32_t i = 0; i < 64u; ++i) {
        const uint64_t pos = p0 + i;
        if (!live || pos < first_bit || pos + 17u + 12u > 8ull * n) continue;
        // 128 bits starting at position i of the 192 loaded
        const uint64_t lo = i ? (q[0] >> i) | (q[1] << (64u - i)) : q[0];
        const uint64_t hi = i ? (q[1] >> i) | (q[2] << (64u - i)) : q[1];
        const uint32_t h = (uint32_t)lo;
        if (((h >> 1) & 3u) != 2u) continue;                 // BTYPE = 10 (dynamic)
        if (((h >> 3) & 31u) > 29u) continue;                // HLIT: at most 286 literal / length codes
        if (((h >> 8) & 31u) > 29u) continue;                // HDIST: at most 30 distance codes
        const uint32_t ncode = ((h >> 13) & 15u) + 4u;
        // the code-length code must be complete: sum of 2^(7 - l) over the lengths = 128
        uint32_t kraft = 0;
        for (uint32_t k = 0; k < ncode; ++k) {
            const uint32_t sh = 17u + 3u * k;
            const uint32_t l = (uint32_t)(sh < 64u ? (sh + 3u <= 64u ? (lo >> sh) : ((lo >> sh) | (hi << (64u - sh)))) : (hi >> (sh - 64u))) & 7u;
            kraft += l ? (128u >> l) : 0u;
        }
        if (kraft != 128u) continue;
        mine |= 1ull << i;
    }
    // One atomic per WAVE (4096 positions, ~8 survivors), not per survivor: some 100 000 atomic adds on this one word, one after the
    // other at the L2, were most of what this kernel took (0.56 ms for a 6.6 MB stream).
    const uint32_t c = (uint32_t)__popcll((unsigned long long)mine);
    const uint32_t incl = zmi_wave_incl_scan(c);
    const uint32_t total = zmi_readlane(incl, 63u);
    if (total == 0u) return;   // (wave-uniform)
    uint32_t base = 0u;
    if (zmi_lane() == 0u) base = atomicAdd(pre_count, total);
    base = zmi_readlane(base, 0u);
    uint32_t k = base + incl - c;
    while (mine) {
        const uint32_t i = (uint32_t)__ffsll((unsigned long long)mine) - 1u;
        mine &= mine - 1ull;
        if (k < pre_cap) pre[k] = (uint32_t)(p0 + i);
        ++k;
    }
}

// Stage 2: one lane per stage-1 survivor.  list[] receives the positions whose code lengths decode exactly (any order),
// count[0] their number (may exceed cap), count[2] = 1 if stage 1 overflowed its list (the result is incomplete).
__global__ void __launch_bounds__(BS_T) zmi_block_validate_kernel(const uint8_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ pre,
                                                                  uint32_t pre_cap, const uint32_t* __restrict__ pre_count,
                                                                  uint32_t* __restrict__ list, uint32_t cap, uint32_t* __restrict__ count) {
    const uint32_t t = blockIdx.x * BS_T + threadIdx.x;
    const uint32_t have = *pre_count;
    if (t == 0u && have > pre_cap) count[2] = 1u;
    if (t >= have || t >= pre_cap) return;
    const uint32_t pos = pre[t];
    if (!bs_validate(in, n, pos)) return;
    const uint32_t k = atomicAdd(count, 1u);
    if (k < cap) list[k] = pos;
}

// d_count: four zeroed words {found, -, stage-1 overflow, stage-1 survivors}; d_pre: pre_cap words of scratch
extern "C" int zmi_launch_block_scan(const uint8_t* d_in, uint32_t n, uint64_t first_bit, uint32_t* d_pre, uint32_t pre_cap, uint32_t* d_list,
                                     uint32_t cap, uint32_t* d_count, hipStream_t stream) {
    if (n == 0 || pre_cap == 0) return 0;
    const uint64_t threads = (8ull * n + 63u) / 64u;
    ZMI_LAUNCH(zmi_block_scan_kernel, dim3((uint32_t)((threads + BS_T - 1u) / BS_T)), dim3(BS_T), 0, stream, d_in, n, first_bit, d_pre, pre_cap,
               d_count + 3);
    ZMI_LAUNCH(zmi_block_validate_kernel, dim3((pre_cap + BS_T - 1u) / BS_T), dim3(BS_T), 0, stream, d_in, n, (const uint32_t*)d_pre, pre_cap,
               (const uint32_t*)(d_count + 3), d_list, cap, d_count);
    return 0;
}
