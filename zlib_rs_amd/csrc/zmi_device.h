// zmi_device.h -- the handful of device-side primitives every kernel in this directory uses.
//
// Product build: hipcc --offload-arch=gfx950 (wave64, CDNA4).  There is no other GPU backend.
// The only alternative definition of these names is the CPU SIMT emulator used by the kernel
// unit tests (tests/emu/hip_emu.h, selected with -DZMI_EMU); it is test infrastructure and is
// never part of libzmi355.so.
#pragma once
#include <stdint.h>

#ifdef ZMI_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
// dynamic LDS base must stay 16-byte aligned (guide: Guideline 17)
#define ZMI_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define ZMI_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
#endif

#define ZMI_WAVE 64

// Orders LDS/global traffic between the lanes of ONE wave (a wave executes in lockstep on the
// hardware, so this only has to stop the compiler from moving memory operations across it).
#ifdef ZMI_EMU
static inline void zmi_wave_sync() { emu::wave_rendezvous(0, false); }
static inline void zmi_wave_order() { emu::wave_rendezvous(0, false); }
#else
// program-order point for cross-lane traffic through LDS or through this wave's own global stores: the
// hardware executes a wave's memory instructions in order, so only the compiler has to be held back
static __device__ __forceinline__ void zmi_wave_order() { __builtin_amdgcn_wave_barrier(); }
static __device__ __forceinline__ void zmi_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
#endif

// nothing moves across this point when the compiler schedules instructions (bounds how many loads of an unrolled loop it
// keeps in flight -- and in registers -- at once)
#ifdef ZMI_EMU
static inline void zmi_sched_fence() {}
#else
static __device__ __forceinline__ void zmi_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
#endif

// the value of the lane below; lane 0 reads 0.  DPP wave_shr:1 with bound_ctrl -- one VALU move, no LDS crossbar round
// trip, no result register parked until it arrives and no fill value to set up (four ds_bpermute per step spilled the
// hash builder of lz77.hip to scratch)
#ifdef ZMI_EMU
static inline uint32_t zmi_lane_up1(uint32_t v) {
    const uint32_t r = (uint32_t)__shfl_up((int)v, 1u);
    return (threadIdx.x & 63u) == 0u ? 0u : r;
}
#else
static __device__ __forceinline__ uint32_t zmi_lane_up1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xF, 0xF, true);
}
#endif

// the value of the lane above; lane 63 reads `fill` (DPP wave_shl:1, lanes without a source keep the old value)
#ifdef ZMI_EMU
static inline uint32_t zmi_lane_down1(uint32_t v, uint32_t fill) {
    const uint32_t r = (uint32_t)__shfl_down((int)v, 1u);
    return (threadIdx.x & 63u) == 63u ? fill : r;
}
#else
static __device__ __forceinline__ uint32_t zmi_lane_down1(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xF, 0xF, false);
}
#endif

// 1 << (lane & 31), computed where it stands: the compiler would keep this loop-invariant value in a register of its own
// across the whole tokeniser loop, and that register is the one that costs the encode kernel a wave per SIMD
#ifdef ZMI_EMU
static inline uint32_t zmi_lane_bit32_here(uint32_t lane) { return 1u << (lane & 31u); }
#else
static __device__ __forceinline__ uint32_t zmi_lane_bit32_here(uint32_t lane) {
    uint32_t r;
    asm volatile("v_lshlrev_b32 %0, %1, 1" : "=v"(r) : "v"(lane));   // the shift count is taken modulo 32
    return r;
}
#endif

// tell the compiler a value is the same in every lane (it then lives in an SGPR and branches on it are scalar)
#ifdef ZMI_EMU
static inline uint32_t zmi_uniform(uint32_t v) { return v; }
#else
static __device__ __forceinline__ uint32_t zmi_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
#endif

// `old` with lane `k` (wave-uniform) replaced by the wave-uniform value `v`
#ifdef ZMI_EMU
static inline uint32_t zmi_writelane(uint32_t old, uint32_t v, uint32_t k) { return (threadIdx.x & 63u) == k ? v : old; }
#else
static __device__ __forceinline__ uint32_t zmi_writelane(uint32_t old, uint32_t v, uint32_t k) {
    // (no writelane builtin in this compiler) value and lane select must be scalar: both go through readfirstlane
    const uint32_t sv = zmi_uniform(v), sk = zmi_uniform(k);
    // two different SGPR operands would exceed the constant-bus limit of a gfx9 VALU instruction: the lane select goes via M0
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(sv), "s"(sk) : "m0");
    return old;
}
#endif


// index of the lowest set bit, 0xFFFFFFFF for 0 (v_ffbl_b32's native result; written as asm because the compiler's
// own ctz forms either add a compare + select for the zero case or make it undefined)
#ifdef ZMI_EMU
static inline uint32_t zmi_ffbl(uint32_t x) { return x ? (uint32_t)__builtin_ctz(x) : 0xFFFFFFFFu; }
#else
static __device__ __forceinline__ uint32_t zmi_ffbl(uint32_t x) {
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
#endif

// value of `v` in lane `k` (k wave-uniform), as a scalar
#ifdef ZMI_EMU
static inline uint32_t zmi_readlane(uint32_t v, uint32_t k) { return (uint32_t)__shfl((int)v, (int)k); }
#else
static __device__ __forceinline__ uint32_t zmi_readlane(uint32_t v, uint32_t k) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)k);
}
#endif

// Workgroup -> work item.  The dispatcher deals consecutive workgroups round-robin over the 8 XCDs and, inside
// an XCD, round-robin over its 4 shader engines (measured: with one workgroup per shard and eight data classes
// laid out with period 8, launch time followed the slowest class pair).  So with the identity mapping every
// XCD / shader engine would see only a few residues of the shard index, and any cost pattern with a small
// period in the batch would load them unevenly while the launch waits for the slowest.  Each group of eight
// consecutive workgroups therefore takes its eight shards in an order rotated by a hash of the group number:
// every XCD and every shader engine sees all residues, neighbouring shards stay on neighbouring workgroups.
static __device__ __forceinline__ unsigned zmi_xcd_spread(unsigned b, unsigned n) {
    const unsigned rot = ((b >> 3) * 0x9E3779B1u) >> 29;
    return b < (n & ~7u) ? ((b & ~7u) | ((b + rot) & 7u)) : b;
}

static __device__ __forceinline__ unsigned zmi_lane() { return threadIdx.x & 63u; }
static __device__ __forceinline__ unsigned zmi_wave() { return threadIdx.x >> 6; }

// number of set bits of `mask` strictly below this lane
static __device__ __forceinline__ unsigned zmi_mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// wave-wide inclusive prefix sum (all 64 lanes must call)
#ifdef ZMI_EMU
static __device__ __forceinline__ unsigned zmi_wave_incl_scan(unsigned v) {
    unsigned lane = zmi_lane();
    for (int d = 1; d < 64; d <<= 1) {
        unsigned t = __shfl_up(v, d);
        if (lane >= (unsigned)d) v += t;
    }
    return v;
}
#else
// DPP version: row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15 (rows 1,3) and
// row_bcast:31 (rows 2,3).  Six VALU instructions, no LDS crossbar (ds_bpermute) round trips.
static __device__ __forceinline__ unsigned zmi_wave_incl_scan(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}
#endif
// wave-wide sum / maximum, the same value in every lane (all 64 lanes must call).  On the device: the DPP prefix steps of the
// scan above and one scalar lane read -- seven instructions; the butterfly of six __shfl_xor it replaces (round 4) is six
// ds_bpermute round trips through the LDS crossbar with their address arithmetic and waits.
#ifdef ZMI_EMU
static __device__ __forceinline__ unsigned zmi_wave_sum(unsigned v) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
static __device__ __forceinline__ unsigned zmi_wave_max(unsigned v) {
    for (int d = 32; d >= 1; d >>= 1) {
        unsigned t = __shfl_xor(v, d);
        v = t > v ? t : v;
    }
    return v;
}
#else
static __device__ __forceinline__ unsigned zmi_wave_sum(unsigned v) {
    return (unsigned)__builtin_amdgcn_readlane((int)zmi_wave_incl_scan(v), 63);
}
static __device__ __forceinline__ unsigned zmi_wave_max(unsigned v) {
    // (lanes without a source read 0: the identity of an unsigned maximum)
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); v = t > v ? t : v;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); v = t > v ? t : v;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); v = t > v ? t : v;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); v = t > v ? t : v;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); v = t > v ? t : v;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); v = t > v ? t : v;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
#endif

// unaligned little-endian 32-bit read from a byte array that lives in LDS or global memory:
// two aligned dword reads + v_alignbyte.  `base` must be 4-byte aligned.
static __device__ __forceinline__ uint32_t zmi_load32u(const unsigned char* base, uint32_t off) {
    const uint32_t* w = (const uint32_t*)(base + (off & ~3u));
    return __builtin_amdgcn_alignbyte(w[1], w[0], off & 3u);
}

struct zmi_b16 {
    uint32_t w[4];
};

// 16 bytes starting at p; bytes past nvalid read as zero.  Fast path needs p 16-byte aligned.
static __device__ __forceinline__ zmi_b16 zmi_ld16(const uint8_t* p, uint32_t nvalid, bool aligned) {
    zmi_b16 r;
    if (aligned && nvalid >= 16u) {
        const uint4* q = (const uint4*)p;
        uint4 v = *q;
        r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
    } else {
        // byte-wise (unaligned start or the last bytes of a shard).  Two 64-bit accumulators selected by a compare: an
        // array indexed with the loop counter would be placed in scratch memory by the compiler.
        uint64_t lo = 0, hi = 0;
        const uint32_t m = nvalid < 16u ? nvalid : 16u;
        for (uint32_t i = 0; i < m; ++i) {
            const uint64_t b = (uint64_t)p[i] << (8u * (i & 7u));
            if (i < 8u) lo |= b; else hi |= b;
        }
        r.w[0] = (uint32_t)lo; r.w[1] = (uint32_t)(lo >> 32); r.w[2] = (uint32_t)hi; r.w[3] = (uint32_t)(hi >> 32);
    }
    return r;
}

