// hip_emu.h -- single-threaded SIMT emulator used ONLY by the CPU-side kernel unit tests.
//
// TEST INFRASTRUCTURE, NOT A PRODUCT PATH.  The shipped library (libzmi355.so) is built by hipcc
// for gfx950 only and never sees this header.  The container that edits this repo has no GPU and
// GPU minutes are rationed, so the HIP kernels in zlib_rs_amd/csrc/*.hip are additionally compiled
// by g++ against this emulator (tests/emu/Makefile, -DZMI_EMU) to shake out logic errors, out of
// bounds LDS/global accesses (ASan) and barrier mismatches before they reach an MI355X.
//
// Model: one OS thread; every HIP thread of a workgroup is a ucontext fiber; workgroups run one
// after the other.  A fiber runs until it reaches a synchronisation point (__syncthreads or a
// wave-level exchange such as __ballot/__shfl) where it parks until its peers arrive.  Wave width
// is 64 as on CDNA4.  Cross-lane operations must be reached by every live lane of the wave
// (wave-uniform control flow) -- the kernels are written that way.
#pragma once
// fiber switch: on x86-64 a dozen instructions of our own (callee-saved registers + stack pointer); glibc's
// swapcontext makes a signal-mask system call per switch, which was half of the CPU test suite's run time
#if defined(__x86_64__) && !defined(EMU_UCONTEXT)
#define EMU_FAST_SWITCH 1
#else
#include <ucontext.h>
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint3_emu { unsigned x, y, z; };

namespace emu {
struct Fiber {
#ifdef EMU_FAST_SWITCH
    void* sp = nullptr;
#else
    ucontext_t ctx;
#endif
    void* stack = nullptr;
    bool done = false;
    volatile int* wait_var = nullptr;
    int wait_val = 0;
    unsigned tid = 0;
};
struct WaveSync {
    int arrived = 0;
    int gen = 0;
    int live = 0;
    uint64_t vals[2][64];
    uint64_t pred[2];
};
struct State {
#ifdef EMU_FAST_SWITCH
    void* sched = nullptr;
#else
    ucontext_t sched;
#endif
    std::vector<Fiber> fibers;
    std::vector<WaveSync> waves;
    int bar_arrived = 0;
    int bar_gen = 0;
    int live_threads = 0;
    unsigned cur = 0;
    unsigned char* dyn_smem = nullptr;
    std::function<void()> body;
    volatile int pass_counter = 0;   // bumped once per scheduler pass (spin_yield parks for one pass)
    bool last_yield_was_spin = false;
};
extern State g;
extern uint3_emu threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

void yield_wait(volatile int* var, int val);
// polling loops (LDS flags between waves) must give the other fibers a turn
void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body);

// rendezvous of all live lanes in the caller's wave; returns generation index used (for vals[])
inline int wave_rendezvous(uint64_t v, bool p) {
    unsigned tid = threadIdx.x;
    WaveSync& w = g.waves[tid >> 6];
    int gen = w.gen;
    unsigned lane = tid & 63;
    if (w.arrived == 0) w.pred[gen & 1] = 0;
    w.vals[gen & 1][lane] = v;
    if (p) w.pred[gen & 1] |= (1ull << lane);
    if (++w.arrived == w.live) {
        w.arrived = 0;
        w.gen = gen + 1;
        // the releasing lane parks for one scheduler pass too, so that after every rendezvous the
        // lanes resume in lane order (same-address LDS atomics of one wave instruction then apply
        // in lane order, as on the hardware)
        yield_wait(&w.gen, gen);
    } else {
        yield_wait(&w.gen, gen);
    }
    return gen & 1;
}
inline void spin_yield() {
    g.last_yield_was_spin = true;
    yield_wait(&g.pass_counter, g.pass_counter);
}
}  // namespace emu

using emu::threadIdx;
using emu::blockIdx;
using emu::blockDim;
using emu::gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

#define ZMI_DYN_SMEM(name) unsigned char* name = emu::g.dyn_smem
#define ZMI_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })

typedef void* hipStream_t;
struct uint4 { unsigned x, y, z, w; };

// ---- host runtime stubs: "device" memory is host memory ----
typedef int hipError_t;
#define hipSuccess 0
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
#define hipHostMallocDefault 0u
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
typedef int hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = 0; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = 0; return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }

inline void __syncthreads() {
    int gen = emu::g.bar_gen;
    if (++emu::g.bar_arrived == emu::g.live_threads) {
        emu::g.bar_arrived = 0;
        emu::g.bar_gen = gen + 1;
    } else {
        emu::yield_wait(&emu::g.bar_gen, gen);
    }
}

// ---- wave-level primitives (64 lanes) ----
inline uint64_t __ballot(int pred) {
    int s = emu::wave_rendezvous(0, pred != 0);
    return emu::g.waves[threadIdx.x >> 6].pred[s];
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) {
    emu::WaveSync& w = emu::g.waves[threadIdx.x >> 6];
    uint64_t b = __ballot(pred);
    uint64_t full = (w.live >= 64) ? ~0ull : ((1ull << w.live) - 1);
    return b == full;
}
template <typename T>
inline T __shfl(T v, int src, int width = 64) {
    (void)width;
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    int s = emu::wave_rendezvous(raw, false);
    uint64_t r = emu::g.waves[threadIdx.x >> 6].vals[s][src & 63];
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
template <typename T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
    (void)width;
    int lane = threadIdx.x & 63;
    int src = lane - (int)d;
    T r = __shfl(v, src < 0 ? lane : src);
    return src < 0 ? v : r;
}
template <typename T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
    (void)width;
    int lane = threadIdx.x & 63;
    int src = lane + (int)d;
    T r = __shfl(v, src > 63 ? lane : src);
    return src > 63 ? v : r;
}
template <typename T>
inline T __shfl_xor(T v, int m, int width = 64) {
    (void)width;
    int lane = threadIdx.x & 63;
    return __shfl(v, lane ^ m);
}
inline int __builtin_amdgcn_readfirstlane(int v) {
    // first live lane == lane 0 in wave-uniform code
    return __shfl(v, 0);
}
inline int __builtin_amdgcn_ds_bpermute(int byte_idx, int v) { return __shfl(v, (byte_idx >> 2) & 63); }
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned add) {
    unsigned lane = threadIdx.x & 63;
    unsigned m = lane >= 32 ? mask : (mask & ((1u << lane) - 1));
    return add + __builtin_popcount(m);
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned add) {
    unsigned lane = threadIdx.x & 63;
    unsigned m = lane < 32 ? 0 : (mask & ((1u << (lane - 32)) - 1));
    return add + __builtin_popcount(m);
}
inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (unsigned)(v >> ((sh & 3) * 8));
}
inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (unsigned)(v >> (sh & 31));
}
// v_bfe_u32: offset and width are taken modulo 32; width 0 gives 0
inline unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned off, unsigned width) {
    off &= 31u; width &= 31u;
    return width ? (v >> off) & ((1u << width) - 1u) : 0u;
}
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
inline float __log2f(float x) { return log2f(x); }
inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(uint64_t v) { return __builtin_ffsll((long long)v); }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
inline int __clzll(uint64_t v) { return v ? __builtin_clzll(v) : 64; }
inline unsigned __brev(unsigned v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}

// ---- atomics (single OS thread: plain RMW) ----
template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
inline void __threadfence() {}
inline void __threadfence_block() {}
