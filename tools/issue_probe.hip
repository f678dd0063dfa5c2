// issue_probe.hip -- how many cycles does a SIMD of gfx950 spend per wave64 instruction?
//
// DESIGN.md priced the integer VALU at 4 cycles per wave64 instruction, MI355X_MICROARCH.md says 2
// (SIMD-32, two passes).  This settles it by measurement, for the instructions the deflate kernels are made of.
// Every test is a loop of 128 copies of one instruction, either INDEPENDENT (8 register chains in rotation: what the
// pipe can issue) or DEPENDENT (one chain: issue + result latency), run with 1 / 2 / 4 / 8 waves per SIMD
// (grid = 256 CUs x W workgroups of 256 threads: every SIMD of every CU holds W waves).
// Output: ns per wave-instruction per SIMD and the same in cycles at the clock measured by s_memtime calibration
// (and at the nominal 2.4 GHz).  Built by tools/gpu_issue_probe.py (hipcc, standalone: no torch).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define REP8(x) x x x x x x x x
#define CHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

enum { OP_ADD = 0, OP_ALIGNBYTE, OP_MUL24, OP_MULLO, OP_XOR, OP_FFBL, OP_CNDMASK, OP_LSHL_OR, OP_MIN3, OP_READLANE, OP_DPP_MOV, OP_SALU, OP_DS_READ, OP_DS_READ2X,
       OP_BPERMUTE, OP_DS_WRITE, OP_MIX_VS, OP_COUNT };
static const char* kNames[OP_COUNT] = {"v_add_u32", "v_alignbyte_b32", "v_mul_u32_u24", "v_mul_lo_u32", "v_xor_b32", "v_ffbl_b32", "v_cndmask_b32", "v_lshl_or_b32",
                                       "v_min3_u32", "v_readlane_b32(+v_add dep)", "v_mov_b32 dpp wave_shr:1", "s_add_u32", "ds_read_b32", "ds_read_b32 x5 (one wait)",
                                       "ds_bpermute_b32", "ds_write_b32", "v_add_u32 + s_add_u32 pairs"};

// 128 instructions per trip of the loop body
template <int OP, bool DEP>
__global__ void __launch_bounds__(256) probe(uint32_t iters, uint32_t* out, unsigned long long* ticks) {
    __shared__ uint32_t lds[4096];
    for (uint32_t i = threadIdx.x; i < 4096u; i += 256u) lds[i] = (i * 4u) & 0x3FFCu;
    __syncthreads();
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t k = blockIdx.x | 1u;
    uint32_t ad = (threadIdx.x * 4u) & 0x3FFCu;   // own LDS slot: conflict-free, every lane chases its own pointer
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t it = 0; it < iters; ++it) {
        if (OP == OP_ADD) {
            if (DEP) asm volatile(".rept 128\n v_add_u32 %0, %0, %1\n .endr" : "+v"(a0) : "v"(k));
            else asm volatile(".rept 16\n v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n .endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if (OP == OP_ALIGNBYTE) {
            if (DEP) asm volatile(".rept 128\n v_alignbyte_b32 %0, %0, %1, 1\n .endr" : "+v"(a0) : "v"(k));
            else asm volatile(".rept 16\n v_alignbyte_b32 %0, %0, %8, 1\n v_alignbyte_b32 %1, %1, %8, 1\n v_alignbyte_b32 %2, %2, %8, 1\n v_alignbyte_b32 %3, %3, %8, 1\n v_alignbyte_b32 %4, %4, %8, 1\n v_alignbyte_b32 %5, %5, %8, 1\n v_alignbyte_b32 %6, %6, %8, 1\n v_alignbyte_b32 %7, %7, %8, 1\n .endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if (OP == OP_MUL24) {
            if (DEP) asm volatile(".rept 128\n v_mul_u32_u24 %0, %0, %1\n .endr" : "+v"(a0) : "v"(k));
            else asm volatile(".rept 16\n v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n .endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if (OP == OP_MULLO) {
            if (DEP) asm volatile(".rept 128\n v_mul_lo_u32 %0, %0, %1\n .endr" : "+v"(a0) : "v"(k));
            else asm volatile(".rept 16\n v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n .endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if (OP == OP_XOR) {
            if (DEP) asm volatile(".rept 128\n v_xor_b32 %0, %0, %1\n .endr" : "+v"(a0) : "v"(k));
            else asm volatile(".rept 16\n v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n .endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if (OP == OP_FFBL) {
            if (DEP) asm volatile(".rept 128\n v_ffbl_b32 %0, %0\n .endr" : "+v"(a0));
            else asm volatile(".rept 16\n v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3\n v_ffbl_b32 %4, %4\n v_ffbl_b32 %5, %5\n v_ffbl_b32 %6, %6\n v_ffbl_b32 %7, %7\n .endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (OP == OP_CNDMASK) {
            if (DEP) asm volatile(".rept 128\n v_cndmask_b32 %0, %0, %1, vcc\n .endr" : "+v"(a0) : "v"(k) : "vcc");
            else asm volatile(".rept 16\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n .endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");
        } else if (OP == OP_LSHL_OR) {
            if (DEP) asm volatile(".rept 128\n v_lshl_or_b32 %0, %0, 1, %1\n .endr" : "+v"(a0) : "v"(k));
            else asm volatile(".rept 16\n v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %8\n v_lshl_or_b32 %2, %2, 1, %8\n v_lshl_or_b32 %3, %3, 1, %8\n v_lshl_or_b32 %4, %4, 1, %8\n v_lshl_or_b32 %5, %5, 1, %8\n v_lshl_or_b32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %8\n .endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if (OP == OP_MIN3) {
            if (DEP) asm volatile(".rept 128\n v_min3_u32 %0, %0, %1, %1\n .endr" : "+v"(a0) : "v"(k));
            else asm volatile(".rept 16\n v_min3_u32 %0, %0, %8, %8\n v_min3_u32 %1, %1, %8, %8\n v_min3_u32 %2, %2, %8, %8\n v_min3_u32 %3, %3, %8, %8\n v_min3_u32 %4, %4, %8, %8\n v_min3_u32 %5, %5, %8, %8\n v_min3_u32 %6, %6, %8, %8\n v_min3_u32 %7, %7, %8, %8\n .endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if (OP == OP_READLANE) {
            // dependent: lane read -> scalar -> vector add (64 pairs = 128 instructions); independent: 128 lane reads into 8 SGPRs
            if (DEP) asm volatile(".rept 64\n v_readlane_b32 s20, %0, 5\n s_nop 0\n v_add_u32 %0, s20, %0\n .endr" : "+v"(a0) : : "s20");
            else asm volatile(".rept 16\n v_readlane_b32 s20, %0, 1\n v_readlane_b32 s21, %0, 2\n v_readlane_b32 s22, %0, 3\n v_readlane_b32 s23, %0, 4\n v_readlane_b32 s24, %0, 5\n v_readlane_b32 s25, %0, 6\n v_readlane_b32 s26, %0, 7\n v_readlane_b32 s27, %0, 8\n .endr"
                              : "+v"(a0) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if (OP == OP_DPP_MOV) {
            if (DEP) asm volatile(".rept 128\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n .endr" : "+v"(a0));
            else asm volatile(".rept 16\n v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n .endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
        } else if (OP == OP_SALU) {
            if (DEP) asm volatile(".rept 128\n s_add_u32 s20, s20, 3\n .endr" : : : "s20", "scc");
            else asm volatile(".rept 16\n s_add_u32 s20, s20, 3\n s_add_u32 s21, s21, 3\n s_add_u32 s22, s22, 3\n s_add_u32 s23, s23, 3\n s_add_u32 s24, s24, 3\n s_add_u32 s25, s25, 3\n s_add_u32 s26, s26, 3\n s_add_u32 s27, s27, 3\n .endr"
                              : : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        } else if (OP == OP_DS_READ) {
            // dependent: pointer chase (address = value read); independent: 8 reads in flight per wait
            if (DEP) asm volatile(".rept 128\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n .endr" : "+v"(ad) : : "memory");
            else asm volatile(".rept 16\n ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n .endr"
                              : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(ad) : "memory");
        } else if (OP == OP_DS_READ2X) {
            // the shape of one lz77 chain step: the link + five window dwords issued together, one wait, the next address
            // depends on the first value (DEP) -- 6 reads per round trip, 126 reads + 21 waits per trip
            if (DEP) asm volatile(".rept 21\n ds_read_b32 %1, %0 offset:4\n ds_read_b32 %2, %0 offset:8\n ds_read_b32 %3, %0 offset:12\n ds_read_b32 %4, %0 offset:16\n ds_read_b32 %5, %0 offset:20\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n .endr"
                                  : "+v"(ad), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5) : : "memory");
            else asm volatile(".rept 21\n ds_read_b32 %1, %6 offset:4\n ds_read_b32 %2, %6 offset:8\n ds_read_b32 %3, %6 offset:12\n ds_read_b32 %4, %6 offset:16\n ds_read_b32 %5, %6 offset:20\n ds_read_b32 %0, %6\n s_waitcnt lgkmcnt(0)\n .endr"
                              : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5) : "v"(ad) : "memory");
        } else if (OP == OP_BPERMUTE) {
            if (DEP) asm volatile(".rept 128\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n .endr" : "+v"(a0) : "v"(ad & 0xFCu) : "memory");
            else asm volatile(".rept 16\n ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n .endr"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ad & 0xFCu) : "memory");
        } else if (OP == OP_DS_WRITE) {
            asm volatile(".rept 128\n ds_write_b32 %0, %1\n .endr\n s_waitcnt lgkmcnt(0)" : : "v"(ad), "v"(a0) : "memory");
        } else if (OP == OP_MIX_VS) {
            asm volatile(".rept 8\n v_add_u32 %0, %0, %8\n s_add_u32 s20, s20, 3\n v_add_u32 %1, %1, %8\n s_add_u32 s21, s21, 3\n v_add_u32 %2, %2, %8\n s_add_u32 s22, s22, 3\n v_add_u32 %3, %3, %8\n s_add_u32 s23, s23, 3\n v_add_u32 %4, %4, %8\n s_add_u32 s24, s24, 3\n v_add_u32 %5, %5, %8\n s_add_u32 s25, s25, 3\n v_add_u32 %6, %6, %8\n s_add_u32 s26, s26, 3\n v_add_u32 %7, %7, %8\n s_add_u32 s27, s27, 3\n .endr"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256u + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + ad;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

struct Result { double ns_per_inst_simd; double ticks_per_inst; };

template <int OP, bool DEP>
static Result run(uint32_t W, uint32_t iters, uint32_t* d_out, unsigned long long* d_ticks, int cus) {
    const uint32_t grid = (uint32_t)cus * W;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<OP, DEP>), dim3(grid), dim3(256), 0, 0, iters / 8u + 1u, d_out, d_ticks);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((probe<OP, DEP>), dim3(grid), dim3(256), 0, 0, iters, d_out, d_ticks);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned long long ticks = 0;
    CHECK(hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost));
    const double insts_per_simd = (double)iters * 128.0 * W;   // W waves share one SIMD
    Result r;
    r.ns_per_inst_simd = ms * 1e6 / insts_per_simd;
    r.ticks_per_inst = (double)ticks / ((double)iters * 128.0);   // s_memtime ticks per instruction of ONE wave
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return r;
}

template <int OP>
static void sweep(uint32_t iters, uint32_t* d_out, unsigned long long* d_ticks, int cus, double ghz) {
    const uint32_t Ws[4] = {1, 2, 4, 8};
    for (int dep = 0; dep < 2; ++dep) {
        if (OP == OP_DS_WRITE && dep) continue;
        if (OP == OP_MIX_VS && dep) continue;
        printf("%-30s %-11s", kNames[OP], dep ? "dependent" : "independent");
        for (uint32_t W : Ws) {
            Result r = dep ? run<OP, true>(W, iters, d_out, d_ticks, cus) : run<OP, false>(W, iters, d_out, d_ticks, cus);
            printf("  W=%u: %6.3f ns = %5.2f cyc", W, r.ns_per_inst_simd, r.ns_per_inst_simd * ghz);
        }
        printf("\n");
    }
}

int main(int argc, char** argv) {
    uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000u;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint32_t* d_out;
    unsigned long long* d_ticks;
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4));
    CHECK(hipMalloc(&d_ticks, 8));
    // clock calibration: a dependent v_add_u32 chain of one wave per SIMD, timed by HIP events and by s_memtime
    Result cal = run<OP_ADD, true>(1, iters * 4u, d_out, d_ticks, cus);
    const double nominal = p.clockRate / 1e6;   // GHz
    printf("# device %s, %d CUs, clockRate %.3f GHz (hipDeviceProp), s_memtime ticks per dependent v_add_u32: %.3f, ns: %.3f  => s_memtime runs at %.1f MHz\n",
           p.name, cus, nominal, cal.ticks_per_inst, cal.ns_per_inst_simd, cal.ticks_per_inst / cal.ns_per_inst_simd * 1e3);
    const double ghz = argc > 2 ? atof(argv[2]) : 2.4;
    printf("# cycles below = ns x %.2f GHz (pass the measured shader clock as argv[2]); one line = ns (cycles) per wave64 instruction PER SIMD with W waves resident on it\n", ghz);
    printf("# independent: 8 register chains in rotation (issue rate); dependent: one chain (issue + latency; with W waves the SIMD interleaves them)\n");
    sweep<OP_ADD>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_XOR>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_ALIGNBYTE>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_LSHL_OR>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_MIN3>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_CNDMASK>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_FFBL>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_MUL24>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_MULLO>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_DPP_MOV>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_READLANE>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_SALU>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_MIX_VS>(iters, d_out, d_ticks, cus, ghz);
    sweep<OP_DS_READ>(iters / 4u, d_out, d_ticks, cus, ghz);
    sweep<OP_DS_READ2X>(iters / 4u, d_out, d_ticks, cus, ghz);
    sweep<OP_BPERMUTE>(iters / 4u, d_out, d_ticks, cus, ghz);
    sweep<OP_DS_WRITE>(iters / 4u, d_out, d_ticks, cus, ghz);
    return 0;
}
