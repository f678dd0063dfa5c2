// issue_probe2.hip -- second table of tools/gpu_issue_probe.py: selects, compares, and UNALIGNED LDS reads.
// (a) v_cndmask_b32 in its encodings (probe 1 measured 23 cycles for the VOP2 form reading an uninitialised VCC),
// (b) the simple VOP2 ops the kernels are made of, (c) ds_read_b32/b64/b128 at byte-unaligned addresses: gfx950 accepts
// them (hipcc emits ds_read_b128 for an align-1 16-byte load), and one unaligned 16-byte read would replace the five
// aligned dword reads + four v_alignbyte of a chain step in lz77.hip.  Address patterns: "seq" = lane i reads at
// base + i (64 consecutive positions), "rnd" = a pseudo-random byte address per lane (chain candidates).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define I8(op, tail) ".rept 16\n " op " %0, %0" tail "\n " op " %1, %1" tail "\n " op " %2, %2" tail "\n " op " %3, %3" tail "\n " op " %4, %4" tail "\n " op " %5, %5" tail "\n " op " %6, %6" tail "\n " op " %7, %7" tail "\n .endr"
#define VOUT "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

enum { P_AND = 0, P_SUB, P_LSHR, P_MINU, P_OR, P_CMP_VCC, P_CMP_SGPR, P_CND_VCC_SET, P_CND_E64, P_CND_E64_CONST, P_BFE, P_ADD3, P_ANDOR, P_PERM, P_MBCNT,
       P_CND_E64_VCC, P_CMPCND_E32, P_CMPCND_E64, P_CMPCND_E64S, P_SDWA, P_OR3, P_LDS32A, P_LDS32U, P_LDS64A, P_LDS64U, P_LDS128A, P_LDS128U, P_LDSU16, P_LDS2X32, P_COUNT };
static const char* kN[P_COUNT] = {"v_and_b32", "v_sub_u32", "v_lshrrev_b32", "v_min_u32", "v_or_b32", "v_cmp_lt_u32 -> vcc", "v_cmp_lt_u32 -> s[20:21]",
    "v_cndmask_b32_e32 (vcc set once)", "v_cndmask_b32_e64 s[20:21]", "v_cndmask_b32_e64 inline consts", "v_bfe_u32", "v_add3_u32", "v_and_or_b32", "v_perm_b32",
    "v_mbcnt_lo_u32_b32", "v_cndmask_b32_e64 vcc", "v_cmp vcc + v_cndmask_e32 vcc (pair)", "v_cmp vcc + v_cndmask_e64 vcc (pair)", "v_cmp s[20:21] + nop + v_cndmask_e64 (pair)", "v_or_b32_sdwa", "v_or3_b32", "ds_read_b32 aligned", "ds_read_b32 unaligned", "ds_read_b64 aligned", "ds_read_b64 unaligned", "ds_read_b128 aligned",
    "ds_read_b128 unaligned", "ds_read_u16", "ds_read2_b32 (8 B, 4-aligned)"};

template <int OP>
__global__ void __launch_bounds__(256) probe(uint32_t iters, uint32_t* out, uint32_t pattern) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[8192 + 64];
    for (uint32_t i = threadIdx.x; i < 8192u + 64u; i += 256u) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t k = blockIdx.x | 1u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // byte address of this lane's reads: seq = consecutive byte positions, rnd = scattered; the alignment variant decides the low bits
    uint32_t ad;
    if (pattern == 0u) ad = wave * 4096u + lane;                       // seq: 64 consecutive byte positions
    else ad = ((lane * 2654435761u + wave * 40503u) >> 17) & 0x7FFFu;   // rnd: anywhere in 32 KiB
    const bool unal = OP == P_LDS32U || OP == P_LDS64U || OP == P_LDS128U;
    if (OP == P_LDS32A || OP == P_LDS2X32) ad = pattern == 0u ? wave * 4096u + lane * 4u : (ad & ~3u);
    if (OP == P_LDS64A) ad = pattern == 0u ? wave * 4096u + lane * 8u : (ad & ~7u);
    if (OP == P_LDS128A) ad = pattern == 0u ? wave * 4096u + lane * 16u : (ad & ~15u);
    if (OP == P_LDSU16) ad = pattern == 0u ? wave * 4096u + lane * 2u : (ad & ~1u);
    if (unal && pattern == 1u) ad |= 1u;   // every lane misaligned
    uint32_t b0, b1, b2, b3;
    uint64_t q0, q1;
    for (uint32_t it = 0; it < iters; ++it) {
        if (OP == P_AND) asm volatile(I8("v_and_b32", ", %8") : VOUT : "v"(k));
        else if (OP == P_SUB) asm volatile(I8("v_sub_u32", ", %8") : VOUT : "v"(k));
        else if (OP == P_LSHR) asm volatile(".rept 16\n v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n v_lshrrev_b32 %4, 1, %4\n v_lshrrev_b32 %5, 1, %5\n v_lshrrev_b32 %6, 1, %6\n v_lshrrev_b32 %7, 1, %7\n .endr" : VOUT);
        else if (OP == P_MINU) asm volatile(I8("v_min_u32", ", %8") : VOUT : "v"(k));
        else if (OP == P_OR) asm volatile(I8("v_or_b32", ", %8") : VOUT : "v"(k));
        else if (OP == P_CMP_VCC) asm volatile(".rept 128\n v_cmp_lt_u32 vcc, %0, %1\n .endr" : : "v"(a0), "v"(k) : "vcc");
        else if (OP == P_CMP_SGPR) asm volatile(".rept 128\n v_cmp_lt_u32 s[20:21], %0, %1\n .endr" : : "v"(a0), "v"(k) : "s20", "s21");
        else if (OP == P_CND_VCC_SET) asm volatile("v_cmp_lt_u32 vcc, %8, %0\n s_nop 4\n" I8("v_cndmask_b32", ", %8, vcc") : VOUT : "v"(k) : "vcc");
        else if (OP == P_CND_E64) asm volatile("v_cmp_lt_u32 s[20:21], %8, %0\n s_nop 4\n" I8("v_cndmask_b32_e64", ", %8, s[20:21]") : VOUT : "v"(k) : "s20", "s21");
        else if (OP == P_CND_E64_CONST) asm volatile("v_cmp_lt_u32 s[20:21], %8, %0\n s_nop 4\n .rept 16\n v_cndmask_b32_e64 %0, 0, 16, s[20:21]\n v_cndmask_b32_e64 %1, 0, 16, s[20:21]\n v_cndmask_b32_e64 %2, 0, 16, s[20:21]\n v_cndmask_b32_e64 %3, 0, 16, s[20:21]\n v_cndmask_b32_e64 %4, 0, 16, s[20:21]\n v_cndmask_b32_e64 %5, 0, 16, s[20:21]\n v_cndmask_b32_e64 %6, 0, 16, s[20:21]\n v_cndmask_b32_e64 %7, 0, 16, s[20:21]\n .endr" : VOUT : "v"(k) : "s20", "s21");
        else if (OP == P_BFE) asm volatile(I8("v_bfe_u32", ", 3, 9") : VOUT);
        else if (OP == P_ADD3) asm volatile(I8("v_add3_u32", ", %8, %8") : VOUT : "v"(k));
        else if (OP == P_ANDOR) asm volatile(I8("v_and_or_b32", ", %8, %8") : VOUT : "v"(k));
        else if (OP == P_PERM) asm volatile(I8("v_perm_b32", ", %8, %8") : VOUT : "v"(k));
        else if (OP == P_MBCNT) asm volatile(I8("v_mbcnt_lo_u32_b32", ", %8") : VOUT : "v"(k));
        else if (OP == P_CND_E64_VCC) asm volatile("v_cmp_lt_u32 vcc, %8, %0\n s_nop 4\n" I8("v_cndmask_b32_e64", ", %8, vcc") : VOUT : "v"(k) : "vcc");
        else if (OP == P_CMPCND_E32) asm volatile(".rept 8\n v_cmp_lt_u32 vcc, %8, %0\n v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cmp_lt_u32 vcc, %8, %1\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %8, %2\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cmp_lt_u32 vcc, %8, %3\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cmp_lt_u32 vcc, %8, %4\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cmp_lt_u32 vcc, %8, %5\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cmp_lt_u32 vcc, %8, %6\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cmp_lt_u32 vcc, %8, %7\n v_cndmask_b32_e32 %7, %7, %8, vcc\n .endr" : VOUT : "v"(k) : "vcc");
        else if (OP == P_CMPCND_E64) asm volatile(".rept 8\n v_cmp_lt_u32 vcc, %8, %0\n v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cmp_lt_u32 vcc, %8, %1\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %8, %2\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cmp_lt_u32 vcc, %8, %3\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cmp_lt_u32 vcc, %8, %4\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cmp_lt_u32 vcc, %8, %5\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cmp_lt_u32 vcc, %8, %6\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cmp_lt_u32 vcc, %8, %7\n v_cndmask_b32_e64 %7, %7, %8, vcc\n .endr" : VOUT : "v"(k) : "vcc");
        else if (OP == P_CMPCND_E64S) asm volatile(".rept 8\n v_cmp_lt_u32 s[20:21], %8, %0\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cmp_lt_u32 s[20:21], %8, %1\n s_nop 1\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cmp_lt_u32 s[20:21], %8, %2\n s_nop 1\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cmp_lt_u32 s[20:21], %8, %3\n s_nop 1\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cmp_lt_u32 s[20:21], %8, %4\n s_nop 1\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cmp_lt_u32 s[20:21], %8, %5\n s_nop 1\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cmp_lt_u32 s[20:21], %8, %6\n s_nop 1\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cmp_lt_u32 s[20:21], %8, %7\n s_nop 1\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n .endr" : VOUT : "v"(k) : "s20", "s21");
        else if (OP == P_SDWA) asm volatile(I8("v_or_b32_sdwa", ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0") : VOUT : "v"(k));
        else if (OP == P_OR3) asm volatile(I8("v_or3_b32", ", %8, %8") : VOUT : "v"(k));
        else if (OP == P_LDS32A || OP == P_LDS32U)
            asm volatile(".rept 16\n ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:1024\n ds_read_b32 %2, %4 offset:2048\n ds_read_b32 %3, %4 offset:3072\n ds_read_b32 %0, %4 offset:4096\n ds_read_b32 %1, %4 offset:5120\n ds_read_b32 %2, %4 offset:6144\n ds_read_b32 %3, %4 offset:7168\n s_waitcnt lgkmcnt(0)\n .endr"
                         : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(ad) : "memory");
        else if (OP == P_LDSU16)
            asm volatile(".rept 16\n ds_read_u16 %0, %4\n ds_read_u16 %1, %4 offset:1024\n ds_read_u16 %2, %4 offset:2048\n ds_read_u16 %3, %4 offset:3072\n ds_read_u16 %0, %4 offset:4096\n ds_read_u16 %1, %4 offset:5120\n ds_read_u16 %2, %4 offset:6144\n ds_read_u16 %3, %4 offset:7168\n s_waitcnt lgkmcnt(0)\n .endr"
                         : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(ad) : "memory");
        else if (OP == P_LDS2X32)
            asm volatile(".rept 16\n ds_read2_b32 %0, %2 offset1:1\n ds_read2_b32 %1, %2 offset0:64 offset1:65\n ds_read2_b32 %0, %2 offset0:128 offset1:129\n ds_read2_b32 %1, %2 offset0:192 offset1:193\n ds_read2_b32 %0, %2 offset0:32 offset1:33\n ds_read2_b32 %1, %2 offset0:96 offset1:97\n ds_read2_b32 %0, %2 offset0:160 offset1:161\n ds_read2_b32 %1, %2 offset0:224 offset1:225\n s_waitcnt lgkmcnt(0)\n .endr"
                         : "=&v"(q0), "=&v"(q1) : "v"(ad) : "memory");
        else if (OP == P_LDS64A || OP == P_LDS64U)
            asm volatile(".rept 16\n ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:1024\n ds_read_b64 %0, %2 offset:2048\n ds_read_b64 %1, %2 offset:3072\n ds_read_b64 %0, %2 offset:4096\n ds_read_b64 %1, %2 offset:5120\n ds_read_b64 %0, %2 offset:6144\n ds_read_b64 %1, %2 offset:7168\n s_waitcnt lgkmcnt(0)\n .endr"
                         : "=&v"(q0), "=&v"(q1) : "v"(ad) : "memory");
        else if (OP == P_LDS128A || OP == P_LDS128U) {
            uint4 r0, r1;
            asm volatile(".rept 16\n ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n ds_read_b128 %0, %2 offset:2048\n ds_read_b128 %1, %2 offset:3072\n ds_read_b128 %0, %2 offset:4096\n ds_read_b128 %1, %2 offset:5120\n ds_read_b128 %0, %2 offset:6144\n ds_read_b128 %1, %2 offset:7168\n s_waitcnt lgkmcnt(0)\n .endr"
                         : "=&v"(r0), "=&v"(r1) : "v"(ad) : "memory");
            a0 ^= r0.x ^ r1.y;
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + ad;
}

template <int OP>
static void sweep(uint32_t iters, uint32_t* d_out, int cus, double ghz) {
    const bool lds = OP >= P_LDS32A;
    for (uint32_t pat = 0; pat < (lds ? 2u : 1u); ++pat) {
        printf("%-34s %-4s", kN[OP], lds ? (pat ? "rnd" : "seq") : "");
        for (uint32_t W : {1u, 2u, 4u, 8u}) {
            const uint32_t grid = (uint32_t)cus * W;
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            hipLaunchKernelGGL((probe<OP>), dim3(grid), dim3(256), 0, 0, iters / 8u + 1u, d_out, pat);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a, 0));
            hipLaunchKernelGGL((probe<OP>), dim3(grid), dim3(256), 0, 0, iters, d_out, pat);
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            const double ns = ms * 1e6 / ((double)iters * 128.0 * W);
            printf("  W=%u: %6.3f ns = %5.2f cyc", W, ns, ns * ghz);
            CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
        }
        printf("\n");
    }
}

int main(int argc, char** argv) {
    uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000u;
    const double ghz = argc > 2 ? atof(argv[2]) : 2.26;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4));
    printf("# table 2: ns (cycles at %.2f GHz) per wave64 instruction per SIMD, W waves per SIMD, 8 independent chains; LDS rows: per ds instruction\n", ghz);
    sweep<P_AND>(iters, d_out, cus, ghz); sweep<P_OR>(iters, d_out, cus, ghz); sweep<P_SUB>(iters, d_out, cus, ghz); sweep<P_LSHR>(iters, d_out, cus, ghz);
    sweep<P_MINU>(iters, d_out, cus, ghz); sweep<P_CMP_VCC>(iters, d_out, cus, ghz); sweep<P_CMP_SGPR>(iters, d_out, cus, ghz);
    sweep<P_CND_VCC_SET>(iters, d_out, cus, ghz); sweep<P_CND_E64>(iters, d_out, cus, ghz); sweep<P_CND_E64_CONST>(iters, d_out, cus, ghz);
    sweep<P_BFE>(iters, d_out, cus, ghz); sweep<P_ADD3>(iters, d_out, cus, ghz); sweep<P_ANDOR>(iters, d_out, cus, ghz); sweep<P_PERM>(iters, d_out, cus, ghz);
    sweep<P_MBCNT>(iters, d_out, cus, ghz);
    sweep<P_CND_E64_VCC>(iters, d_out, cus, ghz); sweep<P_CMPCND_E32>(iters, d_out, cus, ghz); sweep<P_CMPCND_E64>(iters, d_out, cus, ghz); sweep<P_CMPCND_E64S>(iters, d_out, cus, ghz);
    sweep<P_SDWA>(iters, d_out, cus, ghz); sweep<P_OR3>(iters, d_out, cus, ghz);
    sweep<P_LDS32A>(iters / 4u, d_out, cus, ghz); sweep<P_LDS32U>(iters / 4u, d_out, cus, ghz); sweep<P_LDSU16>(iters / 4u, d_out, cus, ghz);
    sweep<P_LDS2X32>(iters / 4u, d_out, cus, ghz);
    sweep<P_LDS64A>(iters / 4u, d_out, cus, ghz); sweep<P_LDS64U>(iters / 4u, d_out, cus, ghz);
    sweep<P_LDS128A>(iters / 4u, d_out, cus, ghz); sweep<P_LDS128U>(iters / 4u, d_out, cus, ghz);
    return 0;
}
