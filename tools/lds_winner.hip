// lds_winner.hip -- which lane wins when several lanes of ONE ds_write_b16 hit the same LDS address?  (lz77.hip inserts
// 64 positions per instruction into 16-bit hash heads with a plain write; ADVICE r02: the winner defines the compressed
// bytes.)  Prints, for groups of g lanes sharing an address (g = 2, 3, 4, 8, 64; contiguous and strided groups), the lane
// whose value the LDS keeps.  Standalone: hipcc --offload-arch=gfx950 -o lds_winner lds_winner.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out, unsigned g, unsigned strided) {
    __shared__ unsigned short lds[64];
    const unsigned lane = threadIdx.x;
    lds[lane] = 0xFFFF;
    __syncthreads();
    const unsigned slot = strided ? lane % (64u / g) : lane / g;   // strided: lanes l, l + 64/g, ... share a slot
    *(volatile unsigned short*)&lds[slot] = (unsigned short)lane;
    __syncthreads();
    out[lane] = lds[lane];
}
int main() {
    unsigned* d; hipMalloc(&d, 256);
    unsigned h[64];
    const unsigned gs[5] = {2, 3, 4, 8, 64};
    for (unsigned strided = 0; strided < 2; ++strided)
        for (unsigned g : gs) {
            if (strided && 64 % g) continue;
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, g, strided);
            hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
            printf("%s groups of %2u: slot 0 keeps lane %u, slot 1 keeps lane %u   (highest lane of slot 0 = %u, lowest = 0)\n", strided ? "strided   " : "contiguous", g, h[0], h[1],
                   strided ? 64 - 64 / g : g - 1);
        }
    return 0;
}
