// Does gfx950 take LDS stores at any byte address?  hipcc --offload-arch=gfx950 -O2 -o ua_bin ua.hip; gpurun -- tools/ua_probe/ua_bin  (MI355X: ok)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
struct __attribute__((packed)) P32 { uint32_t v; };
struct __attribute__((packed)) P64 { uint64_t v; };
struct __attribute__((packed)) P16 { uint16_t v; };
__global__ void k(uint8_t* out, uint32_t off) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[2048];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 2048; i += 64) ring[i] = 0;
    __syncthreads();
    // lane l writes 11 bytes at byte offset 16*l + (l & 7) + off : b64 + b16 + b8
    uint8_t* p = ring + 32u * lane + (lane & 7u) + off;
    ((P64*)p)->v = 0x0807060504030201ull + lane;
    ((P16*)(p + 8))->v = (uint16_t)0x0A09;
    p[10] = 0x0B;
    ((P32*)(p + 11))->v = 0x0F0E0D0Cu;
    __syncthreads();
    for (uint32_t i = lane; i < 2048; i += 64) out[i] = ring[i];
    // unaligned reads
    __syncthreads();
    uint32_t r = ((const P32*)(ring + 16u * lane + (lane & 3u) + 1u))->v;
    ((uint32_t*)(out + 2048))[lane] = r;
}
int main() {
    uint8_t* d; hipMalloc(&d, 4096);
    k<<<1, 64>>>(d, 1);
    uint8_t h[4096]; hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 63; ++l) {
        int base = 32 * l + (l & 7) + 1;
        uint64_t v = 0x0807060504030201ull + l;
        for (int j = 0; j < 8; ++j) if (h[base + j] != (uint8_t)(v >> (8 * j))) bad++;
        if (h[base + 8] != 9 || h[base + 9] != 0xA || h[base + 10] != 0xB) bad++;
        // the 4 bytes at +11 may be overwritten by the next lane's store when they overlap: check only non-overlapping part
        int next = 32 * (l + 1) + ((l + 1) & 7) + 1;
        for (int j = 0; j < 4; ++j) if (base + 11 + j < next && h[base + 11 + j] != 0xC + j) bad++;
    }
    printf("unaligned LDS stores: %s (bad=%d)\n", bad ? "FAIL" : "ok", bad);
    return bad != 0;
}
