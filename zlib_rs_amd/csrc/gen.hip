// gen.hip -- on-device generator for the synthetic Silesia-like benchmark shards.
//
// Workload definition lives in shardgen.h; one HIP thread writes one
// 64-byte line as four 16-byte stores, so a wave writes 4 KiB contiguous.  Not on the timed path:
// bench.py generates the batch once, before the timed region (inputs resident in HBM).
#include "zmi_device.h"
#include "shardgen.h"

__global__ void __launch_bounds__(256) zmi_gen_kernel(uint8_t* __restrict__ out, uint64_t seed, uint32_t first_shard,
                                                      uint32_t shard_step, uint32_t shard_bytes, uint64_t total_lines) {
    uint64_t gl = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gl >= total_lines) return;
    uint32_t lines_per_shard = shard_bytes / ZMI_GEN_LINE;
    uint32_t shard = (uint32_t)(gl / lines_per_shard);
    uint32_t line = (uint32_t)(gl % lines_per_shard);
    uint32_t buf[16];
    zmi_gen_line(seed, first_shard + shard * shard_step, line, lines_per_shard, (uint8_t*)buf);
    uint32_t* dst = (uint32_t*)(out + gl * ZMI_GEN_LINE);
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[i] = buf[i];
}

// shards are laid out back to back: shard first_shard + i * shard_step at out + i*shard_bytes (shard_bytes multiple of 64);
// a step of `world` is the round-robin ownership of a multi-GPU job (BASELINE.json configs[4])
extern "C" int zmi_launch_gen_strided(uint8_t* d_out, uint64_t seed, uint32_t first_shard, uint32_t shard_step, uint32_t n_shards,
                                      uint32_t shard_bytes, hipStream_t stream) {
    uint64_t total_lines = (uint64_t)n_shards * (shard_bytes / ZMI_GEN_LINE);
    if (total_lines == 0) return 0;
    uint64_t nblk = (total_lines + 255) / 256;
    // grid.x limit is 2^31-1; 64 Ki shards x 16 Ki lines = 2^30 lines -> 2^22 blocks
    ZMI_LAUNCH(zmi_gen_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, d_out, seed, first_shard, shard_step, shard_bytes,
               total_lines);
    return 0;
}
extern "C" int zmi_launch_gen(uint8_t* d_out, uint64_t seed, uint32_t first_shard, uint32_t n_shards,
                              uint32_t shard_bytes, hipStream_t stream) {
    return zmi_launch_gen_strided(d_out, seed, first_shard, 1u, n_shards, shard_bytes, stream);
}
