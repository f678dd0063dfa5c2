// zmi_kernels.h -- launch entry points and parameter blocks shared by the kernels and the host API.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef ZMI_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

// match-finder effort (the GPU analogue of CONFIGURATION_TABLE, deflate/algorithm/mod.rs:69-82)
struct zmi_lz_params {
    uint32_t max_chain;  // hash-chain links followed per position
    uint32_t nice_len;   // stop searching once a match this long is found
    uint32_t good_len;   // halve the remaining chain budget above this length
    uint32_t max_dist;   // farthest back-reference (<= 32768 - 5*1024 - 16 = 27632, ring-buffer constraint: LZ_MAX_DIST in lz77.hip)
    uint32_t claim;      // positions a searcher wave claims at once (64, 128, 192 or 256)
    uint32_t hash6;      // 1: chain keyed by a 6-byte hash + one most-recent 4-byte probe; 0: 4-byte hash chain
    uint32_t producers;  // 1..4 hash-building waves per workgroup (tile k is built by wave k mod producers)
    uint32_t dbg;        // measurement aids, 0 in the product (1: producers' pace alone, see lz77.hip)
    uint32_t carry;      // 1: the shards are consecutive segments of one stream; a segment may match into the up to 27 KiB
                         // in front of it (window carry-over, what a preset dictionary is in deflate.rs:499-564)
    uint32_t dict_len;   // carry only: bytes in front of shard 0 that are history too (preset dictionary / earlier input)
    uint32_t barren_chain; // chain links walked in a claim whose 64 probes all missed (an incompressible stretch: whatever the 6-byte
                           // chain holds there is mostly a hash collision).  1: with 0 the benchmark mix keeps its ratio but paper-100k.pdf loses 0.2 %
                           // for 0.5 % of the kernel's time (round 4)
    uint32_t far4, far5; // a 4- (5-) byte match further back than this costs more bits than its literals: dropped
                         // (classic zlib's TOO_FAR idea; the reference itself only drops matches <= 5 under
                         // Z_FILTERED, zlib-rs/src/deflate/algorithm/slow.rs:69-74)
};

struct zmi_enc_params {
    uint32_t max_lazy;    // defer a match shorter than this if the next position has a longer one (0 = greedy)
    uint32_t lazy2, lazy3; // ... or if the position after that (the one after) has one longer by more than this (>= 258: off)
    uint32_t wrap;        // 0 raw deflate, 1 zlib (RFC 1950), 2 gzip (RFC 1952)
    uint32_t level;       // only used for the header's level hint bits
    uint32_t block_span;  // input bytes per deflate block (multiple of 64) ...
    uint32_t block_tokens; // tokens per sub-block: after each one the encoder decides whether it joins the open block or starts
                           // a new one (the reference cuts at 16383 symbols, deflate.rs:321)
    uint32_t split_hdr_bits; // what a block of its own must save: the cost of another dynamic header
    uint32_t min_sub_span;   // input bytes a sub-block covers at least (unless it holds 2 * block_tokens tokens already)
    uint32_t strategy;    // 0 default, 4 = Z_FIXED (static trees only)
    uint32_t far4, far5;  // first block of a piece: a 4- (5-) byte match further back than this is dropped; later blocks derive
                          // their limits from the codes of the block before (enc_far_limits)
    uint32_t chain_mode;  // 0: every shard is its own stream; 1: the shards of the batch are consecutive
                          // segments of ONE raw deflate stream, only shard `last_shard` ends it (BFINAL);
                          // 2: as 1 but nothing ends the stream (Z_SYNC_FLUSH / Z_FULL_FLUSH output)
    uint32_t last_shard;
    uint32_t cost_parse;  // 1: tokens are chosen by a backward cost parse over the matches (levels 3-9, csrc/parse.hip);
                          // 0: by the lazy rule (levels 1, 2 -- greedy -- and Z_HUFFMAN_ONLY / level 0, which have no matches)
};

// per-shard result codes written by the kernels (zlib numbering, zlib-rs/src/c_api.rs:140-148)
#define ZMI_OK 0
#define ZMI_STREAM_END 1
#define ZMI_DATA_ERROR (-3)
#define ZMI_BUF_ERROR (-5)

extern "C" {
int zmi_launch_gen(uint8_t* d_out, uint64_t seed, uint32_t first_shard, uint32_t n_shards, uint32_t shard_bytes,
                   hipStream_t stream);
int zmi_launch_gen_strided(uint8_t* d_out, uint64_t seed, uint32_t first_shard, uint32_t shard_step, uint32_t n_shards,
                           uint32_t shard_bytes, hipStream_t stream);
int zmi_launch_scan_sizes(const uint32_t* d_len, uint32_t n, uint64_t* d_off, hipStream_t stream);
int zmi_launch_copy_ranges(const uint8_t* d_src, const uint64_t* d_src_off, uint64_t src_stride, const uint32_t* d_len,
                           uint32_t n, uint8_t* d_dst, const uint64_t* d_dst_off, uint64_t dst_cap, uint32_t max_len,
                           hipStream_t stream);
int zmi_launch_clamp_lens(const uint32_t* d_len, const uint32_t* d_cap, uint32_t n, uint32_t* d_out, hipStream_t stream);
int zmi_launch_copy_ranges_few(const uint8_t* d_src, const uint64_t* d_src_off, uint64_t src_stride, const uint32_t* d_len,
                               uint32_t n, uint8_t* d_dst, const uint64_t* d_dst_off, uint64_t dst_cap, uint32_t max_len,
                               uint32_t groups, hipStream_t stream);
int zmi_launch_checksum(const uint8_t* d_data, const uint64_t* d_off, const uint32_t* d_len, uint32_t n_shards,
                        uint32_t kind, uint32_t* d_adler, uint32_t* d_crc, hipStream_t stream);
int zmi_launch_lz77(const uint8_t* d_data, const uint64_t* d_off, const uint32_t* d_len, uint32_t first_shard,
                    uint32_t n_shards, uint32_t* d_match, uint64_t match_stride, zmi_lz_params prm, hipStream_t stream);
int zmi_launch_encode(const uint8_t* d_data, const uint64_t* d_off, const uint32_t* d_len, uint32_t first_shard,
                      uint32_t n_shards, uint32_t* d_match, uint64_t match_stride, const uint32_t* d_adler,
                      const uint32_t* d_crc, uint8_t* d_out, uint64_t out_stride, uint32_t out_cap, uint32_t* d_out_len,
                      int32_t* d_status, uint32_t pieces, uint32_t* d_piece_len, const uint32_t* d_dec, uint64_t dec_stride,
                      zmi_enc_params prm, hipStream_t stream);
// the cost parse (levels 3-9, csrc/parse.hip): d_dec receives two bits per position of every shard of the group (dec_stride
// dwords per shard, 16 bytes per segment of 64 positions): 0 literal, 1 the match, 2 the match one byte shorter
int zmi_launch_parse(const uint32_t* d_len, uint32_t first_shard, uint32_t n_shards, uint32_t max_len, const uint32_t* d_match,
                     uint64_t match_stride, uint32_t* d_dec, uint64_t dec_stride, uint32_t pieces, uint32_t strategy, uint32_t span_chunks,
                     hipStream_t stream);
int zmi_launch_inflate(const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n_streams,
                       uint32_t wrap, uint8_t* d_out, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                       uint32_t* d_out_len, uint32_t* d_in_used, uint32_t* d_check, int32_t* d_status,
                       uint64_t* d_bitmap, uint64_t bitmap_words, uint64_t* d_bm_off, const uint32_t* d_out_hist,
                       const uint32_t* d_in_bit, uint32_t* d_resume, uint32_t* d_order, uint32_t mw_max, hipStream_t stream);
int zmi_launch_resolve_jump(uint8_t* d_out, const uint64_t* d_out_off, const uint32_t* d_out_len, uint32_t n_streams,
                            const uint64_t* d_bitmap, const uint64_t* d_bm_off, int32_t* d_ptr, uint64_t n_idx, uint32_t rounds,
                            uint32_t* d_flags, hipStream_t stream);
int zmi_launch_resolve_jump_segments(uint8_t* d_fin, const uint64_t* d_soff, uint32_t nseg, const uint64_t* d_bitmap,
                                     const uint64_t* d_bm_off, int32_t* d_ptr, uint64_t total, uint32_t hist_len, uint32_t rounds,
                                     uint32_t* d_flags, uint32_t* d_err, const uint64_t* d_one_off, const uint32_t* d_one_len,
                                     hipStream_t stream);
int zmi_launch_inflate_resolve(uint8_t* d_out, const uint64_t* d_out_off, const uint32_t* d_out_len, uint32_t n_streams,
                               const uint64_t* d_bitmap, const uint64_t* d_bm_off, const uint32_t* d_out_hist,
                               const uint32_t* d_order, hipStream_t stream);
}
