/* zmi355.h -- C ABI of the MI355X-native DEFLATE engine (libzmi355.so).
 *
 * Two layers, both plain C (pointers + sizes, no C++ / torch types):
 *
 *  1. The zlib stream ABI of the reference, unchanged: z_stream, deflateInit2_/deflate/deflateEnd,
 *     inflateInit2_/inflate/inflateEnd, compress2/uncompress, adler32/crc32 (+_combine) ...
 *     Declared in zmi355_zlib.h with the reference line each symbol replaces
 *     (libz-rs-sys/src/lib.rs) and exported by the drop-in library libz_mi355.so, which is built on this one:
 *     a caller of libz-rs-sys relinks against libz_mi355.so.
 *
 *  2. The batch entry points below (new, additive).  The reference's ABI is one stream per call
 *     (libz-rs-sys/src/lib.rs:1281 deflate, :636 inflate); the north-star workload is thousands of
 *     independent 1 MiB shards, which need one launch per batch, not per stream.  Semantics per
 *     shard are exactly those of
 *         deflateInit2_(level, Z_DEFLATED, windowBits(wrap), 8, strategy) + deflate(Z_FINISH) + deflateEnd
 *         (test-libz-rs-sys/examples/blogpost-compress.rs:43-122), and
 *         inflateInit2_(windowBits(wrap)) + inflate(Z_FINISH) + inflateEnd
 *         (test-libz-rs-sys/examples/blogpost-uncompress.rs:6-44)
 *     with the per-shard return code reported in status[] using zlib's numbering
 *     (zlib-rs/src/c_api.rs:140-148): 0 ok, -3 Z_DATA_ERROR, -5 Z_BUF_ERROR, 2 Z_NEED_DICT.
 *
 * Every function here requires a gfx950 device; there is no CPU fallback (ZMI_E_NODEVICE).
 */
#ifndef ZMI355_H
#define ZMI355_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zmi_ctx zmi_ctx;

/* wrapper selection (windowBits of the reference: -15 raw, 15 zlib, 31 gzip, 47 auto-detect) */
#define ZMI_WRAP_RAW 0
#define ZMI_WRAP_ZLIB 1
#define ZMI_WRAP_GZIP 2
#define ZMI_WRAP_AUTO 3 /* inflate only */

/* library-level return codes (per-shard codes use zlib numbering, see above) */
#define ZMI_E_OK 0
#define ZMI_E_NODEVICE (-101)
#define ZMI_E_HIP (-102)
#define ZMI_E_ARG (-103)
#define ZMI_E_NOMEM (-104)
#define ZMI_E_NORCCL (-105) /* the multi-GPU entry points need RCCL (librccl.so.1) and it could not be loaded */
#define ZMI_E_RCCL (-106)   /* an RCCL call failed; zmi_last_error() holds ncclGetErrorString's text */

const char* zmi_version(void);
const char* zmi_last_error(void);

int zmi_ctx_create(zmi_ctx** ctx, int device);
int zmi_ctx_destroy(zmi_ctx* ctx);
/* upper bound of scratch the context may allocate on the device (default 8 GiB; env ZMI_SCRATCH_MB) */
int zmi_ctx_set_scratch_limit(zmi_ctx* ctx, uint64_t bytes);
/* total output capacity (sum of d_out_cap) one inflate batch may cover.  Inflate keeps 1 bit of scratch
 * per output byte; the capacities live on the device, so the bound comes from here (default: the scratch
 * limit, i.e. 8 GiB of output -> 1 GiB of bitmap).  Streams beyond it report Z_MEM_ERROR (-4). */
int zmi_ctx_set_inflate_out_limit(zmi_ctx* ctx, uint64_t bytes);

/* Host-buffer batches (zmi_deflate_batch / zmi_inflate_batch) stage their chunks through pinned host memory and device slots
 * that the context keeps for the next call (three slots; up to ~6.4 GiB pinned + as much HBM after a multi-GiB batch).
 * zmi_ctx_set_pinned_limit bounds the pinned part (default 10 GiB, env ZMI_PINNED_MB; >= 64 MiB): chunk sizes follow it and a
 * call that ends above it releases the staging.  zmi_ctx_trim releases it now.  If pinned memory cannot be had at all
 * (memlock / container limits) the calls fall back to plain copies instead of failing. */
int zmi_ctx_set_pinned_limit(zmi_ctx* ctx, uint64_t bytes);
int zmi_ctx_trim(zmi_ctx* ctx);

/* hipStream_t the single-stream host wrappers of this context (zmi_inflate_resume) copy and launch on, and the only thing
 * they wait for; default: the null stream.  One context per thread, each with its own stream, run concurrently. */
int zmi_ctx_set_stream(zmi_ctx* ctx, void* stream);

/* Decode-table entries (literal/length + distance tables) the device built for the most recent dynamic block that
 * zmi_inflate_resume calls on this context met; 0 before the first one.  The reference's inflateCodesUsed
 * (libz-rs-sys/src/lib.rs:1252, zlib-rs/src/inflate.rs:2372: state.next) reports the same quantity for its own tables
 * (roots 10 / 9, ENOUGH 1332 + 592); the device tables use roots 9 / 8 with exact-fit sub-tables (at most 852 + 400). */
int zmi_ctx_last_codes_used(zmi_ctx* ctx, uint32_t* entries);
int zmi_ctx_reset_codes_used(zmi_ctx* ctx);

/* per-kernel HIP-event timing for benchmarking: kernels 0 checksum, 1 lz77, 2 encode, 3 inflate (decode),
 * 4 verify, 5 cost parse (deflate levels 3-9), 6 inflate (resolve), 7 pack / stitch copies.  zmi_ctx_get_timing synchronises, returns the sums (ms) / launch counts since the
 * previous call (arrays of 8) and resets them. */
int zmi_ctx_set_timing(zmi_ctx* ctx, int on);
int zmi_ctx_get_timing(zmi_ctx* ctx, double* ms_sums, uint32_t* counts);

/* worst-case compressed size of an n-byte shard; same formula as the reference's compress_bound
 * (zlib-rs/src/deflate.rs:2975-2991) with the wrapper overhead of `wrap`, rounded up to 16. */
uint64_t zmi_deflate_bound(uint64_t n, int wrap);

/* ---- device-resident batch API (all d_* pointers are device memory; stream is a hipStream_t) ---- */

/* shard i = d_in[d_in_off[i] .. + d_in_len[i]); compressed stream i is written at
 * d_out + i*out_stride (out_stride multiple of 16, >= zmi_deflate_bound(max_len, wrap)),
 * its length to d_out_len[i], its zlib return code to d_status[i].
 * level 0..9 or -1, strategy 0..4 as deflateInit2_ (libz-rs-sys/src/lib.rs:2005-2041). */
int zmi_deflate_batch_dev(zmi_ctx* ctx, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                          uint32_t n_shards, uint32_t max_len, int level, int strategy, int wrap, void* d_out,
                          uint64_t out_stride, uint32_t* d_out_len, int32_t* d_status, void* stream);

/* stream i = d_in[d_in_off[i] .. + d_in_len[i]); output i is written at d_out + d_out_off[i]
 * (capacity d_out_cap[i]); d_out_len[i] receives the number of bytes produced, d_status[i] the
 * zlib return code (0 = stream complete and check value correct). */
int zmi_inflate_batch_dev(zmi_ctx* ctx, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                          uint32_t n_streams, int wrap, void* d_out, const uint64_t* d_out_off,
                          const uint32_t* d_out_cap, uint32_t* d_out_len, int32_t* d_status, void* stream);

/* Chained form used by the zlib stream ABI (libz_mi355.so): the shards are consecutive segments of
 * ONE raw deflate stream, contiguous in d_in.  Each starts byte aligned (the empty stored block of
 * Z_SYNC_FLUSH, zlib-rs/src/deflate.rs:2733-2738) and matches into the up to 27 KiB in front of it
 * (window carry-over); finish != 0 makes the last shard end the stream.  The _dict form also treats the
 * dict_len bytes in front of the first segment as history: a preset dictionary (deflateSetDictionary,
 * deflate.rs:499-564) or the tail of the input of an earlier call on the same stream.
 * out_stride: zmi_deflate_bound(max_len, raw) + 16 -- a segment that does not end the stream carries the 5-byte sync
 * marker behind its last block, which the bound of a finished stream does not cover for segments of a few bytes. */
int zmi_deflate_chain_dev(zmi_ctx* ctx, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                          uint32_t n_shards, uint32_t max_len, int level, int strategy, int finish, void* d_out,
                          uint64_t out_stride, uint32_t* d_out_len, int32_t* d_status, void* stream);
int zmi_deflate_chain_dict_dev(zmi_ctx* ctx, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                               uint32_t n_shards, uint32_t max_len, int level, int strategy, int finish,
                               uint32_t dict_len, void* d_out, uint64_t out_stride, uint32_t* d_out_len,
                               int32_t* d_status, void* stream);
/* The chained form for a stream opened with windowBits 9..14 (deflateInit2_, zlib-rs/src/deflate.rs:252-312):
 * back-references reach at most 2^window_bits - 262 bytes (the reference's max_dist, deflate.rs:1423-1425), so an
 * inflater that allocates only the announced window can read the stream.  window_bits 15 = zmi_deflate_chain_dict_dev. */
int zmi_deflate_chain_window_dev(zmi_ctx* ctx, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                 uint32_t n_shards, uint32_t max_len, int level, int strategy, int finish,
                                 uint32_t dict_len, uint32_t window_bits, void* d_out, uint64_t out_stride,
                                 uint32_t* d_out_len, int32_t* d_status, void* stream);
/* As zmi_inflate_batch_dev, additionally reporting the consumed input bytes and why a stream
 * stopped (d_detail: 0 done/error, 1 needs more input, 2 needs more output space). */
int zmi_inflate_batch_dev_ex(zmi_ctx* ctx, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                             uint32_t n_streams, int wrap, void* d_out, const uint64_t* d_out_off,
                             const uint32_t* d_out_cap, uint32_t* d_out_len, int32_t* d_status, uint32_t* d_in_used,
                             int32_t* d_detail, void* stream);
/* As zmi_inflate_batch_dev_ex with preset dictionaries: d_out_hist[i] (array may be NULL) bytes directly in front
 * of stream i's output region are history the stream may refer to (inflateSetDictionary,
 * zlib-rs/src/inflate.rs:2492-2536).  A zlib stream with FDICT set reports Z_NEED_DICT (2) when its entry is 0;
 * checking its DICTID against the dictionary is the caller's job. */
int zmi_inflate_batch_dict_dev(zmi_ctx* ctx, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                               uint32_t n_streams, int wrap, void* d_out, const uint64_t* d_out_off,
                               const uint32_t* d_out_cap, const uint32_t* d_out_hist, uint32_t* d_out_len,
                               int32_t* d_status, uint32_t* d_in_used, int32_t* d_detail, void* stream);
/* Resumable decode of raw deflate streams -- the device half of a streaming inflate() that is fed partial input
 * (the reference keeps Mode / BitReader / Window for this, zlib-rs/src/inflate.rs:288-320; here the state is a
 * block-boundary checkpoint).  Stream i starts at bit d_in_bit[i] (0..7, array may be NULL) of its first byte, with
 * d_out_hist[i] bytes of earlier output directly in front of its output region.  d_resume[4i..4i+3] receives
 * {byte, bit, output bytes, complete}: the start of the block the decode stopped in (status Z_BUF_ERROR, d_detail
 * 1 = more input / 2 = more room needed), or the first bit behind the final block (complete = 1, status Z_OK).
 * d_out_len[i] counts everything decoded including the valid part of the unfinished block; a later call that
 * starts at the checkpoint, with the output in front of it as history, reproduces those bytes and continues.
 * Stops on request (what inflate(Z_BLOCK) / inflate(Z_TREES) are built on, zlib-rs/src/inflate.rs:1276-1284,1323,1369,
 * 1772): bits 8..23 of d_in_bit[i] = stop at the block boundary behind that many complete blocks (0: none), bit 24 = stop
 * behind the header of the first block (code tables read, none of its data decoded).  Both report status Z_BUF_ERROR
 * with d_detail 3; a boundary stop leaves the usual checkpoint, a header stop leaves {byte, bit of the first bit behind
 * the header, output bytes, 2 | BFINAL << 2} in d_resume. */
int zmi_inflate_resume_dev(zmi_ctx* ctx, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                           const uint32_t* d_in_bit, uint32_t n_streams, void* d_out, const uint64_t* d_out_off,
                           const uint32_t* d_out_cap, const uint32_t* d_out_hist, uint32_t* d_out_len,
                           int32_t* d_status, uint32_t* d_in_used, int32_t* d_detail, uint32_t* d_resume, void* stream);

/* Adler-32 (kind bit 0) and/or CRC-32 (kind bit 1) of every shard (zlib-rs/src/adler32.rs:19, crc32.rs:19) */
int zmi_checksum_batch_dev(zmi_ctx* ctx, const void* d_data, const uint64_t* d_off, const uint32_t* d_len,
                           uint32_t n_shards, int kind, uint32_t* d_adler, uint32_t* d_crc, void* stream);

/* synthetic Silesia-like benchmark shards (csrc/shardgen.h): shard first_shard+i at d_out + i*shard_bytes */
int zmi_gen_shards_dev(zmi_ctx* ctx, void* d_out, uint64_t seed, uint32_t first_shard, uint32_t n_shards,
                       uint32_t shard_bytes, void* stream);

/* shard first_shard + i*shard_step at d_out + i*shard_bytes: a step of `world` is the round-robin shard ownership of a
 * multi-GPU job (rank r owns the shards g with g % world == r, BASELINE.json configs[4]) */
int zmi_gen_shards_strided_dev(zmi_ctx* ctx, void* d_out, uint64_t seed, uint32_t first_shard, uint32_t shard_step,
                               uint32_t n_shards, uint32_t shard_bytes, void* stream);

/* ---- the stitch (replaces the append loop of the reference's parallel-deflate recipe, zlib-rs/src/deflate.rs:4145-4221
 * `split_deflate`; multi-member gzip as read by libz-rs-sys/src/gz.rs:1464-1506): the batch leaves its compressed
 * shards in out_stride-strided slots; these calls turn them into dense, ordered bytes on the device ----
 * zmi_scan_sizes_dev   d_off[0..n] = exclusive prefix sum of d_len[0..n) (u64; d_off[n] = total)
 * zmi_copy_ranges_dev  range i: d_len[i] bytes from d_src + (d_src_off ? d_src_off[i] : i*src_stride) to
 *                      d_dst + d_dst_off[i]; ranges that would end behind dst_cap are skipped (the offsets tell).
 *                      max_len bounds d_len[] (it only shapes the launch).
 * zmi_pack_slab_dev    both: the slots of one batch -> one dense slab + its offsets.  One D2H copy / one send per
 *                      peer then moves the batch; after a slab exchange zmi_copy_ranges_dev scatters a peer's slab into
 *                      the globally ordered output (d_src_off = the peer's scan, d_dst_off = global offsets). */
int zmi_scan_sizes_dev(zmi_ctx* ctx, const uint32_t* d_len, uint32_t n, uint64_t* d_off, void* stream);
int zmi_copy_ranges_dev(zmi_ctx* ctx, const void* d_src, const uint64_t* d_src_off, uint64_t src_stride, const uint32_t* d_len,
                        uint32_t n, uint32_t max_len, void* d_dst, const uint64_t* d_dst_off, uint64_t dst_cap, void* stream);
int zmi_pack_slab_dev(zmi_ctx* ctx, const void* d_slots, uint64_t slot_stride, const uint32_t* d_len, uint32_t n,
                      void* d_slab, uint64_t slab_cap, uint64_t* d_off, void* stream);

/* ---- the stitch across the GPUs of a node (BASELINE.json configs[4]; csrc/exchange.hip) -------------------------------
 * Shard g of a job lives on rank g % world (round-robin); every rank compresses its shards with zmi_deflate_batch_dev (no
 * collective in the compression) and packs them into one dense slab (zmi_pack_slab_dev).  "Append the pieces in order" --
 * the loop of the reference's parallel-deflate recipe, zlib-rs/src/deflate.rs:4145-4221 -- then is:
 *   zmi_exchange_sizes        all-gather of the u32 size tables: d_table[r * n_local + j] = size of shard j * world + r.
 *                             Every rank passes the SAME n_local: a shard count that does not divide by the world size is
 *                             padded with zero sizes on the short ranks (checked: ZMI_E_ARG on every rank otherwise; the
 *                             call waits for `stream` when world > 1)
 *   zmi_stitch_plan_dev       from the table alone: d_goff[r * n_local + j] = byte offset of that shard in the stitched
 *                             output, d_soff[r * (n_local + 1) + j] = its offset inside rank r's slab (last entry of a row =
 *                             the slab's size), d_totals[r] = slab size of rank r, d_totals[world] = size of the output;
 *                             totals_host (may be NULL) receives a copy of d_totals (the call then waits for the stream)
 *   zmi_exchange_slabs        the slabs travel point to point: per round of chunk_bytes ONE ncclGroupStart / End holding an
 *                             ncclSend to and an ncclRecv from every peer -- xGMI is a full mesh, so the 7 transfers of a
 *                             round run side by side, one link each; there is no all-gather-v in RCCL and a ring is never
 *                             used.  root < 0: every rank receives every slab (d_recv[p] = room for slab_bytes[p] bytes;
 *                             the own entry is skipped); root >= 0: only that rank receives, the others may pass NULL.
 *                             A receiving rank MUST give room for every peer whose slab is not empty -- every such peer
 *                             sends, and a send without its receive would hang the group: a NULL entry is ZMI_E_ARG
 *   zmi_exchange_slabs_round  one round with bounded memory: peer p's bytes [lo, lo + chunk_bytes) arrive at the start of
 *                             d_stage[p]; the caller consumes them and reuses the staging for the next round
 *   zmi_copy_ranges_dev       (above) scatters a slab or a round of it into the ordered output: d_src_off = the peer's row
 *                             of d_soff, d_dst_off = its row of d_goff.
 * All operations are enqueued on `stream`; slab_bytes / d_recv / d_stage are HOST arrays of `world` entries.
 * The communicator: zmi_comm_unique_id on one rank, the 128 bytes carried to the others by whatever the host has (MPI, a
 * file, torch's store), zmi_comm_create on every rank (ncclCommInitRank on the context's device) -- or zmi_comm_adopt for a
 * host that already holds an ncclComm_t.  RCCL is loaded on first use (dlopen librccl.so.1; ZMI_RCCL_LIB overrides the
 * file): processes that never call these functions do not need it. */
#define ZMI_UNIQUE_ID_BYTES 128
typedef struct zmi_comm zmi_comm;
int zmi_comm_unique_id(void* id128);
int zmi_comm_create(zmi_comm** comm, zmi_ctx* ctx, int world, int rank, const void* id128);
int zmi_comm_adopt(zmi_comm** comm, zmi_ctx* ctx, void* nccl_comm);
int zmi_comm_destroy(zmi_comm* comm);
int zmi_comm_abort(zmi_comm* comm); /* ncclCommAbort: for a communicator whose peers are gone */
int zmi_comm_world(const zmi_comm* comm);
int zmi_comm_rank(const zmi_comm* comm);
int zmi_exchange_sizes(zmi_comm* comm, const uint32_t* d_sizes, uint32_t n_local, uint32_t* d_table, void* stream);
int zmi_stitch_plan_dev(zmi_ctx* ctx, const uint32_t* d_table, uint32_t world, uint32_t n_local, uint64_t* d_goff,
                        uint64_t* d_soff, uint64_t* d_totals, uint64_t* totals_host, void* stream);
int zmi_exchange_slabs(zmi_comm* comm, const void* d_slab, const uint64_t* slab_bytes, void* const* d_recv,
                       uint64_t chunk_bytes, int root, void* stream);
int zmi_exchange_slabs_round(zmi_comm* comm, const void* d_slab, const uint64_t* slab_bytes, uint64_t lo,
                             uint64_t chunk_bytes, void* const* d_stage, int root, void* stream);

/* ---- host-buffer convenience wrappers: copy in, run the batch on the GPU, copy back ---- */
int zmi_deflate_batch(zmi_ctx* ctx, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n_shards,
                      int level, int strategy, int wrap, uint8_t* out, uint64_t out_stride, uint32_t* out_len,
                      int32_t* status);
int zmi_inflate_batch(zmi_ctx* ctx, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n_streams,
                      int wrap, uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len,
                      int32_t* status);
/* One host stream through zmi_inflate_resume_dev (staging buffers are kept in the context): `in` starts at bit in_bit of
 * its first byte, `hist` is the up to 32 KiB of output in front of it, resume[4] as above; out receives
 * min(*out_len, out_cap) bytes.  This is what the stream ABI's inflate() / inflateBack() run on. */
int zmi_inflate_resume(zmi_ctx* ctx, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist,
                       uint32_t hist_len, uint8_t* out, uint32_t out_cap, uint32_t* out_len, int32_t* status,
                       int32_t* detail, uint32_t* in_used, uint32_t* resume);
/* The same call for a stream with flush points (Z_SYNC_FLUSH / Z_FULL_FLUSH markers 00 00 FF FF: pigz, this library's own
 * deflate()): seg_start[0..nseg) are byte offsets into `in` proposed as restart points (seg_start[0] = 0, ascending; normally
 * the byte behind every marker found).  The pieces are decoded side by side on the whole GPU and stitched; every cut is
 * verified (the decode in front of it must end exactly there, on a block boundary), anything else falls back to the
 * serial decode from that point -- results are those of zmi_inflate_resume on the same arguments, whatever seg_start
 * holds.  *segments_used (may be NULL): how many pieces were decoded in parallel (0 = the serial path ran).
 * Reference path it accelerates: zlib-rs/src/inflate.rs:1276 ff. (the Mode::Type block loop), restarted where
 * zlib-rs/src/deflate.rs:2733-2744 (the empty stored block of a flush) made the stream restartable. */
int zmi_inflate_split(zmi_ctx* ctx, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist,
                      uint32_t hist_len, uint8_t* out, uint32_t out_cap, const uint32_t* seg_start, uint32_t nseg,
                      uint32_t* out_len, int32_t* status, int32_t* detail, uint32_t* in_used, uint32_t* resume,
                      uint32_t* segments_used);

/* The same call for a stream WITHOUT flush points -- what every ordinary compressor writes: nothing in it is byte aligned, but
 * its dynamic blocks announce themselves.  The device tries every bit position of the stream for a block header (BTYPE 10, a
 * complete code-length code, code lengths that decode exactly into two complete codes: csrc/blockscan.hip), the stretches
 * between the boundaries found are decoded side by side and stitched as above; a cut counts only if the decode in front of
 * it ended exactly there, at that bit.  Results are those of zmi_inflate_resume on the same arguments; *segments_used (may be
 * NULL) = pieces decoded in parallel, 0 = the serial path ran (stream below 256 KiB, fewer than three boundaries found).
 * Reference path it accelerates: the block loop of zlib-rs/src/inflate.rs:1276 ff. over blocks of 16 383 symbols
 * (zlib-rs/src/deflate.rs:321), driven by test-libz-rs-sys/examples/blogpost-uncompress.rs:6-44. */
int zmi_inflate_blocks(zmi_ctx* ctx, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist,
                       uint32_t hist_len, uint8_t* out, uint32_t out_cap, uint32_t* out_len, int32_t* status, int32_t* detail,
                       uint32_t* in_used, uint32_t* resume, uint32_t* segments_used);

#ifdef __cplusplus
}
#endif
#endif
