// zmi_api.hip -- host side of the batch C ABI declared in include/zmi355.h.
//
// Orchestrates the kernel pipeline on the caller's HIP stream:
//   deflate:  [checksum] -> lz77 (match per position) -> encode (parse + Huffman + bit pack)
//   inflate:  inflate -> checksum of the produced bytes -> verify against the trailer
// Device scratch (4 B per input byte for the match/token array) is owned by the context and the
// batch is cut into groups that fit it.  No CPU fallback exists: without a HIP device every entry
// point fails with ZMI_E_NODEVICE.
#include "zmi_kernels.h"
#include "../../include/zmi355.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <atomic>
#include <algorithm>
#include <string>
#include <vector>

extern "C" int zmi_launch_inflate_verify(const uint8_t* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                         uint32_t n_streams, uint32_t wrap, const uint32_t* d_check, const uint32_t* d_adler,
                                         const uint32_t* d_crc, int32_t* d_status, int32_t* d_detail, hipStream_t stream);

static thread_local std::string g_err;
static int zmi_fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[256];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}
// Tuning overrides (ZMI_CHAIN, ZMI_BLOCK_SPAN, ZMI_HOST_CHUNK ...) exist for the tests and the probe tools.  They are read
// only in a process that was started with ZMI_TUNING set: checked once, so a product process never calls getenv() on its
// hot path and its behaviour does not depend on whatever else is in the environment.
static const char* zmi_tune(const char* name) {
    static const bool enabled = getenv("ZMI_TUNING") != nullptr;
    return enabled ? getenv(name) : nullptr;
}
#define ZMI_HIP(call)                                              \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) return zmi_fail(ZMI_E_HIP, #call, e_); \
    } while (0)

// Every entry point runs on the context's device and leaves the calling thread's current device as it found it (a
// process that drives several GPUs -- one torch process with libz_mi355 loaded -- must not have its device switched
// under it).
struct zmi_dev_guard {
    int prev = -1;
    bool switched = false, ok = true;
    explicit zmi_dev_guard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { ok = hipSetDevice(dev) == hipSuccess; switched = ok && prev >= 0; }
    }
    ~zmi_dev_guard() { if (switched) (void)hipSetDevice(prev); }
};
#define ZMI_ON_DEVICE(c)                                         \
    zmi_dev_guard dev_guard_((c)->device);                       \
    if (!dev_guard_.ok) return zmi_fail(ZMI_E_HIP, "hipSetDevice")

struct zmi_buf {
    void* p = nullptr;
    size_t cap = 0;
};

struct zmi_timer {
    hipEvent_t a, b;
    int kernel;
};
enum { ZMI_K_CHECKSUM = 0, ZMI_K_LZ77 = 1, ZMI_K_ENCODE = 2, ZMI_K_INFLATE = 3, ZMI_K_VERIFY = 4, ZMI_K_PARSE = 5, ZMI_K_RESOLVE = 6, ZMI_K_PACK = 7 };

#define ZMI_HB_SLOTS 3   // host-buffer pipelines: chunks in flight (staging in, on the device, staging out)
struct zmi_ctx {
    bool timing = false;
    std::vector<zmi_timer> timers;
    int device = 0;
    uint64_t scratch_limit = 8ull << 30;
    zmi_buf match;    // u32 per position of the current group
    zmi_buf sums;     // adler[n] crc[n]
    zmi_buf pieces;   // per shard x piece compressed length
    zmi_buf inf_tmp;  // in_used[n] check[n] adler[n] crc[n] bm_off[n] (u64)
    zmi_buf inf_bm;   // inflate: 1 bit per output byte of the batch (where back-references start)
    zmi_buf inf_ptr;  // inflate of few streams: 4 B per output byte, the pointers of the jump resolve (resolve_jump.hip)
    zmi_buf st_in, st_out, st_meta;  // zmi_inflate_resume: staging of one host stream (kept across calls)
    zmi_buf sp_out;                  // zmi_inflate_split: the segments' decode regions
    zmi_buf st_scan;                 // zmi_inflate_blocks: the block scan's counter and list
    zmi_buf st_pin[2];               // zmi_d2h: two pinned 4 MiB pieces the decoded bytes of a single stream leave through
    hipEvent_t st_pin_ev[2]{};
    bool st_pin_ev_live = false;
    // host-buffer batches (zmi_deflate_batch): two slots cycle through copy-in / kernels / copy-out on three streams
    struct hb_slot { zmi_buf in, out, meta; hipEvent_t in_done, k_done, out_done; } hb[ZMI_HB_SLOTS];
    zmi_buf hb_slab[ZMI_HB_SLOTS];                              // device: the chunk's compressed streams packed densely (what travels back)
    zmi_buf hb_pin_in[ZMI_HB_SLOTS], hb_pin_out[ZMI_HB_SLOTS], hb_pin_meta[ZMI_HB_SLOTS];   // pinned host staging of the two slots
    hipStream_t hs_in = nullptr, hs_k = nullptr, hs_out = nullptr, hs_slab = nullptr;
    bool hb_live = false;
    uint64_t pinned_limit = 10ull << 30;   // host-buffer pipelines: pinned staging this context may hold (chunk sizes follow it; env ZMI_PINNED_MB)
    std::atomic<int> hb_out_tight{-1};     // zmi_inflate_batch: did the last chunk decoded fill most of its capacity?  (-1: nothing decoded yet)
    uint32_t last_codes_used = 0;    // zmi_inflate_resume: table entries of the most recent dynamic block of the last call (inflateCodesUsed)
    uint64_t inflate_out_limit = 0;  // output bytes one inflate batch may cover; 0 = scratch_limit
    bool inf_limit_exact = false;    // the caller of zmi_inflate_impl sized inflate_out_limit from the capacities of THIS call (host wrappers)
    uint32_t inf_mw_max = 512u;      // inflate launches of up to this many streams give every stream a 16-wave workgroup (inflate.hip)
    hipStream_t host_stream = nullptr;  // zmi_ctx_set_stream: where the host-buffer wrappers copy and launch
    hipStream_t side = nullptr;         // deflate: the wrapper checksums run here, beside the match search (zmi_deflate_impl)
    hipEvent_t ev_fork{}, ev_join{};
};

static int zmi_reserve(zmi_buf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = (bytes + 0xFFFFFull) & ~(size_t)0xFFFFFull;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) return zmi_fail(ZMI_E_NOMEM, "hipMalloc(scratch)", e);
    b.cap = want;
    return 0;
}

extern "C" const char* zmi_version(void) { return "zmi355 0.1.0 (gfx950; zlib ABI 1.3.0-zlib-rs-0.6.7 compatible)"; }
extern "C" const char* zmi_last_error(void) { return g_err.c_str(); }
// the other translation units of this library (exchange.hip) report through the same thread-local message
extern "C" void zmi_set_last_error(const char* what) { g_err = what ? what : ""; }

extern "C" int zmi_ctx_create(zmi_ctx** out, int device) {
    if (!out) return zmi_fail(ZMI_E_ARG, "zmi_ctx_create: null out pointer");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return zmi_fail(ZMI_E_NODEVICE, "no HIP device (this library has no CPU path)", e);
    if (device < 0 || device >= ndev) return zmi_fail(ZMI_E_ARG, "zmi_ctx_create: device index out of range");
    zmi_ctx* c = new zmi_ctx();
    c->device = device;
    const char* env = getenv("ZMI_SCRATCH_MB");
    if (env && atoll(env) > 0) c->scratch_limit = (uint64_t)atoll(env) << 20;
    env = getenv("ZMI_PINNED_MB");
    if (env && atoll(env) >= 64) c->pinned_limit = (uint64_t)atoll(env) << 20;
    *out = c;
    return ZMI_E_OK;
}

extern "C" int zmi_ctx_destroy(zmi_ctx* c) {
    if (!c) return ZMI_E_OK;
    zmi_dev_guard dev_guard_(c->device);
    if (c->match.p) (void)hipFree(c->match.p);
    if (c->sums.p) (void)hipFree(c->sums.p);
    if (c->pieces.p) (void)hipFree(c->pieces.p);
    if (c->inf_tmp.p) (void)hipFree(c->inf_tmp.p);
    if (c->inf_bm.p) (void)hipFree(c->inf_bm.p);
    if (c->inf_ptr.p) (void)hipFree(c->inf_ptr.p);
    if (c->st_in.p) (void)hipFree(c->st_in.p);
    if (c->st_out.p) (void)hipFree(c->st_out.p);
    if (c->st_meta.p) (void)hipFree(c->st_meta.p);
    if (c->side) { (void)hipStreamDestroy(c->side); (void)hipEventDestroy(c->ev_fork); (void)hipEventDestroy(c->ev_join); }
    if (c->sp_out.p) (void)hipFree(c->sp_out.p);
    if (c->st_scan.p) (void)hipFree(c->st_scan.p);
    for (int k = 0; k < 2; ++k) if (c->st_pin[k].p) (void)hipHostFree(c->st_pin[k].p);
    if (c->st_pin_ev_live) { (void)hipEventDestroy(c->st_pin_ev[0]); (void)hipEventDestroy(c->st_pin_ev[1]); }
    if (c->hb_live) {
        for (auto& sl : c->hb) {
            if (sl.in.p) (void)hipFree(sl.in.p);
            if (sl.out.p) (void)hipFree(sl.out.p);
            if (sl.meta.p) (void)hipFree(sl.meta.p);
            (void)hipEventDestroy(sl.in_done); (void)hipEventDestroy(sl.k_done); (void)hipEventDestroy(sl.out_done);
        }
        for (int k = 0; k < ZMI_HB_SLOTS; ++k) {
            if (c->hb_slab[k].p) (void)hipFree(c->hb_slab[k].p);
            if (c->hb_pin_in[k].p) (void)hipHostFree(c->hb_pin_in[k].p);
            if (c->hb_pin_out[k].p) (void)hipHostFree(c->hb_pin_out[k].p);
            if (c->hb_pin_meta[k].p) (void)hipHostFree(c->hb_pin_meta[k].p);
        }
        (void)hipStreamDestroy(c->hs_in); (void)hipStreamDestroy(c->hs_k); (void)hipStreamDestroy(c->hs_out); if (c->hs_slab) (void)hipStreamDestroy(c->hs_slab);
    }
    delete c;
    return ZMI_E_OK;
}

// per-kernel HIP-event timing (used by bench.py for the roofline line; off by default)
struct zmi_scope_timer {
    zmi_ctx* c;
    hipStream_t st;
    zmi_timer t;
    bool on;
    zmi_scope_timer(zmi_ctx* c_, int kernel, hipStream_t st_) : c(c_), st(st_), on(c_->timing) {
        if (!on) return;
        t.kernel = kernel;
        if (hipEventCreate(&t.a) != hipSuccess || hipEventCreate(&t.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(t.a, st);
    }
    ~zmi_scope_timer() {
        if (!on) return;
        (void)hipEventRecord(t.b, st);
        c->timers.push_back(t);
    }
};

extern "C" int zmi_ctx_device(const zmi_ctx* c) { return c ? c->device : -1; }

// The single-stream host wrappers (zmi_inflate_resume) copy and launch on this stream and wait for it alone -- a caller
// that gives every context its own non-blocking stream can drive several contexts from several threads without one
// call stalling the device for the others (the stream ABI does, zlib_abi.hip).  Default: the null stream.
extern "C" int zmi_ctx_set_stream(zmi_ctx* c, void* stream) {
    if (!c) return zmi_fail(ZMI_E_ARG, "null context");
    c->host_stream = (hipStream_t)stream;
    return ZMI_E_OK;
}

// decode-table entries (literal / length + distance; roots 9 / 8, exact-fit sub-tables) the device built for the most recent
// dynamic block met by zmi_inflate_resume calls on this context; 0 before the first one.  What inflateCodesUsed reports.
extern "C" int zmi_ctx_last_codes_used(zmi_ctx* c, uint32_t* entries) {
    if (!c || !entries) return zmi_fail(ZMI_E_ARG, "null argument");
    *entries = c->last_codes_used;
    return ZMI_E_OK;
}
extern "C" int zmi_ctx_reset_codes_used(zmi_ctx* c) {
    if (!c) return zmi_fail(ZMI_E_ARG, "null context");
    c->last_codes_used = 0;
    return ZMI_E_OK;
}

extern "C" int zmi_ctx_set_timing(zmi_ctx* c, int on) {
    if (!c) return zmi_fail(ZMI_E_ARG, "null context");
    c->timing = on != 0;
    return ZMI_E_OK;
}

// sums[k] = total milliseconds of kernel k since the last call, counts[k] = launches (k < 8)
extern "C" int zmi_ctx_get_timing(zmi_ctx* c, double* sums, uint32_t* counts) {
    if (!c || !sums || !counts) return zmi_fail(ZMI_E_ARG, "null argument");
    for (int k = 0; k < 8; ++k) { sums[k] = 0; counts[k] = 0; }
    for (zmi_timer& t : c->timers) {
        (void)hipEventSynchronize(t.b);
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess && t.kernel >= 0 && t.kernel < 8) {
            sums[t.kernel] += ms;
            counts[t.kernel]++;
        }
        (void)hipEventDestroy(t.a);
        (void)hipEventDestroy(t.b);
    }
    c->timers.clear();
    return ZMI_E_OK;
}

extern "C" int zmi_ctx_set_scratch_limit(zmi_ctx* c, uint64_t bytes) {
    if (!c || bytes < (64ull << 20)) return zmi_fail(ZMI_E_ARG, "scratch limit must be >= 64 MiB");
    c->scratch_limit = bytes;
    return ZMI_E_OK;
}

extern "C" int zmi_ctx_set_inflate_out_limit(zmi_ctx* c, uint64_t bytes) {
    if (!c) return zmi_fail(ZMI_E_ARG, "null context");
    c->inflate_out_limit = bytes < (1ull << 20) ? (1ull << 20) : bytes;
    return ZMI_E_OK;
}

extern "C" uint64_t zmi_deflate_bound(uint64_t n, int wrap) {
    // zlib-rs/src/deflate.rs:2975-2991 (compress_bound_help) with wrap overhead 0 / 6 / 18
    uint64_t w = wrap == ZMI_WRAP_ZLIB ? 6u : (wrap == ZMI_WRAP_GZIP ? 18u : 0u);
    uint64_t b = n + (n == 0) + (n < 9) + ((n + 7) >> 3) + 3 + w;
    // this engine's own worst case is a run of stored blocks (5 bytes per 32 KiB) -- far below the above
    return (b + 15) & ~15ull;
}

// search / parse effort per level: the MI355X analogue of CONFIGURATION_TABLE
// (zlib-rs/src/deflate/algorithm/mod.rs:69-82).  Every position is searched in parallel, so the
// chain budget is what the slowest lane of a wave spends; see DESIGN.md for the measured trade-off.
struct zmi_level_cfg {
    uint32_t chain, nice, good, lazy, tok;
};
// chain = candidates examined per position (the 4-byte probe counts as one); tok = tokens per sub-block of the
// encoder's adaptive block splitting.  Round 2 moved the curve: 16-bit hash heads (twice the buckets: fewer candidates
// that are hash collisions), the probe as a slim 8-byte compare outside the chain loop, limits for short far matches
// that follow the data (encode.hip enc_far_limits), 128 KiB pieces.  Level 6 = budget 5: lcet10.txt 2.79 -> 2.85 (the
// reference's level 6: 2.91), benchmark shards 2.247 -> see DESIGN.md section 9.
static const zmi_level_cfg kLevels[10] = {
    {0, 0, 0, 0, 4096},          // 0: stored
    {2, 16, 8, 0, 8192},         // 1  (greedy, lazy path of the encoder)
    {3, 32, 8, 0, 8192},         // 2  (greedy)
    // 3 ... 9: the token choice is the cost parse's (csrc/parse.hip) -- `lazy` is not used there, what separates the levels is the
    // search: budget, and the length at which a walk is content (good)
    // (round 6: four distinct rungs, as the reference's table has -- deflate/algorithm/mod.rs:72-76.  CPU emulator, 512 KiB inputs, benchmark
    // mix / lcet10.txt: budget 2 = 2.2294 / 2.811, budget 3 good 8 = 2.2476 / 2.869, budget 3 good 16 = 2.2554 / 2.881, budget 4 = 2.2670 / 2.909;
    // until round 5 levels 4 and 5 were one configuration and level 3 ran level 4's search)
    {2, 32, 8, 4, 4096},         // 3
    {3, 64, 8, 8, 4096},         // 4  (a walk is content with 8 equal bytes)
    {3, 128, 16, 16, 4096},      // 5  (... with 16)
    {4, 128, 16, 32, 4096},      // 6  (good 32 -> 16 in round 4: lz77 138.7 -> 130.1 ms, ratio 2.2550 -> 2.2532, lcet10.txt -0.04 %)
    {6, 128, 16, 32, 4096},      // 7
    // 8, 9 (round 5): budgets 10 and 16 on the SHORT-budget kernel (the walk ends at the first candidate equal in 16 bytes, the wave
    // extends it afterwards) instead of 14 and 22 on the deep one, whose in-loop extension makes a candidate cost 22 ms against 7
    // (per 16 Ki shards).  Measured with the cost parse, 8192 shards (profiles/r05_level9_budget_sweep.txt): deep 22 = 2.325 at
    // 24.8 GiB/s, deep 12 = 2.319 at 35.4, short 16 = 2.319 at 39.5, short 10 = 2.314 at 49.2 (round 4's level 9, lazy parse, deep
    // 22: 2.315 at 26.0).  The deep instantiation is gone from lz77.hip.
    {10, 258, 16, 128, 2048},    // 8
    {16, 258, 16, 258, 2048},    // 9
};

extern "C" int zmi_checksum_batch_dev(zmi_ctx* c, const void* d_data, const uint64_t* d_off, const uint32_t* d_len,
                                      uint32_t n, int kind, uint32_t* d_adler, uint32_t* d_crc, void* stream) {
    if (!c) return zmi_fail(ZMI_E_ARG, "null context");
    if (n == 0) return ZMI_E_OK;
    ZMI_ON_DEVICE(c);
    {
        zmi_scope_timer tm(c, ZMI_K_CHECKSUM, (hipStream_t)stream);
        zmi_launch_checksum((const uint8_t*)d_data, d_off, d_len, n, (uint32_t)kind, d_adler, d_crc, (hipStream_t)stream);
    }
    ZMI_HIP(hipGetLastError());
    return ZMI_E_OK;
}

extern "C" int zmi_gen_shards_dev(zmi_ctx* c, void* d_out, uint64_t seed, uint32_t first_shard, uint32_t n_shards,
                                  uint32_t shard_bytes, void* stream) {
    if (!c) return zmi_fail(ZMI_E_ARG, "null context");
    if (shard_bytes % 64u) return zmi_fail(ZMI_E_ARG, "shard_bytes must be a multiple of 64");
    ZMI_ON_DEVICE(c);
    zmi_launch_gen((uint8_t*)d_out, seed, first_shard, n_shards, shard_bytes, (hipStream_t)stream);
    ZMI_HIP(hipGetLastError());
    return ZMI_E_OK;
}

extern "C" int zmi_gen_shards_strided_dev(zmi_ctx* c, void* d_out, uint64_t seed, uint32_t first_shard, uint32_t shard_step,
                                          uint32_t n_shards, uint32_t shard_bytes, void* stream) {
    if (!c) return zmi_fail(ZMI_E_ARG, "null context");
    if (shard_bytes % 64u) return zmi_fail(ZMI_E_ARG, "shard_bytes must be a multiple of 64");
    ZMI_ON_DEVICE(c);
    zmi_launch_gen_strided((uint8_t*)d_out, seed, first_shard, shard_step, n_shards, shard_bytes, (hipStream_t)stream);
    ZMI_HIP(hipGetLastError());
    return ZMI_E_OK;
}

// ---- the stitch: strided slots -> dense slab, slab -> globally ordered output (pack.hip) ----
extern "C" int zmi_scan_sizes_dev(zmi_ctx* c, const uint32_t* d_len, uint32_t n, uint64_t* d_off, void* stream) {
    if (!c || !d_off || (!d_len && n)) return zmi_fail(ZMI_E_ARG, "null argument");
    ZMI_ON_DEVICE(c);
    zmi_launch_scan_sizes(d_len, n, d_off, (hipStream_t)stream);
    ZMI_HIP(hipGetLastError());
    return ZMI_E_OK;
}
extern "C" int zmi_copy_ranges_dev(zmi_ctx* c, const void* d_src, const uint64_t* d_src_off, uint64_t src_stride,
                                   const uint32_t* d_len, uint32_t n, uint32_t max_len, void* d_dst, const uint64_t* d_dst_off,
                                   uint64_t dst_cap, void* stream) {
    if (!c || !d_src || !d_len || !d_dst || !d_dst_off) return zmi_fail(ZMI_E_ARG, "null argument");
    if (n == 0) return ZMI_E_OK;
    ZMI_ON_DEVICE(c);
    {
        zmi_scope_timer tm(c, ZMI_K_PACK, (hipStream_t)stream);
        zmi_launch_copy_ranges((const uint8_t*)d_src, d_src_off, src_stride, d_len, n, (uint8_t*)d_dst, d_dst_off, dst_cap, max_len,
                               (hipStream_t)stream);
    }
    ZMI_HIP(hipGetLastError());
    return ZMI_E_OK;
}
extern "C" int zmi_pack_slab_dev(zmi_ctx* c, const void* d_slots, uint64_t slot_stride, const uint32_t* d_len, uint32_t n,
                                 void* d_slab, uint64_t slab_cap, uint64_t* d_off, void* stream) {
    int rc = zmi_scan_sizes_dev(c, d_len, n, d_off, stream);
    if (rc) return rc;
    const uint32_t max_len = slot_stride > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)slot_stride;
    return zmi_copy_ranges_dev(c, d_slots, nullptr, slot_stride, d_len, n, max_len, d_slab, d_off, slab_cap, stream);
}

static int zmi_deflate_impl(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n,
                            uint32_t max_len, int level, int strategy, int wrap, uint32_t chain_mode, uint32_t dict_len,
                            uint32_t window_bits, void* d_out, uint64_t out_stride, uint32_t* d_out_len, int32_t* d_status,
                            void* stream_);

extern "C" int zmi_deflate_batch_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                     uint32_t n, uint32_t max_len, int level, int strategy, int wrap, void* d_out,
                                     uint64_t out_stride, uint32_t* d_out_len, int32_t* d_status, void* stream_) {
    return zmi_deflate_impl(c, d_in, d_in_off, d_in_len, n, max_len, level, strategy, wrap, 0u, 0u, 15u, d_out, out_stride,
                            d_out_len, d_status, stream_);
}

// The shards are consecutive segments of ONE raw deflate stream, contiguous in d_in (d_in_off[i+1] = d_in_off[i] +
// d_in_len[i]).  Every segment starts byte aligned (the empty stored block of Z_SYNC_FLUSH,
// zlib-rs/src/deflate.rs:2733-2738) and may match into the up to 27 KiB in front of it (window carry-over), so
// splitting a stream costs no cold start; `finish` != 0 makes the last shard end the stream (BFINAL).
extern "C" int zmi_deflate_chain_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                     uint32_t n, uint32_t max_len, int level, int strategy, int finish, void* d_out,
                                     uint64_t out_stride, uint32_t* d_out_len, int32_t* d_status, void* stream_) {
    return zmi_deflate_impl(c, d_in, d_in_off, d_in_len, n, max_len, level, strategy, ZMI_WRAP_RAW, finish ? 1u : 2u, 0u, 15u,
                            d_out, out_stride, d_out_len, d_status, stream_);
}

// As zmi_deflate_chain_dev; additionally the dict_len bytes in front of the first segment (d_in + d_in_off[0] -
// dict_len ...) are history the stream may match into: a preset dictionary (deflateSetDictionary,
// zlib-rs/src/deflate.rs:499-564) or the tail of the input compressed by an earlier call on the same stream.
extern "C" int zmi_deflate_chain_dict_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                          uint32_t n, uint32_t max_len, int level, int strategy, int finish, uint32_t dict_len,
                                          void* d_out, uint64_t out_stride, uint32_t* d_out_len, int32_t* d_status,
                                          void* stream_) {
    return zmi_deflate_impl(c, d_in, d_in_off, d_in_len, n, max_len, level, strategy, ZMI_WRAP_RAW, finish ? 1u : 2u, dict_len,
                            15u, d_out, out_stride, d_out_len, d_status, stream_);
}

// The chained form for a stream opened with windowBits < 15 (deflateInit2_, zlib-rs/src/deflate.rs:252-312): no
// back-reference may reach farther than the reference's max_dist = 2^windowBits - MIN_LOOKAHEAD (deflate.rs:1423-1425),
// or an inflater that allocates the window the header announces rejects the stream ("invalid distance too far back").
extern "C" int zmi_deflate_chain_window_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                            uint32_t n, uint32_t max_len, int level, int strategy, int finish, uint32_t dict_len,
                                            uint32_t window_bits, void* d_out, uint64_t out_stride, uint32_t* d_out_len,
                                            int32_t* d_status, void* stream_) {
    if (window_bits < 9u || window_bits > 15u) return zmi_fail(ZMI_E_ARG, "window_bits must be 9..15");
    return zmi_deflate_impl(c, d_in, d_in_off, d_in_len, n, max_len, level, strategy, ZMI_WRAP_RAW, finish ? 1u : 2u, dict_len,
                            window_bits, d_out, out_stride, d_out_len, d_status, stream_);
}

static int zmi_deflate_impl(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len, uint32_t n,
                            uint32_t max_len, int level, int strategy, int wrap, uint32_t chain_mode, uint32_t dict_len,
                            uint32_t window_bits, void* d_out, uint64_t out_stride, uint32_t* d_out_len, int32_t* d_status,
                            void* stream_) {
    if (!c) return zmi_fail(ZMI_E_ARG, "null context");
    if (level == -1) level = 6;
    if (level < 0 || level > 9) return zmi_fail(ZMI_E_ARG, "level must be -1..9");
    if (strategy < 0 || strategy > 4) return zmi_fail(ZMI_E_ARG, "strategy must be 0..4");
    if (wrap < ZMI_WRAP_RAW || wrap > ZMI_WRAP_GZIP) return zmi_fail(ZMI_E_ARG, "wrap must be raw/zlib/gzip");
    if (out_stride % 16u || out_stride < zmi_deflate_bound(max_len, wrap))
        return zmi_fail(ZMI_E_ARG, "out_stride must be a multiple of 16 and >= zmi_deflate_bound(max_len)");
    if (out_stride > 0xFFFFFFF0ull) return zmi_fail(ZMI_E_ARG, "shards larger than 3.5 GiB are not supported");
    if (n == 0) return ZMI_E_OK;
    hipStream_t stream = (hipStream_t)stream_;
    ZMI_ON_DEVICE(c);

    // wrapper checksums
    int rc = zmi_reserve(c->sums, (size_t)n * 8u);
    if (rc) return rc;
    uint32_t* d_adler = (uint32_t*)c->sums.p;
    uint32_t* d_crc = d_adler + n;
    uint32_t kind = wrap == ZMI_WRAP_ZLIB ? 1u : (wrap == ZMI_WRAP_GZIP ? 2u : 0u);
    // The checksums are needed by the encoder's trailers only: they run on a side stream beside the match search (one is
    // HBM-bound, the other instruction-bound; in a launch of a few hundred shards the checksum kernel is one workgroup per
    // shard with nothing to hide its latency behind: 4.5 ms in front of 6.4 ms of lz77 for 512 shards) and join in front of
    // the first encode launch.
    bool forked = false;
    struct join_guard { zmi_ctx* c; hipStream_t st; bool& f; ~join_guard() { if (f) (void)hipStreamWaitEvent(st, c->ev_join, 0); } } join_on_exit{c, stream, forked};
    if (kind) {
        if (!c->side) {
            ZMI_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
            ZMI_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            ZMI_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        ZMI_HIP(hipEventRecord(c->ev_fork, stream));
        ZMI_HIP(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        {
            zmi_scope_timer tm(c, ZMI_K_CHECKSUM, c->side);
            zmi_launch_checksum((const uint8_t*)d_in, d_in_off, d_in_len, n, kind, d_adler, d_crc, c->side);
        }
        ZMI_HIP(hipEventRecord(c->ev_join, c->side));
        forked = true;
    }

    const zmi_level_cfg& L = kLevels[level];
    zmi_lz_params lp;
    lp.max_chain = L.chain;
    lp.nice_len = L.nice;
    lp.good_len = L.good;
    lp.max_dist = 32768u;  // clamped to the ring-buffer limit by the launcher
    if (window_bits < 15u) lp.max_dist = (1u << window_bits) - 262u;   // w_size - MIN_LOOKAHEAD (deflate.rs:1423-1425)
    lp.hash6 = 1u;  // 6-byte-hash chain + one 4-byte probe: ~1.5x fewer chain steps than a 4-byte chain at equal ratio
    lp.claim = 64u;
    const char* claim_env = zmi_tune("ZMI_CLAIM");
    if (claim_env) lp.claim = (uint32_t)atoi(claim_env);
    const char* md_env = zmi_tune("ZMI_MAXDIST");
    if (md_env && atoi(md_env) > 0 && (uint32_t)atoi(md_env) < lp.max_dist) lp.max_dist = (uint32_t)atoi(md_env);
    const char* h6_env = zmi_tune("ZMI_HASH6");
    if (h6_env) lp.hash6 = atoi(h6_env) ? 1u : 0u;
    zmi_enc_params ep;
    ep.max_lazy = L.lazy;
    ep.lazy2 = 1u;
    ep.lazy3 = 2u;
    if (const char* l3 = zmi_tune("ZMI_LAZY3")) ep.lazy3 = (uint32_t)atoi(l3);
    if (const char* l2 = zmi_tune("ZMI_LAZY2")) ep.lazy2 = (uint32_t)atoi(l2);
    ep.wrap = (uint32_t)wrap;
    ep.level = (uint32_t)level;
    ep.block_span = 65536u;    // one encoder wave per 64 KiB piece, 16 per 1 MiB shard (128 KiB pieces: +0.15 % ratio, but 137 -> 161 ms
                               // per 16 Ki shards: fewer, longer waves fill the chip worse)
    ep.strategy = (uint32_t)strategy;
    ep.chain_mode = chain_mode;
    ep.last_shard = n - 1u;
    if (level == 0) ep.strategy = 100u;           // stored blocks only (deflate_stored)
    ep.cost_parse = (level >= 3 && strategy != 2) ? 1u : 0u;
    if (const char* cp = zmi_tune("ZMI_COST_PARSE")) ep.cost_parse = atoi(cp) ? 1u : 0u;
    if (strategy == 2) { lp.max_chain = 0; lp.max_dist = 0; }   // Z_HUFFMAN_ONLY: literals only -- with no distance allowed no
                                                                 // position gets a candidate (a zero chain budget alone still
                                                                 // examines the first one)
    if (strategy == 3) lp.max_dist = 1;           // Z_RLE: distance-1 matches only
    const char* chain_env = zmi_tune("ZMI_CHAIN");  // tuning aid: override the chain budget of the selected level
    if (chain_env && atoi(chain_env) > 0 && level > 0 && strategy != 2) lp.max_chain = (uint32_t)atoi(chain_env);
    const char* good_env = zmi_tune("ZMI_GOOD");
    if (good_env && atoi(good_env) > 0) lp.good_len = (uint32_t)atoi(good_env);
    const char* nice_env = zmi_tune("ZMI_NICE");
    if (nice_env && atoi(nice_env) > 0) lp.nice_len = (uint32_t)atoi(nice_env);
    const char* lazy_env = zmi_tune("ZMI_LAZY");
    if (lazy_env && atoi(lazy_env) >= 0) ep.max_lazy = (uint32_t)atoi(lazy_env);
    lp.carry = chain_mode != 0u ? 1u : 0u;   // segments of one stream: a segment sees the window in front of it
    lp.dict_len = chain_mode != 0u ? dict_len : 0u;
    if (const char* cv = zmi_tune("ZMI_CARRY")) lp.carry = (chain_mode != 0u && atoi(cv)) ? 1u : 0u;
    // hash-building waves: 1 for the deep chains (the searchers are the whole kernel there), 2 for the short ones (14 searcher
    // waves outrun one producer: budget 5 measured 227 ms with two, 273 ms with one), 3 at level 1, whose searchers do so
    // little per position that even two producers set the pace (97.1 -> 95.2 ms per 16 Ki shards; at level 3 a third
    // producer already costs more as a missing searcher than it brings: 114.2 -> 120.5 ms)
    lp.producers = L.chain > 8u ? 1u : (L.chain <= 2u ? 3u : 2u);   // (budgets 10 / 16: one producer measured 1-2 % ahead of two)
    if (const char* pv = zmi_tune("ZMI_PRODUCERS")) lp.producers = (uint32_t)atoi(pv);
    lp.dbg = 0u;
    if (const char* dv = zmi_tune("ZMI_LZ_DBG")) lp.dbg = (uint32_t)atoi(dv);
    lp.barren_chain = 1u;
    if (const char* bv = zmi_tune("ZMI_BARREN_CHAIN")) lp.barren_chain = (uint32_t)atoi(bv);
    // short far matches are judged by the encoder, block by block, from the codes it just used (enc_far_limits); the
    // search reports every match of 4+ bytes
    lp.far4 = 32768u;
    lp.far5 = 32768u;
    ep.far4 = 2048u;    // the first block of a piece, before any code exists
    ep.far5 = 16384u;
    if (const char* f4 = zmi_tune("ZMI_FAR4")) ep.far4 = (uint32_t)atoi(f4);
    if (const char* f5 = zmi_tune("ZMI_FAR5")) ep.far5 = (uint32_t)atoi(f5);
    ep.block_tokens = L.tok;
    ep.split_hdr_bits = 640u;
    ep.min_sub_span = L.tok >= 4096u ? 5120u : 0u;   // (levels 8 and 9 cut finer, 2048 tokens, and keep every block they can get)
                               // literal-dense data: a sub-block of 4096 tokens is little more than 4 KiB of input, and a block of its own
                               // is a full tree construction; measured on the benchmark mix: 0 -> 103.3 ms encode at ratio 2.2635, 5120 ->
                               // 98.9 ms at 2.2567, 8192 -> 94.6 ms at 2.2440
    if (const char* ms = zmi_tune("ZMI_MIN_SUB_SPAN")) ep.min_sub_span = (uint32_t)atoi(ms);
    if (const char* hb = zmi_tune("ZMI_SPLIT_HDR")) ep.split_hdr_bits = (uint32_t)atoi(hb);
    if (const char* bt = zmi_tune("ZMI_BLOCK_TOKENS")) ep.block_tokens = (uint32_t)atoi(bt) >= 64u ? (uint32_t)atoi(bt) : 64u;
    // a launch of a few shards -- the 64 KiB segments of one deflate() call, a compress2() -- would be a few dozen encoder waves of
    // 0.7 ... 1.8 ms each on 256 CUs (profiles/r05_stream_deflate_trace.txt): pieces of 8 KiB there, eight waves per segment
    // (a marker and a fresh block per 8 KiB, only where the chip would otherwise stand empty: the 15 MiB stream of
    // tools/gpu_stream_deflate_probe.py 2.3362 with 16 KiB pieces -- 0.2 ... 0.5 ms of encoder per 4 MiB call --, 2.3304 with 8 KiB,
    // 2.3076 with 4 KiB)
    // (only for the segments of ONE stream -- chain mode: what a batch of independent shards compresses to does not depend on how
    // many of them a launch holds, tests/test_gpu_parity.py::test_host_batch_pipeline_on_gpu)
    const bool small_launch = chain_mode != 0u && (uint64_t)n * ((max_len + 65535u) / 65536u) < 512u;
    if (small_launch) ep.block_span = 8192u;
    const char* span_env = zmi_tune("ZMI_BLOCK_SPAN");
    if (span_env && atoi(span_env) >= 64) ep.block_span = (uint32_t)atoi(span_env);

    // match/token scratch: one u32 per position, shards padded to a multiple of 64 positions; with the cost parse two more bits
    // per position behind it (its decisions, 16 bytes per 64 positions)
    const uint64_t stride = ((uint64_t)max_len + 63u) & ~63ull;
    uint64_t per_shard = (stride ? stride : 64u) * 4u;
    const uint64_t dec_bytes = ep.cost_parse ? per_shard / 16u : 0u;
    uint64_t group = c->scratch_limit / (per_shard + dec_bytes);
    if (group == 0) return zmi_fail(ZMI_E_NOMEM, "scratch limit too small for one shard");
    if (group > n) group = n;
    rc = zmi_reserve(c->match, (size_t)(group * (per_shard + dec_bytes)));
    if (rc) return rc;
    uint32_t* const d_dec = ep.cost_parse ? (uint32_t*)((uint8_t*)c->match.p + group * per_shard) : nullptr;
    // the encoder runs `pieces` waves per shard (byte-aligned sub-streams, concatenated afterwards):
    // one piece per block_span of input, at most 16
    uint32_t pieces = (max_len + ep.block_span - 1u) / ep.block_span;
    if (pieces < 1u) pieces = 1u;
    if (pieces > 16u) pieces = 16u;
    const char* pieces_env = zmi_tune("ZMI_PIECES");
    if (pieces_env && atoi(pieces_env) >= 1 && atoi(pieces_env) <= 64) pieces = (uint32_t)atoi(pieces_env);
    // every piece region must hold its worst case (stored blocks + marker) and stay 16-byte aligned
    while (pieces > 1u) {
        uint64_t region = (out_stride / pieces) & ~15ull;
        uint64_t psize = (((uint64_t)max_len + pieces - 1u) / pieces + 63u) & ~63ull;
        // worst case: every block stored (5 header bytes + byte alignment); a block holds at least one sub-block of
        // block_tokens tokens (>= as many input bytes), stored sub-blocks are cut at 32 KiB
        const uint64_t min_block = ep.block_tokens < 32768u ? ep.block_tokens : 32768u;
        uint64_t worst = psize + 6u * (psize / min_block + 3u) + 32u;
        if (region >= worst) break;
        --pieces;
    }
    rc = zmi_reserve(c->pieces, (size_t)n * pieces * 4u);
    if (rc) return rc;
    for (uint64_t first = 0; first < n; first += group) {
        uint32_t cnt = (uint32_t)((n - first < group) ? (n - first) : group);
        {
            zmi_scope_timer tm(c, ZMI_K_LZ77, stream);
            int lrc = zmi_launch_lz77((const uint8_t*)d_in, d_in_off, d_in_len, (uint32_t)first, cnt, (uint32_t*)c->match.p,
                                      per_shard / 4u, lp, stream);
            if (lrc) return zmi_fail(ZMI_E_HIP, "lz77 launch setup", (hipError_t)lrc);
        }
        if (ep.cost_parse) {
            zmi_scope_timer tm(c, ZMI_K_PARSE, stream);
            zmi_launch_parse(d_in_len, (uint32_t)first, cnt, max_len, (const uint32_t*)c->match.p, per_shard / 4u, d_dec, dec_bytes / 4u, pieces,
                             (uint32_t)strategy, small_launch ? 4u : 64u, stream);
        }
        if (forked) { ZMI_HIP(hipStreamWaitEvent(stream, c->ev_join, 0)); forked = false; }
        zmi_scope_timer tm2(c, ZMI_K_ENCODE, stream);
        zmi_launch_encode((const uint8_t*)d_in, d_in_off, d_in_len, (uint32_t)first, cnt, (uint32_t*)c->match.p,
                          per_shard / 4u, d_adler, d_crc, (uint8_t*)d_out, out_stride, (uint32_t)out_stride, d_out_len,
                          d_status, pieces, (uint32_t*)c->pieces.p, d_dec, dec_bytes / 4u, ep, stream);
    }
    ZMI_HIP(hipGetLastError());
    return ZMI_E_OK;
}

// extended form used by the zlib stream ABI: also returns consumed input bytes and why a stream stopped
extern "C" int zmi_inflate_batch_dict_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                          uint32_t n, int wrap, void* d_out, const uint64_t* d_out_off,
                                          const uint32_t* d_out_cap, const uint32_t* d_out_hist, uint32_t* d_out_len,
                                          int32_t* d_status, uint32_t* d_in_used, int32_t* d_detail, void* stream_);
extern "C" int zmi_inflate_batch_dev_ex(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                        uint32_t n, int wrap, void* d_out, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                                        uint32_t* d_out_len, int32_t* d_status, uint32_t* d_in_used, int32_t* d_detail,
                                        void* stream_) {
    return zmi_inflate_batch_dict_dev(c, d_in, d_in_off, d_in_len, n, wrap, d_out, d_out_off, d_out_cap, nullptr, d_out_len,
                                      d_status, d_in_used, d_detail, stream_);
}

// d_out_hist (may be null): per stream, the number of bytes directly in front of its output region that hold a
// preset dictionary (inflateSetDictionary, zlib-rs/src/inflate.rs:2492-2536; at most 32768 are ever referenced)
static int zmi_inflate_impl(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                            uint32_t n, int wrap, void* d_out, const uint64_t* d_out_off,
                            const uint32_t* d_out_cap, const uint32_t* d_out_hist, uint32_t* d_out_len,
                            int32_t* d_status, uint32_t* d_in_used, int32_t* d_detail, const uint32_t* d_in_bit,
                            uint32_t* d_resume, void* stream_, bool decode_only = false) {
    if (!c) return zmi_fail(ZMI_E_ARG, "null context");
    if (wrap < ZMI_WRAP_RAW || wrap > ZMI_WRAP_AUTO) return zmi_fail(ZMI_E_ARG, "wrap must be raw/zlib/gzip/auto");
    if (n == 0) return ZMI_E_OK;
    hipStream_t stream = (hipStream_t)stream_;
    ZMI_ON_DEVICE(c);
    int rc = zmi_reserve(c->inf_tmp, (size_t)n * 28u);
    if (rc) return rc;
    uint64_t* d_bm_off = (uint64_t*)c->inf_tmp.p;
    uint32_t* d_used = (uint32_t*)(d_bm_off + n);
    uint32_t* d_check = d_used + n;
    uint32_t* d_adler = d_check + n;
    uint32_t* d_crc = d_adler + n;
    uint32_t* d_order = d_crc + n;   // workgroup -> stream: largest compressed size first (zmi_inflate_order_kernel)
    if (d_in_used) d_used = d_in_used;
    // bitmap scratch: 1 bit per byte of output capacity (+2 words per stream).  The capacities live on the
    // device, so the size comes from the context's limit; streams beyond it report Z_MEM_ERROR.
    const uint64_t out_limit = c->inflate_out_limit ? c->inflate_out_limit : c->scratch_limit;
    rc = zmi_reserve(c->inf_bm, (size_t)(out_limit / 8u) + (size_t)n * 16u);
    if (rc) return rc;
    const uint64_t bm_words = ((out_limit / 8u) + (uint64_t)n * 16u) / 8u;
    uint32_t mw_max = c->inf_mw_max;
    if (const char* mv = zmi_tune("ZMI_INF_MW_MAX")) mw_max = (uint32_t)atoi(mv);   // (tests: both selections, per call)
    {
        zmi_scope_timer tm(c, ZMI_K_INFLATE, stream);
        int lrc = zmi_launch_inflate((const uint8_t*)d_in, d_in_off, d_in_len, n, (uint32_t)wrap, (uint8_t*)d_out, d_out_off, d_out_cap,
                                     d_out_len, d_used, d_check, d_status, (uint64_t*)c->inf_bm.p, bm_words, d_bm_off, d_out_hist, d_in_bit, d_resume,
                                     d_order, mw_max, stream);
        if (lrc) return zmi_fail(ZMI_E_HIP, "inflate launch setup", (hipError_t)lrc);
    }
    // The resolve pass.  Thousands of streams: one wave per stream fills its holes in order.  A handful of streams would leave
    // the chip empty that way (a 1 MiB stream alone: 8 ms): their back-references are resolved by pointer jumping over all
    // output bytes at once (resolve_jump.hip), which costs 4 B of scratch per byte of capacity.
    // (below 256 KiB of capacity the serial pass takes less than the jump pass's two dozen launches)
    // (the pass sweeps out_limit indices whatever the streams' real capacities are -- they are device data: a limit far above what n
    // streams plausibly hold, 64 MiB each, says the caller did not size it for this call, and the serial pass is the safer choice)
    // (callers that sized the limit from this call's capacities -- zmi_inflate_resume, the host-buffer batch -- are exact: one stream of
    // several hundred MiB keeps the jump pass)
    // (round 6: ONE stream whose limit was sized for this call -- a streaming inflate() of the zlib ABI, zmi_inflate_resume -- takes the
    // jump pass from 64 KiB on: its serial pass is one wave alone on the chip, 389 us for the ~40 KiB of an open block, against a dozen
    // launches of a few microseconds; the reference's chunk sweep at 1 / 4 / 16 / 64 KiB pieces: 0.90 / 0.248 / 0.069 / 0.020 s ->
    // 0.59 / 0.153 / 0.046 / 0.018 s for 4 MiB, profiles/r06_eager_inflate_trace.txt)
    const uint64_t jump_min = (n == 1u && c->inf_limit_exact) ? (64ull << 10) : (256ull << 10);
    bool jump = n <= 16u && out_limit <= (1ull << 30) && out_limit >= jump_min &&
                (c->inf_limit_exact || out_limit <= (uint64_t)n * (64ull << 20) + (1ull << 20));
    if (!decode_only) {
    if (const char* jv = zmi_tune("ZMI_INF_JUMP")) jump = atoi(jv) != 0 && out_limit <= (1ull << 30);
    if (jump && zmi_reserve(c->inf_ptr, (size_t)bm_words * 256u + 512u) != 0) jump = false;   // (no room: the serial pass needs none)
    {
        zmi_scope_timer tm(c, ZMI_K_RESOLVE, stream);
        int lrc;
        if (jump) {
            uint32_t rounds = 2u;   // ceil(log3(capacity)) + 1: a round follows two pointers (resolve_jump.hip)
            for (uint64_t span = 3u; rounds < 34u && span < out_limit; span *= 3u) ++rounds;
            lrc = zmi_launch_resolve_jump((uint8_t*)d_out, d_out_off, d_out_len, n, (const uint64_t*)c->inf_bm.p, d_bm_off, (int32_t*)c->inf_ptr.p,
                                          bm_words * 64ull, rounds, (uint32_t*)((uint8_t*)c->inf_ptr.p + (size_t)bm_words * 256u), stream);
        } else {
            lrc = zmi_launch_inflate_resolve((uint8_t*)d_out, d_out_off, d_out_len, n, (const uint64_t*)c->inf_bm.p, d_bm_off, d_out_hist, d_order, stream);
        }
        if (lrc) return zmi_fail(ZMI_E_HIP, "inflate resolve launch setup", (hipError_t)lrc);
    }
    }   // (decode_only: the caller resolves -- zmi_inflate_split stitches the segments of one stream first)
    if (wrap != ZMI_WRAP_RAW) {
        uint32_t kind = wrap == ZMI_WRAP_ZLIB ? 1u : (wrap == ZMI_WRAP_GZIP ? 2u : 3u);
        zmi_scope_timer tm(c, ZMI_K_CHECKSUM, stream);
        zmi_launch_checksum((const uint8_t*)d_out, d_out_off, d_out_len, n, kind, d_adler, d_crc, stream);
    }
    {
        zmi_scope_timer tm(c, ZMI_K_VERIFY, stream);
        zmi_launch_inflate_verify((const uint8_t*)d_in, d_in_off, d_in_len, n, (uint32_t)wrap, d_check, d_adler, d_crc, d_status,
                                  d_detail, stream);
    }
    ZMI_HIP(hipGetLastError());
    return ZMI_E_OK;
}

extern "C" int zmi_inflate_batch_dict_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                          uint32_t n, int wrap, void* d_out, const uint64_t* d_out_off,
                                          const uint32_t* d_out_cap, const uint32_t* d_out_hist, uint32_t* d_out_len,
                                          int32_t* d_status, uint32_t* d_in_used, int32_t* d_detail, void* stream_) {
    return zmi_inflate_impl(c, d_in, d_in_off, d_in_len, n, wrap, d_out, d_out_off, d_out_cap, d_out_hist, d_out_len, d_status,
                            d_in_used, d_detail, nullptr, nullptr, stream_);
}

// Resumable raw-deflate decode (the device half of a streaming inflate; the facts the reference keeps in
// Mode / BitReader / Window, zlib-rs/src/inflate.rs:288-320, reduced to a block-boundary checkpoint).
// Stream i starts at bit d_in_bit[i] (0..7; array may be NULL) of its first byte, with d_out_hist[i] bytes of
// earlier output in front of its output region.  d_resume[4i..4i+3] = {byte, bit, output bytes, complete}: the
// start of the block the decode stopped in (status Z_BUF_ERROR: more input or more room needed), or the first bit
// behind the final block (complete = 1).  d_out_len[i] counts everything decoded, also the valid part of the
// unfinished block; a later call that starts at the checkpoint reproduces those bytes and continues.
extern "C" int zmi_inflate_resume_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                      const uint32_t* d_in_bit, uint32_t n, void* d_out, const uint64_t* d_out_off,
                                      const uint32_t* d_out_cap, const uint32_t* d_out_hist, uint32_t* d_out_len,
                                      int32_t* d_status, uint32_t* d_in_used, int32_t* d_detail, uint32_t* d_resume,
                                      void* stream_) {
    if (!d_resume) return zmi_fail(ZMI_E_ARG, "d_resume is required");
    return zmi_inflate_impl(c, d_in, d_in_off, d_in_len, n, ZMI_WRAP_RAW, d_out, d_out_off, d_out_cap, d_out_hist, d_out_len,
                            d_status, d_in_used, d_detail, d_in_bit, d_resume, stream_);
}

extern "C" int zmi_inflate_batch_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                     uint32_t n, int wrap, void* d_out, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                                     uint32_t* d_out_len, int32_t* d_status, void* stream_) {
    return zmi_inflate_batch_dev_ex(c, d_in, d_in_off, d_in_len, n, wrap, d_out, d_out_off, d_out_cap, d_out_len, d_status, nullptr,
                                    nullptr, stream_);
}

// ---------------- host-buffer wrappers ----------------
static int zmi_host_pipeline_init(zmi_ctx* c) {   // three streams (in / kernels / out) and the events of the two slots
    if (c->hb_live) return 0;
    ZMI_HIP(hipStreamCreateWithFlags(&c->hs_in, hipStreamNonBlocking));
    ZMI_HIP(hipStreamCreateWithFlags(&c->hs_k, hipStreamNonBlocking));
    ZMI_HIP(hipStreamCreateWithFlags(&c->hs_out, hipStreamNonBlocking));
    ZMI_HIP(hipStreamCreateWithFlags(&c->hs_slab, hipStreamNonBlocking));
    for (auto& sl : c->hb) {
        ZMI_HIP(hipEventCreateWithFlags(&sl.in_done, hipEventDisableTiming));
        ZMI_HIP(hipEventCreateWithFlags(&sl.k_done, hipEventDisableTiming));
        ZMI_HIP(hipEventCreateWithFlags(&sl.out_done, hipEventDisableTiming));
    }
    c->hb_live = true;
    return 0;
}
struct zmi_dev_alloc {
    std::vector<void*> ptrs;
    ~zmi_dev_alloc() { for (void* p : ptrs) (void)hipFree(p); }
    void* get(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return p;
    }
};

static int zmi_deflate_batch_simple(zmi_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                 int level, int strategy, int wrap, uint8_t* out, uint64_t out_stride, uint32_t* out_len,
                                 int32_t* status) {
    if (!c || (!in && n) || !in_off || !in_len || !out || !out_len || !status) return zmi_fail(ZMI_E_ARG, "null argument");
    if (n == 0) return ZMI_E_OK;
    ZMI_ON_DEVICE(c);
    // pack the shards back to back on the device, 16-byte aligned starts (fast load path)
    std::vector<uint64_t> doff(n);
    uint64_t total = 0;
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n; ++i) {
        doff[i] = total;
        total += ((uint64_t)in_len[i] + 15u) & ~15ull;
        if (in_len[i] > max_len) max_len = in_len[i];
    }
    if (out_stride % 16u || out_stride < zmi_deflate_bound(max_len, wrap))
        return zmi_fail(ZMI_E_ARG, "out_stride must be a multiple of 16 and >= zmi_deflate_bound(max_len)");
    zmi_dev_alloc A;
    uint8_t* d_in = (uint8_t*)A.get(total + 16);
    uint64_t* d_off = (uint64_t*)A.get((size_t)n * 8);
    uint32_t* d_len = (uint32_t*)A.get((size_t)n * 4);
    uint8_t* d_out = (uint8_t*)A.get((size_t)n * out_stride);
    uint32_t* d_olen = (uint32_t*)A.get((size_t)n * 4);
    int32_t* d_st = (int32_t*)A.get((size_t)n * 4);
    if (!d_in || !d_off || !d_len || !d_out || !d_olen || !d_st) return zmi_fail(ZMI_E_NOMEM, "hipMalloc(batch buffers)");
    for (uint32_t i = 0; i < n;) {   // shards that lie back to back on both sides travel in one copy
        uint32_t j = i;
        uint64_t bytes = in_len[i];
        while (j + 1 < n && in_off[j + 1] == in_off[j] + in_len[j] && doff[j + 1] == doff[j] + in_len[j] && bytes < (1ull << 30)) bytes += in_len[++j];
        if (bytes) ZMI_HIP(hipMemcpy(d_in + doff[i], in + in_off[i], bytes, hipMemcpyHostToDevice));
        i = j + 1;
    }
    ZMI_HIP(hipMemcpy(d_off, doff.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    ZMI_HIP(hipMemcpy(d_len, in_len, (size_t)n * 4, hipMemcpyHostToDevice));
    int rc = zmi_deflate_batch_dev(c, d_in, d_off, d_len, n, max_len, level, strategy, wrap, d_out, out_stride, d_olen, d_st,
                                   nullptr);
    if (rc) return rc;
    ZMI_HIP(hipDeviceSynchronize());
    ZMI_HIP(hipMemcpy(out_len, d_olen, (size_t)n * 4, hipMemcpyDeviceToHost));
    ZMI_HIP(hipMemcpy(status, d_st, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i)
        if (status[i] == 0 && out_len[i])
            ZMI_HIP(hipMemcpy(out + (uint64_t)i * out_stride, d_out + (uint64_t)i * out_stride, out_len[i], hipMemcpyDeviceToHost));
    return ZMI_E_OK;
}


// ---- one host stream decoded as segments (what the stream ABI's inflate() uses for streams with flush points) ----
// A deflate stream that was written with Z_SYNC_FLUSH / Z_FULL_FLUSH points (pigz, this engine's own deflate(): a point every
// 64 KiB of input) carries the marker 00 00 FF FF in front of every byte-aligned restart, and the decode pass needs no window:
// so the pieces between markers can be decoded side by side like the streams of a batch -- thousands of waves instead of
// the one workgroup a single stream gets -- and stitched afterwards (resolve_jump.hip: the back-references of the whole
// output are resolved at once, across segment borders).  `seg_start[0..nseg)` are byte offsets into `in` (ascending,
// seg_start[0] = 0) proposed by the caller, e.g. the bytes behind every marker it found; nothing is taken on trust:
// a segment counts only if the decode of the one before ended exactly at its first byte, on a block boundary, and whatever
// does not check out (a marker that was data, a segment that outgrew its room, a distance that reaches in front of the
// real history) makes the call fall back to the serial decode from there.  Results are those of zmi_inflate_resume on the
// same input: same bytes, status, detail, in_used, resume.
extern "C" int zmi_inflate_resume(zmi_ctx* c, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist,
                                  uint32_t hist_len, uint8_t* out, uint32_t out_cap, uint32_t* out_len, int32_t* status,
                                  int32_t* detail, uint32_t* in_used, uint32_t* resume);
extern "C" int zmi_launch_resolve_jump_segments(uint8_t* d_fin, const uint64_t* d_soff, uint32_t nseg, const uint64_t* d_bitmap,
                                                const uint64_t* d_bm_off, int32_t* d_ptr, uint64_t total, uint32_t hist_len,
                                                uint32_t rounds, uint32_t* d_flags, uint32_t* d_err, const uint64_t* d_one_off,
                                                const uint32_t* d_one_len, hipStream_t stream);
// Device -> caller's (pageable) memory for the single-stream paths.  hipMemcpy into pageable memory is the runtime's own bounce
// through a small pinned buffer, one thread, 2-4 GB/s: the 15 MiB of a decoded stream took longer to leave the device than to
// decode (measured: 6.4 of 7.5 ms).  Here the bytes go out in 4 MiB pieces through two pinned buffers of the context -- the DMA of
// a piece runs while a few host threads copy the piece before it to where it belongs.  Without pinned memory: the plain copy.
struct zmi_copy_job;
static void zmi_parallel_copy(const std::vector<zmi_copy_job>& jobs, unsigned T);
static int zmi_reserve_pinned(zmi_buf& b, size_t bytes);
static unsigned zmi_host_threads();
static int zmi_d2h(zmi_ctx* c, void* dst, const void* src, size_t n, hipStream_t hs);

// (this function copies asynchronously from vectors and locals of its own frame: an error return must not leave such a copy in
// flight -- the stream is drained first)
#define ZMI_HIPS(call)                                                                  \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) { (void)hipStreamSynchronize(c->host_stream); return zmi_fail(ZMI_E_HIP, #call, e_); } \
    } while (0)
// seg_bit (may be null: all zero): segment j starts at bit seg_bit[j] of byte seg_start[j] -- block boundaries found by the scan
// of zmi_inflate_blocks lie anywhere; in_on_device: c->st_in already holds the stream (that scan has copied it)
static int zmi_split_core(zmi_ctx* c, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist,
                          uint32_t hist_len, uint8_t* out, uint32_t out_cap, const uint32_t* seg_start, const uint32_t* seg_bit, uint32_t nseg,
                          bool in_on_device, uint32_t* out_len, int32_t* status, int32_t* detail, uint32_t* in_used, uint32_t* resume,
                          uint32_t* segments_used) {
    if (segments_used) *segments_used = 0;
    if (!c || (!in && in_len) || (!hist && hist_len) || (!out && out_cap) || !out_len || !status || !detail || !in_used || !resume || !seg_start)
        return zmi_fail(ZMI_E_ARG, "null argument");
    if (nseg < 2u || in_bit > 7u || out_cap > (1u << 30) || in_len > 0xFFFFFF00u || seg_start[0] != 0u)
        return zmi_inflate_resume(c, in, in_len, in_bit, hist, hist_len, out, out_cap, out_len, status, detail, in_used, resume);
    for (uint32_t j = 1; j < nseg; ++j)
        if (seg_start[j] <= seg_start[j - 1u] || seg_start[j] >= in_len || (seg_bit && seg_bit[j] > 7u))
            return zmi_inflate_resume(c, in, in_len, in_bit, hist, hist_len, out, out_cap, out_len, status, detail, in_used, resume);
    ZMI_ON_DEVICE(c);
    const bool sp_trace = zmi_tune("ZMI_SPLIT_TRACE") != nullptr;
    auto sp_now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double sp_t0 = sp_trace ? sp_now() : 0.0;
    if (hist_len > 32768u) { hist += hist_len - 32768u; hist_len = 32768u; }
    const size_t base = ((size_t)hist_len + 1023u) & ~(size_t)1023u;
    // every segment decodes into a region of its own: 16 bytes of room per compressed byte (a segment that needs more ends the
    // parallel part in front of it)
    std::vector<uint64_t> ioff(nseg), ooff(nseg);
    std::vector<uint32_t> ilen(nseg), ocap(nseg), shist(nseg), sbit(nseg);
    uint64_t scratch = 0;
    for (uint32_t j = 0; j < nseg; ++j) {
        ioff[j] = seg_start[j];
        // (a cut inside a byte: the segment in front of it sees that byte too -- its decode must stop AT the cut, for want of input)
        ilen[j] = (j + 1u < nseg ? seg_start[j + 1u] + ((seg_bit && seg_bit[j + 1u]) ? 1u : 0u) : in_len) - seg_start[j];
        uint64_t room = (uint64_t)ilen[j] * 16u + 4096u;
        if (room > out_cap) room = out_cap;
        ocap[j] = (uint32_t)room;
        ooff[j] = scratch;
        scratch += (room + 1023u) & ~1023ull;
        shist[j] = j == 0u ? hist_len : 32768u;   // (taken on trust by the decode pass, which reads no history; checked at the stitch)
        sbit[j] = j == 0u ? in_bit : (seg_bit ? seg_bit[j] : 0u);
    }
    if (scratch > (3ull << 30)) return zmi_inflate_resume(c, in, in_len, in_bit, hist, hist_len, out, out_cap, out_len, status, detail, in_used, resume);
    // meta: ioff u64[n] | ooff u64[n] | soff u64[n + 1] | ilen | ocap | hist | bit | olen | status | used | detail (u32[n] each) | resume u32[4n] | one_off u64, one_len u32, err u32
    const size_t n = nseg;
    const size_t o_ioff = 0, o_ooff = 8 * n, o_soff = 16 * n, o_ilen = 24 * n + 8, o_ocap = o_ilen + 4 * n, o_hist = o_ocap + 4 * n, o_bit = o_hist + 4 * n,
                 o_olen = o_bit + 4 * n, o_st = o_olen + 4 * n, o_used = o_st + 4 * n, o_det = o_used + 4 * n, o_res = o_det + 4 * n, o_one = ((o_res + 16 * n) + 7) & ~(size_t)7,
                 meta_bytes = o_one + 32;
    int rc = zmi_reserve(c->st_in, (size_t)in_len + 64u);
    if (!rc) rc = zmi_reserve(c->st_out, base + (size_t)out_cap + 64u);
    if (!rc) rc = zmi_reserve(c->st_meta, meta_bytes);
    if (!rc) rc = zmi_reserve(c->sp_out, (size_t)scratch + 64u);
    if (rc) return rc;
    hipStream_t hs = c->host_stream;
    uint8_t* d = (uint8_t*)c->st_meta.p;
    std::vector<uint8_t> hm(meta_bytes, 0);
    memcpy(hm.data() + o_ioff, ioff.data(), 8 * n); memcpy(hm.data() + o_ooff, ooff.data(), 8 * n);
    memcpy(hm.data() + o_ilen, ilen.data(), 4 * n); memcpy(hm.data() + o_ocap, ocap.data(), 4 * n);
    memcpy(hm.data() + o_hist, shist.data(), 4 * n); memcpy(hm.data() + o_bit, sbit.data(), 4 * n);
    if (!in_on_device) ZMI_HIPS(hipMemcpyAsync(c->st_in.p, in, in_len, hipMemcpyHostToDevice, hs));
    if (hist_len) ZMI_HIPS(hipMemcpyAsync((uint8_t*)c->st_out.p + base - hist_len, hist, hist_len, hipMemcpyHostToDevice, hs));
    ZMI_HIPS(hipMemcpyAsync(d, hm.data(), meta_bytes, hipMemcpyHostToDevice, hs));
    const uint64_t saved_limit = c->inflate_out_limit;
    c->inflate_out_limit = scratch + (1ull << 16);
    struct restore { zmi_ctx* c; uint64_t v; ~restore() { c->inflate_out_limit = v; } } restore_limit{c, saved_limit};
    rc = zmi_inflate_impl(c, c->st_in.p, (const uint64_t*)(d + o_ioff), (const uint32_t*)(d + o_ilen), nseg, ZMI_WRAP_RAW, c->sp_out.p,
                          (const uint64_t*)(d + o_ooff), (const uint32_t*)(d + o_ocap), (const uint32_t*)(d + o_hist), (uint32_t*)(d + o_olen),
                          (int32_t*)(d + o_st), (uint32_t*)(d + o_used), (int32_t*)(d + o_det), (const uint32_t*)(d + o_bit), (uint32_t*)(d + o_res), hs, true);
    if (rc) { (void)hipStreamSynchronize(hs); return rc; }
    std::vector<uint32_t> r(8 * n), codes(nseg, 0u);   // olen | status | used | detail | resume[4n]
    ZMI_HIPS(hipMemcpyAsync(r.data(), d + o_olen, 32 * n, hipMemcpyDeviceToHost, hs));
    // (zmi_inflate_impl's scratch: bm_off u64[nseg] | used[nseg] | check[nseg] -- every segment's tables' size, see zmi_inflate_resume)
    ZMI_HIPS(hipMemcpyAsync(codes.data(), (const uint8_t*)c->inf_tmp.p + 12u * (size_t)nseg, 4u * (size_t)nseg, hipMemcpyDeviceToHost, hs));
    ZMI_HIPS(hipStreamSynchronize(hs));
    const double sp_t1 = sp_trace ? sp_now() : 0.0;
    uint32_t* olen = r.data();
    const uint32_t *used = r.data() + 2 * n, *res = r.data() + 4 * n;
    const int32_t *st = (const int32_t*)(r.data() + n), *det = (const int32_t*)(r.data() + 3 * n);
    // the chain: segment j is CLEAN if its decode ended exactly at its last byte, on a block boundary, with all output complete
    bool trimmed = false;   // a clean segment produced bytes behind its checkpoint: the device copy of the lengths is corrected
    uint32_t tail = 0;   // the first segment that is not clean (or the last one): its result is "the serial decode of the rest"
    std::vector<uint64_t> soff(n + 1, 0);
    uint64_t total = 0;
    for (uint32_t j = 0; j < nseg; ++j) {
        tail = j;
        soff[j] = total;
        // (a cut inside a byte: the decode stops at that bit for want of input, having seen the next block's first bits at most --
        // whatever it made of them lies behind the checkpoint and is dropped)
        const uint32_t cut_byte = j + 1u < nseg ? seg_start[j + 1u] - seg_start[j] : 0u, cut_bit = (j + 1u < nseg && seg_bit) ? seg_bit[j + 1u] : 0u;
        const bool clean = j + 1u < nseg && st[j] == ZMI_BUF_ERROR && det[j] == 1 && res[4 * j] == cut_byte && res[4 * j + 1] == cut_bit &&
                           (cut_bit != 0u || res[4 * j + 2] == olen[j]) && res[4 * j + 2] <= olen[j] && olen[j] <= ocap[j] &&
                           total + res[4 * j + 2] <= out_cap;
        if (!clean) break;
        if (olen[j] != res[4 * j + 2]) { olen[j] = res[4 * j + 2]; trimmed = true; }
        total += olen[j];
    }
    // the tail counts as it stands if it is the stream's real tail (the last segment) or ended the stream; a tail that stopped
    // for any other reason (a marker that was data, no room) is decoded again serially below, from its first byte
    // (and not one that ran out of a region smaller than the room the caller really has left)
    const bool tail_ok = (tail + 1u == nseg || st[tail] == ZMI_OK) && olen[tail] <= ocap[tail] && total + olen[tail] <= out_cap &&
                         (st[tail] == ZMI_OK || (st[tail] == ZMI_BUF_ERROR && (det[tail] == 1 || (det[tail] == 2 && ocap[tail] >= out_cap - total))));
    const uint32_t take = tail + (tail_ok ? 1u : 0u);   // segments whose output is copied
    if (take == 0u || (take == 1u && !tail_ok))        // nothing gained: the serial path
        return zmi_inflate_resume(c, in, in_len, in_bit, hist, hist_len, out, out_cap, out_len, status, detail, in_used, resume);
    if (tail_ok) total += olen[tail] < ocap[tail] ? olen[tail] : ocap[tail];
    soff[take] = total;
    for (uint32_t j = take; j-- > 0u;)   // inflateCodesUsed: the last dynamic block of the part decoded here (a serial tail below overrides it)
        if (codes[j] != 0u) { c->last_codes_used = codes[j]; break; }
    if (total == 0u)
        return zmi_inflate_resume(c, in, in_len, in_bit, hist, hist_len, out, out_cap, out_len, status, detail, in_used, resume);
    // stitch: outputs back to back behind the history, then all back-references at once
    rc = zmi_reserve(c->inf_ptr, (size_t)total * 4u + 512u);
    if (rc) return rc;
    struct { uint64_t off; uint32_t len, err; } one = {0ull, (uint32_t)total, 0u};
    ZMI_HIPS(hipMemcpyAsync(d + o_soff, soff.data(), 8 * (take + 1u), hipMemcpyHostToDevice, hs));
    ZMI_HIPS(hipMemcpyAsync(d + o_one, &one, sizeof(one), hipMemcpyHostToDevice, hs));
    if (trimmed) ZMI_HIPS(hipMemcpyAsync(d + o_olen, olen, 4 * (size_t)take, hipMemcpyHostToDevice, hs));
    uint8_t* d_fin = (uint8_t*)c->st_out.p + base;
    zmi_launch_copy_ranges((const uint8_t*)c->sp_out.p, (const uint64_t*)(d + o_ooff), 0, (const uint32_t*)(d + o_olen), take, d_fin,
                           (const uint64_t*)(d + o_soff), total, 0xFFFFFFFFu, hs);
    uint32_t rounds = 2u;   // ceil(log3(total)) + 1
    for (uint64_t span = 3u; rounds < 34u && span < total; span *= 3u) ++rounds;
    {
        zmi_scope_timer tm(c, ZMI_K_RESOLVE, hs);
        zmi_launch_resolve_jump_segments(d_fin, (const uint64_t*)(d + o_soff), take, (const uint64_t*)c->inf_bm.p, (const uint64_t*)c->inf_tmp.p,
                                         (int32_t*)c->inf_ptr.p, total, hist_len, rounds, (uint32_t*)((uint8_t*)c->inf_ptr.p + (size_t)total * 4u),
                                         (uint32_t*)(d + o_one + 12), (const uint64_t*)(d + o_one), (const uint32_t*)(d + o_one + 8), hs);
    }
    uint32_t err = 0;
    ZMI_HIPS(hipMemcpyAsync(&err, d + o_one + 12, 4, hipMemcpyDeviceToHost, hs));
    double sp_t2 = 0.0;
    if (sp_trace) { ZMI_HIPS(hipStreamSynchronize(hs)); sp_t2 = sp_now(); }
    { const int crc = zmi_d2h(c, out, d_fin, (size_t)total, hs); if (crc) return crc; }
    ZMI_HIPS(hipStreamSynchronize(hs));
    ZMI_HIPS(hipGetLastError());
    if (sp_trace) fprintf(stderr, "[zmi split] %u segments, %u B in, %llu B out: copy-in + decode %.2f ms, stitch + resolve %.2f ms, copy-out %.2f ms\n", nseg, in_len, (unsigned long long)total, sp_t1 - sp_t0, sp_t2 - sp_t1, sp_now() - sp_t2);
    if (err)   // a distance reaches in front of the history that is really there: let the serial decode find and name it
        return zmi_inflate_resume(c, in, in_len, in_bit, hist, hist_len, out, out_cap, out_len, status, detail, in_used, resume);
    if (segments_used) *segments_used = take;
    if (tail_ok) {
        const uint32_t g = (uint32_t)soff[tail];
        *out_len = g + olen[tail];
        *status = st[tail];
        *detail = det[tail];
        *in_used = seg_start[tail] + used[tail];
        resume[0] = seg_start[tail] + res[4 * tail]; resume[1] = res[4 * tail + 1]; resume[2] = g + res[4 * tail + 2]; resume[3] = res[4 * tail + 3];
        return ZMI_E_OK;
    }
    // the clean segments are in `out`; the rest serially, with what has been produced as its history
    const uint32_t at = seg_start[tail], done = (uint32_t)total;
    std::vector<uint8_t> h2;
    if (done >= 32768u) h2.assign(out + done - 32768u, out + done);
    else {
        const uint32_t keep = hist_len < 32768u - done ? hist_len : 32768u - done;
        h2.assign(hist + hist_len - keep, hist + hist_len);
        h2.insert(h2.end(), out, out + done);
    }
    uint32_t ol2 = 0, used2 = 0, res2[4] = {0, 0, 0, 0};
    int32_t st2 = 0, det2 = 0;
    // (a cut found by the block scan lies at a bit: the rest starts there)
    rc = zmi_inflate_resume(c, in + at, in_len - at, tail == 0u ? in_bit : (seg_bit ? seg_bit[tail] : 0u), h2.data(), (uint32_t)h2.size(), out + done,
                            out_cap - done, &ol2, &st2, &det2, &used2, res2);
    if (rc) return rc;
    *out_len = done + ol2; *status = st2; *detail = det2; *in_used = at + used2;
    resume[0] = at + res2[0]; resume[1] = res2[1]; resume[2] = done + res2[2]; resume[3] = res2[3];
    return ZMI_E_OK;
}

#undef ZMI_HIPS
extern "C" int zmi_inflate_split(zmi_ctx* c, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist,
                                 uint32_t hist_len, uint8_t* out, uint32_t out_cap, const uint32_t* seg_start, uint32_t nseg,
                                 uint32_t* out_len, int32_t* status, int32_t* detail, uint32_t* in_used, uint32_t* resume,
                                 uint32_t* segments_used) {
    try {
        return zmi_split_core(c, in, in_len, in_bit, hist, hist_len, out, out_cap, seg_start, nullptr, nseg, false, out_len, status, detail,
                              in_used, resume, segments_used);
    } catch (...) { return zmi_fail(ZMI_E_NOMEM, "zmi_inflate_split: out of host memory"); }
}

// One raw deflate stream WITHOUT flush points, decoded as parallel segments all the same: the device looks for the stream's
// dynamic block headers at every bit position (blockscan.hip), the stretches between the boundaries found are decoded side by
// side and stitched as zmi_inflate_split does -- with the same discipline: a cut counts only if the decode in front of it
// ended exactly there.  Results equal zmi_inflate_resume on the same arguments; *segments_used = 0 says the serial path ran
// (too few boundaries found, or the first cut did not check out).
extern "C" int zmi_launch_block_scan(const uint8_t* d_in, uint32_t n, uint64_t first_bit, uint32_t* d_pre, uint32_t pre_cap, uint32_t* d_list,
                                     uint32_t cap, uint32_t* d_count, hipStream_t stream);
static int zmi_inflate_blocks_body(zmi_ctx* c, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist, uint32_t hist_len,
                                   uint8_t* out, uint32_t out_cap, uint32_t* out_len, int32_t* status, int32_t* detail, uint32_t* in_used,
                                   uint32_t* resume, uint32_t* segments_used) {
    if (segments_used) *segments_used = 0;
    if (!c || (!in && in_len) || (!hist && hist_len) || (!out && out_cap) || !out_len || !status || !detail || !in_used || !resume)
        return zmi_fail(ZMI_E_ARG, "null argument");
    // (below 256 KiB the scan and the stitch cost more than the serial decode of the few blocks there are)
    uint32_t min_len = 256u << 10, gap = 8192u;
    if (const char* e = zmi_tune("ZMI_BLOCKS_MIN")) { if (atoi(e) > 0) min_len = (uint32_t)atoi(e); }
    if (const char* e = zmi_tune("ZMI_BLOCKS_GAP")) { if (atoi(e) > 0) gap = (uint32_t)atoi(e); }
    if (in_len < min_len || in_bit > 7u || out_cap > (1u << 30) || in_len > (1u << 28))
        return zmi_inflate_resume(c, in, in_len, in_bit, hist, hist_len, out, out_cap, out_len, status, detail, in_used, resume);
    ZMI_ON_DEVICE(c);
    const uint32_t cap = 65536u;
    // stage-1 survivors: one bit position in ~250 of ordinary compressed data; room for one in 64 (more: the scan reports an
    // overflow and the serial decode runs)
    const uint32_t pre_cap = in_len / 8u + 4096u;
    int rc = zmi_reserve(c->st_in, (size_t)in_len + 64u);
    if (!rc) rc = zmi_reserve(c->st_scan, ((size_t)cap + pre_cap) * 4u + 64u);
    if (rc) return rc;
    hipStream_t hs = c->host_stream;
    uint32_t* d_count = (uint32_t*)c->st_scan.p;
    uint32_t* d_list = d_count + 16;
    uint32_t* d_pre = d_list + cap;
    std::vector<uint32_t> found;
    uint32_t count = 0, counts[4] = {0, 0, 0, 0};
    {
        hipError_t e = hipMemcpyAsync(c->st_in.p, in, in_len, hipMemcpyHostToDevice, hs);
        if (e == hipSuccess) e = hipMemsetAsync(d_count, 0, 64, hs);
        if (e != hipSuccess) { (void)hipStreamSynchronize(hs); return zmi_fail(ZMI_E_HIP, "zmi_inflate_blocks: copy-in", e); }
        // (the first block starts at in_bit: it needs no finding; the scan starts behind its header bits)
        zmi_launch_block_scan((const uint8_t*)c->st_in.p, in_len, (uint64_t)in_bit + 3u, d_pre, pre_cap, d_list, cap, d_count, hs);
        e = hipMemcpyAsync(counts, d_count, 16, hipMemcpyDeviceToHost, hs);
        if (e == hipSuccess) e = hipStreamSynchronize(hs);
        if (e != hipSuccess) return zmi_fail(ZMI_E_HIP, "zmi_inflate_blocks: scan", e);
        count = counts[2] ? 0u : counts[0];   // (an incomplete survivor list: nothing is taken from it)
        if (count != 0u && count <= cap) {
            found.resize(count);
            e = hipMemcpyAsync(found.data(), d_list, (size_t)count * 4u, hipMemcpyDeviceToHost, hs);
            if (e == hipSuccess) e = hipStreamSynchronize(hs);
            if (e != hipSuccess) return zmi_fail(ZMI_E_HIP, "zmi_inflate_blocks: list", e);
        }
    }
    std::sort(found.begin(), found.end());
    // cuts: boundaries at least `gap` compressed bytes apart (a segment is one wave's work: it should be worth a wave)
    std::vector<uint32_t> seg_start{0u}, seg_bit{in_bit};
    uint64_t last = 0;
    for (uint32_t pos : found) {
        if ((uint64_t)pos < last + 8ull * gap || (pos >> 3) == 0u) continue;
        seg_start.push_back(pos >> 3);
        seg_bit.push_back(pos & 7u);
        last = pos;
        if (seg_start.size() >= 8192u) break;
    }
    if (zmi_tune("ZMI_SPLIT_TRACE")) fprintf(stderr, "[zmi blocks] %u B: %u of %u stage-1 survivors are block headers, %zu segments\n", in_len, count, counts[3], seg_start.size());
    if (seg_start.size() < 4u)
        return zmi_inflate_resume(c, in, in_len, in_bit, hist, hist_len, out, out_cap, out_len, status, detail, in_used, resume);
    return zmi_split_core(c, in, in_len, in_bit, hist, hist_len, out, out_cap, seg_start.data(), seg_bit.data(), (uint32_t)seg_start.size(), true,
                          out_len, status, detail, in_used, resume, segments_used);
}
extern "C" int zmi_inflate_blocks(zmi_ctx* c, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist, uint32_t hist_len,
                                  uint8_t* out, uint32_t out_cap, uint32_t* out_len, int32_t* status, int32_t* detail, uint32_t* in_used,
                                  uint32_t* resume, uint32_t* segments_used) {
    try {
        return zmi_inflate_blocks_body(c, in, in_len, in_bit, hist, hist_len, out, out_cap, out_len, status, detail, in_used, resume, segments_used);
    } catch (...) { return zmi_fail(ZMI_E_NOMEM, "zmi_inflate_blocks: out of host memory"); }
}
// ---- host buffers, pipelined ----
// A caller that hands over host memory pays PCIe both ways (the reference's own caller loop:
// test-libz-rs-sys/examples/blogpost-compress.rs:43-122 -- every real zlib-rs user lives on this path).  Pageable memory is
// the trap: hipMemcpy of pageable memory is the runtime's single-threaded bounce through a small pinned buffer (measured
// round 2: 12-14 GiB/s end to end, with the kernels idle half of the time).  So the batch is cut into chunks that cycle
// through three slots, and per chunk
//   stage in   a few host threads copy the caller's shards into the slot's PINNED staging buffer (memcpy in parallel),
//   H2D        one DMA copy of the whole chunk (stream hs_in),
//   kernels    deflate + zmi_pack_slab_dev: the chunk's compressed streams as one dense slab (stream hs_k),
//   D2H        the slab -- not the compress_bound-strided slots: 0.44 bytes per input byte instead of 1.125 -- into the
//              slot's pinned output staging (stream hs_out),
//   stage out  the host threads scatter the streams to the caller's out + i * out_stride.
// Per chunk the host thread stages chunk k in, hands chunk k-2 out (the slot chunk k is about to use), issues chunk k's
// copies and kernels and chunk k-1's slab copy: the device works on chunks k-1 and k while the host copies.
#include <chrono>
#include <thread>
#include <atomic>
static unsigned zmi_host_threads() {
    unsigned t = std::thread::hardware_concurrency() / 2u;
    t = t < 1u ? 1u : (t > 8u ? 8u : t);
    if (const char* e = zmi_tune("ZMI_HOST_THREADS")) { if (atoi(e) > 0) t = (unsigned)atoi(e) > 64u ? 64u : (unsigned)atoi(e); }
    return t;
}
struct zmi_copy_job { void* dst; const void* src; size_t n; };
// the jobs' bytes split evenly over T threads (the calling thread is one of them)
static void zmi_parallel_copy(const std::vector<zmi_copy_job>& jobs, unsigned T) {
    size_t total = 0;
    for (const zmi_copy_job& j : jobs) total += j.n;
    if (total == 0) return;
    if (T <= 1u || total < ((size_t)4 << 20)) { for (const zmi_copy_job& j : jobs) if (j.n) memcpy(j.dst, j.src, j.n); return; }
    const size_t per = (total + T - 1u) / T;
    auto work = [&](unsigned t) {
        size_t lo = (size_t)t * per, hi = lo + per < total ? lo + per : total, at = 0;
        for (const zmi_copy_job& j : jobs) {
            const size_t a = at, b = at + j.n;
            at = b;
            if (b <= lo || a >= hi) continue;
            const size_t f = a > lo ? a : lo, l = b < hi ? b : hi;
            memcpy((uint8_t*)j.dst + (f - a), (const uint8_t*)j.src + (f - a), l - f);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (std::thread& x : th) x.join();
}
// gives back the cached staging of the host-buffer pipelines (pinned host memory and the device slots); the streams and events stay
static void zmi_hb_release(zmi_ctx* c) {
    for (int k = 0; k < ZMI_HB_SLOTS; ++k) {
        if (c->hb_pin_in[k].p) { (void)hipHostFree(c->hb_pin_in[k].p); c->hb_pin_in[k] = zmi_buf(); }
        if (c->hb_pin_out[k].p) { (void)hipHostFree(c->hb_pin_out[k].p); c->hb_pin_out[k] = zmi_buf(); }
        if (c->hb_pin_meta[k].p) { (void)hipHostFree(c->hb_pin_meta[k].p); c->hb_pin_meta[k] = zmi_buf(); }
        if (c->hb_slab[k].p) { (void)hipFree(c->hb_slab[k].p); c->hb_slab[k] = zmi_buf(); }
        zmi_ctx::hb_slot& sl = c->hb[k];
        if (sl.in.p) { (void)hipFree(sl.in.p); sl.in = zmi_buf(); }
        if (sl.out.p) { (void)hipFree(sl.out.p); sl.out = zmi_buf(); }
        if (sl.meta.p) { (void)hipFree(sl.meta.p); sl.meta = zmi_buf(); }
    }
}
static size_t zmi_hb_pinned_bytes(const zmi_ctx* c) {
    size_t t = 0;
    for (int k = 0; k < ZMI_HB_SLOTS; ++k) t += c->hb_pin_in[k].cap + c->hb_pin_out[k].cap + c->hb_pin_meta[k].cap;
    return t;
}
// zmi_ctx_trim: a long-lived context that once ran a very large host-buffer batch keeps ~6 GiB of pinned host memory and as much
// HBM for the next one; this hands them back (also done automatically when a call leaves more pinned than the context's limit)
extern "C" int zmi_ctx_trim(zmi_ctx* c) {
    if (!c) return zmi_fail(ZMI_E_ARG, "null context");
    ZMI_ON_DEVICE(c);
    zmi_hb_release(c);
    return ZMI_E_OK;
}
extern "C" int zmi_ctx_set_pinned_limit(zmi_ctx* c, uint64_t bytes) {
    if (!c) return zmi_fail(ZMI_E_ARG, "null context");
    c->pinned_limit = bytes < (64ull << 20) ? (64ull << 20) : bytes;
    return ZMI_E_OK;
}
// joins the follower thread of a host-buffer pipeline on every way out of its scope -- also when stage_in / the std::thread
// constructor / a vector throws (std::terminate on a joinable thread, and an exception through a C entry point, otherwise)
struct zmi_joiner {
    std::thread& t;
    std::atomic<int>& failed;
    ~zmi_joiner() {
        if (t.joinable()) { failed.store(ZMI_E_NOMEM, std::memory_order_release); t.join(); }
    }
};
static int zmi_d2h(zmi_ctx* c, void* dst, const void* src, size_t n, hipStream_t hs) {
    const size_t CH = (size_t)4 << 20;
    bool bounce = n >= ((size_t)1 << 20);
    if (bounce && (zmi_reserve_pinned(c->st_pin[0], CH) != 0 || zmi_reserve_pinned(c->st_pin[1], CH) != 0)) bounce = false;
    if (bounce && !c->st_pin_ev_live) {
        if (hipEventCreateWithFlags(&c->st_pin_ev[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->st_pin_ev[1], hipEventDisableTiming) != hipSuccess) bounce = false;
        else c->st_pin_ev_live = true;
    }
    // every error return drains `hs` first: the callers have asynchronous copies into their own frames in flight on it (the
    // `err` word of zmi_split_core), and the DMA into st_pin must not outlive the call either
    if (!bounce) {
        hipError_t e = hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, hs);
        const hipError_t es = hipStreamSynchronize(hs);
        if (e == hipSuccess) e = es;
        if (e != hipSuccess) return zmi_fail(ZMI_E_HIP, "zmi_d2h", e);
        return 0;
    }
    const unsigned T = zmi_host_threads() < 4u ? zmi_host_threads() : 4u;
    const size_t K = (n + CH - 1) / CH;
    for (size_t k = 0; k <= K; ++k) {
        if (k < K) {
            const size_t len = n - k * CH < CH ? n - k * CH : CH;
            hipError_t e = hipMemcpyAsync(c->st_pin[k & 1].p, (const uint8_t*)src + k * CH, len, hipMemcpyDeviceToHost, hs);
            if (e == hipSuccess) e = hipEventRecord(c->st_pin_ev[k & 1], hs);
            if (e != hipSuccess) { (void)hipStreamSynchronize(hs); return zmi_fail(ZMI_E_HIP, "zmi_d2h", e); }
        }
        if (k >= 1) {   // the piece before: wait for its DMA, copy it out (the DMA of piece k runs meanwhile)
            const size_t j = k - 1, len = n - j * CH < CH ? n - j * CH : CH;
            const hipError_t ew = hipEventSynchronize(c->st_pin_ev[j & 1]);
            if (ew != hipSuccess) { (void)hipStreamSynchronize(hs); return zmi_fail(ZMI_E_HIP, "zmi_d2h: hipEventSynchronize", ew); }
            std::vector<zmi_copy_job> jobs{{(uint8_t*)dst + j * CH, c->st_pin[j & 1].p, len}};
            zmi_parallel_copy(jobs, T);
        }
    }
    return 0;
}
static int zmi_reserve_pinned(zmi_buf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) { (void)hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
    const size_t want = (bytes + 0xFFFFFull) & ~(size_t)0xFFFFFull;
    hipError_t e = hipHostMalloc(&b.p, want, hipHostMallocDefault);
    if (e != hipSuccess) { b.p = nullptr; return zmi_fail(ZMI_E_NOMEM, "hipHostMalloc(staging)", e); }
    b.cap = want;
    return 0;
}

static int zmi_deflate_batch_body(zmi_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                  int level, int strategy, int wrap, uint8_t* out, uint64_t out_stride, uint32_t* out_len,
                                  int32_t* status);
// (a C entry point: nothing may unwind through it -- the reference builds with panic = abort for the same reason,
// libz-rs-sys-cdylib/src/lib.rs:5-6)
extern "C" int zmi_deflate_batch(zmi_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                 int level, int strategy, int wrap, uint8_t* out, uint64_t out_stride, uint32_t* out_len,
                                 int32_t* status) {
    try {
        return zmi_deflate_batch_body(c, in, in_off, in_len, n, level, strategy, wrap, out, out_stride, out_len, status);
    } catch (...) {
        return zmi_fail(ZMI_E_NOMEM, "zmi_deflate_batch: out of host memory (or a thread could not be started)");
    }
}
static int zmi_deflate_batch_body(zmi_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                  int level, int strategy, int wrap, uint8_t* out, uint64_t out_stride, uint32_t* out_len,
                                  int32_t* status) {
    if (!c || (!in && n) || !in_off || !in_len || !out || !out_len || !status) return zmi_fail(ZMI_E_ARG, "null argument");
    if (n == 0) return ZMI_E_OK;
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (in_len[i] > max_len) max_len = in_len[i];
    if (out_stride % 16u || out_stride < zmi_deflate_bound(max_len, wrap))
        return zmi_fail(ZMI_E_ARG, "out_stride must be a multiple of 16 and >= zmi_deflate_bound(max_len)");
    // A chunk should fill the chip by itself: one workgroup per shard on 256 CUs wants a few hundred shards per launch (measured:
    // 128-shard chunks made the kernels, not PCIe, the bottleneck; 512 shards of 1 MiB run within 10 % of the per-shard time of
    // the largest launches).  ZMI_HOST_CHUNK (bytes) overrides both bounds (tests).
    // Large batches take larger chunks (an eighth of the input, up to 1 GiB): launches of 512 shards run at 40 GiB/s (two rounds of
    // workgroups on 256 CUs, the slower data classes set the pace), launches of 1024+ at 55-60, above the 46 GiB/s PCIe delivers.
    uint64_t total_in = 0;
    for (uint32_t i = 0; i < n; ++i) total_in += in_len[i];
    uint64_t budget = total_in / 8u;
    budget = budget < (512ull << 20) ? (512ull << 20) : (budget > (1024ull << 20) ? (1024ull << 20) : budget);
    // three slots of pinned staging, each a chunk of input and its worst-case slab (~1.13x): 6.4 chunks in all
    if (budget > c->pinned_limit / 7u) budget = c->pinned_limit / 7u;
    uint32_t min_count = 384u;
    if (const char* e = zmi_tune("ZMI_HOST_CHUNK")) { if (atoll(e) > 0) { budget = (uint64_t)atoll(e); min_count = 1u; } }
    struct chunk { uint32_t first, count; uint64_t bytes; std::vector<uint64_t> doff; };
    std::vector<chunk> chunks;
    for (uint32_t i = 0; i < n;) {
        chunk ck{i, 0, 0, {}};
        while (i < n) {
            const uint64_t a = ((uint64_t)in_len[i] + 15u) & ~15ull;
            if (ck.count >= min_count && ck.bytes + a > budget) break;
            ck.doff.push_back(ck.bytes);
            ck.bytes += a;
            ++ck.count;
            ++i;
        }
        chunks.push_back(std::move(ck));
    }
    const char* pl = zmi_tune("ZMI_HOST_PIPELINE");   // 0: the plain copy-in / kernels / copy-out sequence
    if (chunks.size() < 2 || (pl && !atoi(pl)))
        return zmi_deflate_batch_simple(c, in, in_off, in_len, n, level, strategy, wrap, out, out_stride, out_len, status);
    ZMI_ON_DEVICE(c);
    {
        int irc = zmi_host_pipeline_init(c);
        if (irc) return irc;
    }
    uint64_t max_bytes = 0;
    uint32_t max_count = 0;
    for (const chunk& ck : chunks) { if (ck.bytes > max_bytes) max_bytes = ck.bytes; if (ck.count > max_count) max_count = ck.count; }
    // slab bound: every stream at most out_stride, but never more than the reference's bound for its own length
    uint64_t max_slab = 0;
    for (const chunk& ck : chunks) {
        uint64_t sb = 0;
        for (uint32_t i = 0; i < ck.count; ++i) sb += zmi_deflate_bound(in_len[ck.first + i], wrap) + 16u;
        if (sb > max_slab) max_slab = sb;
    }
    // meta (device and pinned twin): doff u64[count] | in_len u32[count] | out_len u32[count] | status i32[count] | soff u64[count + 1]
    const size_t meta_bytes = (size_t)max_count * 28u + 64u;
    bool zero_copy = true;   // the slab is written into pinned host memory by the pack kernel (0: packed on the device, copied by a copy engine)
    if (const char* zv = zmi_tune("ZMI_HB_ZEROCOPY")) zero_copy = atoi(zv) != 0;
    uint32_t pack_groups = 16u;   // workgroups of the pack kernel that writes to host memory (zmi_launch_copy_ranges_few)
    if (const char* gv = zmi_tune("ZMI_HB_PACK_GROUPS")) { if (atoi(gv) > 0) pack_groups = (uint32_t)atoi(gv); }
    for (int k = 0; k < ZMI_HB_SLOTS; ++k) {
        zmi_ctx::hb_slot& sl = c->hb[k];
        int rc = zmi_reserve(sl.in, (size_t)max_bytes + 64u);
        if (!rc) rc = zmi_reserve(sl.out, (size_t)max_count * out_stride + 64u);
        if (!rc) rc = zmi_reserve(sl.meta, meta_bytes);
        if (!rc && !zero_copy) rc = zmi_reserve(c->hb_slab[k], (size_t)max_slab + 64u);
        if (!rc) rc = zmi_reserve_pinned(c->hb_pin_in[k], (size_t)max_bytes + 64u);
        if (!rc) rc = zmi_reserve_pinned(c->hb_pin_out[k], (size_t)max_slab + 64u);
        if (!rc) rc = zmi_reserve_pinned(c->hb_pin_meta[k], meta_bytes);
        if (rc) {
            // no room for the staging (memlock / container limit, HBM): the plain copy-in / kernels / copy-out sequence needs no
            // pinned memory and a batch that worked before the pipeline existed keeps working
            zmi_hb_release(c);
            return zmi_deflate_batch_simple(c, in, in_off, in_len, n, level, strategy, wrap, out, out_stride, out_len, status);
        }
    }
    const unsigned T = zmi_host_threads();
    const size_t K = chunks.size();
    auto m_doff = [&](void* base) { return (uint64_t*)base; };
    auto m_len = [&](void* base) { return (uint32_t*)((uint8_t*)base + (size_t)max_count * 8u); };
    auto m_olen = [&](void* base) { return (uint32_t*)((uint8_t*)base + (size_t)max_count * 12u); };
    auto m_st = [&](void* base) { return (int32_t*)((uint8_t*)base + (size_t)max_count * 16u); };
    auto m_soff = [&](void* base) { return (uint64_t*)((uint8_t*)base + (size_t)max_count * 20u + 8u - ((size_t)max_count * 20u) % 8u); };
    std::vector<hipEvent_t> sizes_done(ZMI_HB_SLOTS), slab_done(ZMI_HB_SLOTS);
    for (int k = 0; k < ZMI_HB_SLOTS; ++k) {
        ZMI_HIP(hipEventCreateWithFlags(&sizes_done[k], hipEventDisableTiming));
        ZMI_HIP(hipEventCreateWithFlags(&slab_done[k], hipEventDisableTiming));
    }
    struct ev_guard { std::vector<hipEvent_t>& a; std::vector<hipEvent_t>& b; ~ev_guard() { for (hipEvent_t e : a) (void)hipEventDestroy(e); for (hipEvent_t e : b) (void)hipEventDestroy(e); } } evg{sizes_done, slab_done};
    auto stage_in = [&](size_t k) -> int {   // host threads: caller memory -> pinned staging (the slot's previous H2D has finished)
        chunk& ck = chunks[k];
        const int sl = (int)(k % ZMI_HB_SLOTS);
        if (k >= ZMI_HB_SLOTS) ZMI_HIP(hipEventSynchronize(c->hb[sl].in_done));
        std::vector<zmi_copy_job> jobs;
        jobs.reserve(ck.count);
        uint8_t* pin = (uint8_t*)c->hb_pin_in[sl].p;
        for (uint32_t i = 0; i < ck.count; ++i) jobs.push_back({pin + ck.doff[i], in + in_off[ck.first + i], in_len[ck.first + i]});
        zmi_parallel_copy(jobs, T);
        memcpy(m_doff(c->hb_pin_meta[sl].p), ck.doff.data(), (size_t)ck.count * 8u);
        memcpy(m_len(c->hb_pin_meta[sl].p), in_len + ck.first, (size_t)ck.count * 4u);
        return 0;
    };
    auto issue_dev = [&](size_t k) -> int {   // H2D, kernels, pack, sizes back
        chunk& ck = chunks[k];
        const int s2 = (int)(k % ZMI_HB_SLOTS);
        zmi_ctx::hb_slot& sl = c->hb[s2];
        if (k >= ZMI_HB_SLOTS) ZMI_HIP(hipStreamWaitEvent(c->hs_in, sl.k_done, 0));   // the kernels of the chunk that had this slot have read it
        if (k >= ZMI_HB_SLOTS && zero_copy) ZMI_HIP(hipStreamWaitEvent(c->hs_in, slab_done[s2], 0));   // ... and its pack has read the sizes
        ZMI_HIP(hipMemcpyAsync(sl.in.p, c->hb_pin_in[s2].p, (size_t)ck.bytes, hipMemcpyHostToDevice, c->hs_in));
        ZMI_HIP(hipMemcpyAsync(sl.meta.p, c->hb_pin_meta[s2].p, (size_t)max_count * 12u, hipMemcpyHostToDevice, c->hs_in));
        ZMI_HIP(hipEventRecord(sl.in_done, c->hs_in));
        ZMI_HIP(hipStreamWaitEvent(c->hs_k, sl.in_done, 0));
        if (k >= ZMI_HB_SLOTS) ZMI_HIP(hipStreamWaitEvent(c->hs_k, slab_done[s2], 0));   // ... and its slab has left the device
        int rc = zmi_deflate_batch_dev(c, sl.in.p, m_doff(sl.meta.p), m_len(sl.meta.p), ck.count, max_len, level, strategy, wrap, sl.out.p,
                                       out_stride, m_olen(sl.meta.p), m_st(sl.meta.p), c->hs_k);
        if (zero_copy) {
            // The slab does not travel by a copy engine: the pack kernel writes it straight into the pinned host buffer (PCIe
            // writes from the shader, coalesced dwords), on a stream of its own behind the chunk's kernels -- the next chunk's
            // kernels do not wait for it, and the copy engines carry input only.  (Measured: H2D and D2H copies of different
            // streams were served one after the other, a slab queued behind 2.5 GiB of input reached the host 50 ms late.)
            if (rc) return rc;
            ZMI_HIP(hipEventRecord(sl.k_done, c->hs_k));
            ZMI_HIP(hipStreamWaitEvent(c->hs_slab, sl.k_done, 0));
            rc = zmi_scan_sizes_dev(c, m_olen(sl.meta.p), ck.count, m_soff(sl.meta.p), c->hs_slab);
            if (rc) return rc;
            {
                zmi_scope_timer tm(c, ZMI_K_PACK, c->hs_slab);
                zmi_launch_copy_ranges_few((const uint8_t*)sl.out.p, nullptr, out_stride, m_olen(sl.meta.p), ck.count, (uint8_t*)c->hb_pin_out[s2].p,
                                           m_soff(sl.meta.p), c->hb_pin_out[s2].cap, out_stride > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)out_stride,
                                           pack_groups, c->hs_slab);
            }
            ZMI_HIP(hipMemcpyAsync((uint8_t*)c->hb_pin_meta[s2].p + (size_t)max_count * 12u, (uint8_t*)sl.meta.p + (size_t)max_count * 12u,
                                   meta_bytes - (size_t)max_count * 12u, hipMemcpyDeviceToHost, c->hs_slab));
            ZMI_HIP(hipEventRecord(slab_done[s2], c->hs_slab));
            return 0;
        }
        if (!rc) rc = zmi_pack_slab_dev(c, sl.out.p, out_stride, m_olen(sl.meta.p), ck.count, c->hb_slab[s2].p, c->hb_slab[s2].cap,
                                        m_soff(sl.meta.p), c->hs_k);
        if (rc) return rc;
        ZMI_HIP(hipEventRecord(sl.k_done, c->hs_k));
        ZMI_HIP(hipStreamWaitEvent(c->hs_out, sl.k_done, 0));
        // out_len | status | slab offsets travel first: the host needs the slab's size to copy exactly that
        ZMI_HIP(hipMemcpyAsync((uint8_t*)c->hb_pin_meta[s2].p + (size_t)max_count * 12u, (uint8_t*)sl.meta.p + (size_t)max_count * 12u,
                               meta_bytes - (size_t)max_count * 12u, hipMemcpyDeviceToHost, c->hs_out));
        ZMI_HIP(hipEventRecord(sizes_done[s2], c->hs_out));
        return 0;
    };
    auto issue_slab = [&](size_t k) -> int {   // D2H of exactly the slab
        chunk& ck = chunks[k];
        const int s2 = (int)(k % ZMI_HB_SLOTS);
        if (zero_copy) {
            ZMI_HIP(hipEventSynchronize(slab_done[s2]));
            if (m_soff(c->hb_pin_meta[s2].p)[ck.count] > c->hb_pin_out[s2].cap) return zmi_fail(ZMI_E_HIP, "slab larger than its bound");
            return 0;
        }
        ZMI_HIP(hipEventSynchronize(sizes_done[s2]));
        const uint64_t total = m_soff(c->hb_pin_meta[s2].p)[ck.count];
        if (total > c->hb_slab[s2].cap) return zmi_fail(ZMI_E_HIP, "slab larger than its bound");
        // (a stream of its own: hs_out already holds "wait for the kernels of the NEXT chunks" -- the caller's thread runs ahead
        // -- and a slab queued behind those waits left the device only when they had run: 25-29 ms instead of 4)
        if (total) ZMI_HIP(hipMemcpyAsync(c->hb_pin_out[s2].p, c->hb_slab[s2].p, (size_t)total, hipMemcpyDeviceToHost, c->hs_slab));
        ZMI_HIP(hipEventRecord(slab_done[s2], c->hs_slab));
        return 0;
    };
    auto stage_out = [&](size_t k) -> int {   // host threads: pinned slab -> the caller's slots
        chunk& ck = chunks[k];
        const int s2 = (int)(k % ZMI_HB_SLOTS);
        ZMI_HIP(hipEventSynchronize(slab_done[s2]));
        const uint8_t* pm = (const uint8_t*)c->hb_pin_meta[s2].p;
        const uint32_t* ol = m_olen((void*)pm);
        const uint64_t* so = m_soff((void*)pm);
        memcpy(out_len + ck.first, ol, (size_t)ck.count * 4u);
        memcpy(status + ck.first, m_st((void*)pm), (size_t)ck.count * 4u);
        std::vector<zmi_copy_job> jobs;
        jobs.reserve(ck.count);
        const uint8_t* pin = (const uint8_t*)c->hb_pin_out[s2].p;
        for (uint32_t i = 0; i < ck.count; ++i)
            if (m_st((void*)pm)[i] == 0 && ol[i] <= out_stride) jobs.push_back({out + (uint64_t)(ck.first + i) * out_stride, pin + so[i], ol[i]});
        zmi_parallel_copy(jobs, T);
        return 0;
    };
    // Two host threads.  The caller's thread stages chunks in and issues their device work as soon as their slot is free;
    // a second thread follows the results: it waits for a chunk's sizes, issues the copy of exactly its slab, and scatters
    // it.  (One thread doing both waited for results between two issues, and the device idled a third of the time: 18 GiB/s.)
    std::atomic<int> issued{0}, finished{0}, failed{0};
    const int dev_id = c->device;
    const bool hb_trace = zmi_tune("ZMI_HB_TRACE") != nullptr;
    double t_in = 0, t_issue = 0, t_wait_slot = 0, t_slab = 0, t_out = 0, t_wait_issue = 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    std::vector<double> tl(K * 6u, 0.0);   // per chunk: stage_in start, end, sizes known, slab arrived, scattered
    const unsigned T_in = T > 1u ? T - T / 3u : 1u, T_out = T > 2u ? T / 3u : 1u;
    auto consumer = [&]() {
        (void)hipSetDevice(dev_id);
        for (size_t k = 0; k < K; ++k) {
            const double a0 = now();
            while (issued.load(std::memory_order_acquire) <= (int)k) {
                if (failed.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            const double a1 = now();
            int r = issue_slab(k);
            const double a2 = now();
            if (!r && hb_trace) { (void)hipEventSynchronize(slab_done[k % ZMI_HB_SLOTS]); tl[k * 6u + 3u] = now() - t_begin; }
            if (!r) r = stage_out(k);
            tl[k * 6u + 2u] = a2 - t_begin; tl[k * 6u + 4u] = now() - t_begin;
            t_wait_issue += a1 - a0; t_slab += a2 - a1; t_out += now() - a2;
            if (r) { failed.store(r, std::memory_order_release); return; }
            finished.store((int)k + 1, std::memory_order_release);
        }
    };
    (void)T_in; (void)T_out;
    std::thread follower;
    zmi_joiner join_guard{follower, failed};
    follower = std::thread(consumer);
    int rc = 0;
    for (size_t k = 0; k < K && !rc; ++k) {
        while (finished.load(std::memory_order_acquire) + ZMI_HB_SLOTS <= (int)k && !failed.load(std::memory_order_acquire)) std::this_thread::yield();   // slot free
        if (failed.load(std::memory_order_acquire)) break;
        const double b1 = now();
        rc = stage_in(k);
        const double b2 = now();
        if (!rc) rc = issue_dev(k);
        t_wait_slot += 0.0; t_in += b2 - b1; t_issue += now() - b2;
        tl[k * 6u] = b1 - t_begin; tl[k * 6u + 1u] = b2 - t_begin;
        if (!rc) issued.store((int)k + 1, std::memory_order_release);
    }
    const double t_issued_all = now();
    if (rc) failed.store(rc, std::memory_order_release);
    follower.join();
    if (!rc) rc = failed.load();
    // nothing of this call may still be in flight when it returns
    (void)hipStreamSynchronize(c->hs_in);
    (void)hipStreamSynchronize(c->hs_k);
    (void)hipStreamSynchronize(c->hs_out);
    (void)hipStreamSynchronize(c->hs_slab);
    if (hb_trace)
        fprintf(stderr, "[zmi hb] %zu chunks, %u threads: caller stage_in %.1f ms, issue %.1f ms, all issued at %.1f ms; follower waits for issue %.1f, "
                "slab wait+copy-issue %.1f, stage_out %.1f ms; total %.1f ms\n", K, T, t_in, t_issue, t_issued_all - t_begin, t_wait_issue, t_slab, t_out,
                now() - t_begin);
    if (hb_trace)
        for (size_t k = 0; k < K; ++k)
            fprintf(stderr, "[zmi hb]   chunk %zu (%u shards): stage_in %.1f..%.1f, sizes known %.1f, slab on host %.1f, scattered %.1f\n", k, chunks[k].count,
                    tl[k * 6u], tl[k * 6u + 1u], tl[k * 6u + 2u], tl[k * 6u + 3u], tl[k * 6u + 4u]);
    if (zmi_hb_pinned_bytes(c) > c->pinned_limit) zmi_hb_release(c);
    if (rc) return rc;
    ZMI_HIP(hipGetLastError());
    return ZMI_E_OK;
}

static int zmi_inflate_batch_simple(zmi_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                 int wrap, uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len,
                                 int32_t* status) {
    if (!c || (!in && n) || !in_off || !in_len || !out_off || !out_cap || !out_len || !status)
        return zmi_fail(ZMI_E_ARG, "null argument");
    if (n == 0) return ZMI_E_OK;
    ZMI_ON_DEVICE(c);
    std::vector<uint64_t> dioff(n), dooff(n);
    uint64_t tin = 0, tout = 0;
    for (uint32_t i = 0; i < n; ++i) {
        dioff[i] = tin;
        tin += ((uint64_t)in_len[i] + 15u) & ~15ull;
        dooff[i] = tout;
        tout += ((uint64_t)out_cap[i] + 15u) & ~15ull;
    }
    const uint64_t saved_limit = c->inflate_out_limit;
    c->inflate_out_limit = tout + (1ull << 20);   // the capacities are known here: size the bitmap scratch exactly
    c->inf_limit_exact = true;
    struct restore { zmi_ctx* c; uint64_t v; ~restore() { c->inflate_out_limit = v; c->inf_limit_exact = false; } } restore_limit{c, saved_limit};
    zmi_dev_alloc A;
    uint8_t* d_in = (uint8_t*)A.get(tin + 16);
    uint8_t* d_out = (uint8_t*)A.get(tout + 16);
    uint64_t* d_ioff = (uint64_t*)A.get((size_t)n * 8);
    uint64_t* d_ooff = (uint64_t*)A.get((size_t)n * 8);
    uint32_t* d_ilen = (uint32_t*)A.get((size_t)n * 4);
    uint32_t* d_ocap = (uint32_t*)A.get((size_t)n * 4);
    uint32_t* d_olen = (uint32_t*)A.get((size_t)n * 4);
    int32_t* d_st = (int32_t*)A.get((size_t)n * 4);
    if (!d_in || !d_out || !d_ioff || !d_ooff || !d_ilen || !d_ocap || !d_olen || !d_st)
        return zmi_fail(ZMI_E_NOMEM, "hipMalloc(batch buffers)");
    for (uint32_t i = 0; i < n; ++i)
        if (in_len[i]) ZMI_HIP(hipMemcpy(d_in + dioff[i], in + in_off[i], in_len[i], hipMemcpyHostToDevice));
    ZMI_HIP(hipMemcpy(d_ioff, dioff.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    ZMI_HIP(hipMemcpy(d_ooff, dooff.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    ZMI_HIP(hipMemcpy(d_ilen, in_len, (size_t)n * 4, hipMemcpyHostToDevice));
    ZMI_HIP(hipMemcpy(d_ocap, out_cap, (size_t)n * 4, hipMemcpyHostToDevice));
    int rc = zmi_inflate_batch_dev(c, d_in, d_ioff, d_ilen, n, wrap, d_out, d_ooff, d_ocap, d_olen, d_st, nullptr);
    if (rc) return rc;
    ZMI_HIP(hipDeviceSynchronize());
    ZMI_HIP(hipMemcpy(out_len, d_olen, (size_t)n * 4, hipMemcpyDeviceToHost));
    ZMI_HIP(hipMemcpy(status, d_st, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n;) {   // outputs that lie back to back on both sides travel in one copy
        if (out_len[i] == 0 || out_len[i] > out_cap[i]) { ++i; continue; }
        uint32_t j = i;
        uint64_t bytes = out_len[i];
        while (j + 1 < n && out_len[j + 1] && out_len[j + 1] <= out_cap[j + 1] && out_off[j + 1] == out_off[j] + out_len[j] &&
               dooff[j + 1] == dooff[j] + out_len[j] && bytes < (1ull << 30))
            bytes += out_len[++j];
        ZMI_HIP(hipMemcpy(out + out_off[i], d_out + dooff[i], bytes, hipMemcpyDeviceToHost));
        i = j + 1;
    }
    return ZMI_E_OK;
}

// Host buffers, pipelined like zmi_deflate_batch: chunks of consecutive streams (by output capacity) cycle through three slots.
// The caller's thread copies a chunk's compressed streams into pinned staging and issues its H2D copy and kernels; the
// decoded bytes leave the device through a copy kernel of a few workgroups that writes straight into pinned host memory
// (PCIe writes from the shader run beside the copy engine's H2D traffic; two copy-engine directions were served one after
// the other, see zmi_deflate_batch), and a second thread scatters every finished chunk into the caller's regions --
// min(out_len, out_cap) bytes of every stream.
static int zmi_inflate_batch_body(zmi_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                  int wrap, uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len,
                                  int32_t* status);
extern "C" int zmi_inflate_batch(zmi_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                 int wrap, uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len,
                                 int32_t* status) {
    try {
        return zmi_inflate_batch_body(c, in, in_off, in_len, n, wrap, out, out_off, out_cap, out_len, status);
    } catch (...) {
        return zmi_fail(ZMI_E_NOMEM, "zmi_inflate_batch: out of host memory (or a thread could not be started)");
    }
}
static int zmi_inflate_batch_body(zmi_ctx* c, const uint8_t* in, const uint64_t* in_off, const uint32_t* in_len, uint32_t n,
                                  int wrap, uint8_t* out, const uint64_t* out_off, const uint32_t* out_cap, uint32_t* out_len,
                                  int32_t* status) {
    if (!c || (!in && n) || !in_off || !in_len || !out_off || !out_cap || !out_len || !status)
        return zmi_fail(ZMI_E_ARG, "null argument");
    if (n == 0) return ZMI_E_OK;
    // one wave per stream: a launch wants a few thousand streams to keep the chip busy (2048 streams of 1 MiB decode at ~80
    // GiB/s, above what PCIe carries), and a chunk's output has to fit the pinned staging: 2 GiB of capacity per chunk.
    // ZMI_HOST_CHUNK (bytes) overrides both bounds (tests).
    uint64_t budget = 2ull << 30;
    if (budget > c->pinned_limit / 5u) budget = c->pinned_limit / 5u;   // three slots of output staging + their (smaller) input: 4.4 chunks
                                                                         // (the default limit leaves the 2 GiB chunks as they are: a call that
                                                                         // ends above the limit releases the staging, and re-pinning 9 GiB
                                                                         // costs seconds -- measured: 19 -> 3 GiB/s with a limit of 8 GiB)
    uint32_t min_count = 2048u;
    if (const char* e = zmi_tune("ZMI_HOST_CHUNK")) { if (atoll(e) > 0) { budget = (uint64_t)atoll(e); min_count = 1u; } }
    struct chunk { uint32_t first, count; uint64_t in_bytes, out_bytes; std::vector<uint64_t> ioff, ooff; uint32_t max_cap; };
    std::vector<chunk> chunks;
    for (uint32_t i = 0; i < n;) {
        chunk ck{i, 0, 0, 0, {}, {}, 0u};
        while (i < n) {
            const uint64_t ai = ((uint64_t)in_len[i] + 15u) & ~15ull, ao = ((uint64_t)out_cap[i] + 15u) & ~15ull;
            if (ck.count != 0u && (ck.out_bytes + ao > 0xFFFFF000ull || ck.in_bytes + ai > 0xFFFFF000ull)) break;   // (one copy range, u32 length)
            if (ck.count >= min_count && ck.out_bytes + ao > budget) break;
            ck.ioff.push_back(ck.in_bytes);
            ck.ooff.push_back(ck.out_bytes);
            ck.in_bytes += ai;
            ck.out_bytes += ao;
            if (out_cap[i] > ck.max_cap) ck.max_cap = out_cap[i];
            ++ck.count;
            ++i;
        }
        chunks.push_back(std::move(ck));
    }
    const char* pl = zmi_tune("ZMI_HOST_PIPELINE");
    bool fits = true;   // (a chunk leaves the device as one copy range with a 32-bit length)
    for (const chunk& ck : chunks) fits = fits && ck.out_bytes <= 0xFFFFF000ull && ck.in_bytes <= 0xFFFFF000ull;
    if (chunks.size() < 2 || !fits || (pl && !atoi(pl)))
        return zmi_inflate_batch_simple(c, in, in_off, in_len, n, wrap, out, out_off, out_cap, out_len, status);
    ZMI_ON_DEVICE(c);
    int rc = zmi_host_pipeline_init(c);
    if (rc) return rc;
    uint64_t max_in = 0, max_out = 0;
    uint32_t max_count = 0;
    for (const chunk& ck : chunks) {
        if (ck.in_bytes > max_in) max_in = ck.in_bytes;
        if (ck.out_bytes > max_out) max_out = ck.out_bytes;
        if (ck.count > max_count) max_count = ck.count;
    }
    // meta (device and pinned twin): in_off u64 | out_off u64 | in_len | out_cap | out_len | status (u32 each) per stream, then the
    // clamped length min(out_len, out_cap) (what travels back: the decoded bytes, not the capacity)
    const size_t meta_bytes = (size_t)max_count * 36u + 32u;
    for (int k = 0; k < ZMI_HB_SLOTS; ++k) {
        zmi_ctx::hb_slot& sl = c->hb[k];
        rc = zmi_reserve(sl.in, (size_t)max_in + 64u);
        if (!rc) rc = zmi_reserve(sl.out, (size_t)max_out + 64u);
        if (!rc) rc = zmi_reserve(sl.meta, meta_bytes);
        if (!rc) rc = zmi_reserve_pinned(c->hb_pin_in[k], (size_t)max_in + 64u);
        if (!rc) rc = zmi_reserve_pinned(c->hb_pin_out[k], (size_t)max_out + 64u);
        if (!rc) rc = zmi_reserve_pinned(c->hb_pin_meta[k], meta_bytes);
        if (rc) {   // as in zmi_deflate_batch: without the staging the plain path still works
            zmi_hb_release(c);
            return zmi_inflate_batch_simple(c, in, in_off, in_len, n, wrap, out, out_off, out_cap, out_len, status);
        }
    }
    const uint64_t saved_limit = c->inflate_out_limit;
    c->inflate_out_limit = max_out + (1ull << 20);   // bitmap scratch for the largest chunk, once
    struct restore { zmi_ctx* c; uint64_t v; ~restore() { c->inflate_out_limit = v; } } restore_limit{c, saved_limit};
    auto m_ioff = [&](void* b) { return (uint64_t*)b; };
    auto m_ooff = [&](void* b) { return (uint64_t*)((uint8_t*)b + (size_t)max_count * 8u); };
    auto m_ilen = [&](void* b) { return (uint32_t*)((uint8_t*)b + (size_t)max_count * 16u); };
    auto m_ocap = [&](void* b) { return (uint32_t*)((uint8_t*)b + (size_t)max_count * 20u); };
    auto m_olen = [&](void* b) { return (uint32_t*)((uint8_t*)b + (size_t)max_count * 24u); };
    auto m_st = [&](void* b) { return (int32_t*)((uint8_t*)b + (size_t)max_count * 28u); };
    auto m_clen = [&](void* b) { return (uint32_t*)((uint8_t*)b + (size_t)max_count * 32u); };
    const unsigned T = zmi_host_threads();
    const size_t K = chunks.size();
    std::vector<hipEvent_t> out_done(ZMI_HB_SLOTS);
    for (int k = 0; k < ZMI_HB_SLOTS; ++k) ZMI_HIP(hipEventCreateWithFlags(&out_done[k], hipEventDisableTiming));
    struct ev_guard { std::vector<hipEvent_t>& a; ~ev_guard() { for (hipEvent_t e : a) (void)hipEventDestroy(e); } } evg{out_done};
    uint32_t pack_groups = 16u;
    if (const char* gv = zmi_tune("ZMI_HB_PACK_GROUPS")) { if (atoi(gv) > 0) pack_groups = (uint32_t)atoi(gv); }
    bool dma_out = true;
    if (const char* dv = zmi_tune("ZMI_HB_DMA_OUT")) dma_out = atoi(dv) != 0;
    auto stage_in = [&](size_t k) -> int {   // host threads: caller memory -> pinned staging
        chunk& ck = chunks[k];
        const int s2 = (int)(k % ZMI_HB_SLOTS);
        if (k >= ZMI_HB_SLOTS) ZMI_HIP(hipEventSynchronize(c->hb[s2].in_done));
        std::vector<zmi_copy_job> jobs;
        jobs.reserve(ck.count);
        uint8_t* pin = (uint8_t*)c->hb_pin_in[s2].p;
        for (uint32_t i = 0; i < ck.count; ++i) jobs.push_back({pin + ck.ioff[i], in + in_off[ck.first + i], in_len[ck.first + i]});
        zmi_parallel_copy(jobs, T);
        void* pm = c->hb_pin_meta[s2].p;
        memcpy(m_ioff(pm), ck.ioff.data(), (size_t)ck.count * 8u);
        memcpy(m_ooff(pm), ck.ooff.data(), (size_t)ck.count * 8u);
        memcpy(m_ilen(pm), in_len + ck.first, (size_t)ck.count * 4u);
        memcpy(m_ocap(pm), out_cap + ck.first, (size_t)ck.count * 4u);
        return 0;
    };
    auto issue_dev = [&](size_t k) -> int {
        chunk& ck = chunks[k];
        const int s2 = (int)(k % ZMI_HB_SLOTS);
        zmi_ctx::hb_slot& sl = c->hb[s2];
        if (k >= ZMI_HB_SLOTS) {
            ZMI_HIP(hipStreamWaitEvent(c->hs_in, sl.k_done, 0));      // the kernels of the chunk that had this slot have read its input
            ZMI_HIP(hipStreamWaitEvent(c->hs_in, out_done[s2], 0));   // ... and its copy-out has read the job and the results
        }
        ZMI_HIP(hipMemcpyAsync(sl.in.p, c->hb_pin_in[s2].p, (size_t)ck.in_bytes, hipMemcpyHostToDevice, c->hs_in));
        // (two pieces: the results in the middle of the pinned twin belong to the follower thread until it has read them)
        ZMI_HIP(hipMemcpyAsync(sl.meta.p, c->hb_pin_meta[s2].p, (size_t)max_count * 24u, hipMemcpyHostToDevice, c->hs_in));
        ZMI_HIP(hipEventRecord(sl.in_done, c->hs_in));
        ZMI_HIP(hipStreamWaitEvent(c->hs_k, sl.in_done, 0));
        if (k >= ZMI_HB_SLOTS) ZMI_HIP(hipStreamWaitEvent(c->hs_k, out_done[s2], 0));   // the slot's output has left the device
        int r = zmi_inflate_batch_dev(c, sl.in.p, m_ioff(sl.meta.p), m_ilen(sl.meta.p), ck.count, wrap, sl.out.p, m_ooff(sl.meta.p),
                                      m_ocap(sl.meta.p), m_olen(sl.meta.p), m_st(sl.meta.p), c->hs_k);
        if (r) return r;
        ZMI_HIP(hipEventRecord(sl.k_done, c->hs_k));
        ZMI_HIP(hipStreamWaitEvent(c->hs_slab, sl.k_done, 0));
        // How the chunk's bytes travel.  When the chunks decoded so far filled most of their capacity (the usual case: capacities
        // that fit), the whole region leaves as ONE copy by the DMA engine, enqueued here, ahead of time -- 35.6 GiB/s of output
        // against 29.4 for the pack kernel, whose stores leave over PCIe at ~32 GB/s whatever its launch looks like (8 / 16 / 32 /
        // 64 / 256 workgroups: 21.4 / 29.7 / 29.5 / 28.5 / 25.9 GiB/s).  Otherwise -- nothing known yet, or streams that were given
        // much more room than they needed -- range by range through the pack kernel: only the decoded bytes travel.
        // The follower thread keeps the evidence current (stage_out); it is the context's, so a second call starts with it.
        // (Deciding per chunk from its own sizes was built: the copy then cannot be enqueued before the kernels have finished,
        // and a 2 GiB copy enqueued late ran at 27.8 GiB/s.)
        zmi_launch_clamp_lens(m_olen(sl.meta.p), m_ocap(sl.meta.p), ck.count, m_clen(sl.meta.p), c->hs_slab);
        if (dma_out && c->hb_out_tight.load(std::memory_order_acquire) == 1) {
            ZMI_HIP(hipMemcpyAsync(c->hb_pin_out[s2].p, sl.out.p, (size_t)ck.out_bytes, hipMemcpyDeviceToHost, c->hs_slab));
        } else {
            zmi_scope_timer tm(c, ZMI_K_PACK, c->hs_slab);
            // every stream's decoded bytes (min(out_len, out_cap): a stream given four times the room it needs sends a quarter)
            zmi_launch_copy_ranges_few((const uint8_t*)sl.out.p, m_ooff(sl.meta.p), 0, m_clen(sl.meta.p), ck.count, (uint8_t*)c->hb_pin_out[s2].p,
                                       m_ooff(sl.meta.p), c->hb_pin_out[s2].cap, ck.max_cap, pack_groups, c->hs_slab);
        }
        ZMI_HIP(hipMemcpyAsync(m_olen(c->hb_pin_meta[s2].p), m_olen(sl.meta.p), (size_t)max_count * 12u, hipMemcpyDeviceToHost, c->hs_slab));
        ZMI_HIP(hipEventRecord(out_done[s2], c->hs_slab));
        return 0;
    };
    auto stage_out = [&](size_t k) -> int {   // host threads: pinned output -> the caller's regions
        chunk& ck = chunks[k];
        const int s2 = (int)(k % ZMI_HB_SLOTS);
        ZMI_HIP(hipEventSynchronize(out_done[s2]));
        const void* pm = c->hb_pin_meta[s2].p;
        const uint32_t* ol = m_olen((void*)pm);
        memcpy(out_len + ck.first, ol, (size_t)ck.count * 4u);
        memcpy(status + ck.first, m_st((void*)pm), (size_t)ck.count * 4u);
        std::vector<zmi_copy_job> jobs;
        jobs.reserve(ck.count);
        const uint8_t* pin = (const uint8_t*)c->hb_pin_out[s2].p;
        uint64_t decoded = 0;
        for (uint32_t i = 0; i < ck.count; ++i) {
            const uint32_t cap = out_cap[ck.first + i], take = ol[i] < cap ? ol[i] : cap;
            decoded += take;
            if (take) jobs.push_back({out + out_off[ck.first + i], pin + ck.ooff[i], take});
        }
        c->hb_out_tight.store(decoded * 4u >= ck.out_bytes * 3u ? 1 : 0, std::memory_order_release);   // (three quarters of the room used)
        zmi_parallel_copy(jobs, T);
        return 0;
    };
    std::atomic<int> issued{0}, finished{0}, failed{0};
    const int dev_id = c->device;
    auto consumer = [&]() {
        (void)hipSetDevice(dev_id);
        for (size_t k = 0; k < K; ++k) {
            while (issued.load(std::memory_order_acquire) <= (int)k) {
                if (failed.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            const int r = stage_out(k);
            if (r) { failed.store(r, std::memory_order_release); return; }
            finished.store((int)k + 1, std::memory_order_release);
        }
    };
    std::thread follower;
    zmi_joiner join_guard{follower, failed};
    follower = std::thread(consumer);
    rc = 0;
    for (size_t k = 0; k < K && !rc; ++k) {
        while (finished.load(std::memory_order_acquire) + ZMI_HB_SLOTS <= (int)k && !failed.load(std::memory_order_acquire)) std::this_thread::yield();   // slot free
        if (failed.load(std::memory_order_acquire)) break;
        rc = stage_in(k);
        if (!rc) rc = issue_dev(k);
        if (!rc) issued.store((int)k + 1, std::memory_order_release);
    }
    if (rc) failed.store(rc, std::memory_order_release);
    follower.join();
    if (!rc) rc = failed.load();
    (void)hipStreamSynchronize(c->hs_in);
    (void)hipStreamSynchronize(c->hs_k);
    (void)hipStreamSynchronize(c->hs_slab);
    if (zmi_hb_pinned_bytes(c) > c->pinned_limit) zmi_hb_release(c);
    if (rc) return rc;
    ZMI_HIP(hipGetLastError());
    return ZMI_E_OK;
}

// One host stream through zmi_inflate_resume_dev: `in` (raw deflate, starting at bit in_bit of its first byte),
// `hist` = the up to 32 KiB of output in front of it.  The staging buffers belong to the context and are reused,
// a streaming caller pays no allocation per call.  out receives min(*out_len, out_cap) bytes.
extern "C" int zmi_inflate_resume(zmi_ctx* c, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist,
                                  uint32_t hist_len, uint8_t* out, uint32_t out_cap, uint32_t* out_len, int32_t* status,
                                  int32_t* detail, uint32_t* in_used, uint32_t* resume) {
    if (!c || (!in && in_len) || (!hist && hist_len) || (!out && out_cap) || !out_len || !status || !detail || !in_used || !resume)
        return zmi_fail(ZMI_E_ARG, "null argument");
    if (in_len > 0xFFFFFF00u || out_cap > 0xFFFF0000u || (in_bit & 0xFE0000F8u) != 0u) return zmi_fail(ZMI_E_ARG, "stream too large for one call");
    ZMI_ON_DEVICE(c);
    if (hist_len > 32768u) { hist += hist_len - 32768u; hist_len = 32768u; }
    const size_t base = ((size_t)hist_len + 1023u) & ~(size_t)1023u;   // the output region stays aligned; the history ends where it starts
    int rc = zmi_reserve(c->st_in, (size_t)in_len + 64u);
    if (!rc) rc = zmi_reserve(c->st_out, base + (size_t)out_cap + 64u);
    if (!rc) rc = zmi_reserve(c->st_meta, 256u);
    if (rc) return rc;
    hipStream_t hs = c->host_stream;
    if (in_len) ZMI_HIP(hipMemcpyAsync(c->st_in.p, in, in_len, hipMemcpyHostToDevice, hs));
    if (hist_len) ZMI_HIP(hipMemcpyAsync((uint8_t*)c->st_out.p + base - hist_len, hist, hist_len, hipMemcpyHostToDevice, hs));
    // meta: in_off u64 | out_off u64 | in_len | out_cap | hist | in_bit || out_len | status | in_used | detail | resume[4]
    struct { uint64_t in_off, out_off; uint32_t in_len, out_cap, hist, in_bit; } m = {0, (uint64_t)base, in_len, out_cap, hist_len, in_bit};
    uint8_t* d = (uint8_t*)c->st_meta.p;
    ZMI_HIP(hipMemcpyAsync(d, &m, sizeof(m), hipMemcpyHostToDevice, hs));
    const uint64_t saved_limit = c->inflate_out_limit;
    c->inflate_out_limit = (uint64_t)out_cap + (1ull << 16);
    c->inf_limit_exact = true;
    struct restore { zmi_ctx* c; uint64_t v; ~restore() { c->inflate_out_limit = v; c->inf_limit_exact = false; } } restore_limit{c, saved_limit};
    rc = zmi_inflate_resume_dev(c, c->st_in.p, (const uint64_t*)d, (const uint32_t*)(d + 16), (const uint32_t*)(d + 28), 1, c->st_out.p,
                                (const uint64_t*)(d + 8), (const uint32_t*)(d + 20), (const uint32_t*)(d + 24), (uint32_t*)(d + 32),
                                (int32_t*)(d + 36), (uint32_t*)(d + 40), (int32_t*)(d + 44), (uint32_t*)(d + 48), hs);
    if (rc) { (void)hipStreamSynchronize(hs); return rc; }
    uint32_t r[8], codes = 0;
    ZMI_HIP(hipMemcpyAsync(r, d + 32, sizeof(r), hipMemcpyDeviceToHost, hs));
    // (zmi_inflate_impl's scratch for one stream: bm_off u64 | used | check -- a resumable decode leaves its tables' size in `check`)
    ZMI_HIP(hipMemcpyAsync(&codes, (const uint8_t*)c->inf_tmp.p + 12, 4, hipMemcpyDeviceToHost, hs));
    ZMI_HIP(hipStreamSynchronize(hs));   // this stream only: other contexts keep running
    if (codes != 0u) c->last_codes_used = codes;   // (a call that met no dynamic block header keeps the earlier figure)
    *out_len = r[0];
    *status = (int32_t)r[1];
    *in_used = r[2];
    *detail = (int32_t)r[3];
    for (int i = 0; i < 4; ++i) resume[i] = r[4 + i];
    const uint32_t n = r[0] < out_cap ? r[0] : out_cap;
    if (n) {
        const int crc = zmi_d2h(c, out, (const uint8_t*)c->st_out.p + base, n, hs);
        if (crc) return crc;
    }
    return ZMI_E_OK;
}
