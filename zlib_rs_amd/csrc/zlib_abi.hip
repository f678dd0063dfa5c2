// zlib_abi.hip -- the zlib stream ABI (include/zmi355_zlib.h) on top of the GPU batch pipeline.
//
// What the reference does behind these symbols: libz-rs-sys/src/lib.rs is a thin veneer (null /
// version / enum checks) over the stream state machines zlib-rs/src/deflate.rs:2489-2803 and
// zlib-rs/src/inflate.rs:2376-2457.  Here the veneer does the same argument checking and return
// codes, but the state behind `strm->state` is a host-side staging object; the compression work is
// the batch path of zmi_api.hip (deflate: 1 MiB segments compressed in parallel with the window carried
// over, joined by the stitch recipe of zlib-rs/src/deflate.rs:4149-4221; inflate: the resumable device
// decode zmi_inflate_resume, restarted at block checkpoints).
//
// No exception may cross the C boundary (the reference builds with panic=abort,
// libz-rs-sys-cdylib/src/lib.rs:5-6): every export is noexcept and catches std::bad_alloc.
#include "zmi_kernels.h"
#include "../../include/zmi355.h"
#define ZLIB_CONST 1   // the library itself treats next_in / msg as pointers to const
#include "../../include/zmi355_zlib.h"
#include "host_sums.h"
#include <condition_variable>
#include <mutex>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <limits.h>
#include <string.h>
#include <vector>

extern "C" int zmi_deflate_chain_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                     uint32_t n, uint32_t max_len, int level, int strategy, int finish, void* d_out,
                                     uint64_t out_stride, uint32_t* d_out_len, int32_t* d_status, void* stream_);
extern "C" int zmi_deflate_chain_dict_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                          uint32_t n, uint32_t max_len, int level, int strategy, int finish, uint32_t dict_len,
                                          void* d_out, uint64_t out_stride, uint32_t* d_out_len, int32_t* d_status, void* stream);
extern "C" int zmi_deflate_chain_window_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                            uint32_t n, uint32_t max_len, int level, int strategy, int finish, uint32_t dict_len,
                                            uint32_t window_bits, void* d_out, uint64_t out_stride, uint32_t* d_out_len,
                                            int32_t* d_status, void* stream);
extern "C" int zmi_checksum_batch_dev(zmi_ctx* c, const void* d_data, const uint64_t* d_off, const uint32_t* d_len, uint32_t n,
                                      int kind, uint32_t* d_adler, uint32_t* d_crc, void* stream);
extern "C" int zmi_ctx_set_inflate_out_limit(zmi_ctx* c, uint64_t bytes);
extern "C" int zmi_inflate_batch_dict_dev(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                          uint32_t n, int wrap, void* d_out, const uint64_t* d_out_off,
                                          const uint32_t* d_out_cap, const uint32_t* d_out_hist, uint32_t* d_out_len,
                                          int32_t* d_status, uint32_t* d_in_used, int32_t* d_detail, void* stream);
extern "C" int zmi_inflate_split(zmi_ctx* c, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist,
                                 uint32_t hist_len, uint8_t* out, uint32_t out_cap, const uint32_t* seg_start, uint32_t nseg,
                                 uint32_t* out_len, int32_t* status, int32_t* detail, uint32_t* in_used, uint32_t* resume,
                                 uint32_t* segments_used);
extern "C" int zmi_inflate_resume(zmi_ctx* c, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist,
                                  uint32_t hist_len, uint8_t* out, uint32_t out_cap, uint32_t* out_len, int32_t* status,
                                  int32_t* detail, uint32_t* in_used, uint32_t* resume);
extern "C" int zmi_inflate_blocks(zmi_ctx* c, const uint8_t* in, uint32_t in_len, uint32_t in_bit, const uint8_t* hist, uint32_t hist_len,
                                  uint8_t* out, uint32_t out_cap, uint32_t* out_len, int32_t* status, int32_t* detail, uint32_t* in_used,
                                  uint32_t* resume, uint32_t* segments_used);
extern "C" int zmi_inflate_batch_dev_ex(zmi_ctx* c, const void* d_in, const uint64_t* d_in_off, const uint32_t* d_in_len,
                                        uint32_t n, int wrap, void* d_out, const uint64_t* d_out_off, const uint32_t* d_out_cap,
                                        uint32_t* d_out_len, int32_t* d_status, uint32_t* d_in_used, int32_t* d_detail,
                                        void* stream_);

// ------------------------------------------------------------------------------------------------
// host-side scalar utilities of the ABI (adler32 / crc32 / combine on caller memory).  The
// checksums on the deflate/inflate hot path are computed on the GPU (checksum.hip); these are the
// stand-alone utility entry points (zlib-rs/src/adler32.rs:19-87, crc32.rs:19-29, crc32/combine.rs).
// ------------------------------------------------------------------------------------------------
namespace {
const uint32_t kBase = 65521u, kNmax = 5552u, kPoly = 0xEDB88320u;
uint32_t g_crc_table[8][256];   // slicing-by-8: table k advances a byte that is followed by k more bytes
std::once_flag g_crc_once;
void crc_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? kPoly : 0u);
        g_crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int k = 1; k < 8; ++k) g_crc_table[k][i] = g_crc_table[0][g_crc_table[k - 1][i] & 0xFFu] ^ (g_crc_table[k - 1][i] >> 8);
}
// (host_sums.cpp: carry-less-multiply CRC and SSSE3 Adler where the CPU has them -- the check value of an inflate() stream follows
// the bytes handed out, and with the table / byte loops that was 2-5 ms per 4 MiB drained, a quarter of a streaming call)
uint32_t host_adler32(uint32_t adler, const uint8_t* buf, size_t len) { return zmi_host_adler32(adler, buf, len); }
uint32_t host_crc32(uint32_t crc, const uint8_t* buf, size_t len) { return zmi_host_crc32(crc, buf, len); }
uint32_t gf2_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 31; i >= 0; --i) {
        if ((a >> i) & 1u) p ^= b;
        b = (b >> 1) ^ ((b & 1u) ? kPoly : 0u);
    }
    return p;
}
uint32_t gf2_xpow8(uint64_t nbytes) {  // x^(8 n) mod P
    uint32_t r = 0x80000000u, pw = 0x00800000u;
    while (nbytes) {
        if (nbytes & 1u) r = gf2_mul(r, pw);
        pw = gf2_mul(pw, pw);
        nbytes >>= 1;
    }
    return r;
}
uint32_t host_adler_combine(uint32_t a1, uint32_t a2, uint64_t len2) {
    uint32_t rem = (uint32_t)(len2 % kBase);
    uint32_t s1 = a1 & 0xFFFFu;
    uint32_t s2 = (uint32_t)(((uint64_t)rem * s1) % kBase);
    s1 += (a2 & 0xFFFFu) + kBase - 1u;
    s2 += ((a1 >> 16) & 0xFFFFu) + ((a2 >> 16) & 0xFFFFu) + kBase - rem;
    if (s1 >= kBase) s1 -= kBase;
    if (s1 >= kBase) s1 -= kBase;
    if (s2 >= (kBase << 1)) s2 -= (kBase << 1);
    if (s2 >= kBase) s2 -= kBase;
    return s1 | (s2 << 16);
}

// ------------------------------------------------------------------------------------------------
// Device side of the ABI layer: a small pool of engine contexts, each with its own non-blocking HIP stream.
// A call leases one for its duration, copies / launches / waits on that stream only, and gives it back.  Streams of
// different threads therefore run side by side on the GPU (zlib's contract: one thread per stream, streams freely in
// parallel, zlib-rs/src/deflate.rs:53-54); round 1 held one process-wide mutex across copy -> kernels ->
// hipDeviceSynchronize -> copies, which serialised every caller and stalled the whole device per call.  The only locks
// left guard the pool bookkeeping, for a few instructions.
// ------------------------------------------------------------------------------------------------
extern "C" int zmi_ctx_set_stream(zmi_ctx* c, void* stream);
// the ABI layer's own hipMalloc / hipMemcpy calls run on the contexts' device and leave the caller's current device alone
struct AbiDevice {
    int prev = -1;
    bool switched = false;
    AbiDevice() {
        const int dev = 0;   // the ABI layer's contexts live on device 0
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) switched = hipSetDevice(dev) == hipSuccess && prev >= 0;
    }
    ~AbiDevice() { if (switched) (void)hipSetDevice(prev); }
};
struct AbiSlot { zmi_ctx* ctx = nullptr; hipStream_t stream = nullptr; bool busy = false; };
constexpr int kAbiSlots = 16;
AbiSlot g_slots[kAbiSlots];
std::mutex g_mu;                 // pool bookkeeping only (slots, device buffers): never held across device work
std::condition_variable g_slot_free;
struct AbiLease {
    AbiDevice on_dev;
    int slot = -1;
    zmi_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    AbiLease() {
        std::unique_lock<std::mutex> lk(g_mu);
        for (;;) {
            int fresh = -1;
            for (int i = 0; i < kAbiSlots && slot < 0; ++i) {
                if (g_slots[i].busy) continue;
                if (g_slots[i].ctx) slot = i;
                else if (fresh < 0) fresh = i;
            }
            if (slot < 0 && fresh >= 0) {   // no idle context: make another one
                zmi_ctx* c = nullptr;
                if (zmi_ctx_create(&c, 0) != 0) return;   // no HIP device (or out of memory): ctx stays null
                hipStream_t st = nullptr;
                if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
                (void)zmi_ctx_set_stream(c, st);
                g_slots[fresh].ctx = c;
                g_slots[fresh].stream = st;
                slot = fresh;
            }
            if (slot >= 0) break;
            g_slot_free.wait(lk);           // all sixteen in use: wait for one
        }
        g_slots[slot].busy = true;
        ctx = g_slots[slot].ctx;
        stream = g_slots[slot].stream;
        // More than four calls in flight while the HIP runtime runs with its default of 4 hardware queues: the streams of
        // the extra calls share queues and the scaling stops (measured: 3.3x at 4 and at 8 threads; with 16 queues 6.4x at 8
        // and 11.2x at 16).  Said once, and only to a caller that asked for diagnostics (ZMI_VERBOSE).
        int busy = 0;
        for (int i = 0; i < kAbiSlots; ++i) busy += g_slots[i].busy ? 1 : 0;
        static bool hinted = false;
        if (busy > 4 && !hinted) {
            hinted = true;
            if (getenv("ZMI_VERBOSE") && !getenv("GPU_MAX_HW_QUEUES"))
                fprintf(stderr, "[libz_mi355] %d zlib streams are in flight but the HIP runtime multiplexes them onto 4 hardware queues: export "
                                "GPU_MAX_HW_QUEUES=16 (or ZMI_HW_QUEUES=16) before the process starts for multi-threaded scaling\n", busy);
        }
    }
    ~AbiLease() {
        if (slot < 0) return;
        { std::lock_guard<std::mutex> lk(g_mu); g_slots[slot].busy = false; }
        g_slot_free.notify_one();
    }
    AbiLease(const AbiLease&) = delete;
    AbiLease& operator=(const AbiLease&) = delete;
};
bool abi_device_present() { AbiLease l; return l.ctx != nullptr; }
// The HIP runtime multiplexes a process's streams onto 4 hardware queues unless told otherwise (GPU_MAX_HW_QUEUES): with
// the default, four concurrent zlib streams is where the scaling stopped (measured: 1.9x at 2 threads, 3.3x at 4 and at
// 8; with 16 queues 6.4x at 8 and 11.2x at 16 threads).  A drop-in libz does not edit its host's environment on its own:
// an application that drives many streams from many threads exports GPU_MAX_HW_QUEUES=16 itself, or
// asks this library to with ZMI_HW_QUEUES=<n> -- acted on once, at load, only if GPU_MAX_HW_QUEUES is not set, and it
// only takes effect when the HIP runtime has not been initialised yet.
__attribute__((constructor)) void abi_hw_queues_on_request() {
    const char* want = getenv("ZMI_HW_QUEUES");
    if (want && atoi(want) > 0) (void)setenv("GPU_MAX_HW_QUEUES", want, 0);
}
// Device buffers of one call.  They come from a small pool that outlives the call: hipMalloc / hipFree cost far more than
// the kernels of a small compress2() (hipFree also waits for the device), and a caller that compresses many small buffers
// repeats the same sizes.  The pool is touched under g_mu for the moment of taking / returning a buffer; buffers above kPoolKeep are
// returned to the driver at once so that one large call does not pin its memory.
struct PoolSlot { void* p = nullptr; size_t cap = 0; };   // a parked (idle) device buffer
constexpr int kPoolSlots = 192;
constexpr size_t kPoolKeep = (size_t)256 << 20;
constexpr size_t kPoolBudget = (size_t)2 << 30;   // parked bytes in total: a process that merely links this libz never sits on
                                                  // more idle HBM than this, whatever sizes it has compressed in its life
PoolSlot g_pool[kPoolSlots];
size_t g_pool_bytes = 0;
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    // hipMalloc / hipFree never run under the lock (hipFree waits for the device): the lock covers only the moment a
    // buffer is taken out of or put back into the pool
    ~DevBuf() {
        if (!p) return;
        void* drop = p;
        void* evicted[kPoolSlots];
        int n_evicted = 0;
        if (cap <= kPoolKeep) {
            std::lock_guard<std::mutex> lk(g_mu);
            int at = -1;
            for (int i = 0; i < kPoolSlots && at < 0; ++i)
                if (!g_pool[i].p) at = i;
            if (at < 0)                                   // pool full: the smallest parked buffer makes room if this one is larger
                for (int i = 0; i < kPoolSlots; ++i)
                    if (g_pool[i].cap < cap && (at < 0 || g_pool[i].cap < g_pool[at].cap)) at = i;
            if (at >= 0) {
                drop = g_pool[at].p;
                g_pool_bytes -= g_pool[at].p ? g_pool[at].cap : 0;
                g_pool[at].p = p; g_pool[at].cap = cap;
                g_pool_bytes += cap;
                // over the byte budget: the largest parked buffers go back to the driver (never the one just parked --
                // the next call most likely wants exactly that size again)
                while (g_pool_bytes > kPoolBudget) {
                    int big = -1;
                    for (int i = 0; i < kPoolSlots; ++i)
                        if (i != at && g_pool[i].p && (big < 0 || g_pool[i].cap > g_pool[big].cap)) big = i;
                    if (big < 0) break;
                    evicted[n_evicted++] = g_pool[big].p;
                    g_pool_bytes -= g_pool[big].cap;
                    g_pool[big].p = nullptr; g_pool[big].cap = 0;
                }
            }
        }
        if (drop) (void)hipFree(drop);
        for (int i = 0; i < n_evicted; ++i) (void)hipFree(evicted[i]);
    }
    bool alloc(size_t n) {
        if (n == 0) n = 16;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            int fit = -1;
            for (int i = 0; i < kPoolSlots; ++i)           // smallest parked buffer that is large enough
                if (g_pool[i].p && g_pool[i].cap >= n && (fit < 0 || g_pool[i].cap < g_pool[fit].cap)) fit = i;
            if (fit >= 0) { p = g_pool[fit].p; cap = g_pool[fit].cap; g_pool_bytes -= cap; g_pool[fit].p = nullptr; g_pool[fit].cap = 0; return true; }
        }
        const size_t want = n + n / 4 + 256;               // some room for the next, slightly larger, call
        if (hipMalloc(&p, want) == hipSuccess) { cap = want; return true; }
        p = nullptr;
        if (hipMalloc(&p, n) == hipSuccess) { cap = n; return true; }
        p = nullptr;
        return false;
    }
};

// test overrides of the ABI layer's sizes: only in a process started with ZMI_TUNING set (checked once)
const char* abi_tune(const char* name) {
    static const bool enabled = getenv("ZMI_TUNING") != nullptr;
    return enabled ? getenv(name) : nullptr;
}
// 64 KiB segments (one match-search workgroup and one encoder wave each): a 4 MiB deflate() call is 64 workgroups on the 256 CUs
// -- with the 1 MiB segments of rounds 1-4 it was 4, and the single-stream path ran at 1 % of the batch rate.  A segment sees the
// 27 KiB in front of it (window carry-over), the encoder cut its pieces -- with the same marker behind each -- every 64 KiB
// already, so the stream a caller gets is the same one; ZMI_ABI_SEGMENT (bytes, multiple of 64) overrides for tests
// A call that brings less than 8 MiB -- fewer than 128 such workgroups for 256 CUs -- takes 32 KiB segments: the match search of a
// segment is one workgroup's latency (171 us for 64 KiB, of which hashing the 27 KiB carried over is the fixed part), and the
// compressed bytes are the same size (4 MiB calls: 4.80 -> 5.20 GiB/s at ratio 2.3281 both; a 15.7 MB call, 240 segments of 64 KiB,
// would lose 4 % to the doubled carry-over work: tools/gpu_stream_loop_probe.py).
size_t segment_bytes(size_t call_bytes) {
    struct Seg { size_t bytes; bool fixed; };
    static const Seg g = [] {   // (read once, by whichever thread comes first)
        const char* e = abi_tune("ZMI_ABI_SEGMENT");
        const long n = e ? atol(e) : 0;
        const bool fixed = n >= 64 && n <= (1 << 28);
        return Seg{fixed ? ((size_t)n & ~(size_t)63) : ((size_t)64 << 10), fixed};
    }();
    return (!g.fixed && call_bytes < ((size_t)8 << 20)) ? g.bytes / 2 : g.bytes;
}
// deflate(Z_NO_FLUSH) compresses what is buffered once this much has come in (the reference emits whenever its pending buffer
// fills, zlib-rs/src/deflate.rs:2805-2826 flush_pending): a zpipe.c-style caller sees output as it goes and the stream holds a
// few MiB, not its whole input.  ZMI_ABI_EMIT (bytes) overrides for tests.
size_t emit_bytes() {   // (not cached: abi_tune is a null pointer in a product process, and the tests change the value between streams)
    const char* e = abi_tune("ZMI_ABI_EMIT");
    const long long n = e ? atoll(e) : 0;
    return n >= 64 ? (size_t)n : ((size_t)4 << 20);
}

// compress `n` host bytes as consecutive raw-deflate segments; appends the bytes to `out`.  `hist` (hist_len
// bytes, may be 0) is what the stream has in front of these bytes: earlier input or a preset dictionary.
// returns 0 or a negative zlib code
// wrap 1 / 2: *check receives the Adler-32 / CRC-32 of the n bytes (per segment on the GPU, where the data is;
// stitched with the combine algebra, crc32/combine.rs, adler32 combine lib.rs:372)
template <class OutVec>
int gpu_deflate_segments(const uint8_t* in, size_t n, const uint8_t* hist, size_t hist_len, int level, int strategy, bool finish,
                         OutVec& out, int wrap = 0, uint32_t* check = nullptr, size_t* last_at = nullptr,
                         int wbits = 15) {
    if (check) *check = wrap == 1 ? 1u : 0u;
    if (last_at) *last_at = out.size();
    if (n == 0) {
        if (finish) { out.push_back(0x03); out.push_back(0x00); }  // empty final static block (deflate.rs: 03 00)
        else { const uint8_t m[5] = {0x00, 0x00, 0x00, 0xFF, 0xFF}; out.insert(out.end(), m, m + 5); }
        return Z_OK;
    }
    AbiLease lease;
    zmi_ctx* c = lease.ctx;
    hipStream_t hs = lease.stream;
    if (!c) return Z_MEM_ERROR;
    const size_t kSegment = segment_bytes(n);
    const uint32_t nseg = (uint32_t)((n + kSegment - 1) / kSegment);
    std::vector<uint64_t> off(nseg);
    std::vector<uint32_t> len(nseg);
    const size_t base = (hist_len + 1023u) & ~(size_t)1023u;   // the history sits right in front of the (aligned) input
    for (uint32_t i = 0; i < nseg; ++i) {
        const size_t at = (size_t)i * kSegment;
        off[i] = base + at;
        len[i] = (uint32_t)((n - at < kSegment) ? n - at : kSegment);
    }
    const uint32_t max_len = len[0];
    // + 16: a segment that does not end the stream is followed by the 5-byte marker of a sync flush, which compress_bound
    // (the reference's bound for a finished stream) does not cover for very short incompressible segments (11 bytes: 13 bytes
    // of fixed-Huffman block + marker = 17 > bound 16)
    const uint64_t stride = zmi_deflate_bound(max_len, ZMI_WRAP_RAW) + 16u;
    DevBuf d_in, d_off, d_len, d_out, d_olen, d_st, d_slab, d_soff, d_sum;   // (all in front of the Sync guard: destroyed after it)
    if (!d_in.alloc(base + n + 16) || !d_off.alloc(nseg * 8) || !d_len.alloc(nseg * 4) || !d_out.alloc((size_t)nseg * stride) ||
        !d_olen.alloc(nseg * 4) || !d_st.alloc(nseg * 4) || !d_slab.alloc((size_t)nseg * stride) || !d_soff.alloc(((size_t)nseg + 1) * 8))
        return Z_MEM_ERROR;
    struct Sync { hipStream_t s; ~Sync() { (void)hipStreamSynchronize(s); } } sync_at_exit{hs};   // nothing in flight when the buffers go back
    if (hist_len && hipMemcpyAsync((uint8_t*)d_in.p + base - hist_len, hist, hist_len, hipMemcpyHostToDevice, hs) != hipSuccess) return Z_MEM_ERROR;
    if (hipMemcpyAsync((uint8_t*)d_in.p + base, in, n, hipMemcpyHostToDevice, hs) != hipSuccess) return Z_MEM_ERROR;
    if (hipMemcpyAsync(d_off.p, off.data(), nseg * 8, hipMemcpyHostToDevice, hs) != hipSuccess) return Z_MEM_ERROR;
    if (hipMemcpyAsync(d_len.p, len.data(), nseg * 4, hipMemcpyHostToDevice, hs) != hipSuccess) return Z_MEM_ERROR;
    // windowBits < 15: distances stay inside the window the header announces (deflate.rs:1423-1425)
    if (zmi_deflate_chain_window_dev(c, d_in.p, (const uint64_t*)d_off.p, (const uint32_t*)d_len.p, nseg, max_len, level, strategy,
                                     finish ? 1 : 0, (uint32_t)hist_len, (uint32_t)wbits, d_out.p, stride, (uint32_t*)d_olen.p,
                                     (int32_t*)d_st.p, hs) != 0)
        return Z_MEM_ERROR;
    if (check && wrap != 0) {
        if (!d_sum.alloc((size_t)nseg * 8)) return Z_MEM_ERROR;
        if (zmi_checksum_batch_dev(c, d_in.p, (const uint64_t*)d_off.p, (const uint32_t*)d_len.p, nseg, wrap == 1 ? 1 : 2,
                                   (uint32_t*)d_sum.p, (uint32_t*)d_sum.p + nseg, hs) != 0)
            return Z_MEM_ERROR;
    }
    // the segments' slots are packed into one dense slab on the device (csrc/pack.hip): the compressed stream leaves in
    // ONE copy, whatever the number of segments (round 1: a copy per segment)
    if (zmi_pack_slab_dev(c, d_out.p, stride, (const uint32_t*)d_olen.p, nseg, d_slab.p, (uint64_t)nseg * stride, (uint64_t*)d_soff.p, hs) != 0)
        return Z_MEM_ERROR;
    std::vector<uint64_t> soff((size_t)nseg + 1);
    std::vector<int32_t> st(nseg);
    std::vector<uint32_t> sums((size_t)nseg * 2);
    if (hipMemcpyAsync(soff.data(), d_soff.p, ((size_t)nseg + 1) * 8, hipMemcpyDeviceToHost, hs) != hipSuccess) return Z_MEM_ERROR;
    if (hipMemcpyAsync(st.data(), d_st.p, nseg * 4, hipMemcpyDeviceToHost, hs) != hipSuccess) return Z_MEM_ERROR;
    if (check && wrap != 0 && hipMemcpyAsync(sums.data(), d_sum.p, (size_t)nseg * 8, hipMemcpyDeviceToHost, hs) != hipSuccess) return Z_MEM_ERROR;
    if (hipStreamSynchronize(hs) != hipSuccess) return Z_MEM_ERROR;
    if (check && wrap != 0) {
        uint32_t acc = wrap == 1 ? 1u : 0u;
        for (uint32_t i = 0; i < nseg; ++i) {
            if (wrap == 1) acc = host_adler_combine(acc, sums[i], len[i]);
            else acc = gf2_mul(gf2_xpow8(len[i]), acc) ^ sums[nseg + i];
        }
        *check = acc;
    }
    for (uint32_t i = 0; i < nseg; ++i)
        if (st[i] != 0) return Z_BUF_ERROR;
    const size_t at0 = out.size(), total = (size_t)soff[nseg];
    if (last_at) *last_at = at0 + (size_t)soff[nseg - 1];
    out.resize(at0 + total);
    if (total && hipMemcpyAsync(out.data() + at0, d_slab.p, total, hipMemcpyDeviceToHost, hs) != hipSuccess) return Z_MEM_ERROR;
    if (hipStreamSynchronize(hs) != hipSuccess) return Z_MEM_ERROR;
    return Z_OK;
}

// one-stream inflate on the GPU.  status: zlib code (0 = complete), detail 1 = need input, 2 = need output.
// dict / dict_len: preset dictionary (at most its last 32 KiB matter), placed directly in front of the output.
template <class OutVec>
int gpu_inflate_stream(const uint8_t* in, size_t n, int wrap, OutVec& out, size_t cap, uint32_t* in_used,
                       int32_t* status, int32_t* detail, const uint8_t* dict = nullptr, size_t dict_len = 0) {
    AbiLease lease;
    zmi_ctx* c = lease.ctx;
    hipStream_t hs = lease.stream;
    if (!c) return Z_MEM_ERROR;
    if (n > 0xFFFFFFF0ull || cap > 0xFFFFFFF0ull) return Z_MEM_ERROR;
    if (dict_len > 32768u) { dict += dict_len - 32768u; dict_len = 32768u; }
    const size_t base = (dict_len + 1023u) & ~(size_t)1023u;   // output region stays aligned; the dictionary ends where it starts
    DevBuf d_in, d_out, d_meta;
    if (!d_in.alloc(n + 16) || !d_out.alloc(base + cap + 16) || !d_meta.alloc(64)) return Z_MEM_ERROR;
    struct Sync { hipStream_t s; ~Sync() { (void)hipStreamSynchronize(s); } } sync_at_exit{hs};
    if (n && hipMemcpyAsync(d_in.p, in, n, hipMemcpyHostToDevice, hs) != hipSuccess) return Z_MEM_ERROR;
    if (dict_len && hipMemcpyAsync((uint8_t*)d_out.p + base - dict_len, dict, dict_len, hipMemcpyHostToDevice, hs) != hipSuccess) return Z_MEM_ERROR;
    // meta layout: in_off u64 | out_off u64 | in_len u32 | out_cap u32 | out_len u32 | status i32 | in_used u32 | detail i32 | hist u32
    uint64_t offs[2] = {0, (uint64_t)base};
    uint32_t lens[2] = {(uint32_t)n, (uint32_t)cap};
    uint32_t hist = (uint32_t)dict_len;
    uint8_t* m = (uint8_t*)d_meta.p;
    if (hipMemcpyAsync(m, offs, 16, hipMemcpyHostToDevice, hs) != hipSuccess) return Z_MEM_ERROR;
    if (hipMemcpyAsync(m + 16, lens, 8, hipMemcpyHostToDevice, hs) != hipSuccess) return Z_MEM_ERROR;
    if (hipMemcpyAsync(m + 40, &hist, 4, hipMemcpyHostToDevice, hs) != hipSuccess) return Z_MEM_ERROR;
    (void)zmi_ctx_set_inflate_out_limit(c, (uint64_t)cap + 4096u);
    if (zmi_inflate_batch_dict_dev(c, d_in.p, (const uint64_t*)m, (const uint32_t*)(m + 16), 1, wrap, d_out.p,
                                   (const uint64_t*)(m + 8), (const uint32_t*)(m + 20), (const uint32_t*)(m + 40),
                                   (uint32_t*)(m + 24), (int32_t*)(m + 28), (uint32_t*)(m + 32), (int32_t*)(m + 36), hs) != 0)
        return Z_MEM_ERROR;
    uint32_t res[4];
    if (hipMemcpyAsync(res, m + 24, 16, hipMemcpyDeviceToHost, hs) != hipSuccess) return Z_MEM_ERROR;
    if (hipStreamSynchronize(hs) != hipSuccess) return Z_MEM_ERROR;
    uint32_t olen = res[0];
    *status = (int32_t)res[1];
    *in_used = res[2];
    *detail = (int32_t)res[3];
    if (olen > cap) olen = (uint32_t)cap;
    out.resize(olen);
    if (olen && hipMemcpyAsync(out.data(), (const uint8_t*)d_out.p + base, olen, hipMemcpyDeviceToHost, hs) != hipSuccess) return Z_MEM_ERROR;
    if (hipStreamSynchronize(hs) != hipSuccess) return Z_MEM_ERROR;
    return Z_OK;
}

enum { KIND_DEFLATE = 0x5A44, KIND_INFLATE = 0x5A49 };

// Every buffer a stream owns comes from the stream's zalloc and goes back through its zfree, as the reference's whole
// arena does (zlib-rs/src/deflate.rs:252-439, zlib-rs/src/allocate.rs:200-222): a caller that counts or limits its
// allocations (the reference's own tests do: mem_setup / mem_limit in test-libz-rs-sys) sees all host memory of the
// stream.  (Device memory is the GPU's: it has no zalloc.)  A failed zalloc surfaces as std::bad_alloc, which every entry
// point turns into Z_MEM_ERROR.
struct ZCalls { alloc_func za = nullptr; free_func zf = nullptr; voidpf op = nullptr; };
template <class T>
struct ZAlloc {
    using value_type = T;
    using propagate_on_container_copy_assignment = std::false_type;   // a copied state keeps the allocator of ITS stream
    using propagate_on_container_move_assignment = std::false_type;
    using propagate_on_container_swap = std::false_type;
    ZCalls c;
    ZAlloc() = default;
    explicit ZAlloc(const ZCalls& cc) : c(cc) {}
    template <class U> ZAlloc(const ZAlloc<U>& o) : c(o.c) {}
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        void* p;
        if (!c.za) p = malloc(bytes ? bytes : 1);
        else if (bytes <= 0xFFFFFFFFull) p = c.za(c.op, (uInt)(bytes ? bytes : 1), 1u);   // zalloc(opaque, items, size): 32-bit counts
        else p = c.za(c.op, (uInt)((bytes + 65535u) >> 16), 65536u);
        if (!p) throw std::bad_alloc();
        return (T*)p;
    }
    void deallocate(T* p, size_t) { if (c.zf) c.zf(c.op, p); else free(p); }
    template <class U> bool operator==(const ZAlloc<U>& o) const { return c.za == o.c.za && c.zf == o.c.zf && c.op == o.c.op; }
    template <class U> bool operator!=(const ZAlloc<U>& o) const { return !(*this == o); }
};
using Bytes = std::vector<uint8_t, ZAlloc<uint8_t>>;

struct DeflateState {
    ZCalls zc;                     // the stream's allocator: every Bytes member below allocates through it
    explicit DeflateState(const ZCalls& z) : zc(z), in(ZAlloc<uint8_t>(z)), pending(ZAlloc<uint8_t>(z)), hist(ZAlloc<uint8_t>(z)), last_seg(ZAlloc<uint8_t>(z)) {}
    int kind = KIND_DEFLATE;
    int level = 6, strategy = 0, wrap = 1, wbits = 15;
    bool header_done = false, finished = false, trailer_done = false;
    Bytes in;       // input not yet compressed
    Bytes pending;  // compressed bytes not yet handed to the caller
    size_t pending_pos = 0;
    uint32_t adler = 1, crc = 0;
    uint64_t total_len = 0;
    int last_flush = -2;
    Bytes hist;     // the up to 32 KiB the stream has in front of `in`: earlier input or the preset dictionary
    bool dict_set = false;         // zlib wrapper: header announces the dictionary (FDICT + DICTID)
    uint32_t dictid = 0;
    gz_headerp gzhead = nullptr;   // deflateSetHeader: read when the header is written, as the reference does
    uint32_t prime_val = 0;        // deflatePrime: bits (< 8) waiting in front of the next compressed data
    int prime_bits = 0;
    Bytes last_seg; // compressed bytes of the most recent segment (deflateUsed decodes them on demand)
    bool last_seg_final = false;
    int used_bits = 0;             // deflateUsed: 0 = no flush yet, -1 = not computed for last_seg yet
};
// The inflate side is a host state machine around the resumable device decode (zmi_inflate_resume): wrapper
// header and trailer are parsed here, the deflate data goes to the GPU from the last block boundary the decode
// reached (the checkpoint), with the 32 KiB of output in front of it as history.  The reference keeps the
// equivalent facts in Mode / BitReader / Window (zlib-rs/src/inflate.rs:288-320).
enum { IM_HEAD = 0, IM_DICT, IM_BLOCKS, IM_TRAILER, IM_DONE, IM_BAD };
struct InflateState {
    ZCalls zc;
    explicit InflateState(const ZCalls& z) : zc(z), in(ZAlloc<uint8_t>(z)), hist(ZAlloc<uint8_t>(z)), out(ZAlloc<uint8_t>(z)), tmp(ZAlloc<uint8_t>(z)),
                                             window(ZAlloc<uint8_t>(z)), dict(ZAlloc<uint8_t>(z)) {}
    int kind = KIND_INFLATE;
    int wrap = 1, wbits = 15;
    int hdr_wbits = 0;             // windowBits 0: the window size the zlib header announces
    int mode = IM_HEAD;
    int form = -1;                 // wrapper found: 0 raw, 1 zlib, 2 gzip (-1: no header seen yet)
    Bytes in;       // input from the checkpoint on; the next block starts at bit `sbit` of in[0]
    uint32_t sbit = 0;
    size_t tried = (size_t)-1;     // in.size() at the last decode attempt
    uInt prev_in0 = 0;             // avail_in of the previous inflate() call (ZMI_INFLATE_DEFER: a shorter piece ends the deferral)
    size_t stop = 0;               // how far into `in` the decoder got (inflateSync searches from there)
    Bytes hist;     // the up to 32 KiB of output in front of the checkpoint; starts as the preset dictionary
    size_t pend = 0;               // decoded bytes behind the checkpoint that are already queued
    Bytes out;      // decoded bytes not yet handed to the caller
    size_t out_pos = 0;
    Bytes tmp;      // decode target of one attempt
    uint32_t check = 1;            // Adler-32 / CRC-32 of the bytes handed to the caller
    uint32_t want_check = 0;       // trailer values, compared when the last byte has been handed out
    uint32_t want_len = 0;
    bool check_seen = false;       // the trailer's check value has arrived
    uint64_t total = 0;            // bytes decoded
    bool verify = true;            // inflateValidate
    int error = 0;
    const char* errmsg = nullptr;
    bool have_dict = false;
    uint32_t dictid = 0;
    gz_headerp gzhead = nullptr;   // inflateGetHeader
    Bytes window;   // the last 32 KiB handed to the caller (inflateGetDictionary)
    uint64_t bias = 0;             // buffered bytes inflateSync reported as not yet consumed
    int sync_have = 0;             // inflateSync: marker bytes matched so far
    bool in_sync = false;
    int last_block = 0;
    uint32_t primed = 0;           // bits at the front of `in` that came from inflatePrime
    Bytes dict;     // the preset dictionary (inflateGetDictionary shows it in front of the output)
    uint8_t* back_window = nullptr;   // inflateBack: the caller's window
    // inflate(Z_BLOCK) / inflate(Z_TREES) (inflate.rs:1276-1284,1323,1369,1772,1856-1873): the device decode stops at the
    // next block boundary / behind the next block header; the call that reaches the stop reports it in data_type
    uint32_t codes_used = 0;       // inflateCodesUsed: table entries of the most recent dynamic block the device decoded (0: none yet)
    int stop_state = 0;            // 0 none, 1 at a block boundary (the reference's Mode::Type), 2 behind a block header (Len_ / CopyBlock)
    bool stop_reported = false;    // a call has returned with the stop in data_type: the next one moves on
    bool hdr_seen = false;         // Z_TREES: the header of the block at the checkpoint has been reported
    uint32_t stop_bits = 0;        // unused bits of the last byte consumed at the stop
    bool hdr_last = false;         // the block whose header was reported is the final one
};

// length of the gzip header at the start of `in` (inflate.rs:1063-1275): 0 while it is incomplete, -1 with *err
// set when it is invalid; fills *h (may be null) once the header is complete
template <class InVec>
long gzip_header_len(const InVec& in, gz_header* h, bool verify, const char** err) {
    if (in.size() < 10) return 0;
    if (in[2] != 8) { *err = "unknown compression method"; return -1; }
    const uint8_t flg = in[3];
    if (flg & 0xE0) { *err = "unknown header flags set"; return -1; }
    size_t p = 10;
    size_t xoff = 0, xlen = 0, noff = 0, coff = 0;
    if (flg & 4) {
        if (p + 2 > in.size()) return 0;
        xlen = in[p] | ((size_t)in[p + 1] << 8);
        xoff = p + 2;
        p += 2 + xlen;
        if (p > in.size()) return 0;
    }
    if (flg & 8) { noff = p; while (p < in.size() && in[p]) ++p; if (p >= in.size()) return 0; ++p; }
    if (flg & 16) { coff = p; while (p < in.size() && in[p]) ++p; if (p >= in.size()) return 0; ++p; }
    if (flg & 2) {
        if (p + 2 > in.size()) return 0;
        const uint32_t c = host_crc32(0, in.data(), p) & 0xFFFFu;
        if (verify && c != (uint32_t)(in[p] | (in[p + 1] << 8))) { *err = "header crc mismatch"; return -1; }
        p += 2;
    }
    if (h) {
        h->text = flg & 1;
        h->time = (uLong)(in[4] | ((uint32_t)in[5] << 8) | ((uint32_t)in[6] << 16) | ((uint32_t)in[7] << 24));
        h->xflags = in[8];
        h->os = in[9];
        h->hcrc = (flg >> 1) & 1;
        h->extra_len = (uInt)xlen;
        if ((flg & 4) && h->extra) memcpy(h->extra, in.data() + xoff, xlen < h->extra_max ? xlen : h->extra_max);
        auto copy_str = [&](size_t off, Bytef* dst, uInt cap) {
            if (!dst || cap == 0) return;
            size_t n = strlen((const char*)in.data() + off) + 1;   // the terminator is stored when it fits (inflate.rs:1176-1222)
            memcpy(dst, in.data() + off, n < cap ? n : cap);
        };
        if (flg & 8) copy_str(noff, h->name, h->name_max); else h->name = nullptr;
        if (flg & 16) copy_str(coff, h->comment, h->comm_max); else h->comment = nullptr;
        h->done = 1;
    }
    return (long)p;
}


const char* const kErrMsg[10] = {"need dictionary", "stream end", "", "file error", "stream error", "data error",
                                 "insufficient memory", "buffer error", "incompatible version", ""};

bool version_ok(const char* version, int stream_size) {  // lib.rs:2133-2143
    return version && version[0] == '1' && stream_size == (int)sizeof(z_stream);
}
// The reference installs its default allocator pair as soon as either callback is missing, so that the state is always
// freed by the partner of what allocated it (zlib-rs/src/deflate.rs:264-280, inflate.rs:2240-2251; c_api.rs:120-122).
voidpf default_zalloc(voidpf, uInt items, uInt size) { return malloc((size_t)items * size); }
void default_zfree(voidpf, voidpf p) { free(p); }
template <typename T>
T* alloc_state(z_streamp strm) {
    if (!strm->zalloc || !strm->zfree) { strm->zalloc = default_zalloc; strm->zfree = default_zfree; strm->opaque = nullptr; }
    void* mem = strm->zalloc(strm->opaque, 1, (uInt)sizeof(T));
    if (!mem) return nullptr;
    return new (mem) T(ZCalls{strm->zalloc, strm->zfree, strm->opaque});
}
template <typename T>
void free_state(z_streamp strm, T* st) {
    st->~T();
    if (strm->zfree) strm->zfree(strm->opaque, st);
    else free(st);   // (a stream whose callbacks were cleared after init: what the default pair would have done)
}
DeflateState* dstate(z_streamp strm) {
    if (!strm || !strm->state) return nullptr;
    DeflateState* s = (DeflateState*)strm->state;
    return s->kind == KIND_DEFLATE ? s : nullptr;
}
InflateState* istate(z_streamp strm) {
    if (!strm || !strm->state) return nullptr;
    InflateState* s = (InflateState*)strm->state;
    return s->kind == KIND_INFLATE ? s : nullptr;
}
void put_header(DeflateState* s) {
    if (s->wrap == 1) {  // deflate.rs:1572-1601
        unsigned lf = (s->strategy >= 2 || s->level < 2) ? 0 : (s->level < 6 ? 1 : (s->level == 6 ? 2 : 3));
        unsigned h = ((8u + ((unsigned)(s->wbits - 8) << 4)) << 8) | (lf << 6) | (s->dict_set ? 0x20u : 0u);
        h += 31 - (h % 31);
        s->pending.push_back((uint8_t)(h >> 8));
        s->pending.push_back((uint8_t)h);
        if (s->dict_set)   // DICTID: Adler-32 of the dictionary, big-endian (deflate.rs:1596-1600)
            for (int i = 3; i >= 0; --i) s->pending.push_back((uint8_t)(s->dictid >> (8 * i)));
    } else if (s->wrap == 2) {  // deflate.rs:2574-2700
        const uint8_t xfl = (uint8_t)(s->level == 9 ? 2 : ((s->strategy >= 2 || s->level < 2) ? 4 : 0));
        const size_t at = s->pending.size();
        if (!s->gzhead) {
            const uint8_t g[10] = {0x1F, 0x8B, 8, 0, 0, 0, 0, 0, xfl, 3};
            s->pending.insert(s->pending.end(), g, g + 10);
        } else {
            const gz_header& h = *s->gzhead;
            const uint8_t flg = (uint8_t)((h.text ? 1 : 0) | (h.hcrc ? 2 : 0) | (h.extra ? 4 : 0) | (h.name ? 8 : 0) | (h.comment ? 16 : 0));
            const uint32_t t = (uint32_t)h.time;
            const uint8_t g[10] = {0x1F, 0x8B, 8, flg, (uint8_t)t, (uint8_t)(t >> 8), (uint8_t)(t >> 16), (uint8_t)(t >> 24), xfl, (uint8_t)h.os};
            s->pending.insert(s->pending.end(), g, g + 10);
            if (h.extra) {
                const uint32_t xl = h.extra_len & 0xFFFFu;
                s->pending.push_back((uint8_t)xl);
                s->pending.push_back((uint8_t)(xl >> 8));
                s->pending.insert(s->pending.end(), h.extra, h.extra + xl);
            }
            if (h.name) s->pending.insert(s->pending.end(), h.name, h.name + strlen((const char*)h.name) + 1);
            if (h.comment) s->pending.insert(s->pending.end(), h.comment, h.comment + strlen((const char*)h.comment) + 1);
            if (h.hcrc) {   // CRC-16 = low half of the CRC-32 of the header so far
                const uint32_t c = host_crc32(0, s->pending.data() + at, s->pending.size() - at);
                s->pending.push_back((uint8_t)c);
                s->pending.push_back((uint8_t)(c >> 8));
            }
        }
    }
    s->header_done = true;
}
// full_flush: the caller asked for Z_FULL_FLUSH -- the data after it must not refer to anything before it
// (data, n): what to compress -- the stream's buffer (the default), or the caller's own buffer when a call brings a whole launch's
// worth and nothing is buffered in front of it: the bytes then go from the caller's memory to the device without a stop in the
// stream's buffer (a 4 MiB deflate() call: 0.3 of its 1.3 ms).  The call stays synchronous: the caller's buffer is read only while
// deflate() runs (zlib-rs/src/deflate.rs:1697-1698 copies its input for the same reason).
int compress_buffered(DeflateState* s, bool finish, bool full_flush = false, const uint8_t* data = nullptr, size_t n = 0) {
    const bool direct = data != nullptr;
    if (!direct) { data = s->in.data(); n = s->in.size(); }
    if (!s->header_done) put_header(s);
    if (s->prime_bits) {
        // deflatePrime left a partial byte.  Every segment of this engine starts on a byte boundary, so an empty
        // stored block (3 header bits, padding, 00 00 FF FF) follows the primed bits: valid deflate, byte aligned again
        s->pending.push_back((uint8_t)s->prime_val);
        if (s->prime_bits > 5) s->pending.push_back(0);   // the block header does not fit into that byte
        const uint8_t m[4] = {0x00, 0x00, 0xFF, 0xFF};
        s->pending.insert(s->pending.end(), m, m + 4);
        s->prime_bits = 0;
        s->prime_val = 0;
    }
    s->total_len += n;
    uint32_t part = 0;   // checksum of this call's input, computed on the GPU next to the compression
    size_t last_at = 0;
    int rc = gpu_deflate_segments(data, n, s->hist.data(), s->hist.size(), s->level, s->strategy, finish, s->pending,
                                  s->wrap, &part, &last_at, s->wbits);
    if (rc == Z_OK) {
        s->last_seg.assign(s->pending.begin() + last_at, s->pending.end());
        s->last_seg_final = finish;
        s->used_bits = -1;
    }
    if (rc == Z_OK && s->wrap == 1) s->adler = host_adler_combine(s->adler, part, n);
    if (rc == Z_OK && s->wrap == 2) s->crc = gf2_mul(gf2_xpow8(n), s->crc) ^ part;
    // window carry-over to the next call: the last 32 KiB of what the stream has seen (deflate.rs:2739-2752: only
    // Z_FULL_FLUSH forgets it)
    // (Z_FINISH leaves the window as it is -- deflateGetDictionary after the last deflate() still shows it, as
    // libz-rs-sys-cdylib/example.c test_deflate_get_dict expects; nothing is compressed against it any more)
    if (full_flush) s->hist.clear();
    else {
        if (n >= 32768u) s->hist.assign(data + (n - 32768u), data + n);
        else {
            s->hist.insert(s->hist.end(), data, data + n);
            if (s->hist.size() > 32768u) s->hist.erase(s->hist.begin(), s->hist.end() - 32768);
        }
    }
    if (!direct) s->in.clear();
    if (rc != Z_OK) return rc;
    if (finish) {
        if (s->wrap == 1) {  // deflate.rs:2786-2788
            for (int i = 3; i >= 0; --i) s->pending.push_back((uint8_t)(s->adler >> (8 * i)));
        } else if (s->wrap == 2) {  // deflate.rs:2773-2785
            for (int i = 0; i < 4; ++i) s->pending.push_back((uint8_t)(s->crc >> (8 * i)));
            for (int i = 0; i < 4; ++i) s->pending.push_back((uint8_t)((uint32_t)s->total_len >> (8 * i)));
        }
        s->finished = true;
    }
    return Z_OK;
}
template <class Vec>
size_t drain(z_streamp strm, Vec& buf, size_t& pos) {
    size_t n = buf.size() - pos;
    if (n > strm->avail_out) n = strm->avail_out;
    if (n) {
        memcpy(strm->next_out, buf.data() + pos, n);
        strm->next_out += n;
        strm->avail_out -= (uInt)n;
        strm->total_out += n;
        pos += n;
    }
    if (pos == buf.size()) { buf.clear(); pos = 0; }
    return n;
}
// ---- inflate machinery ----
void inf_bad(InflateState* s, const char* msg) {
    s->mode = IM_BAD;
    s->error = Z_DATA_ERROR;
    s->errmsg = msg;
    s->in.clear();
    s->sbit = 0;
}
// The history a restart sees is the window the stream was opened with (1 << windowBits; the header's size for windowBits
// 0): a distance that reaches further back than window + bytes decoded since the checkpoint is "invalid distance too far
// back", as the reference reports it for window.have() + bytes written in the call (inflate.rs:839-850,2042-2047).
void inf_keep_hist(InflateState* s, const uint8_t* p, size_t n) {
    const size_t w = s->wbits >= 8 && s->wbits <= 15 ? (size_t)1 << s->wbits : (s->hdr_wbits ? (size_t)1 << s->hdr_wbits : 32768u);
    if (n >= w) s->hist.assign(p + (n - w), p + n);
    else {
        s->hist.insert(s->hist.end(), p, p + n);
        if (s->hist.size() > w) s->hist.erase(s->hist.begin(), s->hist.end() - w);
    }
}
// decoded bytes waiting for the caller before decoding pauses / input bytes handed to one device decode.
// ZMI_ABI_QUEUE and ZMI_ABI_TAKE (bytes) override them so that tests reach these paths with small streams.
size_t abi_limit(const char* name, size_t dflt) {
    const char* e = abi_tune(name);
    const long long v = e ? atoll(e) : 0;
    return v > 0 ? (size_t)v : dflt;
}
size_t queue_limit() { return abi_limit("ZMI_ABI_QUEUE", (size_t)32 << 20); }
// ZMI_INFLATE_DEFER=BYTES (with ZMI_TUNING=1; read once; default 0 = off): inflate(Z_NO_FLUSH) may take its input and decode
// LATER, once BYTES have come in since the last decode -- what zlib calls output latency ("may introduce some output latency
// (reading input without producing any output) except when forced to flush").  A device decode costs a launch and a round trip
// (~170 us) whatever it is given, and restarts at the last block boundary: a caller feeding 16-byte pieces pays that a thousand
// times per block.  Off by default, because it is not free of consequences: the end of the stream may be found in a LATER call
// than the one that delivered its last byte (the caller asks again with avail_in = 0, or flushes), and input behind the end of
// the stream that arrived in earlier calls cannot be handed back (avail_in is exact only for the call that finds the end) --
// a reader of concatenated streams in small pieces must leave it off.  A call decodes at once when it flushes, brings no input,
// brings LESS than the call before (the last piece of a file), asks for a block stop, or the threshold is reached.
size_t defer_bytes() {
    // (honoured only with ZMI_TUNING set, like every other override -- an inherited environment variable must not change
    // what the ABI does for every consumer in the process)
    static const size_t v = [] { const char* e = abi_tune("ZMI_INFLATE_DEFER"); const long long n = e ? atoll(e) : 0; return n > 0 ? (size_t)n : (size_t)0; }();
    return v;
}
size_t take_limit() { return abi_limit("ZMI_ABI_TAKE", (size_t)256 << 20); }

// Flush points in what is buffered: a stream written with Z_SYNC_FLUSH / Z_FULL_FLUSH points (pigz, a logger, this library's
// own deflate(): one every 64 KiB of input) has the bytes 00 00 FF FF in front of every restart, and the decode pass needs no
// window -- zmi_inflate_split decodes the pieces side by side and checks every proposed cut (the same four bytes can be data).
// Candidates at least 8 KiB apart, the bytes BEHIND each marker; seg[0] = 0.  Empty unless the split is worth it.
bool split_enabled() {
    static const bool on = [] { const char* e = getenv("ZMI_ABI_SPLIT"); return !(e && (e[0] == '0' || e[0] == 'o')); }();
    return on;
}
void find_flush_points(const uint8_t* in, size_t n, std::vector<uint32_t>& seg) {
    seg.clear();
    // (ZMI_ABI_SPLIT_MIN / ZMI_ABI_SPLIT_GAP, bytes, under ZMI_TUNING: the tests split small streams)
    static const size_t kMin = [] { const char* e = abi_tune("ZMI_ABI_SPLIT_MIN"); return e && atol(e) > 0 ? (size_t)atol(e) : (size_t)256 << 10; }();
    static const size_t kGap = [] { const char* e = abi_tune("ZMI_ABI_SPLIT_GAP"); return e && atol(e) > 0 ? (size_t)atol(e) : (size_t)8192; }();
    if (n < kMin || !split_enabled()) return;
    // 00 00 FF FF, found through its third byte: memchr runs at memory speed and 0xFF is every 256th byte of compressed data
    // (memmem with a four-byte needle: 2.9 GB/s, 1.5 ms per 4 MiB piece -- an eighth of the call)
    seg.push_back(0u);
    size_t at = kGap;   // where the next marker may start
    while (at + 4 < n && seg.size() < 8192u) {
        const uint8_t* p = (const uint8_t*)memchr(in + at + 2, 0xFF, n - (at + 2) - 1);   // candidate for the marker's third byte
        if (!p) break;
        const size_t i = (size_t)(p - in);   // i >= at + 2, i + 1 < n
        if (in[i + 1] == 0xFF && in[i - 1] == 0x00 && in[i - 2] == 0x00) {
            const size_t cut = i + 2;
            if (cut >= n) break;
            seg.push_back((uint32_t)cut);
            at = cut + kGap;
        } else at = i - 1;   // the next candidate third byte is at i + 1 at the earliest
    }
    if (seg.size() < 4u) seg.clear();
}

// n decoded bytes (n <= avail_out) go to the caller: the copy, the wrapper's running check, the window inflateGetDictionary shows
void inf_deliver(z_streamp strm, InflateState* s, const uint8_t* p, size_t n) {
    memcpy(strm->next_out, p, n);
    if (s->verify && s->form == 1) s->check = host_adler32(s->check, p, n);
    else if (s->verify && s->form == 2) s->check = host_crc32(s->check, p, n);
    if (n >= 32768u) s->window.assign(p + (n - 32768u), p + n);
    else {
        s->window.insert(s->window.end(), p, p + n);
        if (s->window.size() > 32768u) s->window.erase(s->window.begin(), s->window.end() - 32768);
    }
    strm->next_out += n;
    strm->avail_out -= (uInt)n;
    strm->total_out += n;
    if (s->form > 0) strm->adler = s->check;
}

// Decode what is buffered, from the checkpoint.  Every new byte goes to the caller (`strm`, while nothing is queued in front of it
// and the caller has room: one pass over the bytes instead of two -- a 4 MiB piece of a stream is 9 MiB of output) or into the queue,
// the checkpoint moves to the last block boundary reached, and the mode changes when the final block ended or the data is invalid.
int inflate_attempt(InflateState* s, int stop_mode = 0, z_streamp strm = nullptr) {   // stop_mode 1: decode one block, 2: only the next block header
    AbiLease lease;
    zmi_ctx* c = lease.ctx;
    if (!c) return Z_MEM_ERROR;
    const size_t kTake = take_limit(), kQueueLimit = queue_limit();
    size_t take = s->in.size() < kTake ? s->in.size() : kTake;
    size_t cap = take * 4 + 65536;
    if (cap > ((size_t)64 << 20)) cap = (size_t)64 << 20;
    for (;;) {
        if (s->tmp.size() < cap) s->tmp.resize(cap);
        uint32_t olen = 0, used = 0, res[4] = {0, 0, 0, 0};
        int32_t st = 0, det = 0;
        const uint32_t in_bit = s->sbit | (stop_mode == 1 ? (1u << 8) : 0u) | (stop_mode == 2 ? (1u << 24) : 0u);
        std::vector<uint32_t> seg;
        if (stop_mode == 0 && take <= ((size_t)16 << 20)) find_flush_points(s->in.data(), take, seg);
        int rc;
        (void)zmi_ctx_reset_codes_used(c);   // (contexts are leased per call: the figure belongs to this stream's attempt)
        if (!seg.empty()) {
            uint32_t used_seg = 0;
            rc = zmi_inflate_split(c, s->in.data(), (uint32_t)take, in_bit, s->hist.data(), (uint32_t)s->hist.size(), s->tmp.data(), (uint32_t)cap,
                                   seg.data(), (uint32_t)seg.size(), &olen, &st, &det, &used, res, &used_seg);
            static const bool trace = abi_tune("ZMI_ABI_TRACE") != nullptr;
            if (trace) fprintf(stderr, "[zmi abi] inflate: %zu bytes buffered, %zu cuts proposed, %u pieces decoded side by side\n", take, seg.size(), used_seg);
        }
        else if (stop_mode == 0 && split_enabled()) {
            // no flush points (an ordinary compressor's stream): the device looks for the block headers itself and decodes the
            // blocks side by side (zmi_inflate_blocks; below 256 KiB, or with nothing found, this IS zmi_inflate_resume)
            uint32_t used_seg = 0;
            rc = zmi_inflate_blocks(c, s->in.data(), (uint32_t)take, in_bit, s->hist.data(), (uint32_t)s->hist.size(), s->tmp.data(), (uint32_t)cap,
                                    &olen, &st, &det, &used, res, &used_seg);
            static const bool trace = abi_tune("ZMI_ABI_TRACE") != nullptr;
            if (trace) fprintf(stderr, "[zmi abi] inflate: %zu bytes buffered, no flush points, %u blocks decoded side by side\n", take, used_seg);
        }
        else
            rc = zmi_inflate_resume(c, s->in.data(), (uint32_t)take, in_bit, s->hist.data(), (uint32_t)s->hist.size(), s->tmp.data(),
                                    (uint32_t)cap, &olen, &st, &det, &used, res);
        if (rc != 0) return Z_MEM_ERROR;
        if (st == Z_MEM_ERROR) return Z_MEM_ERROR;
        {
            uint32_t cu = 0;
            if (zmi_ctx_last_codes_used(c, &cu) == 0 && cu != 0u) s->codes_used = cu;
        }
        const size_t eff = olen < cap ? olen : cap;
        if (st == Z_BUF_ERROR && det == 3 && (res[3] & 2u)) {   // behind the block header: nothing decoded, the checkpoint stays
            s->hdr_seen = true;
            s->hdr_last = ((res[3] >> 2) & 1u) != 0u;
            s->stop_state = 2;
            s->stop_reported = false;
            s->stop_bits = res[1] ? 8u - res[1] : 0u;
            s->stop = (size_t)res[0] + (res[1] ? 1u : 0u);   // bytes of `in` the reference has consumed by now
            s->tried = (size_t)-1;
            return Z_OK;
        }
        if (st == Z_BUF_ERROR && det == 2 && res[2] == 0) {   // one block that is larger than the room: more room
            if (cap >= 0xE0000000ull) return Z_MEM_ERROR;
            cap *= 2;
            continue;
        }
        if (eff > s->pend) {
            const uint8_t* fresh = s->tmp.data() + s->pend;
            size_t nfresh = eff - s->pend;
            if (strm && strm->avail_out && s->out_pos >= s->out.size()) {
                const size_t direct = nfresh < strm->avail_out ? nfresh : strm->avail_out;
                inf_deliver(strm, s, fresh, direct);
                fresh += direct;
                nfresh -= direct;
            }
            if (nfresh) s->out.insert(s->out.end(), fresh, fresh + nfresh);
            s->total += eff - s->pend;
        }
        if (st == Z_OK) {   // the final block ended `used` bytes in
            if (stop_mode) { s->stop_state = 1; s->stop_reported = false; s->stop_bits = res[1] ? 8u - res[1] : 0u; s->hdr_seen = false; }
            s->in.erase(s->in.begin(), s->in.begin() + (used < s->in.size() ? used : s->in.size()));
            s->sbit = 0;
            s->primed = 0;
            s->pend = 0;
            s->stop = 0;
            s->last_block = 1;
            inf_keep_hist(s, s->tmp.data(), eff);
            s->mode = IM_TRAILER;
            break;
        }
        if (st != Z_BUF_ERROR) {
            // the bytes decoded in front of the error are valid output: they are handed out first, as the reference
            // does, and the error is reported once they are gone
            // the device reports the cause in `detail` (16 + k, inflate.hip DE_*): the reference's messages (inflate.rs State::bad)
            static const char* const kCause[] = {"invalid or corrupt deflate stream", "invalid stored block lengths", "invalid block type",
                                                 "too many length or distance symbols", "invalid code lengths set", "invalid bit length repeat",
                                                 "invalid code -- missing end-of-block", "invalid literal/lengths set", "invalid distances set",
                                                 "invalid distance too far back", "incorrect header check", "invalid literal/length or distance code"};
            const int k = det >= 16 && det < 16 + (int)(sizeof(kCause) / sizeof(kCause[0])) ? det - 16 : 0;
            inf_bad(s, kCause[k]);
            break;
        }
        // more input or more room needed: everything in front of the checkpoint is settled
        // (an attempt on a shorter slice of the input may decode less than an earlier one has already queued)
        inf_keep_hist(s, s->tmp.data(), res[2]);
        s->pend = (eff > s->pend ? eff : s->pend) - res[2];
        s->in.erase(s->in.begin(), s->in.begin() + res[0]);
        if (res[0] || res[1] != s->sbit) s->primed = 0;
        s->sbit = res[1];
        s->stop = used > res[0] ? used - res[0] : 0;
        if (take > res[0]) take -= res[0]; else take = 0;
        if (det == 3) {   // the block boundary the caller asked for
            s->stop_state = 1; s->stop_reported = false; s->stop_bits = s->sbit ? 8u - s->sbit : 0u; s->hdr_seen = false;
            s->stop = s->sbit ? 1u : 0u;
            s->tried = (size_t)-1;
            return Z_OK;
        }
        const bool more_buffered = take < s->in.size();
        if (det == 2 || more_buffered) {
            if (s->out.size() - s->out_pos > kQueueLimit) { s->tried = (size_t)-1; return Z_OK; }   // let the caller drain first
            if (more_buffered) {
                if (res[0] == 0 && det != 2) {   // no block boundary inside what was given: give more
                    if (take >= 0xF0000000ull || take == s->in.size()) break;
                    take = s->in.size() < take * 2 ? s->in.size() : take * 2;
                } else take = s->in.size() < kTake ? s->in.size() : kTake;
            }
            continue;
        }
        break;
    }
    s->tried = s->in.size();
    return Z_OK;
}

// header / blocks / trailer.  Returns Z_OK, Z_NEED_DICT or Z_MEM_ERROR.
int inflate_run(z_streamp strm, InflateState* s, int stop_mode = 0) {
    for (;;) {
        switch (s->mode) {
        case IM_HEAD: {
            if (s->wrap == ZMI_WRAP_RAW) { s->form = 0; s->mode = IM_BLOCKS; break; }
            if (s->sbit != 0) { inf_bad(s, "incorrect header check"); break; }   // primed bits in front of a wrapper
            if (s->in.size() < 2) return Z_OK;
            const bool magic = s->in[0] == 0x1F && s->in[1] == 0x8B;
            if (s->wrap == ZMI_WRAP_GZIP || (s->wrap == ZMI_WRAP_AUTO && magic)) {   // inflate.rs:1000-1030
                if (!magic) { inf_bad(s, "incorrect header check"); break; }
                const char* err = nullptr;
                const long l = gzip_header_len(s->in, s->gzhead, s->verify, &err);
                if (l < 0) { inf_bad(s, err); break; }
                if (l == 0) return Z_OK;
                s->in.erase(s->in.begin(), s->in.begin() + l);
                s->form = 2;
                s->check = 0;
                strm->adler = 0;
                s->mode = IM_BLOCKS;
                if (stop_mode) { s->stop_state = 1; s->stop_reported = false; s->stop_bits = 0; s->stop = 0; return Z_OK; }   // Mode::Type behind the header
                break;
            }
            if (s->gzhead) s->gzhead->done = -1;   // not a gzip stream (inflate.rs:1024-1028)
            const unsigned h = ((unsigned)s->in[0] << 8) | s->in[1];
            if (h % 31u) { inf_bad(s, "incorrect header check"); break; }
            if ((s->in[0] & 0x0F) != 8) { inf_bad(s, "unknown compression method"); break; }
            const int len = (s->in[0] >> 4) + 8;
            if (len > 15 || (s->wbits != 0 && len > s->wbits)) { inf_bad(s, "invalid window size"); break; }
            if (s->wbits == 0) s->hdr_wbits = len;
            s->form = 1;
            s->check = 1;
            if (s->in[1] & 0x20) {   // FDICT: the DICTID follows; wait for inflateSetDictionary (inflate.rs:1036-1062)
                if (s->in.size() < 6) return Z_OK;
                s->dictid = ((uint32_t)s->in[2] << 24) | ((uint32_t)s->in[3] << 16) | ((uint32_t)s->in[4] << 8) | s->in[5];
                s->in.erase(s->in.begin(), s->in.begin() + 6);
                s->mode = IM_DICT;
                break;
            }
            s->in.erase(s->in.begin(), s->in.begin() + 2);
            strm->adler = 1;
            s->mode = IM_BLOCKS;
            if (stop_mode) { s->stop_state = 1; s->stop_reported = false; s->stop_bits = 0; s->stop = 0; return Z_OK; }   // Mode::Type behind the header
            break;
        }
        case IM_DICT:
            strm->adler = s->dictid;
            strm->msg = kErrMsg[0];
            return Z_NEED_DICT;
        case IM_BLOCKS: {
            if (s->in.empty() || s->tried == s->in.size()) return Z_OK;
            if (s->out.size() - s->out_pos > queue_limit()) return Z_OK;
            const int rc = inflate_attempt(s, stop_mode == 2 && s->hdr_seen ? 1 : stop_mode, strm);
            if (rc != Z_OK) return rc;
            if (s->mode == IM_BLOCKS || s->stop_state) return Z_OK;   // (a stop behind the final block comes before its trailer)
            break;
        }
        case IM_TRAILER: {
            if (s->form == 0) { s->mode = IM_DONE; return Z_OK; }
            if (s->form == 1) {
                if (s->in.size() < 4) return Z_OK;
                s->want_check = ((uint32_t)s->in[0] << 24) | ((uint32_t)s->in[1] << 16) | ((uint32_t)s->in[2] << 8) | s->in[3];
                s->in.erase(s->in.begin(), s->in.begin() + 4);
                s->check_seen = true;
                s->mode = IM_DONE;
                return Z_OK;
            }
            // gzip: the CRC is judged as soon as it is there, before the length arrives (inflate.rs Mode::Check, Mode::Length)
            if (!s->check_seen) {
                if (s->in.size() < 4) return Z_OK;
                s->want_check = s->in[0] | ((uint32_t)s->in[1] << 8) | ((uint32_t)s->in[2] << 16) | ((uint32_t)s->in[3] << 24);
                s->in.erase(s->in.begin(), s->in.begin() + 4);
                s->check_seen = true;
            }
            if (s->in.size() < 4) return Z_OK;
            s->want_len = s->in[0] | ((uint32_t)s->in[1] << 8) | ((uint32_t)s->in[2] << 16) | ((uint32_t)s->in[3] << 24);
            s->in.erase(s->in.begin(), s->in.begin() + 4);
            s->mode = IM_DONE;
            return Z_OK;
        }
        default:
            return Z_OK;
        }
    }
}

// hands decoded bytes to the caller; the check value follows what has been handed out (inflate/window.rs:95-168)
size_t inflate_drain(z_streamp strm, InflateState* s) {
    size_t n = s->out.size() > s->out_pos ? s->out.size() - s->out_pos : 0;
    if (n > strm->avail_out) n = strm->avail_out;
    if (n) {
        inf_deliver(strm, s, s->out.data() + s->out_pos, n);
        s->out_pos += n;
    }
    if (s->out_pos >= s->out.size()) { s->out.clear(); s->out_pos = 0; }
    else if (s->out_pos > ((size_t)8 << 20)) { s->out.erase(s->out.begin(), s->out.begin() + s->out_pos); s->out_pos = 0; }
    return n;
}
}  // namespace

#define ZMI_ABI_TRY try {
#define ZMI_ABI_CATCH(ret) } catch (...) { return (ret); }

extern "C" {

const char* zlibVersion(void) { return ZLIB_VERSION; }
// lib.rs:2219-2270: sizes of uInt / uLong / voidpf / z_off_t in two bits each (2 -> 0, 4 -> 1, 8 -> 2, other -> 3); every
// optional-feature bit is 0, as in the reference
static constexpr uLong size_code(size_t n) { return n == 2 ? 0u : (n == 4 ? 1u : (n == 8 ? 2u : 3u)); }
uLong zlibCompileFlags(void) {
    return size_code(sizeof(uInt)) | (size_code(sizeof(uLong)) << 2) | (size_code(sizeof(void*)) << 4) | (size_code(sizeof(long)) << 6);
}
const char* zError(int err) { int i = 2 - err; return (i >= 0 && i < 10) ? kErrMsg[i] : ""; }

// ---------------------------------------------------------------- deflate
int deflateInit2_(z_streamp strm, int level, int method, int windowBits, int memLevel, int strategy, const char* version,
                  int stream_size) {
    ZMI_ABI_TRY
    if (!version_ok(version, stream_size)) return Z_VERSION_ERROR;
    if (!strm) return Z_STREAM_ERROR;
    strm->msg = nullptr;
    if (level == Z_DEFAULT_COMPRESSION) level = 6;
    int wrap = 1;
    if (windowBits < 0) { if (windowBits < -15) return Z_STREAM_ERROR; wrap = 0; windowBits = -windowBits; }
    else if (windowBits > 15) { wrap = 2; windowBits -= 16; }
    if (memLevel < 1 || memLevel > 9 || method != Z_DEFLATED || windowBits < 8 || windowBits > 15 || level < 0 || level > 9 ||
        strategy < 0 || strategy > Z_FIXED || (windowBits == 8 && wrap != 1))
        return Z_STREAM_ERROR;  // deflate.rs:299-306
    if (windowBits == 8) windowBits = 9;
    if (!abi_device_present()) { strm->msg = "no HIP device"; return Z_MEM_ERROR; }
    DeflateState* s = alloc_state<DeflateState>(strm);
    if (!s) return Z_MEM_ERROR;
    s->level = level; s->strategy = strategy; s->wrap = wrap; s->wbits = windowBits;
    s->adler = 1; s->crc = 0;
    strm->state = (internal_state*)s;
    strm->total_in = strm->total_out = 0;
    strm->data_type = Z_UNKNOWN;
    strm->adler = wrap == 2 ? 0 : 1;
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int deflateInit_(z_streamp strm, int level, const char* version, int stream_size) {
    return deflateInit2_(strm, level, Z_DEFLATED, MAX_WBITS, 8, Z_DEFAULT_STRATEGY, version, stream_size);
}
int deflate(z_streamp strm, int flush) {
    ZMI_ABI_TRY
    DeflateState* s = dstate(strm);
    if (!s || flush < 0 || flush > Z_BLOCK) return Z_STREAM_ERROR;
    if (!strm->next_out || (strm->avail_in != 0 && !strm->next_in)) { strm->msg = kErrMsg[4]; return Z_STREAM_ERROR; }
    if (s->finished && s->pending.empty() && flush != Z_FINISH) { strm->msg = kErrMsg[4]; return Z_STREAM_ERROR; }
    if (strm->avail_out == 0) { strm->msg = kErrMsg[7]; return Z_BUF_ERROR; }
    const uInt in0 = strm->avail_in, out0 = strm->avail_out;
    if (s->finished && strm->avail_in != 0) { strm->msg = kErrMsg[7]; return Z_BUF_ERROR; }
    // a call that brings a launch's worth (or flushes / finishes) with nothing buffered in front of it is compressed straight from
    // the caller's buffer (compress_buffered)
    const uint8_t* direct = nullptr;
    size_t direct_n = 0;
    if (strm->avail_in) {
        if (s->in.empty() && !s->finished && (flush != Z_NO_FLUSH || strm->avail_in >= emit_bytes())) { direct = strm->next_in; direct_n = strm->avail_in; }
        else s->in.insert(s->in.end(), strm->next_in, strm->next_in + strm->avail_in);
        strm->next_in += strm->avail_in;
        strm->total_in += strm->avail_in;
        strm->avail_in = 0;
    }
    int rc = Z_OK;
    // the first call writes the wrapper's header even without input, and from then on the stream counts as started
    // (deflate.rs:2543-2627: Status::Init -> Busy; deflateEnd then reports Z_DATA_ERROR, deflate.rs:728-743)
    if (!s->header_done && !s->finished) put_header(s);
    const int old_flush = s->last_flush;
    if (!s->finished) {
        if (flush == Z_FINISH) rc = compress_buffered(s, true, false, direct, direct_n);
        else if (flush != Z_NO_FLUSH) { if (!s->in.empty() || s->last_flush != flush || in0) rc = compress_buffered(s, false, flush == Z_FULL_FLUSH, direct, direct_n); }
        else if (direct || s->in.size() >= emit_bytes()) rc = compress_buffered(s, false, false, direct, direct_n);
        if (rc != Z_OK) { strm->msg = zError(rc); return rc; }
    }
    s->last_flush = flush;
    drain(strm, s->pending, s->pending_pos);
    strm->adler = s->wrap == 2 ? s->crc : s->adler;
    if (s->finished && s->pending.empty()) return Z_STREAM_END;
    // a call that had nothing to do is an error only if it asks for no more than the call before it (deflate.rs:2526-2533:
    // rank_flush(flush) <= rank_flush(old_flush); a fresh stream starts at -2, so its first empty call is Z_OK)
    auto rank = [](int f) { return f * 2 - (f > 4 ? 9 : 0); };
    if (in0 == 0 && out0 == strm->avail_out && flush != Z_FINISH && rank(flush) <= rank(old_flush)) { strm->msg = kErrMsg[7]; return Z_BUF_ERROR; }
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int deflateEnd(z_streamp strm) {
    DeflateState* s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    bool busy = !s->finished && (s->header_done || !s->in.empty());  // deflate.rs:728-743: freed, but reports the loss
    free_state(strm, s);
    strm->state = nullptr;
    return busy ? Z_DATA_ERROR : Z_OK;
}
int deflateReset(z_streamp strm) {
    DeflateState* s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    int level = s->level, strategy = s->strategy, wrap = s->wrap, wbits = s->wbits;
    const ZCalls zc = s->zc;
    s->~DeflateState();
    new (s) DeflateState(zc);
    s->level = level; s->strategy = strategy; s->wrap = wrap; s->wbits = wbits;
    strm->total_in = strm->total_out = 0;
    strm->msg = nullptr;
    strm->data_type = Z_UNKNOWN;
    strm->adler = wrap == 2 ? 0 : 1;
    return Z_OK;
}
int deflateParams(z_streamp strm, int level, int strategy) {
    ZMI_ABI_TRY
    DeflateState* s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (level == Z_DEFAULT_COMPRESSION) level = 6;
    if (level < 0 || level > 9 || strategy < 0 || strategy > Z_FIXED) return Z_STREAM_ERROR;
    if ((level != s->level || strategy != s->strategy) && !s->in.empty()) {
        int rc = compress_buffered(s, false);  // data supplied so far keeps the old parameters
        if (rc != Z_OK) return rc;
        drain(strm, s->pending, s->pending_pos);
        if (!s->pending.empty()) { s->level = level; s->strategy = strategy; return Z_BUF_ERROR; }
    }
    s->level = level; s->strategy = strategy;
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int deflateTune(z_streamp strm, int, int, int, int) { return dstate(strm) ? Z_OK : Z_STREAM_ERROR; }
z_size_t deflateBound_z(z_streamp strm, z_size_t sourceLen) {
    // bound (deflate.rs:3193-3287): the wrapper's length follows the stream (DICTID, the gzip header fields handed in with
    // deflateSetHeader), a window other than 32 KiB gets the conservative formula, the default configuration
    // compress_bound_help (deflate.rs:2975-2991).  This engine's own overhead on incompressible input -- one call's input is cut
    // into segments of 32 KiB (64 KiB in calls of 8 MiB and more) and encoder pieces of 8 KiB, every piece ends with the 5-byte
    // marker, a stored block (5 bytes of header) holds at least 4096 bytes: some 10 bytes per 4 KiB, 0.25 % -- stays far inside the
    // (n + 7) / 8 term, and inside the tightest branch too, `n + n / 32 + ...` (3.9 %) for level 0 with a window below 32 KiB
    // (tests/zlib_abi_harness.py misc_symbol_checks: random configurations, and levels 0 / 1 x small windows x incompressible input).
    const z_size_t n = sourceLen;
    const z_size_t comp_len = n + ((n + 7) >> 3) + ((n + 63) >> 6) + 5;
    DeflateState* s = dstate(strm);
    if (!s) return comp_len + 6;
    z_size_t wrap_len = 0;
    if (s->wrap == 1) wrap_len = 6 + ((s->dict_set || s->total_len != 0 || !s->in.empty()) ? 4 : 0);
    else if (s->wrap == 2) {
        wrap_len = 18;
        if (const gz_header* h = s->gzhead) {
            if (h->extra) wrap_len += 2 + h->extra_len;
            if (h->name) wrap_len += strlen((const char*)h->name) + 1;
            if (h->comment) wrap_len += strlen((const char*)h->comment) + 1;
            if (h->hcrc) wrap_len += 2;
        }
    }
    if (s->wbits != 15) {
        if (s->level == 0) return n + (n >> 5) + (n >> 7) + (n >> 11) + 7 + wrap_len;
        return comp_len + wrap_len;
    }
    return n + (n == 0) + (n < 9) + ((n + 7) >> 3) + 3 + wrap_len;
}
uLong deflateBound(z_streamp strm, uLong sourceLen) { return (uLong)deflateBound_z(strm, sourceLen); }
int deflatePending(z_streamp strm, unsigned* pending, int* bits) {
    DeflateState* s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (pending) *pending = (unsigned)(s->pending.size() - s->pending_pos);
    if (bits) *bits = s->prime_bits;
    return Z_OK;
}
int deflateSetDictionary(z_streamp strm, const Bytef* dictionary, uInt dictLength) {
    // deflate.rs:499-564: not for gzip streams, for zlib streams only before the first deflate() call, never with
    // input pending; the last 32 KiB of the dictionary become the window in front of the data
    ZMI_ABI_TRY
    DeflateState* s = dstate(strm);
    if (!s || !dictionary) return Z_STREAM_ERROR;
    if (s->wrap == 2 || (s->wrap == 1 && s->header_done) || !s->in.empty() || s->finished) return Z_STREAM_ERROR;
    if (s->wrap == 1) {
        s->dictid = host_adler32(1, dictionary, dictLength);
        s->dict_set = true;
        strm->adler = s->dictid;
    }
    const uInt keep = dictLength > 32768u ? 32768u : dictLength;
    s->hist.assign(dictionary + (dictLength - keep), dictionary + dictLength);
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int deflatePrime(z_streamp strm, int bits, int value) {
    // deflate.rs:566-606: whole bytes go to the pending output at once (in front of a header that is not written yet,
    // as in the reference); a rest below 8 bits waits for the next compressed data (compress_buffered)
    ZMI_ABI_TRY
    DeflateState* s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (bits < 0 || bits > 32) return Z_BUF_ERROR;
    uint64_t acc = (uint64_t)s->prime_val | (((uint64_t)(uint32_t)value & ((1ull << bits) - 1ull)) << s->prime_bits);
    int total = s->prime_bits + bits;
    while (total >= 8) { s->pending.push_back((uint8_t)acc); acc >>= 8; total -= 8; }
    s->prime_val = (uint32_t)acc;
    s->prime_bits = total;
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int deflateUsed(z_streamp strm, int* bits) {
    // deflate.rs:129 bits_used: bits of the last byte in use when the output last went to a byte boundary (1..8, 0 before
    // any flush).  The encoder kernels do not report bit positions; the most recent segment is decoded on the device to
    // the block that ends it (zmi_inflate_resume's checkpoint), only when somebody asks.
    ZMI_ABI_TRY
    DeflateState* s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (s->used_bits < 0) {
        AbiLease lease;
        zmi_ctx* c = lease.ctx;
        if (!c) return Z_MEM_ERROR;
        // a flush ends with the empty stored block 00 00 FF FF: without those four bytes the decode stops at its header
        size_t n = s->last_seg.size();
        if (!s->last_seg_final) n = n >= 4 ? n - 4 : 0;
        const std::vector<uint8_t> zeros(32768, 0);   // any history will do: only positions are wanted
        std::vector<uint8_t> tmp;
        size_t base = 0, cap = (size_t)4 << 20;
        uint32_t bit = 0;
        for (;;) {
            tmp.resize(cap);
            uint32_t olen = 0, used = 0, res[4] = {0, 0, 0, 0};
            int32_t st = 0, det = 0;
            if (zmi_inflate_resume(c, s->last_seg.data() + base, (uint32_t)(n - base), bit, zeros.data(), 32768u, tmp.data(), (uint32_t)cap,
                                   &olen, &st, &det, &used, res) != 0)
                return Z_MEM_ERROR;
            if (st == Z_BUF_ERROR && det == 2) {   // out of room: go on from the checkpoint (or with more room)
                if (res[2] == 0) { if (cap >= 0xE0000000ull) return Z_MEM_ERROR; cap *= 2; }
                base += res[0]; bit = res[1];
                continue;
            }
            if (st == Z_OK) s->used_bits = res[1] ? (int)res[1] : 8;
            else if (st == Z_BUF_ERROR) s->used_bits = (int)((res[1] + 2u) & 7u) + 1;
            else return Z_STREAM_ERROR;
            break;
        }
    }
    if (bits) *bits = s->used_bits;
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int deflateGetDictionary(z_streamp strm, Bytef* dictionary, uInt* dictLength) {   // deflate.rs: the current window
    DeflateState* s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    std::vector<uint8_t> w(s->hist.begin(), s->hist.end());
    w.insert(w.end(), s->in.begin(), s->in.end());
    if (w.size() > 32768u) w.erase(w.begin(), w.end() - 32768);
    if (dictionary && !w.empty()) memcpy(dictionary, w.data(), w.size());
    if (dictLength) *dictLength = (uInt)w.size();
    return Z_OK;
}
int deflateSetHeader(z_streamp strm, gz_headerp head) {   // deflate.rs:3145-3155
    DeflateState* s = dstate(strm);
    if (!s || s->wrap != 2) return Z_STREAM_ERROR;
    s->gzhead = head;
    return Z_OK;
}
int deflateCopy(z_streamp dest, z_streamp source) {   // deflate.rs: a deep copy of the stream state
    ZMI_ABI_TRY
    DeflateState* s = dstate(source);
    if (!s || !dest) return Z_STREAM_ERROR;
    *dest = *source;
    DeflateState* d = alloc_state<DeflateState>(dest);
    if (!d) { dest->state = nullptr; return Z_MEM_ERROR; }
    { const ZCalls own = d->zc; *d = *s; d->zc = own; }   // (the buffers are copied into dest's allocator, which stays dest's)
    dest->state = (internal_state*)d;
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int deflateResetKeep(z_streamp strm) { return deflateReset(strm); }   // no window allocation to keep here

z_size_t compressBound_z(z_size_t n) { return n + (n == 0) + (n < 9) + ((n + 7) >> 3) + 3 + 6; }
uLong compressBound(uLong n) { return (uLong)compressBound_z(n); }
int compress2_z(Bytef* dest, z_size_t* destLen, const Bytef* source, z_size_t sourceLen, int level) {
    ZMI_ABI_TRY
    if (!dest || !destLen || (!source && sourceLen)) return Z_STREAM_ERROR;
    if (level == Z_DEFAULT_COMPRESSION) level = 6;
    if (level < 0 || level > 9) return Z_STREAM_ERROR;
    DeflateState s{ZCalls{}};   // (compress2 has no z_stream: malloc / free, as the reference's own)
    s.level = level;
    s.in.assign(source, source + sourceLen);
    int rc = compress_buffered(&s, true);
    if (rc != Z_OK) return rc;
    if (s.pending.size() > *destLen) return Z_BUF_ERROR;
    memcpy(dest, s.pending.data(), s.pending.size());
    *destLen = s.pending.size();
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int compress_z(Bytef* dest, z_size_t* destLen, const Bytef* source, z_size_t sourceLen) { return compress2_z(dest, destLen, source, sourceLen, 6); }
int compress2(Bytef* dest, uLongf* destLen, const Bytef* source, uLong sourceLen, int level) {
    if (!destLen) return Z_STREAM_ERROR;
    z_size_t dl = *destLen;
    int rc = compress2_z(dest, &dl, source, sourceLen, level);
    *destLen = (uLongf)dl;
    return rc;
}
int compress(Bytef* dest, uLongf* destLen, const Bytef* source, uLong sourceLen) { return compress2(dest, destLen, source, sourceLen, 6); }

// ---------------------------------------------------------------- inflate
namespace {
static int parse_window_bits(int windowBits, int* wrap, int* wb) {   // inflate.rs:2298-2327
    int w = windowBits;
    if (w < 0) { if (w < -15) return Z_STREAM_ERROR; *wrap = ZMI_WRAP_RAW; w = -w; }
    else if (w >= 32) { *wrap = ZMI_WRAP_AUTO; w -= 32; }
    else if (w >= 16) { *wrap = ZMI_WRAP_GZIP; w -= 16; }
    else *wrap = ZMI_WRAP_ZLIB;
    if (w != 0 && (w < 8 || w > 15)) return Z_STREAM_ERROR;
    *wb = w;
    return Z_OK;
}
static void inflate_reset_state(z_streamp strm, InflateState* s) {
    const int wrap = s->wrap, wb = s->wbits;
    uint8_t* bw = s->back_window;
    const ZCalls zc = s->zc;
    s->~InflateState();
    new (s) InflateState(zc);
    s->wrap = wrap; s->wbits = wb; s->back_window = bw;
    strm->total_in = strm->total_out = 0;
    strm->msg = nullptr;
    strm->adler = 1;
    strm->data_type = 0;
}
}  // namespace

int inflateInit2_(z_streamp strm, int windowBits, const char* version, int stream_size) {
    ZMI_ABI_TRY
    if (!version_ok(version, stream_size)) return Z_VERSION_ERROR;
    if (!strm) return Z_STREAM_ERROR;
    strm->msg = nullptr;
    int wrap = 0, wb = 0;
    if (parse_window_bits(windowBits, &wrap, &wb) != Z_OK) return Z_STREAM_ERROR;
    if (!abi_device_present()) { strm->msg = "no HIP device"; return Z_MEM_ERROR; }
    InflateState* s = alloc_state<InflateState>(strm);
    if (!s) return Z_MEM_ERROR;
    s->wrap = wrap; s->wbits = wb;
    strm->state = (internal_state*)s;
    strm->total_in = strm->total_out = 0;
    strm->adler = 1;
    strm->data_type = 0;
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int inflateInit_(z_streamp strm, const char* version, int stream_size) { return inflateInit2_(strm, MAX_WBITS, version, stream_size); }
int inflate(z_streamp strm, int flush) {
    // Every call takes all of avail_in and decodes as far as the buffered input allows ("provides as much output as
    // possible", lib.rs:637).  Z_BLOCK stops at the next block boundary (also right behind a zlib / gzip header), Z_TREES
    // also behind the next block header; data_type reports the stop (inflate.rs:1276-1284,1323,1369,1772,1856-1873).
    ZMI_ABI_TRY
    InflateState* s = istate(strm);
    if (!s || s->back_window || !strm->next_out || (strm->avail_in != 0 && !strm->next_in)) return Z_STREAM_ERROR;
    if (s->bias) { strm->total_in += (uLong)s->bias; s->bias = 0; }
    const int stop_mode = flush == Z_BLOCK ? 1 : (flush == Z_TREES ? 2 : 0);
    // a stop that an earlier call has reported is left behind (the reference: Mode::Type -> TypeDo on entry); one that was
    // reached while the caller's buffer was full is reported by the call that drains the queue
    if (s->stop_state && (s->stop_reported || !stop_mode)) { s->stop_state = 0; s->stop_reported = false; }
    if (!stop_mode) s->hdr_seen = false;
    const uInt in0 = strm->avail_in, out0 = strm->avail_out;
    uInt taken = 0;
    // Bytes can only be handed back to the caller in the call that took them (next_in still points behind them).  So no
    // call may return with bytes of its take buffered that the decoder has not read yet: whenever decoding stops before
    // the buffered input is used up (Z_NEED_DICT; the pause while much output is queued), the unread tail of this call's
    // take goes back -- the end of the stream is then always found in the call that delivered the bytes behind it, and
    // those are returned exactly (the reference consumes only what it decodes, inflate.rs:2376-2457).
    auto hand_back = [&](size_t keep) {
        if (s->in.size() <= keep || taken == 0) return;
        size_t n = s->in.size() - keep;
        if (n > taken) n = taken;
        strm->next_in -= n; strm->total_in -= (uLong)n; strm->avail_in += (uInt)n;
        s->in.resize(s->in.size() - n);
        taken -= (uInt)n;
        s->tried = (size_t)-1;
    };
    // the caller's input is taken a piece at a time, and only while the decoder can use it: what a pause hands back (and
    // the next call copies again) stays bounded, however much the caller offers
    const size_t kAbsorb = abi_limit("ZMI_ABI_ABSORB", (size_t)16 << 20);
    bool stopped_now = false;
    const size_t kDefer = defer_bytes();
    const bool may_defer = kDefer != 0 && flush == Z_NO_FLUSH && stop_mode == 0 && in0 != 0 && in0 >= s->prev_in0;
    bool deferred = false;
    s->prev_in0 = in0;
    for (;;) {
        if (s->stop_state) { inflate_drain(strm, s); break; }   // a pending stop: only the queue is handed out
        const bool paused = s->mode == IM_BLOCKS && s->out.size() - s->out_pos > queue_limit();
        const bool wants_input = s->mode != IM_BLOCKS || s->in.empty() || s->tried == s->in.size() || may_defer;
        if (s->mode != IM_DONE && s->mode != IM_BAD && strm->avail_in && !paused && wants_input) {
            const uInt n = strm->avail_in < kAbsorb ? strm->avail_in : (uInt)kAbsorb;
            s->in.insert(s->in.end(), strm->next_in, strm->next_in + n);
            strm->next_in += n; strm->total_in += n; strm->avail_in -= n;
            taken += n;
            s->in_sync = false;
        }
        const int was = s->mode;
        if (may_defer && s->mode == IM_BLOCKS && s->in.size() - (s->tried == (size_t)-1 || s->tried > s->in.size() ? 0 : s->tried) < kDefer &&
            s->out_pos >= s->out.size()) {
            deferred = true;   // taken, not decoded yet (ZMI_INFLATE_DEFER)
            break;
        }
        const int rc = inflate_run(strm, s, stop_mode);
        if (rc == Z_NEED_DICT) { hand_back(0); return Z_NEED_DICT; }
        if (rc != Z_OK) { strm->msg = zError(rc); return rc; }
        if (s->stop_state) {
            // the reference has consumed the input up to the byte that holds the last bit in front of the stop: the rest of
            // this call's take goes back (s->stop = bytes of `in` up to there)
            hand_back(s->stop);
            stopped_now = true;
            inflate_drain(strm, s);
            break;
        }
        if (s->mode == IM_DONE && was != IM_DONE && !s->in.empty()) {
            hand_back(0);   // bytes behind the end of the stream belong to the caller
            s->in.clear();
        }
        inflate_drain(strm, s);
        if (strm->avail_out == 0 || s->mode == IM_DONE || s->mode == IM_BAD) break;
        // the caller still has room: decoding that paused on queued output goes on (the queue is empty now), and a
        // decoder that has used up what is buffered gets the next piece
        const bool undecoded = s->mode == IM_BLOCKS && !s->in.empty() && s->tried != s->in.size();
        if (!undecoded && strm->avail_in == 0) break;
    }
    (void)stopped_now;
    // paused (output queued beyond the limit, or the caller's buffer is full) with input the decoder has not reached:
    // `stop` is how far it read (its bit reader runs a few bytes ahead, hence the margin)
    if (!deferred && !s->stop_state && s->mode == IM_BLOCKS && s->tried != s->in.size()) hand_back(s->stop > 64 ? s->stop - 64 : 0);
    const bool drained = s->out_pos >= s->out.size();
    // data_type as the reference reports it (inflate.rs:1856-1873): unused bits of the last byte, +64 in the last
    // block, +128 right behind a block (or a wrapper header), +256 behind a block header
    if (s->stop_state && drained) {
        s->stop_reported = true;
        const bool lastb = s->stop_state == 2 ? s->hdr_last : s->last_block != 0;
        strm->data_type = (int)(s->stop_bits + (lastb ? 64u : 0u) + (s->stop_state == 1 ? 128u : 256u));
    } else {
        const bool at_boundary = !s->stop_state && s->mode == IM_BLOCKS && s->pend == 0 && s->in.size() <= (s->sbit ? 1u : 0u) && s->form >= 0;
        strm->data_type = (int)((s->sbit && s->mode == IM_BLOCKS ? 8u - s->sbit : 0u) + (s->last_block ? 64 : 0) + (at_boundary ? 128 : 0));
    }
    if (s->stop_state) {   // (the trailer behind a final block is looked at by the next call)
        // (no progress is Z_BUF_ERROR also at a stop, as in the reference: inflate.rs:2450-2456 -- e.g. the boundary behind
        // an empty stored block whose header the call before has reported)
        if (in0 == strm->avail_in && out0 == strm->avail_out) { strm->msg = kErrMsg[7]; return Z_BUF_ERROR; }
        return Z_OK;
    }
    if (s->mode == IM_TRAILER && s->check_seen && drained && s->verify && s->check != s->want_check) {
        inf_bad(s, "incorrect data check");   // a gzip trailer that ends behind a wrong CRC is an error already
        strm->msg = s->errmsg;
        return Z_DATA_ERROR;
    }
    if (s->mode == IM_DONE && drained) {
        if (s->verify && s->form > 0 && s->check != s->want_check) { inf_bad(s, "incorrect data check"); strm->msg = s->errmsg; return Z_DATA_ERROR; }
        if (s->verify && s->form == 2 && (uint32_t)s->total != s->want_len) { inf_bad(s, "incorrect length check"); strm->msg = s->errmsg; return Z_DATA_ERROR; }
        return Z_STREAM_END;
    }
    if (s->mode == IM_BAD && drained) { strm->msg = s->errmsg; return s->error; }
    if (in0 == strm->avail_in && out0 == strm->avail_out) { strm->msg = kErrMsg[7]; return Z_BUF_ERROR; }
    if (flush == Z_FINISH) { strm->msg = kErrMsg[7]; return Z_BUF_ERROR; }   // inflate.rs:2450-2456
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int inflateEnd(z_streamp strm) {
    InflateState* s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    free_state(strm, s);
    strm->state = nullptr;
    return Z_OK;
}
int inflateReset(z_streamp strm) {
    InflateState* s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    inflate_reset_state(strm, s);
    return Z_OK;
}
int inflateReset2(z_streamp strm, int windowBits) {
    InflateState* s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    int wrap = 0, wb = 0;
    if (parse_window_bits(windowBits, &wrap, &wb) != Z_OK) return Z_STREAM_ERROR;
    s->wrap = wrap; s->wbits = wb;
    return inflateReset(strm);
}
int inflateResetKeep(z_streamp strm) { return inflateReset(strm); }   // no window allocation to keep here
int inflateSetDictionary(z_streamp strm, const Bytef* dictionary, uInt dictLength) {
    // inflate.rs:2492-2536: for a wrapped stream only after inflate() returned Z_NEED_DICT, and the dictionary must
    // have the announced Adler-32; for a raw stream before any output
    ZMI_ABI_TRY
    InflateState* s = istate(strm);
    if (!s || !dictionary) return Z_STREAM_ERROR;
    if (s->wrap != ZMI_WRAP_RAW && s->mode != IM_DICT) return Z_STREAM_ERROR;
    if (s->wrap == ZMI_WRAP_RAW && (s->total != 0 || s->mode == IM_DONE || s->mode == IM_BAD)) return Z_STREAM_ERROR;
    if (s->mode == IM_DICT && host_adler32(1, dictionary, dictLength) != s->dictid) return Z_DATA_ERROR;
    const uInt keep = dictLength > 32768u ? 32768u : dictLength;
    s->hist.assign(dictionary + (dictLength - keep), dictionary + dictLength);
    s->dict = s->hist;
    s->have_dict = true;
    if (s->mode == IM_DICT) { s->mode = IM_BLOCKS; strm->adler = 1; }
    s->tried = (size_t)-1;   // decode what is buffered with the dictionary in place
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int inflateGetDictionary(z_streamp strm, Bytef* dictionary, uInt* dictLength) {   // inflate.rs: the sliding window so far
    ZMI_ABI_TRY
    InflateState* s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    // what the stream has produced up to the caller's read position, preceded by the preset dictionary
    std::vector<uint8_t> w(s->dict.begin(), s->dict.end());
    w.insert(w.end(), s->window.begin(), s->window.end());
    if (w.size() > 32768u) w.erase(w.begin(), w.end() - 32768);
    if (dictionary && !w.empty()) memcpy(dictionary, w.data(), w.size());
    if (dictLength) *dictLength = (uInt)w.size();
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int inflateGetHeader(z_streamp strm, gz_headerp head) {   // inflate.rs: only for streams that may be gzip
    InflateState* s = istate(strm);
    if (!s || s->wrap == ZMI_WRAP_RAW || s->wrap == ZMI_WRAP_ZLIB) return Z_STREAM_ERROR;
    s->gzhead = head;
    if (head) head->done = 0;
    return Z_OK;
}
int inflateCopy(z_streamp dest, z_streamp source) {
    ZMI_ABI_TRY
    InflateState* s = istate(source);
    if (!s || !dest) return Z_STREAM_ERROR;
    *dest = *source;
    InflateState* d = alloc_state<InflateState>(dest);
    if (!d) { dest->state = nullptr; return Z_MEM_ERROR; }
    Bytes scratch(s->tmp.get_allocator());
    scratch.swap(s->tmp);   // the decode target of the last attempt is no state: not copied
    { const ZCalls own = d->zc; *d = *s; d->zc = own; }
    scratch.swap(s->tmp);
    dest->state = (internal_state*)d;
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int inflatePrime(z_streamp strm, int bits, int value) {
    // inflate.rs:2160-2172: bits go in front of the input still to come.  Here that position exists while nothing but
    // the rest of a partly used byte (or earlier primed bits) is buffered; with undecoded input buffered the call
    // is refused (Z_STREAM_ERROR).
    ZMI_ABI_TRY
    InflateState* s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (bits == 0) return Z_OK;
    if (bits < 0) {   // forget the bits held in front of the next input byte
        if (s->primed) { s->in.erase(s->in.begin(), s->in.begin() + (s->sbit + s->primed) / 8u); s->sbit = 0; s->primed = 0; }
        else if (s->sbit && !s->in.empty()) { s->in.erase(s->in.begin()); s->sbit = 0; }
        s->tried = (size_t)-1;
        return Z_OK;
    }
    if (s->in.size() > 4 || s->pend != 0 || s->mode == IM_DONE || s->mode == IM_BAD) return Z_STREAM_ERROR;
    const uint32_t held = (uint32_t)s->in.size() * 8u - (s->in.empty() ? 0u : s->sbit);
    if (bits > 16 || held + (uint32_t)bits > 32u) return Z_STREAM_ERROR;
    uint64_t v = 0;
    for (size_t i = 0; i < s->in.size(); ++i) v |= (uint64_t)s->in[i] << (8u * i);
    v >>= s->in.empty() ? 0u : s->sbit;
    v |= ((uint64_t)(uint32_t)value & ((1ull << bits) - 1ull)) << held;
    const uint32_t total = held + (uint32_t)bits;
    const uint32_t nbytes = (total + 7u) / 8u;
    s->sbit = nbytes * 8u - total;   // top-aligned: the input byte that comes next continues the bit sequence
    v <<= s->sbit;
    s->in.resize(nbytes);
    for (uint32_t i = 0; i < nbytes; ++i) s->in[i] = (uint8_t)(v >> (8u * i));
    s->primed = total;
    s->tried = (size_t)-1;
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int inflateSync(z_streamp strm) {
    // inflate.rs:2458-2535: skip input up to the 00 00 FF FF of a flush point; decoding restarts behind it with an
    // empty window (so only a Z_FULL_FLUSH point really works), without check value verification
    ZMI_ABI_TRY
    InflateState* s = istate(strm);
    if (!s || (strm->avail_in != 0 && !strm->next_in)) return Z_STREAM_ERROR;
    if (s->bias) { strm->total_in += (uLong)s->bias; s->bias = 0; }
    auto search = [&](const uint8_t* buf, size_t len, size_t* next) {
        int got = s->sync_have;
        size_t i = 0;
        while (i < len && got < 4) {
            if (buf[i] == (got < 2 ? 0 : 0xFF)) ++got;
            else if (buf[i] != 0) got = 0;
            else got = 4 - got;
            ++i;
        }
        s->sync_have = got;
        *next = i;
    };
    size_t from = s->stop;
    if (s->sbit && from == 0) from = 1;
    if (from > s->in.size()) from = s->in.size();
    if (strm->avail_in == 0 && (s->in_sync || s->in.size() == from)) return Z_BUF_ERROR;
    if (!s->in_sync) { s->in_sync = true; s->sync_have = 0; }
    bool found = false;
    if (!s->in.empty()) {   // first what is buffered, from where the decoder stopped
        size_t next = 0;
        search(s->in.data() + from, s->in.size() - from, &next);
        if (s->sync_have == 4) {
            found = true;
            s->in.erase(s->in.begin(), s->in.begin() + from + next);
            s->bias = s->in.size();          // these count as not yet consumed, as they are for the reference
            strm->total_in -= (uLong)s->bias;
        } else s->in.clear();
        s->sbit = 0;
        s->primed = 0;
        s->stop = 0;
    }
    if (!found) {
        size_t next = 0;
        search(strm->next_in, strm->avail_in, &next);
        strm->next_in += next; strm->avail_in -= (uInt)next; strm->total_in += (uLong)next;
        if (s->sync_have != 4) return Z_DATA_ERROR;
    }
    // restart on a new block
    if (s->form < 0) { s->wrap = ZMI_WRAP_RAW; s->form = 0; }   // no header yet: treat the rest as raw
    else s->verify = false;                                      // no point in computing a check value now
    s->mode = IM_BLOCKS;
    s->hist.clear();
    s->pend = 0;
    s->tried = (size_t)-1;
    s->error = 0; s->errmsg = nullptr;
    s->in_sync = false; s->sync_have = 0;
    s->last_block = 0;
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int inflateSyncPoint(z_streamp strm) {   // inflate.rs:2537: waiting for the LEN of a stored block with no bits held
    InflateState* s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (s->mode != IM_BLOCKS || s->in.empty() || s->pend != 0) return 0;
    const uint32_t avail = (uint32_t)(s->in.size() > 2 ? 16 : s->in.size() * 8) - s->sbit;
    if (avail < 3u) return 0;
    const uint32_t v = ((uint32_t)s->in[0] | (s->in.size() > 1 ? (uint32_t)s->in[1] << 8 : 0u)) >> s->sbit;
    if (((v >> 1) & 3u) != 0u) return 0;
    return s->in.size() == (s->sbit + 3u + 7u) / 8u;
}
long inflateMark(z_streamp strm) {   // inflate.rs:2605-2619: -1 in the upper half while not inside a code
    InflateState* s = istate(strm);
    if (!s) return -65536;
    if (!strm->next_out || (!strm->next_in && strm->avail_in != 0)) return LONG_MIN;
    return (s->pend == 0 && s->in.size() <= (s->sbit ? 1u : 0u)) ? -65536 : 0;
}
int inflateValidate(z_streamp strm, int check) {   // inflate.rs:2595
    InflateState* s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    s->verify = check != 0;
    return Z_OK;
}
int inflateUndermine(z_streamp strm, int) { return istate(strm) ? Z_OK : Z_STREAM_ERROR; }   // inflate.rs:2588
// libz-rs-sys/src/lib.rs:1252 -> zlib-rs/src/inflate.rs:2372 (state.next: entries of the code tables in use).  The tables live on
// the device (roots 9 / 8, exact-fit sub-tables -- other sizes than the reference's 10 / 9): the figure is theirs, for the most
// recent dynamic block decoded; 0 before the first one and after a reset, (unsigned long)-1 for an invalid stream as in the reference.
unsigned long inflateCodesUsed(z_streamp strm) { return istate(strm) ? (unsigned long)istate(strm)->codes_used : (unsigned long)-1; }

// ---- inflateBack: raw deflate, input pulled and output pushed through callbacks (inflate/infback.rs:17-722)
int inflateBackInit_(z_streamp strm, int windowBits, unsigned char* window, const char* version, int stream_size) {
    ZMI_ABI_TRY
    if (!version_ok(version, stream_size)) return Z_VERSION_ERROR;
    if (!strm || !window || windowBits < 8 || windowBits > 15) return Z_STREAM_ERROR;
    strm->msg = nullptr;
    if (!abi_device_present()) { strm->msg = "no HIP device"; return Z_MEM_ERROR; }
    InflateState* s = alloc_state<InflateState>(strm);
    if (!s) return Z_MEM_ERROR;
    s->wrap = ZMI_WRAP_RAW; s->wbits = windowBits; s->back_window = window;
    strm->state = (internal_state*)s;
    return Z_OK;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int inflateBack(z_streamp strm, in_func in, void* in_desc, out_func out, void* out_desc) {
    ZMI_ABI_TRY
    InflateState* s = istate(strm);
    if (!s || !s->back_window || !in || !out) return Z_STREAM_ERROR;
    const uLong tin = strm->total_in, tout = strm->total_out;
    inflate_reset_state(strm, s);
    strm->total_in = tin; strm->total_out = tout;
    s->mode = IM_BLOCKS; s->form = 0;
    const size_t wsize = (size_t)1 << s->wbits;
    const unsigned char* next = strm->next_in;
    unsigned have = next ? strm->avail_in : 0;
    size_t wpos = 0;
    int ret = Z_OK;
    if (have) { s->in.assign(next, next + have); next += have; }
    unsigned chunk = have;
    for (;;) {
        if (s->in.empty() || s->tried == s->in.size()) {   // everything buffered has been tried: pull
            have = in(in_desc, (unsigned char**)&next);
            if (have == 0) { next = nullptr; ret = Z_BUF_ERROR; break; }
            s->in.insert(s->in.end(), next, next + have);
            next += have;
            chunk = have;
        }
        const int rc = inflate_attempt(s);
        if (rc != Z_OK) { ret = rc; break; }
        bool stop = false;
        while (s->out_pos < s->out.size()) {   // push through the caller's window, one window at a time
            size_t n = s->out.size() - s->out_pos;
            if (n > wsize - wpos) n = wsize - wpos;
            memcpy(s->back_window + wpos, s->out.data() + s->out_pos, n);
            s->out_pos += n;
            wpos += n;
            if (wpos == wsize) {
                wpos = 0;
                if (out(out_desc, s->back_window, (unsigned)wsize) != 0) { ret = Z_BUF_ERROR; stop = true; break; }
            }
        }
        s->out.clear(); s->out_pos = 0;
        if (stop) break;
        if (s->mode == IM_TRAILER) { s->mode = IM_DONE; ret = Z_STREAM_END; break; }
        if (s->mode == IM_BAD) { ret = Z_DATA_ERROR; strm->msg = s->errmsg; break; }
    }
    if (wpos != 0 && out(out_desc, s->back_window, (unsigned)wpos) != 0 && ret == Z_STREAM_END) ret = Z_BUF_ERROR;
    unsigned unused = 0;
    if (ret == Z_STREAM_END || (ret == Z_BUF_ERROR && next)) {   // what lies behind the end of the stream stays with the caller
        unused = (unsigned)(s->in.size() < chunk ? s->in.size() : chunk);
        if (ret != Z_STREAM_END) unused = 0;
    }
    strm->next_in = next ? (Bytef*)(next - unused) : nullptr;
    strm->avail_in = next ? unused : 0;
    return ret;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int inflateBackEnd(z_streamp strm) {
    InflateState* s = istate(strm);
    if (!s || !s->back_window) return Z_STREAM_ERROR;
    free_state(strm, s);
    strm->state = nullptr;
    return Z_OK;
}

int uncompress2_z(Bytef* dest, z_size_t* destLen, const Bytef* source, z_size_t* sourceLen) {
    ZMI_ABI_TRY
    if (!dest || !destLen || !source || !sourceLen) return Z_STREAM_ERROR;
    const z_size_t room = *destLen;
    // A large stream: the blocks are found on the device and decoded side by side, straight into the caller's buffer
    // (zmi_inflate_blocks).  Only the plain outcome is taken from here -- a complete stream that fits, with its Adler-32 right;
    // everything else (errors, short room, FDICT) goes the way it always went, which names the reference's codes.
    static const z_size_t kBlocksMin = [] { const char* e = abi_tune("ZMI_ABI_BLOCKS_MIN"); return e && atol(e) > 0 ? (z_size_t)atol(e) : (z_size_t)256 << 10; }();
    if (*sourceLen >= kBlocksMin && *sourceLen < ((z_size_t)1 << 28) && room >= 1 && room <= ((z_size_t)1 << 30) && split_enabled()) {
        const uint32_t cmf = source[0], flg = source[1];
        if ((cmf & 0x0Fu) == 8u && (cmf >> 4) <= 7u && ((cmf << 8) | flg) % 31u == 0u && !(flg & 0x20u)) {
            AbiLease lease;
            if (lease.ctx) {
                uint32_t olen = 0, used2 = 0, res[4] = {0, 0, 0, 0}, segs = 0;
                int32_t st2 = 0, det2 = 0;
                const uint32_t n2 = (uint32_t)(*sourceLen - 2);
                if (zmi_inflate_blocks(lease.ctx, source + 2, n2, 0u, nullptr, 0u, dest, (uint32_t)room, &olen, &st2, &det2, &used2, res, &segs) == 0 &&
                    st2 == Z_OK && olen <= room && (uint64_t)used2 + 4u <= n2) {
                    const uint8_t* t = source + 2 + used2;
                    const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
                    if (host_adler32(1u, dest, olen) == want) {
                        *sourceLen = (z_size_t)used2 + 6u;
                        *destLen = olen;
                        return Z_OK;
                    }
                }
            }
        }
    }
    std::vector<uint8_t> out;
    uint32_t used = 0;
    int32_t st = 0, detail = 0;
    int rc = gpu_inflate_stream(source, *sourceLen, ZMI_WRAP_ZLIB, out, room, &used, &st, &detail);
    if (rc != Z_OK) return rc;
    *sourceLen = used;
    // whatever the outcome, *destLen reports what was written (inflate.rs:260-283: dest_len = total_out)
    const size_t got = out.size() < room ? out.size() : room;
    if (got) memcpy(dest, out.data(), got);
    *destLen = got;
    if (st == Z_OK) return Z_OK;
    // no room at all: the reference decodes into a one-byte buffer of its own only to tell a complete empty stream (Z_OK
    // above) from everything else, which is a data error (inflate.rs:202-217, :263-275; uncompress_edge_cases)
    if (st == Z_BUF_ERROR && detail == 2) return room == 0 ? Z_DATA_ERROR : Z_BUF_ERROR;
    if (st == Z_BUF_ERROR) return Z_DATA_ERROR;  // input ended inside the stream: uncompress reports a data error (inflate.rs:268-284)
    return st == Z_NEED_DICT ? Z_DATA_ERROR : st;
    ZMI_ABI_CATCH(Z_MEM_ERROR)
}
int uncompress_z(Bytef* dest, z_size_t* destLen, const Bytef* source, z_size_t sourceLen) { return uncompress2_z(dest, destLen, source, &sourceLen); }
int uncompress2(Bytef* dest, uLongf* destLen, const Bytef* source, uLong* sourceLen) {
    if (!destLen || !sourceLen) return Z_STREAM_ERROR;
    z_size_t dl = *destLen, sl = *sourceLen;
    int rc = uncompress2_z(dest, &dl, source, &sl);
    *destLen = (uLongf)dl; *sourceLen = (uLong)sl;
    return rc;
}
int uncompress(Bytef* dest, uLongf* destLen, const Bytef* source, uLong sourceLen) { return uncompress2(dest, destLen, source, &sourceLen); }

// ---------------------------------------------------------------- checksums
uLong adler32_z(uLong adler, const Bytef* buf, z_size_t len) { return buf ? host_adler32((uint32_t)adler, buf, len) : 1; }
uLong adler32(uLong adler, const Bytef* buf, uInt len) { return adler32_z(adler, buf, len); }
uLong adler32_combine64(uLong a1, uLong a2, long long len2) { return len2 < 0 ? 0xFFFFFFFFul : host_adler_combine((uint32_t)a1, (uint32_t)a2, (uint64_t)len2); }
uLong adler32_combine(uLong a1, uLong a2, long len2) { return adler32_combine64(a1, a2, len2); }
uLong crc32_z(uLong crc, const Bytef* buf, z_size_t len) { return buf ? host_crc32((uint32_t)crc, buf, len) : 0; }
uLong crc32(uLong crc, const Bytef* buf, uInt len) { return crc32_z(crc, buf, len); }
uLong crc32_combine_gen64(long long len2) { return gf2_xpow8((uint64_t)len2); }
uLong crc32_combine_gen(long len2) { return crc32_combine_gen64(len2); }
uLong crc32_combine_op(uLong crc1, uLong crc2, uLong op) { return gf2_mul((uint32_t)op, (uint32_t)crc1) ^ (uint32_t)crc2; }
uLong crc32_combine64(uLong crc1, uLong crc2, long long len2) { return crc32_combine_op(crc1, crc2, crc32_combine_gen64(len2)); }
uLong crc32_combine(uLong crc1, uLong crc2, long len2) { return crc32_combine64(crc1, crc2, len2); }
const uint32_t* get_crc_table(void) { std::call_once(g_crc_once, crc_init); return g_crc_table[0]; }

}  // extern "C"
