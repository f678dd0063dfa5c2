// exchange.hip -- the multi-GPU stitch behind the C ABI: size-table all-gather, stitch plan, point-to-point slab exchange.
//
// What the reference does here: its parallel-deflate recipe compresses the pieces independently and appends the finished byte
// strings in piece order (zlib-rs/src/deflate.rs:4145-4221; multi-member gzip is read back that way, libz-rs-sys/src/gz.rs:
// 1464-1506).  With the pieces spread round-robin over the GPUs of a node (BASELINE.json configs[4]: shard g lives on rank
// g % world) "append in order" becomes an exchange:
//   1. zmi_exchange_sizes     ncclAllGather of the u32 size tables (fixed size: 4 B per shard)
//   2. zmi_stitch_plan_dev    one small kernel: global byte offset of every shard, offsets inside every rank's slab, slab sizes
//   3. zmi_exchange_slabs(_round)  the slabs have different sizes and RCCL has no all-gather-v; xGMI is a full mesh of
//                             point-to-point links (7 x ~153 GB/s per GPU), so every rank posts one ncclSend and one ncclRecv
//                             per peer inside ONE ncclGroupStart/End: 7 concurrent transfers, each on its own link.  A ring
//                             would push all slabs through one link and is never used.
//   4. zmi_copy_ranges_dev    (pack.hip) scatters a received slab into the globally ordered output.
// RCCL is bound at run time (dlopen of librccl.so.1: in a process that already holds RCCL -- torch, a Rust host that linked
// it -- that is the loaded instance); a single-GPU user of libzmi355.so never loads it.  ZMI_RCCL_LIB names another library
// file (the tests bind a two-process mock that moves the bytes through files: tests/emu/mock_rccl.c).
#include "zmi_device.h"
#include "zmi_kernels.h"
#include "../../include/zmi355.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

extern "C" int zmi_ctx_device(const zmi_ctx* c);
extern "C" void zmi_set_last_error(const char* what);

// ---- RCCL binding (prototypes as in rccl.h; declared here so that neither this file nor the emulator build needs the header)
typedef struct { char internal[ZMI_UNIQUE_ID_BYTES]; } zx_uid;   // ncclUniqueId
typedef void* zx_comm_t;                                          // ncclComm_t
enum { ZX_UINT8 = 1, ZX_UINT32 = 3 };                             // ncclUint8, ncclUint32
struct zx_api {
    void* handle = nullptr;
    int (*GetUniqueId)(zx_uid*) = nullptr;
    int (*CommInitRank)(zx_comm_t*, int, zx_uid, int) = nullptr;
    int (*CommDestroy)(zx_comm_t) = nullptr;
    int (*CommAbort)(zx_comm_t) = nullptr;
    int (*CommCount)(zx_comm_t, int*) = nullptr;
    int (*CommUserRank)(zx_comm_t, int*) = nullptr;
    int (*CommCuDevice)(zx_comm_t, int*) = nullptr;   // optional: only zmi_comm_adopt's device check uses it
    int (*AllGather)(const void*, void*, size_t, int, zx_comm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, zx_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, zx_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
static zx_api g_rccl;
static std::once_flag g_rccl_once;

static int zx_fail(int code, const char* what, const char* detail = nullptr) {
    char buf[384];
    if (detail) snprintf(buf, sizeof buf, "%s: %s", what, detail);
    else snprintf(buf, sizeof buf, "%s", what);
    zmi_set_last_error(buf);
    return code;
}

static void zx_load() {
    const char* names[4] = {getenv("ZMI_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
        g_rccl.why = dlerror();
        if (n == names[0]) break;   // an explicit choice is not second-guessed
    }
    if (!g_rccl.handle) return;
    bool ok = true;
#define ZX_SYM(field, name)                                                   \
    do {                                                                      \
        *(void**)(&g_rccl.field) = dlsym(g_rccl.handle, name);                \
        if (!g_rccl.field) { ok = false; g_rccl.why = std::string("missing symbol ") + name; } \
    } while (0)
    ZX_SYM(GetUniqueId, "ncclGetUniqueId");
    ZX_SYM(CommInitRank, "ncclCommInitRank");
    ZX_SYM(CommDestroy, "ncclCommDestroy");
    ZX_SYM(CommAbort, "ncclCommAbort");
    ZX_SYM(CommCount, "ncclCommCount");
    ZX_SYM(CommUserRank, "ncclCommUserRank");
    ZX_SYM(AllGather, "ncclAllGather");
    ZX_SYM(Send, "ncclSend");
    ZX_SYM(Recv, "ncclRecv");
    ZX_SYM(GroupStart, "ncclGroupStart");
    ZX_SYM(GroupEnd, "ncclGroupEnd");
    ZX_SYM(GetErrorString, "ncclGetErrorString");
#undef ZX_SYM
    *(void**)(&g_rccl.CommCuDevice) = dlsym(g_rccl.handle, "ncclCommCuDevice");
    if (!ok) { dlclose(g_rccl.handle); g_rccl.handle = nullptr; }
}
static int zx_need_rccl() {
    std::call_once(g_rccl_once, zx_load);
    if (!g_rccl.handle) return zx_fail(ZMI_E_NORCCL, "RCCL is not available (librccl.so.1)", g_rccl.why.c_str());
    return 0;
}
#define ZX_NCCL(call)                                                                        \
    do {                                                                                     \
        int r_ = (call);                                                                     \
        if (r_ != 0) return zx_fail(ZMI_E_RCCL, #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
    } while (0)

struct zmi_comm {
    zx_comm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    bool owned = false;   // created by zmi_comm_create (destroyed with the handle) or adopted from the host
    uint32_t* nl_dev = nullptr;   // 65 dwords: the ranks' n_local (zmi_exchange_sizes checks that they agree)
};

struct zx_dev_guard {
    int prev = -1;
    bool switched = false;
    explicit zx_dev_guard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) switched = hipSetDevice(dev) == hipSuccess && prev >= 0;
    }
    ~zx_dev_guard() { if (switched) (void)hipSetDevice(prev); }
};

extern "C" int zmi_comm_unique_id(void* id128) {
    if (!id128) return zx_fail(ZMI_E_ARG, "zmi_comm_unique_id: null buffer");
    if (int rc = zx_need_rccl()) return rc;
    zx_uid id;
    memset(&id, 0, sizeof id);
    ZX_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return ZMI_E_OK;
}

extern "C" int zmi_comm_create(zmi_comm** out, zmi_ctx* ctx, int world, int rank, const void* id128) {
    if (!out || !ctx || !id128 || world < 1 || rank < 0 || rank >= world) return zx_fail(ZMI_E_ARG, "zmi_comm_create: bad argument");
    if (int rc = zx_need_rccl()) return rc;
    zx_dev_guard g(zmi_ctx_device(ctx));   // ncclCommInitRank binds the communicator to the calling thread's current device
    zx_uid id;
    memcpy(&id, id128, sizeof id);
    zmi_comm* c = new zmi_comm();
    c->world = world; c->rank = rank; c->device = zmi_ctx_device(ctx); c->owned = true;
    int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) { delete c; return zx_fail(ZMI_E_RCCL, "ncclCommInitRank", g_rccl.GetErrorString(r)); }
    *out = c;
    return ZMI_E_OK;
}

extern "C" int zmi_comm_adopt(zmi_comm** out, zmi_ctx* ctx, void* nccl_comm) {
    if (!out || !ctx || !nccl_comm) return zx_fail(ZMI_E_ARG, "zmi_comm_adopt: bad argument");
    if (int rc = zx_need_rccl()) return rc;
    zmi_comm* c = new zmi_comm();
    c->comm = (zx_comm_t)nccl_comm; c->device = zmi_ctx_device(ctx); c->owned = false;
    int r = g_rccl.CommCount(c->comm, &c->world);
    if (r == 0) r = g_rccl.CommUserRank(c->comm, &c->rank);
    if (r != 0) { delete c; return zx_fail(ZMI_E_RCCL, "ncclCommCount / ncclCommUserRank", g_rccl.GetErrorString(r)); }
    // the exchange runs under a device guard for the CONTEXT's device: a communicator that lives on another GPU is an argument error
    int cdev = c->device;
    if (g_rccl.CommCuDevice && g_rccl.CommCuDevice(c->comm, &cdev) == 0 && cdev != c->device) {
        delete c;
        return zx_fail(ZMI_E_ARG, "zmi_comm_adopt: the communicator belongs to another device than the context");
    }
    *out = c;
    return ZMI_E_OK;
}

extern "C" int zmi_comm_destroy(zmi_comm* c) {
    if (!c) return ZMI_E_OK;
    int r = 0;
    if (c->owned && c->comm && g_rccl.handle) { zx_dev_guard g(c->device); r = g_rccl.CommDestroy(c->comm); }
    if (c->nl_dev) { zx_dev_guard g(c->device); (void)hipFree(c->nl_dev); }
    delete c;
    return r == 0 ? ZMI_E_OK : zx_fail(ZMI_E_RCCL, "ncclCommDestroy", g_rccl.GetErrorString(r));
}

// gives up on a communicator whose operations cannot complete (a peer died): ncclCommAbort instead of ncclCommDestroy
extern "C" int zmi_comm_abort(zmi_comm* c) {
    if (!c) return ZMI_E_OK;
    if (c->comm && g_rccl.handle) { zx_dev_guard g(c->device); (void)g_rccl.CommAbort(c->comm); }
    if (c->nl_dev) { zx_dev_guard g(c->device); (void)hipFree(c->nl_dev); }
    delete c;
    return ZMI_E_OK;
}

extern "C" int zmi_comm_world(const zmi_comm* c) { return c ? c->world : 0; }
extern "C" int zmi_comm_rank(const zmi_comm* c) { return c ? c->rank : -1; }

// d_table[r * n_local + j] = d_sizes[j] of rank r, on every rank
extern "C" int zmi_exchange_sizes(zmi_comm* c, const uint32_t* d_sizes, uint32_t n_local, uint32_t* d_table, void* stream) {
    if (!c || !d_sizes || !d_table) return zx_fail(ZMI_E_ARG, "zmi_exchange_sizes: null argument");
    // Every rank must pass the SAME n_local (a job whose shard count does not divide by the world size pads the short ranks'
    // tables with zero sizes): a mismatch would be an all-gather with different counts -- a hang or a garbage table.  n_local = 0
    // on every rank is a no-op; on one rank only it is the same mismatch, so it takes part like any other value.
    //  Everything that can fail on THIS rank alone -- argument checks, allocations, the copy of n_local -- happens and is
    // waited for before the first collective: a rank that returned from here after its peers entered ncclAllGather would leave them
    // blocked in it.  A failure behind that point (the collective itself, the read-back) leaves the communicator in an unknown
    // state: the stream is drained, the error says so, and the caller is expected to zmi_comm_abort().
    if (c->world > 64) return zx_fail(ZMI_E_ARG, "zmi_exchange_sizes: more than 64 ranks");
    zx_dev_guard g(c->device);
    if (c->world > 1) {   // one dword per rank in front of the table's all-gather: 4 * world bytes, checked on the host
        if (c->nl_dev == nullptr && hipMalloc((void**)&c->nl_dev, 4u * 65u) != hipSuccess) return zx_fail(ZMI_E_NOMEM, "zmi_exchange_sizes: hipMalloc");
        uint32_t mine = n_local;
        std::vector<uint32_t> all;
        try { all.resize((size_t)c->world); } catch (...) { return zx_fail(ZMI_E_NOMEM, "zmi_exchange_sizes: host table"); }
        if (hipMemcpyAsync(c->nl_dev + 64, &mine, 4u, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
            hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
            return zx_fail(ZMI_E_HIP, "zmi_exchange_sizes: copy of n_local");
        {
            const int nr = g_rccl.AllGather(c->nl_dev + 64, c->nl_dev, 1, ZX_UINT32, c->comm, (hipStream_t)stream);
            if (nr != 0) {
                (void)hipStreamSynchronize((hipStream_t)stream);
                return zx_fail(ZMI_E_RCCL, "zmi_exchange_sizes: all-gather of n_local failed (abort the communicator: zmi_comm_abort)", g_rccl.GetErrorString(nr));
            }
        }
        if (hipMemcpyAsync(all.data(), c->nl_dev, 4u * (size_t)c->world, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
            hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
            (void)hipStreamSynchronize((hipStream_t)stream);
            return zx_fail(ZMI_E_HIP, "zmi_exchange_sizes: read-back of the ranks' n_local (abort the communicator: zmi_comm_abort)");
        }
        for (int p = 0; p < c->world; ++p)
            if (all[(size_t)p] != n_local) return zx_fail(ZMI_E_ARG, "zmi_exchange_sizes: the ranks passed different n_local (pad with zero sizes)");
    }
    if (n_local == 0) return ZMI_E_OK;
    ZX_NCCL(g_rccl.AllGather(d_sizes, d_table, n_local, ZX_UINT32, c->comm, (hipStream_t)stream));
    return ZMI_E_OK;
}

// ---- the plan: everything "append in order" needs, from the size table alone -------------------------------------------
// table[r][j] = compressed size of global shard g = j * world + r.
//   goff[r][j]  byte offset of that shard in the stitched output (exclusive scan in g order)
//   soff[r][j]  byte offset inside rank r's dense slab, soff[r][n_local] = the slab's size
//   totals[r]   = soff[r][n_local];  totals[world] = size of the stitched output
// One workgroup: the table is 4 B per shard (2 MiB at 512 Ki shards), the scans are over in microseconds.
static __device__ uint64_t zx_block_scan(uint64_t* part, uint64_t sum) {   // exclusive prefix of `sum` over the 1024 threads; part[1023] = total
    const uint32_t t = threadIdx.x;
    __syncthreads();
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint64_t v = t >= d ? part[t - d] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    return part[t] - sum;
}

__global__ void __launch_bounds__(1024) zmi_stitch_plan_kernel(const uint32_t* __restrict__ table, uint32_t world, uint32_t n_local,
                                                                uint64_t* __restrict__ goff, uint64_t* __restrict__ soff,
                                                                uint64_t* __restrict__ totals) {
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    {   // global order
        const uint64_t n = (uint64_t)world * n_local;
        const uint64_t per = (n + 1023u) / 1024u;
        const uint64_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
        uint64_t sum = 0;
        for (uint64_t g = lo; g < hi; ++g) sum += table[(g % world) * n_local + g / world];
        uint64_t o = zx_block_scan(part, sum);
        for (uint64_t g = lo; g < hi; ++g) {
            const uint64_t k = (g % world) * n_local + g / world;
            goff[k] = o;
            o += table[k];
        }
        if (t == 1023u) totals[world] = part[1023];
    }
    for (uint32_t r = 0; r < world; ++r) {   // every rank's slab
        const uint32_t* row = table + (uint64_t)r * n_local;
        uint64_t* so = soff + (uint64_t)r * (n_local + 1u);
        const uint32_t per = (n_local + 1023u) / 1024u;
        const uint32_t lo = t * per < n_local ? t * per : n_local, hi = lo + per < n_local ? lo + per : n_local;
        uint64_t sum = 0;
        for (uint32_t j = lo; j < hi; ++j) sum += row[j];
        uint64_t o = zx_block_scan(part, sum);
        for (uint32_t j = lo; j < hi; ++j) { so[j] = o; o += row[j]; }
        if (t == 1023u) { so[n_local] = part[1023]; totals[r] = part[1023]; }
    }
}

extern "C" int zmi_stitch_plan_dev(zmi_ctx* ctx, const uint32_t* d_table, uint32_t world, uint32_t n_local, uint64_t* d_goff,
                                   uint64_t* d_soff, uint64_t* d_totals, uint64_t* totals_host, void* stream) {
    if (!ctx || !d_table || !d_goff || !d_soff || !d_totals || world == 0) return zx_fail(ZMI_E_ARG, "zmi_stitch_plan_dev: bad argument");
    zx_dev_guard g(zmi_ctx_device(ctx));
    ZMI_LAUNCH(zmi_stitch_plan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_table, world, n_local, d_goff, d_soff, d_totals);
    if (hipGetLastError() != hipSuccess) return zx_fail(ZMI_E_HIP, "zmi_stitch_plan_kernel launch");
    if (totals_host) {   // the slab sizes are what the host posts its sends and receives with
        if (hipMemcpyAsync(totals_host, d_totals, (size_t)(world + 1u) * 8u, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
            hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
            return zx_fail(ZMI_E_HIP, "zmi_stitch_plan_dev: copy of the totals");
    }
    return ZMI_E_OK;
}

// One round of the exchange: bytes [lo, lo + chunk_bytes) of every slab.  This rank sends that piece of its own slab to every
// peer (root < 0) or to `root`; it receives peer p's piece into d_recv[p] + (recv_at_offset ? lo : 0) when it is a receiver.
// Sends and receives are decided by the SAME data on both sides -- the slab sizes every rank holds and the root -- never by what
// a receiver happens to pass: a receiver has to take every peer's bytes (zx_check refuses a null entry for a peer whose slab is
// not empty), otherwise that peer's ncclSend would have no partner and the group would hang.
// All of it inside one group: the transfers of a round run concurrently, one link each.
static int zx_round(zmi_comm* c, const uint8_t* d_slab, const uint64_t* slab_bytes, uint64_t lo, uint64_t chunk, void* const* d_recv,
                    bool recv_at_offset, int root, hipStream_t stream) {
    const bool receives = root < 0 || root == c->rank;
    const uint64_t mine = slab_bytes[c->rank];
    ZX_NCCL(g_rccl.GroupStart());
    int rc = 0;
    for (int p = 0; p < c->world && rc == 0; ++p) {
        if (p == c->rank) continue;
        if (lo < mine && (root < 0 || p == root)) {
            const uint64_t n = mine - lo < chunk ? mine - lo : chunk;
            rc = g_rccl.Send(d_slab + lo, (size_t)n, ZX_UINT8, p, c->comm, stream);
        }
        if (rc == 0 && receives && lo < slab_bytes[p]) {
            const uint64_t n = slab_bytes[p] - lo < chunk ? slab_bytes[p] - lo : chunk;
            rc = g_rccl.Recv((uint8_t*)d_recv[p] + (recv_at_offset ? lo : 0), (size_t)n, ZX_UINT8, p, c->comm, stream);
        }
    }
    const int re = g_rccl.GroupEnd();
    if (rc != 0) return zx_fail(ZMI_E_RCCL, "ncclSend / ncclRecv", g_rccl.GetErrorString(rc));
    if (re != 0) return zx_fail(ZMI_E_RCCL, "ncclGroupEnd", g_rccl.GetErrorString(re));
    return 0;
}

static int zx_check(zmi_comm* c, const void* d_slab, const uint64_t* slab_bytes, uint64_t chunk, int root, void* const* d_recv,
                    const char* who) {
    if (!c || !slab_bytes || chunk == 0) return zx_fail(ZMI_E_ARG, who, "null argument or zero chunk");
    if (root >= c->world) return zx_fail(ZMI_E_ARG, who, "root out of range");
    if (!d_slab && slab_bytes[c->rank]) return zx_fail(ZMI_E_ARG, who, "null slab");
    if (root < 0 || root == c->rank) {   // this rank receives: every peer with bytes sends to it, so it needs room for each
        for (int p = 0; p < c->world; ++p)
            if (p != c->rank && slab_bytes[p] != 0 && (!d_recv || !d_recv[p]))
                return zx_fail(ZMI_E_ARG, who, "a receiving rank must give room for every peer's slab (null d_recv entry)");
    }
    return 0;
}

// whole slabs: d_recv[p] holds peer p's slab (slab_bytes[p] bytes) afterwards; rounds of chunk_bytes keep RCCL's staging bounded
extern "C" int zmi_exchange_slabs(zmi_comm* c, const void* d_slab, const uint64_t* slab_bytes, void* const* d_recv,
                                  uint64_t chunk_bytes, int root, void* stream) {
    if (int rc = zx_check(c, d_slab, slab_bytes, chunk_bytes, root, d_recv, "zmi_exchange_slabs")) return rc;
    zx_dev_guard g(c->device);
    uint64_t biggest = 0;
    for (int p = 0; p < c->world; ++p) biggest = slab_bytes[p] > biggest ? slab_bytes[p] : biggest;
    for (uint64_t lo = 0; lo < biggest; lo += chunk_bytes)
        if (int rc = zx_round(c, (const uint8_t*)d_slab, slab_bytes, lo, chunk_bytes, d_recv, true, root, (hipStream_t)stream)) return rc;
    return ZMI_E_OK;
}

// bounded memory: ONE round; peer p's bytes [lo, lo + chunk_bytes) land at the start of d_stage[p] (chunk_bytes of staging per
// peer, reused by the next round once the caller has consumed it -- scattered it with zmi_copy_ranges_dev, written it out)
extern "C" int zmi_exchange_slabs_round(zmi_comm* c, const void* d_slab, const uint64_t* slab_bytes, uint64_t lo, uint64_t chunk_bytes,
                                        void* const* d_stage, int root, void* stream) {
    if (int rc = zx_check(c, d_slab, slab_bytes, chunk_bytes, root, d_stage, "zmi_exchange_slabs_round")) return rc;
    zx_dev_guard g(c->device);
    return zx_round(c, (const uint8_t*)d_slab, slab_bytes, lo, chunk_bytes, d_stage, false, root, (hipStream_t)stream);
}
