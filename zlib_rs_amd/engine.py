"""Device-resident batch deflate / inflate on one MI355X.

torch is plumbing here: it owns the HBM buffers and the HIP stream; all compute is in the
hand-written HIP kernels of csrc/ reached through the C ABI (include/zmi355.h).

Mirrors the per-stream contract of the reference's C API
(libz-rs-sys/src/lib.rs: deflateInit2_ :2005, deflate :1281, inflateInit2_ :967, inflate :636):
every shard is compressed as deflateInit2_(level, Z_DEFLATED, wbits, 8, strategy) +
deflate(Z_FINISH) would, and the per-shard return code uses zlib's numbering.
"""
import torch

from . import _lib

WRAP_RAW, WRAP_ZLIB, WRAP_GZIP, WRAP_AUTO = 0, 1, 2, 3
GEN_SEED = 0x5A4C4942


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


class Engine:
    def __init__(self, device=None, scratch_bytes=None):
        if not torch.cuda.is_available():
            raise RuntimeError("zlib_rs_amd needs an MI355X (no HIP device visible); there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.L = _lib.lib()
        import ctypes as C
        self._ctx = C.c_void_p()
        _lib.check(self.L.zmi_ctx_create(C.byref(self._ctx), self.device.index), "zmi_ctx_create")
        if scratch_bytes:
            _lib.check(self.L.zmi_ctx_set_scratch_limit(self._ctx, int(scratch_bytes)), "zmi_ctx_set_scratch_limit")

    def close(self):
        if self._ctx:
            self.L.zmi_ctx_destroy(self._ctx)
            self._ctx = None

    def deflate_bound(self, n, wrap=WRAP_ZLIB):
        return int(self.L.zmi_deflate_bound(int(n), int(wrap)))

    # ---- synthetic benchmark shards (csrc/shardgen.h) ----
    def gen_shards(self, n_shards, shard_bytes=1 << 20, first_shard=0, seed=GEN_SEED, out=None, shard_step=1):
        """shard first_shard + i*shard_step at out[i*shard_bytes:]; shard_step = world size gives a rank its round-robin
        share of a multi-GPU job (dist.shards_of_rank)"""
        if out is None:
            out = torch.empty(n_shards * shard_bytes, dtype=torch.uint8, device=self.device)
        # the generator kernel indexes lines with 32-bit block ids: chunk very large batches
        step = 16384
        for s0 in range(0, n_shards, step):
            cnt = min(step, n_shards - s0)
            _lib.check(self.L.zmi_gen_shards_strided_dev(self._ctx, out.data_ptr() + s0 * shard_bytes, seed,
                                                         first_shard + s0 * shard_step, shard_step, cnt, shard_bytes,
                                                         _stream_ptr()), "zmi_gen_shards_strided_dev")
        return out

    # ---- the stitch: strided slots -> dense slab -> globally ordered output (csrc/pack.hip) ----
    def scan_sizes(self, lengths, out=None):
        """exclusive prefix sum of the int32 sizes -> int64 offsets, n + 1 entries (last = total)"""
        n = int(lengths.numel())
        if out is None:
            out = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        _lib.check(self.L.zmi_scan_sizes_dev(self._ctx, lengths.data_ptr(), n, out.data_ptr(), _stream_ptr()), "zmi_scan_sizes_dev")
        return out

    def pack_slab(self, slots, lengths, slab=None, offsets=None):
        """slots [n, stride] uint8 with lengths[i] valid bytes each -> (slab uint8, offsets int64[n + 1]).  Without a
        preallocated slab this synchronises once to size it."""
        n = int(lengths.numel())
        if offsets is None:
            offsets = self.scan_sizes(lengths)
        if slab is None:
            slab = torch.empty(int(offsets[n].item()) + 16, dtype=torch.uint8, device=self.device)
        self.copy_ranges(slots, None, slots.stride(0), lengths, slots.stride(0), slab, offsets)
        return slab, offsets

    def copy_ranges(self, src, src_off, src_stride, lengths, max_len, dst, dst_off):
        n = int(lengths.numel())
        _lib.check(self.L.zmi_copy_ranges_dev(self._ctx, src.data_ptr(), src_off.data_ptr() if src_off is not None else None,
                                              int(src_stride), lengths.data_ptr(), n, int(min(max_len, 0xFFFFFFFF)), dst.data_ptr(),
                                              dst_off.data_ptr(), int(dst.numel()), _stream_ptr()), "zmi_copy_ranges_dev")
        return dst

    # ---- the stitch across GPUs (csrc/exchange.hip: RCCL behind the C ABI) ----
    def comm_unique_id(self):
        """128 bytes made by one rank and carried to the others by whatever the host has (here: torch's process group)"""
        import ctypes as C
        buf = C.create_string_buffer(128)
        _lib.check(self.L.zmi_comm_unique_id(buf), "zmi_comm_unique_id")
        return buf.raw

    def comm_create(self, world, rank, uid):
        import ctypes as C
        comm = C.c_void_p()
        _lib.check(self.L.zmi_comm_create(C.byref(comm), self._ctx, int(world), int(rank), C.create_string_buffer(bytes(uid), 128)),
                   "zmi_comm_create")
        return comm

    def comm_destroy(self, comm, abort=False):
        (self.L.zmi_comm_abort if abort else self.L.zmi_comm_destroy)(comm)

    def exchange_sizes(self, comm, sizes, world):
        """all-gather of the int32 size tables -> [world, n_local] on every rank"""
        table = torch.empty((world, int(sizes.numel())), dtype=torch.int32, device=self.device)
        _lib.check(self.L.zmi_exchange_sizes(comm, sizes.data_ptr(), int(sizes.numel()), table.data_ptr(), _stream_ptr()),
                   "zmi_exchange_sizes")
        return table

    def stitch_plan(self, table):
        """table [world, n_local] -> (goff [world, n_local], soff [world, n_local + 1], totals: list of world + 1 ints;
        the last one is the size of the stitched output).  Waits for the stream (the totals are host data)."""
        import ctypes as C
        world, n_local = int(table.shape[0]), int(table.shape[1])
        goff = torch.empty((world, n_local), dtype=torch.int64, device=self.device)
        soff = torch.empty((world, n_local + 1), dtype=torch.int64, device=self.device)
        d_tot = torch.empty(world + 1, dtype=torch.int64, device=self.device)
        host = (C.c_uint64 * (world + 1))()
        _lib.check(self.L.zmi_stitch_plan_dev(self._ctx, table.data_ptr(), world, n_local, goff.data_ptr(), soff.data_ptr(),
                                              d_tot.data_ptr(), host, _stream_ptr()), "zmi_stitch_plan_dev")
        return goff, soff, [int(x) for x in host]

    def exchange_round(self, comm, slab, totals, lo, chunk_bytes, stage, root=-1):
        """one bounded round of the slab exchange: stage[p] (a uint8 tensor of chunk_bytes; None for this rank, and on a rank that does
        not receive -- root >= 0 and not this rank; a receiving rank must give room for every peer: ZMI_E_ARG otherwise) receives peer p's slab bytes [lo, lo + chunk_bytes)"""
        import ctypes as C
        world = len(stage)
        tb = (C.c_uint64 * world)(*[int(t) for t in totals[:world]])
        ptrs = (C.c_void_p * world)(*[None if t is None else t.data_ptr() for t in stage])
        _lib.check(self.L.zmi_exchange_slabs_round(comm, slab.data_ptr(), tb, int(lo), int(chunk_bytes), ptrs, int(root), _stream_ptr()),
                   "zmi_exchange_slabs_round")

    # ---- deflate ----
    def deflate_batch(self, data, offsets, lengths, max_len, level=6, strategy=0, wrap=WRAP_ZLIB, out=None, out_len=None,
                      status=None):
        """data: uint8 device tensor; offsets (uint64 as int64) / lengths (uint32 as int32) device tensors.
        Returns (out [n, stride] uint8, out_len [n] int32, status [n] int32)."""
        n = int(lengths.numel())
        stride = self.deflate_bound(max_len, wrap)
        if out is None:
            out = torch.empty((n, stride), dtype=torch.uint8, device=self.device)
        if out_len is None:
            out_len = torch.empty(n, dtype=torch.int32, device=self.device)
        if status is None:
            status = torch.empty(n, dtype=torch.int32, device=self.device)
        _lib.check(self.L.zmi_deflate_batch_dev(self._ctx, data.data_ptr(), offsets.data_ptr(), lengths.data_ptr(), n,
                                                int(max_len), int(level), int(strategy), int(wrap), out.data_ptr(),
                                                out.stride(0) if out.dim() == 2 else stride, out_len.data_ptr(),
                                                status.data_ptr(), _stream_ptr()), "zmi_deflate_batch_dev")
        return out, out_len, status

    # ---- inflate ----
    def inflate_batch(self, data, offsets, lengths, out, out_offsets, out_caps, wrap=WRAP_ZLIB, out_len=None, status=None):
        n = int(lengths.numel())
        if out_len is None:
            out_len = torch.empty(n, dtype=torch.int32, device=self.device)
        if status is None:
            status = torch.empty(n, dtype=torch.int32, device=self.device)
        # inflate keeps 1 bit of scratch per byte of output capacity; the capacities are device data, `out` bounds them
        _lib.check(self.L.zmi_ctx_set_inflate_out_limit(self._ctx, int(out.numel()) + (1 << 20)), "zmi_ctx_set_inflate_out_limit")
        _lib.check(self.L.zmi_inflate_batch_dev(self._ctx, data.data_ptr(), offsets.data_ptr(), lengths.data_ptr(), n,
                                                int(wrap), out.data_ptr(), out_offsets.data_ptr(), out_caps.data_ptr(),
                                                out_len.data_ptr(), status.data_ptr(), _stream_ptr()),
                   "zmi_inflate_batch_dev")
        return out_len, status

    # ---- checksums ----
    def checksums(self, data, offsets, lengths, adler=True, crc=True):
        n = int(lengths.numel())
        a = torch.zeros(n, dtype=torch.int32, device=self.device)
        c = torch.zeros(n, dtype=torch.int32, device=self.device)
        kind = (1 if adler else 0) | (2 if crc else 0)
        _lib.check(self.L.zmi_checksum_batch_dev(self._ctx, data.data_ptr(), offsets.data_ptr(), lengths.data_ptr(), n, kind,
                                                 a.data_ptr(), c.data_ptr(), _stream_ptr()), "zmi_checksum_batch_dev")
        return a, c


def uniform_layout(n, shard_bytes, device):
    """offsets/lengths tensors for n back-to-back shards of equal size"""
    off = (torch.arange(n, dtype=torch.int64, device=device) * shard_bytes)
    ln = torch.full((n,), shard_bytes, dtype=torch.int32, device=device)
    return off, ln
