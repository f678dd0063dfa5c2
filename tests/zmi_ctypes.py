"""ctypes binding of the batch C ABI (include/zmi355.h) used by the tests.

Loads either the product library (zlib_rs_amd/libzmi355.so, needs an MI355X) or the CPU SIMT
emulator build of the same kernels (tests/emu/libzmi355_emu.so, test infrastructure only).
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bind(lib):
    u8p, u32p, u64p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
    lib.zmi_version.restype = C.c_char_p
    lib.zmi_last_error.restype = C.c_char_p
    lib.zmi_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.zmi_ctx_destroy.argtypes = [C.c_void_p]
    lib.zmi_deflate_bound.restype = C.c_uint64
    lib.zmi_deflate_bound.argtypes = [C.c_uint64, C.c_int]
    lib.zmi_deflate_batch.argtypes = [C.c_void_p, C.c_void_p, u64p, u32p, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_uint64, u32p, i32p]
    lib.zmi_inflate_batch.argtypes = [C.c_void_p, C.c_void_p, u64p, u32p, C.c_uint32, C.c_int, C.c_void_p, u64p, u32p,
                                      u32p, i32p]
    vp = C.c_void_p
    lib.zmi_inflate_batch_dev.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.zmi_ctx_set_inflate_out_limit.argtypes = [vp, C.c_uint64]
    lib.zmi_scan_sizes_dev.argtypes = [vp, vp, C.c_uint32, vp, vp]
    lib.zmi_copy_ranges_dev.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_uint64, vp]
    lib.zmi_pack_slab_dev.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, vp, C.c_uint64, vp, vp]
    lib.zmi_inflate_resume.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, u32p, i32p, i32p, u32p, u32p]
    lib.zmi_inflate_split.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, u32p, i32p, i32p, u32p,
                                      u32p, u32p]
    lib.zmi_inflate_blocks.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, u32p, i32p, i32p, u32p, u32p, u32p]
    # the multi-GPU stitch (csrc/exchange.hip)
    lib.zmi_comm_unique_id.argtypes = [vp]
    lib.zmi_comm_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int, vp]
    lib.zmi_comm_destroy.argtypes = [vp]
    lib.zmi_comm_world.argtypes = [vp]
    lib.zmi_comm_rank.argtypes = [vp]
    lib.zmi_exchange_sizes.argtypes = [vp, vp, C.c_uint32, vp, vp]
    lib.zmi_stitch_plan_dev.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]
    lib.zmi_exchange_slabs.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_int, vp]
    lib.zmi_exchange_slabs_round.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, vp, C.c_int, vp]
    return lib


def load_emu(rebuild=True):
    d = os.path.join(ROOT, "tests", "emu")
    if rebuild:
        subprocess.run(["make", "-s", "-C", d], check=True)
    return _bind(C.CDLL(os.path.join(d, "libzmi355_emu.so")))


def load_product():
    return _bind(C.CDLL(os.path.join(ROOT, "zlib_rs_amd", "libzmi355.so")))


class Engine:
    """Host-buffer view of the batch API (works for both the product and the emulator build)."""

    def __init__(self, lib, device=0):
        self.lib = lib
        self.ctx = C.c_void_p()
        rc = lib.zmi_ctx_create(C.byref(self.ctx), device)
        if rc != 0:
            raise RuntimeError("zmi_ctx_create failed: %d %s" % (rc, lib.zmi_last_error().decode()))

    def close(self):
        if self.ctx:
            self.lib.zmi_ctx_destroy(self.ctx)
            self.ctx = None

    def deflate(self, shards, level=6, strategy=0, wrap=1):
        """shards: list of bytes -> (list of compressed bytes, list of status)"""
        n = len(shards)
        lens = np.array([len(s) for s in shards], dtype=np.uint32)
        offs = np.zeros(n, dtype=np.uint64)
        if n:
            offs[1:] = np.cumsum(lens[:-1].astype(np.uint64))
        blob = np.frombuffer(b"".join(shards) + b"\0", dtype=np.uint8).copy()
        stride = int(self.lib.zmi_deflate_bound(int(lens.max()) if n else 0, wrap))
        out = np.zeros(max(1, n * stride), dtype=np.uint8)
        olen = np.zeros(max(1, n), dtype=np.uint32)
        st = np.zeros(max(1, n), dtype=np.int32)
        rc = self.lib.zmi_deflate_batch(self.ctx, blob.ctypes.data, offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        lens.ctypes.data_as(C.POINTER(C.c_uint32)), n, level, strategy, wrap,
                                        out.ctypes.data, stride, olen.ctypes.data_as(C.POINTER(C.c_uint32)),
                                        st.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc != 0:
            raise RuntimeError("zmi_deflate_batch failed: %d %s" % (rc, self.lib.zmi_last_error().decode()))
        return [bytes(out[i * stride:i * stride + int(olen[i])]) for i in range(n)], [int(x) for x in st[:n]]

    def inflate(self, streams, caps, wrap=1):
        """streams: list of bytes, caps: list of output capacities -> (list of bytes, list of status)"""
        n = len(streams)
        lens = np.array([len(s) for s in streams], dtype=np.uint32)
        offs = np.zeros(n, dtype=np.uint64)
        if n:
            offs[1:] = np.cumsum(lens[:-1].astype(np.uint64))
        blob = np.frombuffer(b"".join(streams) + b"\0", dtype=np.uint8).copy()
        ocap = np.array(caps, dtype=np.uint32)
        ooff = np.zeros(n, dtype=np.uint64)
        if n:
            ooff[1:] = np.cumsum(ocap[:-1].astype(np.uint64))
        out = np.zeros(max(1, int(ocap.astype(np.uint64).sum())), dtype=np.uint8)
        olen = np.zeros(max(1, n), dtype=np.uint32)
        st = np.zeros(max(1, n), dtype=np.int32)
        rc = self.lib.zmi_inflate_batch(self.ctx, blob.ctypes.data, offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        lens.ctypes.data_as(C.POINTER(C.c_uint32)), n, wrap, out.ctypes.data,
                                        ooff.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        ocap.ctypes.data_as(C.POINTER(C.c_uint32)),
                                        olen.ctypes.data_as(C.POINTER(C.c_uint32)), st.ctypes.data_as(C.POINTER(C.c_int32)))
        if rc != 0:
            raise RuntimeError("zmi_inflate_batch failed: %d %s" % (rc, self.lib.zmi_last_error().decode()))
        return [bytes(out[int(ooff[i]):int(ooff[i]) + min(int(olen[i]), int(ocap[i]))]) for i in range(n)], [int(x) for x in st[:n]]

    def inflate_dev(self, streams, caps, in_offsets, out_offsets, wrap=1, out_limit=None):
        """EMULATOR ONLY (its "device" memory is host memory): the device-pointer entry point with an
        arbitrary input / output layout -> (list of bytes, list of status)"""
        n = len(streams)
        lens = np.array([len(s) for s in streams], dtype=np.uint32)
        ioff = np.array(in_offsets, dtype=np.uint64)
        ooff = np.array(out_offsets, dtype=np.uint64)
        ocap = np.array(caps, dtype=np.uint32)
        blob = np.zeros(int(max(int(o) + len(s) for o, s in zip(in_offsets, streams))) + 64, dtype=np.uint8)
        for o, s_ in zip(in_offsets, streams):
            blob[int(o):int(o) + len(s_)] = np.frombuffer(s_, dtype=np.uint8)
        out = np.full(int(max(int(o) + int(c) for o, c in zip(out_offsets, caps))) + 64, 0xEE, dtype=np.uint8)
        olen = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.int32)
        if out_limit is not None:
            self.lib.zmi_ctx_set_inflate_out_limit(self.ctx, int(out_limit))
        rc = self.lib.zmi_inflate_batch_dev(self.ctx, blob.ctypes.data, ioff.ctypes.data, lens.ctypes.data, n, wrap,
                                            out.ctypes.data, ooff.ctypes.data, ocap.ctypes.data, olen.ctypes.data,
                                            st.ctypes.data, None)
        if rc != 0:
            raise RuntimeError("zmi_inflate_batch_dev failed: %d %s" % (rc, self.lib.zmi_last_error().decode()))
        res = [bytes(out[int(ooff[i]):int(ooff[i]) + min(int(olen[i]), int(ocap[i]))]) for i in range(n)]
        guard = [bytes(out[int(ooff[i]) + int(ocap[i]):int(ooff[i]) + int(ocap[i]) + 1]) for i in range(n)]
        return res, [int(x) for x in st], guard

    def pack_slab(self, members, stride, slot_base=0, slab_base=0):
        """EMULATOR ONLY: members laid out in stride-d slots (first slot at byte slot_base of its buffer) ->
        (slab bytes, offsets list of n + 1) through zmi_pack_slab_dev; the slab starts at byte slab_base of its buffer"""
        n = len(members)
        slots = np.full(slot_base + n * stride + 64, 0xA5, dtype=np.uint8)
        for i, m in enumerate(members):
            slots[slot_base + i * stride:slot_base + i * stride + len(m)] = np.frombuffer(m, dtype=np.uint8)
        lens = np.array([len(m) for m in members], dtype=np.uint32)
        total = int(lens.sum())
        slab = np.full(slab_base + total + 64, 0xEE, dtype=np.uint8)
        off = np.zeros(n + 1, dtype=np.uint64)
        rc = self.lib.zmi_pack_slab_dev(self.ctx, slots.ctypes.data + slot_base, stride, lens.ctypes.data, n,
                                        slab.ctypes.data + slab_base, total, off.ctypes.data, None)
        if rc != 0:
            raise RuntimeError("zmi_pack_slab_dev failed: %d %s" % (rc, self.lib.zmi_last_error().decode()))
        assert bytes(slab[:slab_base]) == b"\xee" * slab_base and bytes(slab[slab_base + total:]) == b"\xee" * 64   # nothing outside
        return bytes(slab[slab_base:slab_base + total]), [int(x) for x in off]

    def inflate_resume(self, data, in_bit=0, hist=b"", cap=1 << 16):
        """one call of the resumable raw-deflate decode (zmi_inflate_resume) ->
        (output bytes incl. the valid part of an unfinished block, status, detail, in_used, [byte, bit, out, complete])"""
        src = np.frombuffer(bytes(data) + b"\0", dtype=np.uint8).copy()
        h = np.frombuffer(bytes(hist) + b"\0", dtype=np.uint8).copy()
        out = np.zeros(max(1, cap), dtype=np.uint8)
        olen, used = C.c_uint32(0), C.c_uint32(0)
        st, det = C.c_int32(0), C.c_int32(0)
        res = (C.c_uint32 * 4)()
        rc = self.lib.zmi_inflate_resume(self.ctx, src.ctypes.data, len(data), in_bit, h.ctypes.data, len(hist), out.ctypes.data, cap,
                                         C.byref(olen), C.byref(st), C.byref(det), C.byref(used), res)
        if rc != 0:
            raise RuntimeError("zmi_inflate_resume failed: %d %s" % (rc, self.lib.zmi_last_error().decode()))
        return bytes(out[:min(olen.value, cap)]), st.value, det.value, used.value, list(res)

    def inflate_blocks(self, data, in_bit=0, hist=b"", cap=1 << 16):
        """zmi_inflate_blocks: the same call, the restart points found by the device's block scan -> the same tuple + segments"""
        src = np.frombuffer(bytes(data) + b"\0", dtype=np.uint8).copy()
        h = np.frombuffer(bytes(hist) + b"\0", dtype=np.uint8).copy()
        out = np.zeros(max(1, cap), dtype=np.uint8)
        olen, used, nused = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        st, det = C.c_int32(0), C.c_int32(0)
        res = (C.c_uint32 * 4)()
        rc = self.lib.zmi_inflate_blocks(self.ctx, src.ctypes.data, len(data), in_bit, h.ctypes.data, len(hist), out.ctypes.data, cap,
                                         C.byref(olen), C.byref(st), C.byref(det), C.byref(used), res, C.byref(nused))
        if rc != 0:
            raise RuntimeError("zmi_inflate_blocks failed: %d %s" % (rc, self.lib.zmi_last_error().decode()))
        return bytes(out[:min(olen.value, cap)]), st.value, det.value, used.value, list(res), nused.value

    def inflate_split(self, data, seg_start, in_bit=0, hist=b"", cap=1 << 16):
        """zmi_inflate_split: the same call with proposed restart points -> the same tuple + segments decoded in parallel"""
        src = np.frombuffer(bytes(data) + b"\0", dtype=np.uint8).copy()
        h = np.frombuffer(bytes(hist) + b"\0", dtype=np.uint8).copy()
        out = np.zeros(max(1, cap), dtype=np.uint8)
        seg = (C.c_uint32 * max(1, len(seg_start)))(*seg_start)
        olen, used, nused = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        st, det = C.c_int32(0), C.c_int32(0)
        res = (C.c_uint32 * 4)()
        rc = self.lib.zmi_inflate_split(self.ctx, src.ctypes.data, len(data), in_bit, h.ctypes.data, len(hist), out.ctypes.data, cap,
                                        seg, len(seg_start), C.byref(olen), C.byref(st), C.byref(det), C.byref(used), res, C.byref(nused))
        if rc != 0:
            raise RuntimeError("zmi_inflate_split failed: %d %s" % (rc, self.lib.zmi_last_error().decode()))
        return bytes(out[:min(olen.value, cap)]), st.value, det.value, used.value, list(res), nused.value
