"""PCIe-inclusive rate of the host-buffer entry points (zmi_deflate_batch / zmi_inflate_batch, include/zmi355.h):
what a caller sees who hands over host memory.  DESIGN.md section 4 quotes it; it is never bench.py's `value`.
PROBE_S shards of 1 MiB (default 2048), level 6, pageable and pinned host memory."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from zlib_rs_amd.engine import Engine  # noqa: E402


def main():
    S = int(os.environ.get("PROBE_S", "2048"))
    B = 1 << 20
    e = Engine(0)
    L = e.L
    u64p, u32p, i32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
    L.zmi_deflate_bound.restype = C.c_uint64
    L.zmi_deflate_batch.argtypes = [C.c_void_p, C.c_void_p, u64p, u32p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64,
                                    u32p, i32p]
    L.zmi_inflate_batch.argtypes = [C.c_void_p, C.c_void_p, u64p, u32p, C.c_uint32, C.c_int, C.c_void_p, u64p, u32p, u32p, i32p]
    dev = e.gen_shards(S, B)
    stride = int(L.zmi_deflate_bound(B, 1))
    off = (np.arange(S, dtype=np.uint64) * B)
    ln = np.full(S, B, dtype=np.uint32)
    olen = np.zeros(S, dtype=np.uint32)
    st = np.zeros(S, dtype=np.int32)
    for kind in os.environ.get("PROBE_KINDS", "pageable,pinned").split(","):
        host_in = dev.cpu()
        host_out = torch.empty(S * stride, dtype=torch.uint8)
        if kind == "pinned":
            host_in, host_out = host_in.pin_memory(), host_out.pin_memory()
        for rep in range(2):
            t = time.perf_counter()
            rc = L.zmi_deflate_batch(e._ctx, host_in.data_ptr(), off.ctypes.data_as(u64p), ln.ctypes.data_as(u32p), S, 6, 0, 1,
                                     host_out.data_ptr(), stride, olen.ctypes.data_as(u32p), st.ctypes.data_as(i32p))
            dt = time.perf_counter() - t
        assert rc == 0 and not st.any()
        ratio = S * B / float(olen.astype(np.int64).sum())
        print("deflate host buffers (%s): %.2f GiB/s of input incl. PCIe both ways, ratio %.3f, %d shards" %
              (kind, S * B / 2**30 / dt, ratio, S))
        if os.environ.get("PROBE_DEFLATE_ONLY"):
            continue
        # and back: compressed streams from host memory, output to host memory
        back = torch.empty(S * B, dtype=torch.uint8)
        if kind == "pinned":
            back = back.pin_memory()
        coff = (np.arange(S, dtype=np.uint64) * stride)
        ocap = np.full(S, B, dtype=np.uint32)
        blen = np.zeros(S, dtype=np.uint32)
        for rep in range(2):
            t = time.perf_counter()
            rc = L.zmi_inflate_batch(e._ctx, host_out.data_ptr(), coff.ctypes.data_as(u64p), olen.ctypes.data_as(u32p), S, 1,
                                     back.data_ptr(), off.ctypes.data_as(u64p), ocap.ctypes.data_as(u32p), blen.ctypes.data_as(u32p),
                                     st.ctypes.data_as(i32p))
            dt = time.perf_counter() - t
        assert rc == 0 and not st.any() and torch.equal(back, host_in)
        print("inflate host buffers (%s): %.2f GiB/s of output incl. PCIe both ways, bit-exact" % (kind, S * B / 2**30 / dt))
    e.close()


if __name__ == "__main__":
    main()
