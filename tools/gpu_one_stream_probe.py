#!/usr/bin/env python3
"""One CPU-made zlib stream (python zlib, level 6, no flush points) through zmi_inflate_batch: time per call; with
ZMI_LIB=variants/libzmi355_prof.so the decode kernel prints its phase profile."""
import ctypes as C
import os
import sys
import time
import zlib

os.environ.setdefault("ZMI_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_fast_probe as P  # noqa: E402
import numpy as np  # noqa: E402
import oracle_lib  # noqa: E402

L, path = P.load_lib()
vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
L.zmi_inflate_batch.argtypes = [vp, vp, vp, vp, u32, i32, vp, vp, vp, vp, vp]
ctx = C.c_void_p()
assert L.zmi_ctx_create(C.byref(ctx), 0) == 0
o = oracle_lib.load(rebuild=False)
for cls in (0, 5):
    raw = o.gen_shard(cls, 1 << 20)
    comp = np.frombuffer(zlib.compress(raw, 6), dtype=np.uint8).copy()
    out = np.zeros(len(raw), dtype=np.uint8)
    ioff = np.zeros(1, dtype=np.uint64); ilen = np.array([len(comp)], dtype=np.uint32)
    ooff = np.zeros(1, dtype=np.uint64); ocap = np.array([len(raw)], dtype=np.uint32)
    olen = np.zeros(1, dtype=np.uint32); st = np.zeros(1, dtype=np.int32)
    for rep in range(3):
        t = time.perf_counter()
        rc = L.zmi_inflate_batch(ctx, comp.ctypes.data, ioff.ctypes.data, ilen.ctypes.data, 1, 1, out.ctypes.data, ooff.ctypes.data,
                                 ocap.ctypes.data, olen.ctypes.data, st.ctypes.data)
        dt = time.perf_counter() - t
    assert rc == 0 and st[0] == 0 and bytes(out) == raw
    print("class %d: 1 MiB from %d B: %.2f ms per call" % (cls, len(comp), dt * 1e3))
