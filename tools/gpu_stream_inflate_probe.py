#!/usr/bin/env python3
"""bench.py's CPU-made stream (15.74 MB, no flush points) through uncompress() and through inflate() in 4 MiB pieces, twice each --
meant to run under `rocprofv3 --kernel-trace --memory-copy-trace` to see where a call's time goes on the device."""
import ctypes as C
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench                         # noqa: E402
import zlib_abi_harness as H         # noqa: E402
from zlib_rs_amd import _build       # noqa: E402

lib = H.bind(C.CDLL(_build.ABI_LIB))
o = bench._oracle()
total = 15740000
data = b"".join(o.gen_shard(i, 1 << 20) for i in range(15))
data += o.gen_shard(15, 1 << 20)[:total - len(data)]
zc = zlib.compress(data, 6)
dst = C.create_string_buffer(len(data))
for rep in range(3):
    dl = C.c_ulong(len(data))
    t0 = time.perf_counter()
    rc = lib.uncompress(dst, C.byref(dl), zc, len(zc))
    dt = time.perf_counter() - t0
    assert rc == 0 and dst.raw[:len(data)] == data
    print("uncompress: %.2f ms = %.3f GiB/s" % (dt * 1e3, len(data) / 2**30 / dt))
rc, oc = o.deflate(data, 6, 2)
for rep in range(3):
    dt, rc2, back, unused = bench._inflate_loop(H, lib, oc, 31, len(data))
    assert rc2 == 1 and back == data
    print("inflate() 4 MiB pieces of the oracle's gzip stream: %.2f ms = %.3f GiB/s" % (dt * 1e3, len(data) / 2**30 / dt))
