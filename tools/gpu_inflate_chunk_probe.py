#!/usr/bin/env python3
"""inflate() of one CPU-made gzip stream (no flush points) through libz_mi355.so in pieces of PROBE_CHUNK bytes (default 65536),
the whole output buffer available -- the loop of tools/chunk_sweep.c from Python, for tracing one chunk size with rocprofv3."""
import ctypes as C
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib                    # noqa: E402
import zlib_abi_harness as H         # noqa: E402
from zlib_rs_amd import _build       # noqa: E402

lib = H.bind(C.CDLL(_build.ABI_LIB))
o = oracle_lib.load(rebuild=False)
data = b"".join(o.gen_shard(i, 1 << 20) for i in range(4))
co = zlib.compressobj(6, zlib.DEFLATED, 31)
comp = co.compress(data) + co.flush()
chunk = int(os.environ.get("PROBE_CHUNK", "65536"))
for rep in range(3):
    strm = H.ZStream()
    assert lib.inflateInit2_(C.byref(strm), 31, lib.zlibVersion(), C.sizeof(H.ZStream)) == 0
    src = C.create_string_buffer(comp, len(comp))
    dst = C.create_string_buffer(len(data) + 64)
    strm.next_out, strm.avail_out = C.addressof(dst), len(dst)
    t = time.perf_counter()
    rc, calls = 0, 0
    for pos in range(0, len(comp), chunk):
        strm.next_in, strm.avail_in = C.addressof(src) + pos, min(chunk, len(comp) - pos)
        rc = lib.inflate(C.byref(strm), 0)
        calls += 1
        if rc != 0:
            break
    dt = time.perf_counter() - t
    ok = rc == 1 and dst.raw[:len(data)] == data
    lib.inflateEnd(C.byref(strm))
    print("chunk %d: %d calls, %.2f ms = %.3f GiB/s of output, %.1f us per call, ok %s" % (chunk, calls, dt * 1e3, len(data) / 2**30 / dt, dt * 1e6 / calls, ok))
