#!/usr/bin/env python3
"""Where does a streaming inflate() spend its time?  One 15.7 MB gzip stream through inflate() of libz_mi355.so in 4 MiB
pieces (the loop of bench.py's stream_abi leg), every call timed; the one-shot uncompress() of the same data beside it."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib                    # noqa: E402
import zlib_abi_harness as H         # noqa: E402
from zlib_rs_amd import _build       # noqa: E402

lib = H.bind(C.CDLL(_build.ABI_LIB))
o = oracle_lib.load(rebuild=False)
data = b"".join(o.gen_shard(i, 1 << 20) for i in range(15))
comp_own = H.deflate_stream(lib, data, level=6, wbits=31, chunk_in=1 << 22, chunk_out=1 << 22)
import zlib as _z
_co = _z.compressobj(6, _z.DEFLATED, 31)
comp_cpu = _co.compress(data) + _co.flush()
for rep in range(4):
    comp = comp_own if rep < 2 else comp_cpu
    print("own stream (flush points)" if rep < 2 else "CPU-made stream (no flush points)")
    strm = H.ZStream()
    assert lib.inflateInit2_(C.byref(strm), 31, lib.zlibVersion(), C.sizeof(H.ZStream)) == 0
    src = C.create_string_buffer(comp, len(comp))
    out = C.create_string_buffer(1 << 22)
    pos, calls, t_all = 0, [], time.perf_counter()
    got = 0
    while True:
        if strm.avail_in == 0 and pos < len(comp):
            n = min(1 << 22, len(comp) - pos)
            strm.next_in, strm.avail_in = C.addressof(src) + pos, n
            pos += n
        strm.next_out, strm.avail_out = C.addressof(out), len(out)
        t = time.perf_counter()
        rc = lib.inflate(C.byref(strm), 0)
        calls.append((time.perf_counter() - t, len(out) - strm.avail_out, strm.avail_in))
        got += len(out) - strm.avail_out
        if rc != 0:
            break
    dt = time.perf_counter() - t_all
    lib.inflateEnd(C.byref(strm))
    print("rep %d: rc %d, %d bytes in %.1f ms = %.3f GiB/s; calls (ms, out, avail_in): %s" % (rep, rc, got, dt * 1e3, got / 2**30 / dt,
          " ".join("%.1f/%d/%d" % (c[0] * 1e3, c[1] >> 10, c[2] >> 10) for c in calls)))
import zlib
z = zlib.compress(data, 6)
dst = C.create_string_buffer(len(data))
for rep in range(2):
    dl = C.c_ulong(len(data))
    t = time.perf_counter()
    rc = lib.uncompress(dst, C.byref(dl), z, len(z))
    print("uncompress: rc %d %.1f ms = %.3f GiB/s" % (rc, (time.perf_counter() - t) * 1e3, len(data) / 2**30 / (time.perf_counter() - t)))
