#!/usr/bin/env python3
"""deflate() of one 15 MiB stream through libz_mi355.so in 4 MiB pieces, call by call; ZMI_ABI_SEGMENT=<bytes> varies the segment size."""
import ctypes as C
import os
import sys
import time
import zlib

os.environ.setdefault("ZMI_TUNING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib                    # noqa: E402
import zlib_abi_harness as H         # noqa: E402
from zlib_rs_amd import _build       # noqa: E402

lib = H.bind(C.CDLL(_build.ABI_LIB))
o = oracle_lib.load(rebuild=False)
data = b"".join(o.gen_shard(i, 1 << 20) for i in range(15))
H.deflate_stream(lib, data[:1 << 20], level=6, wbits=31, chunk_in=1 << 20, chunk_out=1 << 20)
for rep in range(3):
    t = time.perf_counter()
    comp = H.deflate_stream(lib, data, level=6, wbits=31, chunk_in=1 << 22, chunk_out=1 << 22)
    dt = time.perf_counter() - t
    ok = zlib.decompress(comp, 31) == data
    t = time.perf_counter()
    rc, back, unused = H.inflate_stream(lib, comp, 31, chunk_in=1 << 22, chunk_out=1 << 22)
    di = time.perf_counter() - t
    print("segment %s: deflate %.1f ms = %.3f GiB/s, ratio %.4f, valid %s; inflate back %.1f ms = %.3f GiB/s" % (
        os.environ.get("ZMI_ABI_SEGMENT", "default"), dt * 1e3, len(data) / 2**30 / dt, len(data) / len(comp), ok and back == data,
        di * 1e3, len(data) / 2**30 / di))
