#!/usr/bin/env python3
"""deflate() of bench.py's 15.74 MB stream through bench.py's own ctypes loop (no Python-side copies in the timed region), for sweeps
of the tuning knobs: ZMI_TUNING=1 ZMI_ABI_SEGMENT=32768 ZMI_BLOCK_SPAN=8192 python tools/gpu_stream_loop_probe.py"""
import ctypes as C
import os
import sys
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
H.deflate_stream(lib, data[:1 << 20], level=6, wbits=31, chunk_in=1 << 20, chunk_out=1 << 20)
for chunk in (1 << 22, len(data)):
    best = None
    for _ in range(4):
        dt, comp = bench._deflate_loop(H, lib, data, 6, 31, chunk=chunk)
        best = dt if best is None else min(best, dt)
    assert zlib.decompress(comp, 31) == data
    print("segment %s span %s chunk %9d: %.2f ms = %.3f GiB/s  ratio %.4f" % (os.environ.get("ZMI_ABI_SEGMENT", "default"), os.environ.get("ZMI_BLOCK_SPAN", "default"),
                                                                                chunk, best * 1e3, len(data) / 2**30 / best, len(data) / len(comp)))
