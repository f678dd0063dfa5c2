#!/usr/bin/env python3
"""Ratio probe on the CPU emulator build: lcet10.txt (real English text) and 8 synthetic shards at one level, with
whatever ZMI_* tuning variables the environment carries.  Used to tune the level table on something other than the
synthetic generator ("tuned on the synthetic Zipf generator only")."""
import os
import sys

os.environ.setdefault("ZMI_TUNING", "1")   # the ZMI_* overrides are honoured only with this set
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib      # noqa: E402
import parity_checks   # noqa: E402
import zmi_ctypes      # noqa: E402

o = oracle_lib.load()
e = zmi_ctypes.Engine(zmi_ctypes.load_emu(rebuild=False))
fx = parity_checks.real_fixtures()
syn = [o.gen_shard(i, 1 << 17) for i in range(8)]
lvl = int(os.environ.get("LVL", "6"))
blobs = [raw for _, raw in fx[:2]] + syn
outs, st = e.deflate(blobs, level=lvl, wrap=2)
for b, c in zip(blobs, outs):
    assert zlib.decompress(c, 31) == b
print("%-40s L%d lcet10 %.4f  paper %.4f  synthetic %.4f" % (os.environ.get("TAG", ""), lvl, len(blobs[0]) / len(outs[0]),
                                                           len(blobs[1]) / len(outs[1]), sum(map(len, syn)) / sum(map(len, outs[2:]))))
