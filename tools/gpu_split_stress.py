#!/usr/bin/env python3
"""Randomised streams with flush points through inflate() of libz_mi355.so (segment-parallel decode) for STRESS_SECONDS,
system zlib as the judge; exits non-zero on the first difference."""
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
budget = float(os.environ.get("STRESS_SECONDS", "120"))
seed = int(os.environ.get("STRESS_SEED", "5000"))
t0 = time.time()
n = 0
while time.time() - t0 < budget:
    n += H.flush_point_stream_checks(lib, o, seeds=range(seed, seed + 4), big=True)
    seed += 4
print("split stress ok: %d cases in %.0f s, next seed %d" % (n, time.time() - t0, seed))
