"""Per-call cost of the one-shot stream ABI on the device path (compress2 / uncompress of one buffer): what a caller that
compresses many small buffers sees -- launch + copy + allocation overhead, not kernel throughput.
usage (on an MI355X): python tools/gpu_abi_latency.py        prints  size, compress2 us/call, uncompress us/call, MiB/s"""
import ctypes as C
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
import zlib_abi_harness as H
from zlib_rs_amd import _build

lib = H.bind(C.CDLL(_build.ABI_LIB))
o = oracle_lib.load(rebuild=False)
print("size_bytes,compress2_us,uncompress_us,compress2_MiB_s,uncompress_MiB_s")
for n in (1 << 10, 1 << 14, 1 << 17, 1 << 20, 1 << 22, 1 << 24):
    data = o.gen_shard(0, n)
    cap = C.c_ulong(lib.compressBound(n))
    dst = C.create_string_buffer(cap.value)
    back = C.create_string_buffer(n)
    reps = max(3, min(200, (64 << 20) // n))
    for timed in (False, True):
        t0 = time.perf_counter()
        for _ in range(reps if timed else 2):
            cap.value = len(dst)
            assert lib.compress2(dst, C.byref(cap), data, n, 6) == 0
        tc = (time.perf_counter() - t0) / reps
        comp = dst.raw[:cap.value]
        t0 = time.perf_counter()
        for _ in range(reps if timed else 2):
            ocap = C.c_ulong(n)
            assert lib.uncompress(back, C.byref(ocap), comp, len(comp)) == 0
        tu = (time.perf_counter() - t0) / reps
    assert back.raw == data and zlib.decompress(comp) == data
    print("%d,%.0f,%.0f,%.1f,%.1f" % (n, tc * 1e6, tu * 1e6, n / tc / (1 << 20), n / tu / (1 << 20)))

# ---- thread scaling: N threads, each compressing and expanding its own 1 MiB buffers through compress2 / uncompress.
# Round 1 held one process-wide lock across every call (copy, kernels, device synchronisation, copies): N threads ran
# no faster than one.  Every call now leases a context with its own HIP stream (csrc/zlib_abi.hip AbiLease).
import threading

n = 1 << 20
datas = [o.gen_shard(i % 8, n) for i in range(16)]


def worker(t, calls, out):
    cap = C.c_ulong(lib.compressBound(n))
    dst = C.create_string_buffer(cap.value)
    back = C.create_string_buffer(n)
    ok = True
    for k in range(calls):
        cap.value = len(dst)
        ok &= lib.compress2(dst, C.byref(cap), datas[(t + k) % 16], n, 6) == 0
        ocap = C.c_ulong(n)
        ok &= lib.uncompress(back, C.byref(ocap), dst.raw[:cap.value], cap.value) == 0
        ok &= back.raw == datas[(t + k) % 16]
    out[t] = ok


print("threads,calls_per_thread,seconds,roundtrips_per_s,speedup_vs_1")
base = None
for nt in (1, 2, 4, 8, 16):
    calls = 12
    res = [None] * nt
    ths = [threading.Thread(target=worker, args=(t, calls, res)) for t in range(nt)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    assert all(res), res
    rate = nt * calls / dt
    base = base or rate
    print("%d,%d,%.3f,%.1f,%.2f" % (nt, calls, dt, rate, rate / base))
