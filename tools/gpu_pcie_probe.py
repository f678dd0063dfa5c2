#!/usr/bin/env python3
"""PCIe copy rates with pinned memory of different kinds (hipHostMalloc flags), one way and both ways at once."""
import ctypes as C
import threading
import time

hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
vp = C.c_void_p
hip.hipMalloc.argtypes = [C.POINTER(vp), C.c_size_t]
hip.hipHostMalloc.argtypes = [C.POINTER(vp), C.c_size_t, C.c_uint]
hip.hipMemcpy.argtypes = [vp, vp, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
hip.hipStreamSynchronize.argtypes = [vp]
N = 512 << 20
d1, d2 = vp(), vp()
assert hip.hipMalloc(C.byref(d1), N) == 0 and hip.hipMalloc(C.byref(d2), N) == 0
s1, s2 = vp(), vp()
hip.hipStreamCreateWithFlags(C.byref(s1), 1)
hip.hipStreamCreateWithFlags(C.byref(s2), 1)
for name, flags in (("default", 0), ("noncoherent", 0x80000000), ("coherent", 0x40000000), ("numa-user", 0x20000000), ("write-combined", 0x4)):
    h1, h2 = vp(), vp()
    if hip.hipHostMalloc(C.byref(h1), N, flags) != 0 or hip.hipHostMalloc(C.byref(h2), N, flags) != 0:
        print(name, "alloc failed")
        continue
    C.memset(h1, 1, N); C.memset(h2, 2, N)
    res = {}
    for what, (a, b, kind, st) in (("h2d", (d1, h1, 1, s1)), ("d2h", (h2, d2, 2, s2))):
        hip.hipMemcpyAsync(a, b, N, kind, st); hip.hipStreamSynchronize(st)
        t = time.perf_counter()
        for _ in range(4):
            hip.hipMemcpyAsync(a, b, N, kind, st)
        hip.hipStreamSynchronize(st)
        res[what] = 4 * N / 1e9 / (time.perf_counter() - t)
    t = time.perf_counter()
    for _ in range(4):
        hip.hipMemcpyAsync(d1, h1, N, 1, s1)
        hip.hipMemcpyAsync(h2, d2, N, 2, s2)
    hip.hipStreamSynchronize(s1); hip.hipStreamSynchronize(s2)
    dt = time.perf_counter() - t
    print("%-14s h2d %.1f GB/s  d2h %.1f GB/s  both at once: %.1f GB/s each way (%.1f total)" % (name, res["h2d"], res["d2h"], 4 * N / 1e9 / dt, 8 * N / 1e9 / dt))
