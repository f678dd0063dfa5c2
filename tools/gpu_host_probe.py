#!/usr/bin/env python3
"""Torch-free probe of the host-buffer path (zmi_deflate_batch / zmi_inflate_batch: pageable host memory in and out).
    ZMI_HOST_THREADS=12 python tools/gpu_host_probe.py --shards 4096
Prints GiB/s of raw data per call (best of --reps), the device-resident rate of the same shards beside it."""
import argparse
import ctypes as C
import os
import sys
import time

if "--json" not in sys.argv:
    os.environ.setdefault("ZMI_TUNING", "1")   # (the --json form is bench.py's leg: the product configuration, no overrides)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_fast_probe as P  # noqa: E402
import numpy as np  # noqa: E402


def main():
    if os.environ.get("PROBE_WITH_TORCH"):   # (does a process that also holds torch's runtime see the same rates?)
        import torch
        torch.zeros(1 << 20, device="cuda").sum().item()
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=4096)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--scratch-gib", type=float, default=0.0, help="the context's scratch limit (bench.py: 70)")
    ap.add_argument("--hog-gib", type=float, default=0.0, help="device memory held beside the run (bench.py holds ~206 GiB)")
    ap.add_argument("--json", action="store_true", help="one warm-up call, then --reps timed calls of each direction; prints one JSON object with medians (bench.py's pcie_inclusive leg)")
    ap.add_argument("--empty-out", action="store_true", help="np.empty output buffers (pages never touched before the first call), as bench.py has them")
    a = ap.parse_args()
    L, path = P.load_lib()
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.zmi_deflate_batch.argtypes = [vp, vp, vp, vp, u32, i32, i32, i32, vp, u64, vp, vp]
    L.zmi_inflate_batch.argtypes = [vp, vp, vp, vp, u32, i32, vp, vp, vp, vp, vp]
    ctx = C.c_void_p()
    assert L.zmi_ctx_create(C.byref(ctx), 0) == 0
    if a.scratch_gib > 0:
        L.zmi_ctx_set_scratch_limit.argtypes = [vp, u64]
        assert L.zmi_ctx_set_scratch_limit(ctx, int(a.scratch_gib * 2**30)) == 0
    hog = [P.dmalloc(1 << 30) for _ in range(int(a.hog_gib))]
    B, S = 1 << 20, a.shards
    stride = int(L.zmi_deflate_bound(B, 1))
    h_in = np.empty(S * B, dtype=np.uint8)
    piece = min(S, 1024)                      # (1 GiB of device memory at a time: bench.py runs this beside its own tensors)
    d_in = P.dmalloc(piece * B)
    for s0 in range(0, S, piece):
        cnt = min(piece, S - s0)
        L.zmi_gen_shards_dev(ctx, d_in, 0x5A4C4942, s0, cnt, B, None)
        P.hip.hipDeviceSynchronize()
        P.ck(P.hip.hipMemcpy(h_in.ctypes.data + s0 * B, d_in, cnt * B, 2), "d2h")
    P.hip.hipFree(d_in)
    h_off = np.arange(S, dtype=np.uint64) * B
    h_len = np.full(S, B, dtype=np.uint32)
    h_out = np.empty(S * stride, dtype=np.uint8) if a.empty_out else np.zeros(S * stride, dtype=np.uint8)
    h_olen = np.zeros(S, dtype=np.uint32)
    h_st = np.zeros(S, dtype=np.int32)
    best = None
    tdef, tinf = [], []
    for _ in range(a.reps + (1 if a.json else 0)):
        t = time.perf_counter()
        rc = L.zmi_deflate_batch(ctx, h_in.ctypes.data, h_off.ctypes.data, h_len.ctypes.data, S, a.level, 0, 1, h_out.ctypes.data, stride,
                                 h_olen.ctypes.data, h_st.ctypes.data)
        dt = time.perf_counter() - t
        assert rc == 0 and not h_st.any(), L.zmi_last_error()
        best = dt if best is None else min(best, dt)
        tdef.append(dt)
    if not a.json:
        print("deflate host path: %d shards %.1f ms = %.2f GiB/s (threads env %s)" % (S, best * 1e3, S * B / 2**30 / best, os.environ.get("ZMI_HOST_THREADS", "default")))
    # inflate back through the host path
    c_off = np.arange(S, dtype=np.uint64) * stride
    o_off = np.arange(S, dtype=np.uint64) * B
    o_cap = np.full(S, B, dtype=np.uint32)
    h_back = np.zeros(S * B, dtype=np.uint8)
    b_len = np.zeros(S, dtype=np.uint32)
    b_st = np.zeros(S, dtype=np.int32)
    best = None
    for _ in range(a.reps + (1 if a.json else 0)):
        t = time.perf_counter()
        rc = L.zmi_inflate_batch(ctx, h_out.ctypes.data, c_off.ctypes.data, h_olen.ctypes.data, S, 1, h_back.ctypes.data, o_off.ctypes.data,
                                 o_cap.ctypes.data, b_len.ctypes.data, b_st.ctypes.data)
        dt = time.perf_counter() - t
        assert rc == 0 and not b_st.any(), L.zmi_last_error()
        best = dt if best is None else min(best, dt)
        tinf.append(dt)
    assert np.array_equal(h_back, h_in)
    if a.json:
        import json

        def rate(ts):
            ts = sorted(ts[1:])                  # (the first call pays for the staging buffers)
            g = S * B / 2**30
            return {"median": g / ts[len(ts) // 2], "min": g / ts[-1], "max": g / ts[0], "runs": len(ts)}
        print(json.dumps({"shards": S, "deflate": rate(tdef), "inflate": rate(tinf), "ratio": S * B / float(h_olen.astype(np.int64).sum()),
                          "round_trip": "bit-exact", "process": "a process of its own, without torch: the HIP runtime is the system's (/opt/rocm), as for any "
                          "caller of libzmi355.so -- inside bench.py's process the library binds to the runtime torch ships"}))
        return
    print("inflate host path: %d streams %.1f ms = %.2f GiB/s of output" % (S, best * 1e3, S * B / 2**30 / best))
    # host memcpy bandwidth for reference (one thread)
    t = time.perf_counter()
    tmp = h_in.copy()
    dt = time.perf_counter() - t
    print("numpy copy of the input (one thread): %.2f GiB/s; cores %d" % (S * B / 2**30 / dt, os.cpu_count()))


if __name__ == "__main__":
    main()
