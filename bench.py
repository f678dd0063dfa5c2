#!/usr/bin/env python3
"""bench.py -- GiB/s of raw input compressed at level 6 over 1 MiB synthetic Silesia-like shards.

One process per GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks timing:
shards are independent, so the data path has no collective; the shard-size table is all-gathered
after the timed region).  A step = one deflate pass over this rank's whole
batch of shards, inputs resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (lz77) and `cpu_baseline` (the oracle's level-6 restatement on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GIB = float(1 << 30)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def usable_cores():
    """host cores this process may actually use: affinity mask capped by the cgroup CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:  # noqa: BLE001
            pass
    return max(1, n)


def cpu_baseline(shard_bytes, level, budget_s=15.0):
    """oracle level-`level` deflate (C restatement of the reference) of the same synthetic shards on all
    host cores: POSIX threads inside the oracle library (zo_bench_deflate), bounded sample."""
    import ctypes as C
    import oracle_lib
    o = oracle_lib.load(rebuild=False)
    if not hasattr(o.lib, "zo_bench_deflate"):
        return None
    o.lib.zo_bench_deflate.restype = C.c_double
    o.lib.zo_bench_deflate.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    cores = usable_cores()
    tot = C.c_uint64(0)
    # single thread: 8 shards (one of each class)
    t1 = o.lib.zo_bench_deflate(0x5A4C4942, 0, 8, shard_bytes, level, 1, C.byref(tot))
    one = 8 * shard_bytes / GIB / t1
    # all cores: size the sample to ~budget_s seconds, a multiple of 8 shards per thread
    per_thread = max(8, int(budget_s / (t1 / 8.0)) // 8 * 8)
    n = min(cores * per_thread, 16384, max(cores * 8, int(24 * GIB / shard_bytes)))
    n -= n % 8
    tall = o.lib.zo_bench_deflate(0x5A4C4942, 0, n, shard_bytes, level, cores, C.byref(tot))
    # labelled secondary reference: the system's zlib, compress2(level) of the same 8 shards, one thread
    secondary = None
    try:
        import zlib
        shards = [o.gen_shard(i, shard_bytes) for i in range(8)]
        ts = time.perf_counter()
        csz = sum(len(zlib.compress(d, level)) for d in shards)
        tz = time.perf_counter() - ts
        secondary = {"library": "system zlib " + zlib.ZLIB_RUNTIME_VERSION, "single_thread_GiB_s": 8 * shard_bytes / GIB / tz,
                     "ratio": 8 * shard_bytes / float(csz), "sample": "8 x %d B (classes 0-7)" % shard_bytes}
    except Exception:  # noqa: BLE001
        pass
    return {"value": n * shard_bytes / GIB / tall, "unit": "GiB/s", "cores": cores, "kind": "port", "system_zlib": secondary,
            "sample": "%d x %d B synthetic shards (classes 0-7), oracle zo_deflate level %d, %d POSIX threads, %.1f s"
                      % (n, shard_bytes, level, cores, tall),
            "single_thread_GiB_s": one, "ratio": n * shard_bytes / float(tot.value)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shards", type=int, default=int(os.environ.get("ZMI_BENCH_SHARDS", 65536)), help="shards per GPU")
    ap.add_argument("--shard-bytes", type=int, default=1 << 20)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--verify", type=int, default=64, help="shards checked on the host with the oracle after timing")
    ap.add_argument("--inflate-streams", type=int, default=4096,
                    help="streams of the step's own output inflated on the GPU afterwards (BASELINE.json configs[3]: 4096 x 1 MiB)")
    ap.add_argument("--scratch-gib", type=float, default=float(os.environ.get("ZMI_BENCH_SCRATCH_GIB", 70)),
                    help="device scratch of the engine (4 B per input byte of one launch group): 70 GiB = 16384 shards per launch")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from zlib_rs_amd.engine import Engine, uniform_layout, WRAP_ZLIB

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    e = Engine(local, scratch_bytes=int(args.scratch_gib * GIB))
    S, B = args.shards, args.shard_bytes
    first = rank * S

    data = e.gen_shards(S, B, first_shard=first)
    off, ln = uniform_layout(S, B, dev)
    stride = e.deflate_bound(B, WRAP_ZLIB)
    out = torch.empty((S, stride), dtype=torch.uint8, device=dev)
    olen = torch.empty(S, dtype=torch.int32, device=dev)
    st = torch.empty(S, dtype=torch.int32, device=dev)

    def step():
        e.deflate_batch(data, off, ln, B, level=args.level, wrap=WRAP_ZLIB, out=out, out_len=olen, status=st)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    import ctypes as C
    e.L.zmi_ctx_set_timing(e._ctx, 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sums = (C.c_double * 8)()
    cnts = (C.c_uint32 * 8)()
    e.L.zmi_ctx_get_timing(e._ctx, sums, cnts)
    e.L.zmi_ctx_set_timing(e._ctx, 0)
    from zlib_rs_amd import dist as zdist
    elapsed = zdist.max_over_ranks(elapsed, dev)

    # ---- correctness outside the timed region ----
    assert int((st != 0).sum().item()) == 0, "deflate reported errors"
    csum = olen.to(torch.int64).sum()
    if world > 1:
        # shard-size table exchange (the fixed-size part of the stitch)
        table = zdist.exchange_sizes(olen)                 # [world, S] on every rank
        offs, stitched_total = zdist.stitch_offsets(table)  # byte offset of every shard in the stitched output
        dist.all_reduce(csum)
        assert stitched_total == int(csum.item())
    comp_total = int(csum.item())
    raw_total = S * B * world
    ratio = raw_total / comp_total
    if rank == 0 and args.verify > 0:
        import oracle_lib
        o = oracle_lib.load(rebuild=False)
        idx = sorted(set([0, S - 1] + list(range(7, S, max(1, S // args.verify)))))[:args.verify + 2]
        hl = olen.cpu().numpy()
        for i in idx:
            comp = bytes(out[i, :int(hl[i])].cpu().numpy())
            rc, back, _, msg = o.inflate(comp, B, 1)
            assert rc == 1 and back == o.gen_shard(first + i, B), "round trip failed for shard %d: rc=%d %s" % (i, rc, msg)
    # on-device round trip with the GPU inflater: the compressed shards of the step, back to their input (bit-exact);
    # timed separately (second pass), reported in the "inflate" object -- it is not part of `value`
    nv = max(1, min(S, args.inflate_streams))
    back = torch.empty(nv * B, dtype=torch.uint8, device=dev)
    cap = torch.full((nv,), B, dtype=torch.int32, device=dev)
    ooff = torch.arange(nv, dtype=torch.int64, device=dev) * B
    coff = torch.arange(nv, dtype=torch.int64, device=dev) * out.stride(0)
    olen_v = olen[:nv].contiguous()
    blen, bst = e.inflate_batch(out, coff, olen_v, back, ooff, cap, wrap=WRAP_ZLIB)
    torch.cuda.synchronize()
    assert int((bst != 0).sum().item()) == 0 and torch.equal(back, data[:nv * B]), "device round trip failed"
    e.L.zmi_ctx_set_timing(e._ctx, 1)
    torch.cuda.synchronize()
    ti = time.perf_counter()
    e.inflate_batch(out, coff, olen_v, back, ooff, cap, wrap=WRAP_ZLIB, out_len=blen, status=bst)
    torch.cuda.synchronize()
    inf_s = time.perf_counter() - ti
    isums = (C.c_double * 8)()
    icnts = (C.c_uint32 * 8)()
    e.L.zmi_ctx_get_timing(e._ctx, isums, icnts)
    e.L.zmi_ctx_set_timing(e._ctx, 0)
    inflate_obj = {"streams": nv, "stream_bytes": B, "value": nv * B / GIB / inf_s, "unit": "GiB/s of output", "ms": inf_s * 1e3,
                   "kernel_ms": {"decode": isums[3], "resolve": isums[6], "checksum": isums[0], "verify": isums[4]},
                   "input": "the step's own level-%d zlib streams, output compared bit-exactly with the shards" % args.level}
    del back

    if rank == 0:
        value = raw_total * args.steps / GIB / elapsed
        lz_ms = sums[1] / max(1, cnts[1])
        launches_per_step = max(1, cnts[1] // max(1, args.steps))
        shards_per_launch = S / launches_per_step
        algo_bytes = shards_per_launch * B * (1.0 + 1.0 / ratio)
        achieved = algo_bytes / (lz_ms * 1e-3) / 1e9 if lz_ms > 0 else 0.0
        # measured HBM traffic of the dominant kernel (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes,
        # tools/prof_final.sh) -- recorded per 2048-shard launch in profiles/, scaled to this run's launch size
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))["zmi_lz77_kernel"]
            traffic = (tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"]) * shards_per_launch / tj["shards_per_launch"]
        except Exception:  # noqa: BLE001
            pass
        line = {
            "metric": "GiB/s raw input compressed (level %d, 1 MiB shards)" % args.level,
            "value": value, "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d x %d B synthetic Silesia-like shards per GPU, level %d, zlib wrapper, round-trip verified"
                                   % (S, B, args.level), "shards_per_gpu": S, "shard_bytes": B, "level": args.level,
                       "parallelism": "shard-parallel x%d (no data-path collective)" % world},
            "ratio": ratio,
            "roofline": {"bound": "hbm", "kernel": "zmi_lz77_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "read_only_frac": value * GIB / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms": {"checksum": sums[0] / max(1, cnts[0]), "lz77": lz_ms, "encode": sums[2] / max(1, cnts[2])},
                         "launches_per_step": int(launches_per_step)},
            "inflate": inflate_obj,
        }
        if world == 1 and not args.no_cpu:
            cb = cpu_baseline(B, args.level)
            if cb is not None:
                line["cpu_baseline"] = cb
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    e.close()


if __name__ == "__main__":
    main()
