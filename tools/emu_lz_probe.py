#!/usr/bin/env python3
"""Emulator experiment: ratio and the searcher's step counts (chain steps / extension rounds, per lane and per wave) under
ZMI_* tuning settings.  Usage: LVL=6 python tools/emu_lz_probe.py 'ZMI_CHAIN=6 ZMI_MIN_LIVE=16' '' ...  (one process per setting)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("_LZ_CHILD"):
    os.environ.setdefault("ZMI_TUNING", "1")
    import zlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib, parity_checks, zmi_ctypes
    o = oracle_lib.load()
    lib = zmi_ctypes.load_emu(rebuild=False)
    e = zmi_ctypes.Engine(lib)
    fx = parity_checks.real_fixtures()
    syn = [o.gen_shard(i, 1 << 17) for i in range(8)]
    lvl = int(os.environ.get("LVL", "6"))
    blobs = [raw for _, raw in fx[:2]] + syn
    cnt = (C.c_uint64 * 8)()
    res = []
    for b in blobs:
        lib.zmi_emu_lz_counts(cnt, 1)
        outs, st = e.deflate([b], level=lvl, wrap=2)
        assert zlib.decompress(outs[0], 31) == b
        lib.zmi_emu_lz_counts(cnt, 1)
        res.append((len(b), len(outs[0]), list(cnt)))
    def fmt(r):
        n, c, k = r
        return "%.3f w%.2f x%.2f" % (n / c, k[1] / max(1, k[4]), k[3] / max(1, k[4]))
    tot = [sum(r[2][i] for r in res[2:]) for i in range(8)]
    print("%-40s L%d lcet10 %s | paper %s | syn %.4f w%.2f x%.2f lane-util %.2f\n    per class: %s" % (
        os.environ.get("TAG", ""), lvl, fmt(res[0]), fmt(res[1]), sum(r[0] for r in res[2:]) / sum(r[1] for r in res[2:]),
        tot[1] / max(1, tot[4]), tot[3] / max(1, tot[4]), tot[0] / max(1, 64 * tot[1]), "  ".join(fmt(r) for r in res[2:])))
    sys.exit(0)
for setting in sys.argv[1:] or [""]:
    env = dict(os.environ, _LZ_CHILD="1", TAG=setting)
    for kv in setting.split():
        k, v = kv.split("=")
        env[k] = v
    subprocess.run([sys.executable, os.path.abspath(__file__)], env=env)
