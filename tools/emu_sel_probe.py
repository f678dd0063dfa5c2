#!/usr/bin/env python3
"""Emulator experiment: ratio and the share of positions that get the deep chain walk under the ZMI_SEL* selection rules
(lz77.hip).  Usage: LVL=6 python tools/emu_sel_probe.py 'ZMI_SEL=2 ZMI_SEL_LEN=16' 'ZMI_SEL=1' ...  (one process per setting)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("_SEL_CHILD"):
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
    outs, st = e.deflate(blobs, level=lvl, wrap=2)
    for b, c in zip(blobs, outs):
        assert zlib.decompress(c, 31) == b
    d, a = C.c_uint64(), C.c_uint64()
    try:
        lib.zmi_emu_sel_counts(C.byref(d), C.byref(a))
    except AttributeError:
        pass
    vr, va = C.c_uint64(), C.c_uint64()
    lib.zmi_emu_vis_counts(C.byref(vr), C.byref(va))
    lib.zmi_emu_hops.restype = C.c_uint64
    print("visited candidates %d, real (>= 6 bytes) %.3f, hops %d" % (va.value, vr.value / max(1, va.value), lib.zmi_emu_hops()))
    print("%-44s L%d lcet10 %.4f paper %.4f syn %.4f [%s] deep %.3f" % (
        os.environ.get("TAG", ""), lvl, len(blobs[0]) / len(outs[0]), len(blobs[1]) / len(outs[1]), sum(map(len, syn)) / sum(map(len, outs[2:])),
        " ".join("%.2f" % (len(s) / len(c)) for s, c in zip(syn, outs[2:])), d.value / max(1, a.value)))
    sys.exit(0)
for setting in sys.argv[1:] or [""]:
    env = dict(os.environ, _SEL_CHILD="1", TAG=setting)
    for kv in setting.split():
        k, v = kv.split("=")
        env[k] = v
    subprocess.run([sys.executable, os.path.abspath(__file__)], env=env)
