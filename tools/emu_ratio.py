"""ANALYSIS TOOL: level / parse comparison on the CPU emulator build -- compressed sizes of the real fixtures and of benchmark
shards, every stream checked by zlib.  usage: ZMI_TUNING=1 python tools/emu_ratio.py [level] [size]"""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ZMI_TUNING", "1")
import oracle_lib  # noqa: E402
import parity_checks  # noqa: E402
import zmi_ctypes  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
o = oracle_lib.load(rebuild=False)
eng = zmi_ctypes.Engine(zmi_ctypes.load_emu(rebuild=False))
blobs = [(name, parity_checks.tile(raw, size)) for name, raw in parity_checks.real_fixtures()] + [("shard%d" % i, o.gen_shard(i, size)) for i in range(8)]
res = {}
for cp in ("0", "1"):
    os.environ["ZMI_COST_PARSE"] = cp
    comp, st = eng.deflate([b for _, b in blobs], level=level, wrap=1)
    assert all(s == 0 for s in st), st
    for (name, b), c in zip(blobs, comp):
        assert zlib.decompress(c) == b, name
    res[cp] = [len(c) for c in comp]
tot = {cp: sum(res[cp][3:]) for cp in res}
for i, (name, b) in enumerate(blobs):
    rc, oc, _ = o.deflate(b, level, 1) if False else (0, None, None)
    print("%-16s lazy %8d (%.4f)  cost %8d (%.4f)  %+.2f %%" % (name, res["0"][i], len(b) / res["0"][i], res["1"][i], len(b) / res["1"][i],
                                                            100.0 * (res["0"][i] / res["1"][i] - 1.0)))
print("benchmark mix (shards 0-7): lazy %.4f  cost %.4f" % (8 * size / tot["0"], 8 * size / tot["1"]))
