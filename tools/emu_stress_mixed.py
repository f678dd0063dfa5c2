"""Deflate stress on the CPU emulator with shards stitched from runs of very different compressibility (random / zero / high bytes /
periodic / generator classes), run lengths around the piece size (shard / 16): status 0, system zlib and the engine's own inflate read it back,
zmi_deflate_bound holds.  usage: python tools/emu_stress_mixed.py SEED SECONDS   (test infrastructure)"""
import os, sys, zlib, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import zmi_ctypes, oracle_lib
o = oracle_lib.load(False)
eng = zmi_ctypes.Engine(zmi_ctypes.load_emu(False))
rnd = random.Random(int(sys.argv[1])); t = time.time(); rounds = 0
wb = {0: -15, 1: 15, 2: 31}
while time.time() - t < float(sys.argv[2]):
    shards = []
    for i in range(rnd.randrange(1, 4)):
        ln = rnd.choice([4096, 65536, 70000, 131072, rnd.randrange(1, 200000)])
        parts = []; tot = 0
        while tot < ln:
            k = rnd.choice([1, 7, 100, ln // 16, ln // 16 + 1, ln // 16 - 1, rnd.randrange(1, 30000)]); k = max(1, k)
            kind = rnd.randrange(5)
            if kind == 0: p = os.urandom(k)
            elif kind == 1: p = bytes(k)
            elif kind == 2: p = bytes(rnd.randrange(200, 256) for _ in range(min(k, 3000))) * (k // 3000 + 1)
            elif kind == 3: p = (b"abcdefghij" * (k // 10 + 1))
            else: p = o.gen_shard(rnd.randrange(8), k)
            parts.append(p[:k]); tot += k
        shards.append(b"".join(parts)[:ln])
    lvl = rnd.randrange(0, 10); strat = rnd.choice([0, 0, 1, 2, 3, 4]); wrap = rnd.randrange(3)
    outs, st = eng.deflate(shards, level=lvl, strategy=strat, wrap=wrap)
    assert st == [0] * len(shards), (st, lvl, strat, [len(s) for s in shards])
    for c, d in zip(outs, shards):
        assert zlib.decompressobj(wb[wrap]).decompress(c) == d
        assert len(c) <= eng.lib.zmi_deflate_bound(len(d), wrap)
    back, st2 = eng.inflate(outs, [len(d) for d in shards], wrap)
    assert st2 == [0] * len(shards) and back == shards
    rounds += 1
print("adversarial mix ok:", rounds, "rounds")
