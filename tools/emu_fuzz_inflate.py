"""Mutation fuzz of the batch inflate on the CPU SIMT emulator (tests/emu), system zlib as the judge -- the idea of the
reference's fuzz/fuzz_targets/uncompress.rs (arbitrary bytes into inflate must never crash, hang or write out of bounds)
plus agreement: a stream zlib accepts must decode to the same bytes, a stream zlib rejects must be rejected.
usage: python tools/emu_fuzz_inflate.py SEED SECONDS   (test infrastructure; the product path needs an MI355X)"""
import os
import random
import sys
import time
import zlib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib
import zmi_ctypes

o = oracle_lib.load()
eng = zmi_ctypes.Engine(zmi_ctypes.load_emu(False))
rnd = random.Random(int(sys.argv[1]))
WB = {0: -15, 1: 15, 2: 31}
t0, rounds, accepted, rejected = time.time(), 0, 0, 0
while time.time() - t0 < float(sys.argv[2]):
    wrap = rnd.randrange(3)
    streams, caps, want = [], [], []
    for _ in range(rnd.randrange(1, 5)):
        n = rnd.choice([0, 5, 300, 5000, rnd.randrange(40000), rnd.randrange(40000), 40000 + rnd.randrange(260000)])   # the larger ones reach the decode kernel's lane-serial fast pass (>= 4 KiB of input)
        d = o.gen_shard(rnd.randrange(8), n) if rnd.random() < 0.8 else bytes(rnd.randrange(3) for _ in range(n))
        co = zlib.compressobj(rnd.choice([0, 1, 6, 9]), zlib.DEFLATED, WB[wrap], 8, rnd.choice([0, 0, 2, 3, 4]))
        c = bytearray(co.compress(d) + co.flush())
        kind = rnd.random()
        if kind < 0.45 and c:                       # flip bits
            for _ in range(rnd.choice([1, 1, 2, 8])):
                c[rnd.randrange(len(c))] ^= 1 << rnd.randrange(8)
        elif kind < 0.6 and c:                      # truncate
            del c[rnd.randrange(len(c)):]
        elif kind < 0.7:                            # garbage
            c = bytearray(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 400)))
        elif kind < 0.8 and len(c) > 8:             # splice two halves
            k = rnd.randrange(len(c))
            c = c[:k] + c[rnd.randrange(len(c)):]
        cap = len(d) + rnd.choice([0, 0, 64, 70000])
        dec = zlib.decompressobj(WB[wrap])
        try:
            out = dec.decompress(bytes(c), cap + 1)
            ok = dec.eof and len(out) <= cap
        except zlib.error:
            out, ok = b"", False
        streams.append(bytes(c)); caps.append(cap); want.append((ok, out))
    back, st = eng.inflate(streams, caps, wrap)
    for i, (ok, out) in enumerate(want):
        if ok:
            assert st[i] == 0 and back[i] == out, (rounds, i, st[i], len(out), len(back[i]))
            accepted += 1
        else:
            assert st[i] != 0, (rounds, i, "zlib rejects, engine accepts", streams[i][:40].hex())
            rejected += 1
    rounds += 1
print("emu inflate fuzz ok: %d rounds, %d accepted, %d rejected, seed %s" % (rounds, accepted, rejected, sys.argv[1]))
