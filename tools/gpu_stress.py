"""Randomised round-trip stress on the GPU (run through gpurun): STRESS_SECONDS of ragged batches both ways against
system zlib.  Prints one line per round and a summary; exits non-zero on the first mismatch."""
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from zlib_rs_amd.engine import Engine  # noqa: E402
import test_gpu_parity as T  # noqa: E402  (helpers only)


def main():
    budget = float(os.environ.get("STRESS_SECONDS", "240"))
    seed = int(os.environ.get("STRESS_SEED", "1"))
    rng = np.random.default_rng(seed)
    e = Engine(0)
    pool = b"".join(T._gen(e, 16, 1 << 20))
    t0 = time.time()
    rounds = total = 0
    while time.time() - t0 < budget:
        big = rng.random() < 0.2
        n = int(rng.integers(1, 12)) if big else int(rng.integers(10, 400))
        shards = []
        for _ in range(n):
            hi = 4 << 20 if big else 200000
            ln = int(rng.choice([0, 1, 3, 4, 5, 63, 64, 65, 1023, 1024, 1025, 32767, 32768, 32769, 65535, 65536, 65537,
                                 int(rng.integers(0, hi)), int(rng.integers(0, hi))]))
            at = int(rng.integers(0, len(pool) - ln + 1))
            s = pool[at:at + ln]
            kind = rng.random()
            if kind < 0.1:
                s = bytes(ln)                                   # runs
            elif kind < 0.2:
                s = (s[:max(1, ln // 7)] * 8)[:ln]             # long-distance repeats
            elif kind < 0.3:
                s = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()   # incompressible
            shards.append(s)
        level, wrap, strategy = int(rng.integers(0, 10)), int(rng.integers(0, 3)), int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
        outs, st = T._deflate(e, shards, level=level, wrap=wrap, strategy=strategy)
        assert (st == 0).all(), ("deflate status", level, wrap, strategy)
        for s, o in zip(shards, outs):
            assert T._dec(o, wrap) == s, ("system zlib disagrees", level, wrap, strategy, len(s))
        back, st2 = T._inflate(e, outs, [len(s) for s in shards], wrap=wrap)
        assert (st2 == 0).all() and back == shards, ("gpu inflate of gpu deflate", level, wrap, strategy)
        wbits = int(rng.integers(9, 16))
        zs = []
        for s in shards:
            co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, {0: -wbits, 1: wbits, 2: 16 + wbits}[wrap], 8,
                                  int(rng.choice([0, 1, 2, 3, 4])))
            zs.append(co.compress(s) + co.flush())
        back, st3 = T._inflate(e, zs, [len(s) for s in shards], wrap=wrap)
        assert (st3 == 0).all() and back == shards, ("gpu inflate of zlib streams", wrap, wbits)
        rounds += 1
        total += sum(len(s) for s in shards)
        print("round %d: %d shards, %.1f MiB, level %d wrap %d strategy %d ok" % (rounds, n, sum(map(len, shards)) / 2**20, level, wrap,
                                                                              strategy), flush=True)
    print("stress ok: %d rounds, %.1f MiB each way, %.0f s, seed %d" % (rounds, total / 2**20, time.time() - t0, seed))
    e.close()


if __name__ == "__main__":
    main()
