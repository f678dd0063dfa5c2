"""Randomised stress of the zlib stream ABI (libz_mi355.so) on the GPU: random sizes, chunkings, flush patterns,
wrappers, levels, dictionaries; every result is checked against system zlib."""
import ctypes as C
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import zlib_abi_harness as H  # noqa: E402
from zlib_rs_amd import _build  # noqa: E402


def main():
    budget = float(os.environ.get("STRESS_SECONDS", "120"))
    rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "2")))
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    o = oracle_lib.load(rebuild=False)
    pool = b"".join(o.gen_shard(c, 1 << 20) for c in range(8))
    t0 = time.time()
    rounds = 0
    while time.time() - t0 < budget:
        ln = int(rng.choice([0, 1, 100, 5000, 70000, int(rng.integers(0, 3 << 20))]))
        at = int(rng.integers(0, len(pool) - ln + 1))
        data = pool[at:at + ln]
        wbits = int(rng.choice([15, -15, 31]))
        level = int(rng.integers(0, 10))
        chunk_in = int(rng.choice([ln or 1, 1 << 16, int(rng.integers(1, max(2, ln + 1)))]))
        if ln > 200000 and chunk_in < 4096:
            chunk_in = 4096 + chunk_in      # keep the number of ABI calls per round bounded
        flush_every = int(rng.choice([0, 0, 1, 3]))
        comp = H.deflate_stream(lib, data, level=level, wbits=wbits, chunk_in=chunk_in, chunk_out=int(rng.choice([4096, 65536, 1 << 20])),
                                flush_every=flush_every or None)
        assert zlib.decompress(comp, wbits) == data, ("deflate", ln, wbits, level, chunk_in, flush_every)
        co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, wbits)
        ref = co.compress(data) + co.flush()
        for stream in (comp, ref):
            ci = int(rng.choice([1 << 30, 1 << 16, int(rng.integers(1, max(2, len(stream) + 1)))]))
            if len(stream) > 100000 and ci < 2048:
                ci += 2048
            rc, back, unused = H.inflate_stream(lib, stream, wbits=wbits, chunk_in=ci, chunk_out=int(rng.choice([8192, 1 << 20])))
            assert rc == H.Z_STREAM_END and back == data and unused == 0, ("inflate", ln, wbits, ci, rc)
        if ln >= 2000 and wbits != 31 and rng.random() < 0.3:
            H.dictionary_checks(lib, data[ln // 2:ln // 2 + min(ln // 2, 300000)], data[:min(ln // 2, 40000)], expect_gain=False)
        rounds += 1
    print("abi stress ok: %d rounds in %.0f s" % (rounds, time.time() - t0))


if __name__ == "__main__":
    main()
