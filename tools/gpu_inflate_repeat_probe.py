"""GPU probe: the stream ABI's inflate() of one 15.74 MB stream, the same call sequence repeated -- does a process slow down
over repeated inflateInit / inflate / inflateEnd cycles?  (round 6: the median of five runs read 1.0 GiB/s where best-of-two read 1.65)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import zlib_abi_harness as H  # noqa: E402
from zlib_rs_amd import _build  # noqa: E402

lib = H.bind(C.CDLL(_build.ABI_LIB))
o = bench._oracle()
data = b"".join(o.gen_shard(i, 1 << 20) for i in range(15))
rc, ocomp = o.deflate(data, 6, 2)
dt, comp = bench._deflate_loop(H, lib, data, 6, 31)
for name, c in (("own", comp), ("cpu-made", ocomp)):
    ts = []
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        dt, rc, back, unused = bench._inflate_loop(H, lib, c, 31, len(data))
        assert rc == 1 and back == data
        ts.append(dt * 1e3)
    print(name, " ".join("%.2f" % t for t in ts), "ms")
