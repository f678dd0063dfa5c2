"""ANALYSIS TOOL: a hash of what the CPU emulator build compresses the real fixtures and benchmark shards to (levels 3, 6, 9, whole
shards and ragged sizes) -- a kernel change that must not change a single output byte is checked with this before and after.
usage: [HASH_LEVELS=1,2,6] [HASH_STRATEGIES=0,2,3] python tools/emu_hash.py [size]"""
import hashlib
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import parity_checks  # noqa: E402
import zmi_ctypes  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
o = oracle_lib.load(rebuild=False)
eng = zmi_ctypes.Engine(zmi_ctypes.load_emu(rebuild=False))
blobs = [parity_checks.tile(raw, size) for _, raw in parity_checks.real_fixtures()] + [o.gen_shard(i, size) for i in range(8)]
blobs += [blobs[0][:70001], blobs[3][:4097], blobs[5][:65], blobs[7][:262145], b"", b"a", blobs[2][:300000]]
levels = [int(x) for x in os.environ.get("HASH_LEVELS", "3,6,9").split(",")]
strategies = [int(x) for x in os.environ.get("HASH_STRATEGIES", "0,1,4").split(",")]
for level in levels:
    for strategy in strategies:
        comp, st = eng.deflate(blobs, level=level, wrap=1, strategy=strategy)
        assert all(s == 0 for s in st), st
        h = hashlib.sha256()
        for b, c in zip(blobs, comp):
            assert zlib.decompress(c) == b
            h.update(c)
        print("level %d strategy %d: %d bytes, sha256 %s" % (level, strategy, sum(len(c) for c in comp), h.hexdigest()[:24]))
