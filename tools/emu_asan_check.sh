#!/bin/bash
# The kernels and the host ABI code on the CPU emulator under AddressSanitizer: LDS / global overruns, host buffer
# arithmetic (gzungetc, the inflate checkpoint bookkeeping).  Test infrastructure; takes about two minutes.
set -e
cd "$(dirname "$0")/.."
make -s -C tests/emu libzmi355_emu_asan.so
ASAN=$(gcc -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0
export ZMI_NO_ALLOC_FAULTS=1
export ZMI_TUNING=1 ZMI_ABI_SPLIT_MIN=20000 ZMI_ABI_SPLIT_GAP=1500   # (streams of the tests' sizes are cut at their flush points)
LD_PRELOAD=$ASAN python - <<'PY'
import ctypes as C, json, os, sys, tempfile, zlib
sys.path.insert(0, "tests")
import oracle_lib, resume_checks, zlib_abi_harness as H, zmi_ctypes
so = os.path.join("tests", "emu", "libzmi355_emu_asan.so")
o = oracle_lib.load()
lib = H.bind(C.CDLL(so))
with tempfile.TemporaryDirectory() as d:
    H.gz_checks(lib, d, o.gen_shard(1, 60000))
H.streaming_checks(lib, o.gen_shard(0, 40000) + o.gen_shard(3, 30000))
for seed in range(10):
    H.random_streaming_roundtrips(lib, o, 4, seed, max_len=30000)
H.golden_inflate_checks(lib, json.load(open("tests/golden/inflate_vectors.json")))
os.environ["ZMI_TUNING"] = "1"
os.environ["ZMI_ABI_SEGMENT"] = "4096"
H.run_abi_checks(lib, o, sizes=(0, 1, 100, 5000, 20000))
H.header_copy_checks(lib, o.gen_shard(2, 40000))
H.misc_symbol_checks(lib, o)                      # device-buffer pool reuse with changing sizes, allocators, bounds
H.config_matrix_roundtrips(lib, o, 25, seed=3, max_len=40000)
H.threaded_roundtrips(lib, o, threads=3, rounds=2)
eng = zmi_ctypes.Engine(zmi_ctypes._bind(C.CDLL(so)))
resume_checks.resume_chain_checks(eng, o, sizes=(60000, 40000, 20000, 20000), trials=2)
shards = [o.gen_shard(i % 8, n) for i, n in enumerate([0, 1, 15, 16, 17, 1000, 4096, 9999, 20000] * 3)]
for lvl in (1, 6, 9):
    for strat in (0, 2, 3, 4):
        outs, st = eng.deflate(shards, level=lvl, strategy=strat, wrap=2)
        assert [zlib.decompress(x, 31) for x in outs] == shards
# round 3: segment-parallel inflate, pointer-jumping resolve, the host-buffer pipelines in small chunks, truncated stored blocks
import parity_checks
parity_checks.split_inflate_checks(eng, o)
parity_checks.jump_resolve_checks(eng, o)
H.flush_point_stream_checks(lib, o, seeds=range(700, 702))
os.environ["ZMI_HOST_CHUNK"] = "30000"
blobs = [o.gen_shard(i % 8, 9000 + 700 * i) for i in range(24)]
outs, st = eng.deflate_host(blobs, level=6, wrap=1) if hasattr(eng, "deflate_host") else eng.deflate(blobs, level=6, wrap=1)
assert [zlib.decompress(x) for x in outs] == blobs
back, st2 = eng.inflate_host(outs, [len(b) for b in blobs], wrap=1) if hasattr(eng, "inflate_host") else eng.inflate(outs, [len(b) for b in blobs], wrap=1)
assert back == blobs and not any(st2)
print("asan check ok")
PY
