"""GPU tests (-m gpu) of the stream ABI added at the end of round 1 -- so far validated on the CPU emulator only, hence in a
file that sorts behind the parity tests: the configuration matrix of end_to_end.rs, the entry points nothing else drives,
streams on concurrent threads."""
import ctypes as C

import pytest

import oracle_lib
import zlib_abi_harness as H

pytestmark = pytest.mark.gpu


@pytest.mark.usefixtures("inf_selection")
def test_config_matrix_roundtrips_on_gpu():
    """end_to_end.rs's property over level x windowBits x memLevel x strategy on the device path"""
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    H.config_matrix_roundtrips(lib, oracle_lib.load(rebuild=False), 40, seed=11, max_len=300000)


def test_misc_entry_points_on_gpu():
    """allocators, deflateBound as a guarantee, deflateParams / Tune / ResetKeep / inflateReset2, _z one-shots, combine operators"""
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    H.misc_symbol_checks(lib, oracle_lib.load(rebuild=False))


@pytest.mark.usefixtures("inf_selection")
def test_streams_on_concurrent_threads_on_gpu():
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    H.threaded_roundtrips(lib, oracle_lib.load(rebuild=False), threads=8, rounds=5)


def test_product_process_without_tuning_on_gpu():
    """A process started WITHOUT ZMI_TUNING (what a user of the libraries runs: no override is read, the product's own kernel
    selection, segment sizes and queue limits): the stream-ABI families and the batch parity families once more, in one go."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes as C, json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
assert not [k for k in os.environ if k.startswith("ZMI_")], "the point of this process is an untouched environment"
import oracle_lib, parity_checks, zmi_ctypes
import zlib_abi_harness as H
from zlib_rs_amd import _build
o = oracle_lib.load(rebuild=False)
lib = H.bind(C.CDLL(_build.ABI_LIB))
H.run_abi_checks(lib, o, sizes=(0, 1, 100, 5000, 70000, 3 << 20))
assert H.flush_point_stream_checks(lib, o, seeds=range(800, 806), big=True) == 18
assert H.uncompress_large_checks(lib, o, 1 << 20) == 6
vectors = json.load(open(os.path.join(%r, "tests", "golden", "inflate_vectors.json")))
assert H.golden_inflate_checks(lib, vectors, steps=(0, 1, 7)) > 40
assert H.block_stop_checks(lib, C.CDLL("libz.so.1"), o.gen_shard(0, 400000) + o.gen_shard(3, 300000)) > 50
for seed in range(1000, 1003):
    H.random_streaming_roundtrips(lib, o, 3, seed, max_len=200000)
eng = zmi_ctypes.Engine(zmi_ctypes.load_product())
assert parity_checks.golden_bitstreams_exact(eng.inflate, o) >= 20
assert parity_checks.golden_files_exact(eng.inflate, o) >= 8
assert parity_checks.corrupt_streams_exact(eng.inflate, o, o.gen_shard(1, 1 << 18)) >= 30
assert parity_checks.truncated_stored_checks(eng.inflate, o) > 0
assert parity_checks.split_inflate_checks(eng, o, big=True) == 12
assert parity_checks.blocks_inflate_checks(eng, o, big=True) == 14
assert parity_checks.jump_resolve_checks(eng, o, big=True) == 8
eng.close()
print("product process ok")
''' % (root, root, root)
    env = {k: v for k, v in os.environ.items() if not k.startswith("ZMI_")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "product process ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
