"""GPU tests (-m gpu) of the stream ABI added at the end of round 1 -- so far validated on the CPU emulator only, hence in a
file that sorts behind the parity tests: the configuration matrix of end_to_end.rs, the entry points nothing else drives,
streams on concurrent threads."""
import ctypes as C

import pytest

import oracle_lib
import zlib_abi_harness as H

pytestmark = pytest.mark.gpu


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


def test_streams_on_concurrent_threads_on_gpu():
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    H.threaded_roundtrips(lib, oracle_lib.load(rebuild=False), threads=8, rounds=5)
