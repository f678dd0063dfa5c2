"""CPU test of the zlib stream ABI layer (zlib_abi.hip) running on the emulator build."""
import ctypes as C
import os

import oracle_lib
import zlib_abi_harness as H
import zmi_ctypes


def test_zlib_abi_on_emulator(monkeypatch):
    monkeypatch.setenv("ZMI_ABI_SEGMENT", "4096")  # several chained segments even for small inputs
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    H.run_abi_checks(lib, oracle_lib.load(), sizes=(0, 1, 100, 5000, 20000))
