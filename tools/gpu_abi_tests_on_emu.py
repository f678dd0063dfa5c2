"""Runs the bodies of the -m gpu stream-ABI tests (tests/test_gpu_abi.py, tests/test_gpu_zz_abi_matrix.py) with their GPU-size
parameters against the CPU emulator build of the same sources -- a dry run of the host logic when no GPU time is left
(about 12 minutes; the compiled-C-program test is skipped).  Test infrastructure only.
usage: python tools/gpu_abi_tests_on_emu.py"""
import os, sys, time, tempfile, pathlib, inspect
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import zmi_ctypes
zmi_ctypes.load_emu(False)
from zlib_rs_amd import _build
_build.ABI_LIB = os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")
import test_gpu_abi as T, test_gpu_zz_abi_matrix as Z
class MP:
    def setenv(self, k, v): os.environ[k] = v
for mod in (T, Z):
    for name, fn in inspect.getmembers(mod, inspect.isfunction):
        if not name.startswith("test_") or "c_program" in name: continue
        t = time.time()
        params = inspect.signature(fn).parameters
        kw = {}
        if "tmp_path" in params: kw["tmp_path"] = pathlib.Path(tempfile.mkdtemp())
        if "monkeypatch" in params: kw["monkeypatch"] = MP()
        try:
            fn(**kw); print("ok  ", name, "%.1f s" % (time.time() - t), flush=True)
        except Exception as e:
            import traceback; traceback.print_exc(); print("FAIL", name, repr(e)[:300], flush=True)
