"""CPU test: the shipped libraries load and export every symbol the headers in include/ declare
(no compute calls -- there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#define[^\n]*(\\\n[^\n]*)*", "", src)
    src = re.sub(r"typedef[^;]*;", "", src)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("alloc_func", "free_func", "sizeof")))


def _exports(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return set(l.split()[-1] for l in out.splitlines() if l.strip())


def test_libraries_export_declared_symbols():
    from zlib_rs_amd import _build
    _build.build()
    core, abi = _exports(_build.LIB), _exports(_build.ABI_LIB)
    missing = [n for n in _declared("zmi355.h") if n not in core]
    assert not missing, missing
    missing = [n for n in _declared("zmi355_zlib.h") if n not in abi]
    assert not missing, missing
    # the engine library itself must not define zlib-named symbols (they would interpose libz)
    assert not ({"deflate", "inflate", "crc32", "adler32", "compress", "uncompress"} & core)


def test_libraries_load_and_fail_loudly_without_gpu():
    import torch
    from zlib_rs_amd import _build
    core = C.CDLL(_build.LIB)
    core.zmi_version.restype = C.c_char_p
    assert b"zmi355" in core.zmi_version()
    abi = C.CDLL(_build.ABI_LIB)
    abi.zlibVersion.restype = C.c_char_p
    assert abi.zlibVersion().startswith(b"1.")
    if not torch.cuda.is_available():
        ctx = C.c_void_p()
        rc = core.zmi_ctx_create(C.byref(ctx), 0)
        assert rc != 0, "context creation must fail without a HIP device (no CPU fallback)"
        from zlib_rs_amd.engine import Engine
        try:
            Engine()
            raise AssertionError("Engine() must raise without a GPU")
        except RuntimeError:
            pass
