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


def _exports_versions(lib):
    """{symbol: version node or None} of the dynamic symbol table"""
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    res = {}
    for l in out.splitlines():
        if not l.strip():
            continue
        name = l.split()[-1]
        if "@" in name:
            sym, ver = name.split("@", 1)
            res[sym] = ver.lstrip("@")
        else:
            res[name] = None
    return res


def _exports(lib):
    return set(k for k in _exports_versions(lib) if not k.startswith("ZLIB_"))


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


def test_drop_in_library_exports_exactly_the_zlib_abi_with_its_version_nodes():
    """libz_mi355.so against the reference's libz-rs-sys/include/zlib.map (extracted to tests/golden/
    zlib_symbol_versions.json): every versioned entry point carries its node, the pre-1.2.0 entry points carry none,
    and nothing but the declared entry points is exported (no C++ helpers, no std:: instantiations)."""
    import json
    from zlib_rs_amd import _build
    _build.build()
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "zlib_symbol_versions.json")))
    got = _exports_versions(_build.ABI_LIB)
    got = {k: v for k, v in got.items() if not k.startswith("ZLIB_")}     # the version nodes themselves
    declared = set(_declared("zmi355_zlib.h"))
    assert set(got) == declared, (sorted(set(got) - declared), sorted(declared - set(got)))
    for sym, node in golden["versions"].items():
        assert got.get(sym) == node, (sym, got.get(sym), node)
    for sym, ver in got.items():
        if sym not in golden["versions"]:
            assert ver is None, (sym, ver)
    for sym in golden["local"]:
        assert sym.rstrip("*") not in got or sym.endswith("*")
    assert not [s for s in got if s.startswith("_")]


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
