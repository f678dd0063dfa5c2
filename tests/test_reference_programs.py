"""The reference's own C programs -- libz-rs-sys-cdylib/example.c (every part of the stream ABI a C caller uses:
compress/uncompress, gz* I/O, small-buffer and large deflate/inflate with deflateParams, Z_FULL_FLUSH + inflateSync,
dictionaries, deflateBound/Copy/GetDictionary/SetHeader/Tune/Pending/Prime) and zpipe.c -- compiled UNMODIFIED against
include/ of this repo and run against the drop-in library.  The reference's CI does exactly this with its own cdylib
(.github/workflows/checks.yaml:436-478).

The sources stay in /root/reference (never copied): oracle/Makefile compiles them where they lie into oracle/_ref/
(git-ignored; __graft_entry__.build() does it whenever /root/reference exists, and the binaries travel to the GPU
box with the snapshot).  CPU variant: linked with the emulator build of the same ABI code.  GPU variant: the product
library on the MI355X."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
HAVE_REF = os.path.exists("/root/reference/libz-rs-sys-cdylib/example.c")

EXPECT = ["uncompress(): hello, hello!", "gzread(): hello, hello!", "gzgets() after gzseek:  hello!", "gzgets(): hello, hello!",
          "inflate(): hello, hello!", "large_inflate(): OK", "after inflateSync(): hello, hello!",
          "inflate with dictionary: hello, hello!", "deflateBound(): OK", "deflateGetDictionary(): hello, hello!",
          "deflateSetHeader(): OK", "deflateTune(): OK", "deflatePending(): OK", "deflatePrime(): OK", "gzclose -> 0"]


def _run_programs(suffix, tmp_path, payload):
    ex, zp = os.path.join(REFDIR, "example_" + suffix), os.path.join(REFDIR, "zpipe_" + suffix)
    if not (os.path.exists(ex) and os.path.exists(zp)):
        pytest.skip("oracle/_ref/*_%s not built: needs /root/reference at build time (make -C oracle ref ref-emu)" % suffix)
    r = subprocess.run([ex, str(tmp_path / "example.gz")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr)
    for line in EXPECT:
        assert line in r.stdout, (line, r.stdout, r.stderr)
    # zpipe: compress | decompress gives the input back (checks.yaml:442-444), and a conformant inflater reads the middle
    import zlib
    comp = subprocess.run([zp], input=payload, capture_output=True, timeout=600)
    assert comp.returncode == 0, comp.stderr
    assert zlib.decompress(comp.stdout) == payload
    back = subprocess.run([zp, "-d"], input=comp.stdout, capture_output=True, timeout=600)
    assert back.returncode == 0 and back.stdout == payload
    bad = subprocess.run([zp, "-d"], input=comp.stdout[:len(comp.stdout) // 2], capture_output=True, timeout=600)
    assert bad.returncode != 0                              # truncated input: zpipe reports Z_DATA_ERROR


def test_reference_example_c_and_zpipe_c_on_the_emulator(tmp_path):
    import oracle_lib
    import zmi_ctypes
    zmi_ctypes.load_emu()
    if HAVE_REF:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref-emu"], check=True)
    _run_programs("emu", tmp_path, oracle_lib.load().gen_shard(0, 200000))


@pytest.mark.gpu
def test_reference_example_c_and_zpipe_c_on_gpu(tmp_path):
    import oracle_lib
    o = oracle_lib.load(rebuild=False)
    # 11 MiB through zpipe.c's 16 KiB deflate(Z_NO_FLUSH) calls: the stream emits twice on the way (every 4 MiB) and once at Z_FINISH
    _run_programs("zmi", tmp_path, o.gen_shard(0, 3 << 20) + o.gen_shard(5, 1 << 20) + o.gen_shard(3, 4 << 20) + o.gen_shard(6, 3 << 20))
