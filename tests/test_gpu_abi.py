"""GPU test (-m gpu): the zlib stream ABI of libz_mi355.so driven the way a C caller / the
reference's examples drive libz-rs-sys, plus a compiled C program against include/zmi355_zlib.h."""
import ctypes as C
import os
import subprocess
import sys

import pytest

import oracle_lib
import zlib_abi_harness as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.usefixtures("inf_selection")
def test_zlib_abi_on_gpu():
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    H.run_abi_checks(lib, oracle_lib.load(rebuild=False), sizes=(0, 1, 100, 5000, 70000, 3 << 20))


@pytest.mark.usefixtures("inf_selection")
def test_preset_dictionary_and_window_carry_on_gpu():
    import zlib
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    o = oracle_lib.load(rebuild=False)
    text = o.gen_shard(1, 3 << 20)
    H.dictionary_checks(lib, text[100000:1400000], text[:40000])   # > 1 MiB of data: two chained segments behind the dictionary
    H.dictionary_checks(lib, text[3000:9000], text[:1500])
    # a 3 MiB stream = three 1 MiB segments compressed in parallel with the window carried over; fed in pieces with
    # Z_SYNC_FLUSH in between the history also survives across deflate() calls
    one = H.deflate_stream(lib, text, level=6, wbits=15)
    assert zlib.decompress(one) == text
    chunked = H.deflate_stream(lib, text, level=6, wbits=15, chunk_in=300000, flush_every=1)
    assert zlib.decompress(chunked) == text
    assert len(chunked) < len(one) * 1.01, (len(chunked), len(one))


@pytest.mark.usefixtures("inf_selection")
def test_gzip_header_copy_and_dictionary_queries_on_gpu():
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    H.header_copy_checks(lib, oracle_lib.load(rebuild=False).gen_shard(2, 1500000))


@pytest.mark.usefixtures("inf_selection")
def test_inflate_hands_out_output_progressively_on_gpu():
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    H.progressive_inflate_checks(lib, oracle_lib.load(rebuild=False).gen_shard(0, 2 << 20))


@pytest.mark.usefixtures("inf_selection")
def test_three_gzip_members_in_8k_pieces_through_the_default_mode_on_gpu():
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    assert H.multi_member_reader_checks(lib, oracle_lib.load(rebuild=False)) == 3
    assert H.multi_member_reader_checks(lib, oracle_lib.load(rebuild=False), member_bytes=(8192, 1, 300000), piece=1000) == 3


def test_c_program_links_and_roundtrips(tmp_path):
    from zlib_rs_amd import _build
    exe = str(tmp_path / "abi_smoke")
    lib_dir = os.path.dirname(_build.ABI_LIB)
    subprocess.run(["gcc", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe,
                    "-L" + lib_dir, "-lz_mi355", "-lzmi355", "-Wl,-rpath," + lib_dir], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_smoke ok" in r.stdout


@pytest.mark.usefixtures("inf_selection")
def test_streaming_entry_points_on_gpu():
    """packet-wise and piece-wise inflate on the resumable device decode, inflateSync / Prime / Mark / Validate /
    SyncPoint, inflateBack, deflatePrime / deflateUsed; the system's zlib reads the primed stream too"""
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    o = oracle_lib.load(rebuild=False)
    H.streaming_checks(lib, o.gen_shard(0, 400000) + o.gen_shard(3, 300000), syslib=C.CDLL("libz.so.1"))


@pytest.mark.usefixtures("inf_selection")
def test_inflate_block_and_trees_stops_on_gpu():
    """inflate(Z_BLOCK) / inflate(Z_TREES) on the device decode's block and header stops: call by call the system zlib's
    return code, input left, output and data_type"""
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    o = oracle_lib.load(rebuild=False)
    assert H.block_stop_checks(lib, C.CDLL("libz.so.1"), o.gen_shard(0, 400000) + o.gen_shard(3, 300000)) > 50


@pytest.mark.usefixtures("inf_selection")
def test_streaming_inflate_large_in_small_chunks_on_gpu():
    """24 MiB through inflate() in 64 KiB pieces with a 256 KiB output buffer (the zpipe.c loop): bit-exact, and the
    host state stays small -- the decode restarts at block checkpoints instead of buffering the stream"""
    import zlib
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    o = oracle_lib.load(rebuild=False)
    data = b"".join(o.gen_shard(i, 1 << 20) for i in range(24))
    for wbits in (15, 31):
        co = zlib.compressobj(6, zlib.DEFLATED, wbits)
        comp = co.compress(data) + co.flush()
        rc, out, unused = H.inflate_stream(lib, comp + b"tail", wbits=wbits, chunk_in=1 << 16, chunk_out=1 << 18)
        assert rc == H.Z_STREAM_END and out == data and unused == 4


@pytest.mark.usefixtures("inf_selection")
def test_gz_file_api_on_gpu(tmp_path):
    """gzopen ... gzclose (libz-rs-sys/src/gz.rs) against Python's gzip module and the system's libz, 3 MiB members"""
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    H.gz_checks(lib, tmp_path, oracle_lib.load(rebuild=False).gen_shard(1, 3 << 20), syslib=C.CDLL("libz.so.1"))


@pytest.mark.usefixtures("inf_selection")
def test_reference_inflate_vectors_through_the_stream_abi_on_gpu():
    """the golden bitstreams / fixtures of the reference's tests through inflate(), whole and in steps"""
    import json
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    vectors = json.load(open(os.path.join(ROOT, "tests", "golden", "inflate_vectors.json")))
    assert H.golden_inflate_checks(lib, vectors, steps=(0, 1, 2, 3, 5, 17, 64)) > 100


@pytest.mark.usefixtures("inf_selection")
def test_streams_with_flush_points_are_decoded_as_segments_on_gpu():
    """inflate() of streams with sync / full flush points (marker look-alikes inside stored blocks included): the
    segment-parallel decode (zmi_inflate_split) gives the serial decode's bytes and codes; truncated and corrupted variants"""
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    assert H.flush_point_stream_checks(lib, oracle_lib.load(rebuild=False), seeds=range(800, 812), big=True) == 36


@pytest.mark.usefixtures("inf_selection")
def test_random_streaming_roundtrips_on_gpu():
    """randomised pieces / rooms / flush arguments through inflate() against streams of the system's zlib"""
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    o = oracle_lib.load(rebuild=False)
    for seed in range(1000, 1008):
        H.random_streaming_roundtrips(lib, o, 3, seed, max_len=200000)

