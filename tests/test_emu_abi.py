"""CPU test of the zlib stream ABI layer (zlib_abi.hip) running on the emulator build."""
import ctypes as C
import os

import pytest

import oracle_lib
import zlib_abi_harness as H
import zmi_ctypes


@pytest.mark.usefixtures("inf_selection")
def test_zlib_abi_on_emulator(monkeypatch):
    monkeypatch.setenv("ZMI_ABI_SEGMENT", "4096")  # several chained segments even for small inputs
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    H.run_abi_checks(lib, oracle_lib.load(), sizes=(0, 1, 100, 5000, 20000))


def test_window_carry_over_between_segments(monkeypatch):
    """a stream split into chained segments keeps its window: a segment matches into the bytes in front of it,
    so the split costs (almost) nothing; ZMI_CARRY=0 gives the cold-start variant (Z_FULL_FLUSH semantics)"""
    import zlib
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    o = oracle_lib.load()
    data = o.gen_shard(0, 40000)
    sizes = {}
    for carry in ("1", "0"):
        monkeypatch.setenv("ZMI_ABI_SEGMENT", "4096")
        monkeypatch.setenv("ZMI_CARRY", carry)
        comp = H.deflate_stream(lib, data, level=6, wbits=15)
        assert zlib.decompress(comp) == data
        sizes[carry] = len(comp)
    monkeypatch.setenv("ZMI_ABI_SEGMENT", "65536")
    whole = len(H.deflate_stream(lib, data, level=6, wbits=15))
    assert sizes["1"] < sizes["0"] * 0.93, sizes          # ten 4 KiB cold starts cost > 7 % on this text
    assert sizes["1"] < whole * 1.03, (sizes, whole)       # with the window carried the split is nearly free


@pytest.mark.usefixtures("inf_selection")
def test_preset_dictionary_and_history_across_calls(monkeypatch):
    import zlib
    monkeypatch.setenv("ZMI_ABI_SEGMENT", "8192")
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    o = oracle_lib.load()
    text = o.gen_shard(1, 60000)
    H.dictionary_checks(lib, text[40000:52000], text[:33000])      # dictionary longer than the window: its tail is used
    H.dictionary_checks(lib, text[3000:9000], text[:1500])         # short dictionary, unaligned length
    # history survives Z_SYNC_FLUSH between deflate() calls (only Z_FULL_FLUSH forgets it, deflate.rs:2739-2752)
    data = text[:30000]
    one = H.deflate_stream(lib, data, level=6, wbits=15)
    chunked = H.deflate_stream(lib, data, level=6, wbits=15, chunk_in=3000, flush_every=1)
    assert zlib.decompress(chunked) == data
    assert len(chunked) < len(one) * 1.06, (len(chunked), len(one))   # ten flushes: markers + block headers only


@pytest.mark.usefixtures("inf_selection")
def test_gzip_header_copy_and_dictionary_queries(monkeypatch):
    monkeypatch.setenv("ZMI_ABI_SEGMENT", "8192")
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    H.header_copy_checks(lib, oracle_lib.load().gen_shard(2, 40000))


@pytest.mark.usefixtures("inf_selection")
def test_inflate_hands_out_output_progressively():
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    H.progressive_inflate_checks(lib, oracle_lib.load().gen_shard(0, 60000))


@pytest.mark.usefixtures("inf_selection")
def test_streaming_entry_points():
    """packet-wise and byte-wise inflate on the resumable device decode, inflateSync / Prime / Mark / Validate /
    SyncPoint, inflateBack, deflatePrime / deflateUsed"""
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    o = oracle_lib.load()
    H.streaming_checks(lib, o.gen_shard(0, 40000) + o.gen_shard(3, 30000), syslib=C.CDLL("libz.so.1"))


@pytest.mark.usefixtures("inf_selection")
def test_inflate_block_and_trees_stops():
    """inflate(Z_BLOCK) / inflate(Z_TREES): every call's return code, input left, output and data_type equal the system
    zlib's (the reference's own tests: test-libz-rs-sys/src/inflate.rs:640-676, :2036-2078)"""
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    o = oracle_lib.load()
    assert H.block_stop_checks(lib, C.CDLL("libz.so.1"), o.gen_shard(0, 40000) + o.gen_shard(3, 30000)) > 50


@pytest.mark.usefixtures("inf_selection")
def test_gz_file_api(tmp_path):
    """gzopen ... gzclose against Python's gzip module and the system's libz (libz-rs-sys/src/gz.rs)"""
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    H.gz_checks(lib, tmp_path, oracle_lib.load().gen_shard(1, 60000), syslib=C.CDLL("libz.so.1"))


@pytest.mark.usefixtures("inf_selection")
def test_reference_inflate_vectors_through_the_stream_abi():
    """the golden bitstreams / fixtures of the reference's tests through inflate(), whole and in steps"""
    import json
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    vectors = json.load(open(os.path.join(zmi_ctypes.ROOT, "tests", "golden", "inflate_vectors.json")))
    assert H.golden_inflate_checks(lib, vectors) > 50


@pytest.mark.usefixtures("inf_selection")
def test_inflate_attempt_limits(monkeypatch):
    """the bounds inside inflate(): one device decode sees a limited slice of the buffered input, and decoding pauses
    while the caller has not fetched what is queued -- shrunk through the environment so that a small stream gets there"""
    import zlib
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    o = oracle_lib.load()
    data = o.gen_shard(0, 90000) + o.gen_shard(6, 60000) + o.gen_shard(3, 50000)
    comp = zlib.compress(data, 6) + b"rest"
    for take, queue in (("3000", "20000"), ("100", "1000000"), ("1000000", "5000"), ("100", "3000")):
        monkeypatch.setenv("ZMI_ABI_TAKE", take)
        monkeypatch.setenv("ZMI_ABI_QUEUE", queue)
        rc, out, unused = H.inflate_stream(lib, comp, wbits=15, chunk_in=1 << 30, chunk_out=7000)
        # (bytes behind the end of the stream come back only from the call that delivered them)
        assert rc == H.Z_STREAM_END and out == data and unused in (0, 4), (take, queue, rc, len(out), unused)


@pytest.mark.usefixtures("inf_selection")
def test_random_streaming_roundtrips():
    """randomised pieces / rooms / flush arguments through inflate() against streams of the system's zlib"""
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    o = oracle_lib.load()
    for seed in range(40):
        H.random_streaming_roundtrips(lib, o, 6, seed)


@pytest.mark.usefixtures("inf_selection")
def test_streams_with_flush_points_are_decoded_as_segments():
    """inflate() of streams with sync / full flush points: the segment-parallel decode gives the serial decode's results.
    (A process of its own: the library reads its tuning variables once.)"""
    import re
    import subprocess
    import sys
    code = ("import ctypes as C, os, sys\n"
            "sys.path.insert(0, %r)\n"
            "import oracle_lib, zlib_abi_harness as H, zmi_ctypes\n"
            "zmi_ctypes.load_emu(rebuild=False)\n"
            "lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, 'tests', 'emu', 'libzmi355_emu.so')))\n"
            "print('cases', H.flush_point_stream_checks(lib, oracle_lib.load(), seeds=range(700, 703)))\n") % os.path.join(zmi_ctypes.ROOT, "tests")
    zmi_ctypes.load_emu()
    env = dict(os.environ, ZMI_TUNING="1", ZMI_ABI_SPLIT_MIN="20000", ZMI_ABI_SPLIT_GAP="1500", ZMI_ABI_TRACE="1")   # (the product splits from 256 KiB)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "cases 9" in r.stdout, (r.stdout[-400:], r.stderr[-800:])
    used = [int(m) for m in re.findall(r"(\d+) pieces decoded side by side", r.stderr)]
    assert used and max(used) >= 4, used   # the parallel path ran


@pytest.mark.usefixtures("inf_selection")
def test_streams_without_flush_points_are_decoded_as_blocks():
    """inflate() and uncompress() of ordinary streams (no flush points): the device finds the dynamic block headers and decodes the
    blocks side by side (zmi_inflate_blocks); results and error codes are those of the serial path.  (A process of its own: the
    library reads its tuning variables once.)"""
    import re
    import subprocess
    import sys
    code = ("import ctypes as C, os, sys, zlib\n"
            "sys.path.insert(0, %r)\n"
            "import oracle_lib, zlib_abi_harness as H, zmi_ctypes\n"
            "zmi_ctypes.load_emu(rebuild=False)\n"
            "lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, 'tests', 'emu', 'libzmi355_emu.so')))\n"
            "o = oracle_lib.load()\n"
            "print('uncompress cases', H.uncompress_large_checks(lib, o, 1 << 16))\n"
            "data = b''.join(o.gen_shard(c, 1 << 16) for c in (1, 3, 6, 0))\n"
            "for wb, comp in ((31, zlib.compressobj(6, zlib.DEFLATED, 31)), (15, zlib.compressobj(9, zlib.DEFLATED, 15)), (-15, zlib.compressobj(1, zlib.DEFLATED, -15))):\n"
            "    blob = comp.compress(data) + comp.flush()\n"
            "    for chunk in (len(blob), 50000):\n"
            "        rc, back, unused = H.inflate_stream(lib, blob + b'xyz', wb, chunk_in=chunk, chunk_out=1 << 17)\n"
            "        assert rc == 1 and back == data and unused == 3, (wb, chunk, rc, len(back), unused)\n"
            "print('streams ok')\n") % os.path.join(zmi_ctypes.ROOT, "tests")
    zmi_ctypes.load_emu()
    env = dict(os.environ, ZMI_TUNING="1", ZMI_BLOCKS_MIN="20000", ZMI_BLOCKS_GAP="1500", ZMI_ABI_BLOCKS_MIN="20000", ZMI_ABI_TRACE="1", ZMI_SPLIT_TRACE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "uncompress cases 6" in r.stdout and "streams ok" in r.stdout, (r.stdout[-400:], r.stderr[-1200:])
    used = [int(m) for m in re.findall(r"(\d+) blocks decoded side by side", r.stderr)]
    assert used and max(used) >= 4, used   # the parallel path ran through inflate()
    segs = [int(m) for m in re.findall(r"are block headers, (\d+) segments", r.stderr)]
    assert segs and max(segs) >= 4, segs


def test_host_checksums_match_zlib_at_every_length_and_alignment():
    """crc32() / adler32() of the drop-in library (csrc/host_sums.cpp: carry-less-multiply folding and SSSE3 sums with scalar
    heads and tails) against Python's zlib: lengths around the block sizes of the vector paths, odd offsets, running values"""
    import random
    import zlib
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    rng = random.Random(3)
    pool = bytes(rng.randrange(256) for _ in range(70000)) + b"\xff" * 12000
    lens = list(range(0, 200)) + [255, 256, 257, 5535, 5536, 5537, 5551, 5552, 5553, 11072, 65535, 65536, 70001] + [rng.randrange(82000) for _ in range(300)]
    for n in lens:
        off = rng.randrange(0, len(pool) - n + 1)
        d = pool[off:off + n]
        c0, a0 = rng.randrange(1 << 32), (rng.randrange(65521) << 16) | rng.randrange(65521)
        assert lib.crc32(c0, d, n) == zlib.crc32(d, c0), n
        assert lib.adler32(a0, d, n) == zlib.adler32(d, a0), n


def test_random_deflate_streams(monkeypatch):
    """randomised deflate(): level, strategy, wrapper, input pieces, output room, flush points, primed bits; the
    system's zlib must read every stream back"""
    import random
    import zlib
    monkeypatch.setenv("ZMI_ABI_SEGMENT", "8192")
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    o = oracle_lib.load()
    rnd = random.Random(77)
    for r in range(60):
        n = rnd.choice([0, 1, 700, 9000, rnd.randrange(40000), rnd.randrange(150000)])
        data = o.gen_shard(rnd.randrange(8), n)
        wbits = rnd.choice([15, 31, -15])
        comp = H.deflate_stream(lib, data, level=rnd.choice([1, 3, 6, 9]), wbits=wbits, chunk_in=rnd.choice([None, 1000, 7777]),
                                chunk_out=rnd.choice([64, 4096, 100000]), flush_every=rnd.choice([None, 1, 3]),
                                strategy=rnd.choice([0, 0, 1, 2, 3, 4]))
        assert zlib.decompressobj(wbits).decompress(comp) == data, (r, n, wbits)


@pytest.mark.usefixtures("inf_selection")
def test_config_matrix_roundtrips(monkeypatch):
    """end_to_end.rs's property over level x windowBits x memLevel x strategy; small windows must bound the distances"""
    monkeypatch.setenv("ZMI_ABI_SEGMENT", "16384")
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    H.config_matrix_roundtrips(lib, oracle_lib.load(), 60, seed=5)


def test_misc_entry_points():
    """allocators, deflateBound as a guarantee, deflateParams / Tune / ResetKeep / inflateReset2, _z one-shots, combine operators"""
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    H.misc_symbol_checks(lib, oracle_lib.load())


@pytest.mark.usefixtures("inf_selection")
def test_streams_on_concurrent_threads():
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    H.threaded_roundtrips(lib, oracle_lib.load(), threads=4, rounds=3)


@pytest.mark.usefixtures("inf_selection")
def test_input_handback_after_a_paused_decode(monkeypatch, tmp_path):
    monkeypatch.setenv("ZMI_ABI_QUEUE", "4096")
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    H.handback_checks(lib, oracle_lib.load(), tmp_path)
    monkeypatch.setenv("ZMI_ABI_ABSORB", "1000")     # the caller's input is taken in small pieces
    H.handback_checks(lib, oracle_lib.load(), tmp_path, size=60000)
    assert H.multi_member_reader_checks(lib, oracle_lib.load(), member_bytes=(30000, 1, 20000), piece=8192) == 3   # Python's gzip reader loop
    assert H.multi_member_reader_checks(lib, oracle_lib.load(), member_bytes=(9000, 7000, 100), piece=777) == 3


def test_deflate_emits_as_input_arrives(monkeypatch):
    """deflate(Z_NO_FLUSH) hands out compressed data once a few segments' worth of input has come in (the reference: whenever its
    pending buffer fills, zlib-rs/src/deflate.rs:2805-2826) -- a zpipe.c-style caller sees output before Z_FINISH and the stream
    does not hold its whole input; the pieces form one valid stream"""
    import zlib
    monkeypatch.setenv("ZMI_ABI_SEGMENT", "4096")
    monkeypatch.setenv("ZMI_ABI_EMIT", "20000")
    zmi_ctypes.load_emu()
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
    data = oracle_lib.load().gen_shard(0, 90000)
    strm = H.ZStream()
    assert lib.deflateInit2_(C.byref(strm), 6, 8, 15, 8, 0, lib.zlibVersion(), C.sizeof(H.ZStream)) == H.Z_OK
    src = C.create_string_buffer(data, len(data))
    obuf = C.create_string_buffer(1 << 17)
    out = bytearray()
    seen_before_finish = 0
    for pos in range(0, len(data), 3000):
        strm.next_in, strm.avail_in = C.addressof(src) + pos, min(3000, len(data) - pos)
        strm.next_out, strm.avail_out = C.addressof(obuf), len(obuf)
        assert lib.deflate(C.byref(strm), H.Z_NO_FLUSH) == H.Z_OK and strm.avail_in == 0
        got = len(obuf) - strm.avail_out
        out += obuf.raw[:got]
        seen_before_finish += got
    assert seen_before_finish > 20000 // 4, "no output before Z_FINISH: deflate() is buffering the whole input again"
    rc = H.Z_OK
    while rc != H.Z_STREAM_END:
        strm.next_out, strm.avail_out = C.addressof(obuf), len(obuf)
        rc = lib.deflate(C.byref(strm), H.Z_FINISH)
        assert rc in (H.Z_OK, H.Z_STREAM_END)
        out += obuf.raw[:len(obuf) - strm.avail_out]
    assert lib.deflateEnd(C.byref(strm)) == H.Z_OK
    assert zlib.decompress(bytes(out)) == data


def test_inflate_input_deferral_is_opt_in_and_exact_at_the_end():
    """ZMI_INFLATE_DEFER (off by default): inflate(Z_NO_FLUSH) takes small pieces without a device decode per call; the output is
    complete once the caller asks again with avail_in = 0 (or flushes), a shorter last piece ends the deferral by itself, and the
    number of device decodes drops from one per call to one per threshold.  In its own process: the setting is read once."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, os, sys, zlib
sys.path.insert(0, %r)
import oracle_lib, zmi_ctypes
import zlib_abi_harness as H
zmi_ctypes.load_emu(rebuild=False)
lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
o = oracle_lib.load(rebuild=False)
data = o.gen_shard(0, 60000)
for wbits, tail in ((15, b""), (31, b""), (15, b"x" * 7)):
    co = zlib.compressobj(6, zlib.DEFLATED, wbits)
    comp = co.compress(data) + co.flush()
    for chunk in (16, 100, 1000):
        strm = H.ZStream()
        assert lib.inflateInit2_(C.byref(strm), wbits, lib.zlibVersion(), C.sizeof(H.ZStream)) == 0
        feed = comp + (tail if chunk == 16 else b"")
        src = C.create_string_buffer(feed, len(feed))
        dst = C.create_string_buffer(len(data) + 64)
        strm.next_out, strm.avail_out = C.addressof(dst), len(dst)
        rc, calls_with_output = 0, 0
        for pos in range(0, len(feed), chunk):
            before = strm.avail_out
            strm.next_in, strm.avail_in = C.addressof(src) + pos, min(chunk, len(feed) - pos)
            rc = lib.inflate(C.byref(strm), 0)
            assert rc in (0, 1), rc
            calls_with_output += before != strm.avail_out
            if rc == 1:
                break
        polls = 0
        while rc == 0 and polls < 10:
            strm.avail_in = 0
            rc = lib.inflate(C.byref(strm), 0)
            polls += 1
        assert rc == 1 and dst.raw[:len(data)] == data and strm.total_out == len(data), (wbits, chunk, rc, polls)
        ncalls = (len(feed) + chunk - 1) // chunk
        assert calls_with_output <= ncalls // 4 + 4, ("a decode per call?", calls_with_output, ncalls)
        lib.inflateEnd(C.byref(strm))
print("deferral ok")
''' % os.path.join(zmi_ctypes.ROOT, "tests")
    env = dict(os.environ)
    env["ZMI_INFLATE_DEFER"] = "8192"
    env["ZMI_TUNING"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "deferral ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
