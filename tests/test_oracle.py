"""CPU tests: the oracle (oracle/zoracle*.c) pinned against the reference's own golden vectors.

Vectors were extracted from the reference's test sources by tests/golden/extract_reference_vectors.py
(committed JSON; nothing here reads /root/reference).  Second opinion: system zlib.
"""
import base64
import json
import os
import zlib

import pytest

import oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
DEF = json.load(open(os.path.join(HERE, "golden", "deflate_vectors.json")))["vectors"]
INF = json.load(open(os.path.join(HERE, "golden", "inflate_vectors.json")))


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


@pytest.mark.parametrize("v", DEF, ids=[v["source"].split("/")[-1] for v in DEF])
def test_deflate_golden_vectors_byte_exact(o, v):
    c = v["config"]
    wb, wrap = c["window_bits"], 1
    if wb > 15:
        wrap, wb = 2, wb - 16
    elif wb < 0:
        wrap, wb = 0, -wb
    rc, out = o.deflate(bytes.fromhex(v["input"]), c["level"], wrap, c["strategy"], c["mem_level"], wb)
    assert rc == 0
    assert out == bytes.fromhex(v["expected"])


@pytest.mark.parametrize("v", INF["bitstreams"], ids=[v["source"].split(":")[-1] for v in INF["bitstreams"]])
def test_inflate_handmade_bitstreams(o, v):
    data = bytes.fromhex(v["input"])
    rc, out, used, msg = o.inflate(data, 8 * len(data) + 64, v["wrap"])
    if v["expect"] == "data_error":
        assert rc == -3, (rc, msg)
    else:
        assert rc in (1, -5), (rc, msg)  # complete stream, or valid-so-far but unfinished
        if rc == 1:
            assert zlib.decompress(data, -15) == out


def test_inflate_error_messages(o):
    # names of the reference tests double as the message the reference reports (inflate.rs:694-1822)
    want = {"invalid_block_type": "invalid block type", "invalid_stored_block_length": "invalid stored block lengths",
            "too_many_length_or_distance_symbols": "too many length or distance symbols",
            "invalid_code_lengths_set": "invalid code lengths set", "invalid_bit_length_repeat_1": "invalid bit length repeat",
            "invalid_code_missing_end_of_block": "invalid code -- missing end-of-block",
            "invalid_literal_lengths_set": "invalid literal/lengths set", "invalid_distances_set": "invalid distances set",
            "invalid_literal_length_code": "invalid literal/length code", "invalid_distance_code": "invalid distance code",
            "invalid_distance_too_far_back": "invalid distance too far back", "incorrect_data_check": "incorrect data check",
            "incorrect_length_check": "incorrect length check"}
    seen = 0
    for v in INF["bitstreams"]:
        name = v["source"].split(":")[-1]
        if name in want:
            data = bytes.fromhex(v["input"])
            rc, out, used, msg = o.inflate(data, 8 * len(data) + 64, v["wrap"])
            assert msg == want[name], (name, msg)
            seen += 1
    assert seen >= 12


@pytest.mark.parametrize("f", INF["files"], ids=[f["source"].split("/")[-1] for f in INF["files"]])
def test_inflate_reference_fixtures(o, f):
    data = base64.b64decode(f["data_b64"])
    rc, out, used, msg = o.inflate(data, f["out_len"] + 16, f["wrap"])
    assert rc == 1, msg
    assert len(out) == f["out_len"] and zlib.crc32(out) == f["out_crc32"] and zlib.adler32(out) == f["out_adler32"]


def test_checksum_known_answers(o):
    # libz-rs-sys/src/lib.rs:146,179: crc32_z(0,[1,2,3]) == 1438416925 ; zlib-rs/src/crc32.rs:228
    assert o.crc32(bytes([1, 2, 3])) == 1438416925
    assert o.adler32(b"") == 1 and o.crc32(b"") == 0
    for n in (0, 1, 5551, 5552, 5553, 70000):
        d = o.prng_bytes(314159, n, 3)
        assert o.adler32(d) == zlib.adler32(d) and o.crc32(d) == zlib.crc32(d)
        k = n // 3
        assert o.lib.zo_crc32_combine(zlib.crc32(d[:k]), zlib.crc32(d[k:]), n - k) == zlib.crc32(d)
        assert o.lib.zo_adler32_combine(zlib.adler32(d[:k]), zlib.adler32(d[k:]), n - k) == zlib.adler32(d)


def test_compress_bound_doctest(o):
    # zlib-rs/src/deflate.rs:2966-2968
    assert [o.lib.zo_compress_bound(n, 1) for n in (1024, 4096, 65536)] == [1161, 4617, 73737]


def test_split_deflate_stitch_algebra(o):
    # zlib-rs/src/deflate.rs:4149-4221: crc of a concatenation from the parts' crcs
    a, b = o.gen_shard(0, 4096), o.gen_shard(3, 8192)
    assert o.lib.zo_crc32_combine(o.crc32(a), o.crc32(b), len(b)) == o.crc32(a + b)


@pytest.mark.parametrize("level", range(0, 10))
def test_oracle_deflate_roundtrip_all_levels(o, level):
    for cls in range(8):
        d = o.gen_shard(cls, 1 << 15)
        for wrap, wb in ((0, -15), (1, 15), (2, 31)):
            rc, c = o.deflate(d, level, wrap)
            assert rc == 0 and zlib.decompress(c, wb) == d
            assert len(c) <= o.lib.zo_compress_bound(len(d), wrap)
            rc2, back, _, msg = o.inflate(c, len(d), wrap)
            assert rc2 == 1 and back == d, msg


def test_oracle_tree_writer_matches_system_zlib(o):
    # Z_HUFFMAN_ONLY / Z_RLE have no match-finder freedom: byte-identical output pins the Huffman
    # tree builder, the tree transmission and the bit writer independently of the golden vectors
    for cls in range(8):
        d = o.gen_shard(cls, 1 << 16)
        for strat in (2, 3):
            rc, c = o.deflate(d, 6, 1, strat)
            co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, strat)
            assert c == co.compress(d) + co.flush()


def test_prng_matches_reference_lcg(o):
    # test-libz-rs-sys/src/inflate.rs:1981-1993
    state, out = 314159, bytearray()
    for _ in range(10):
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        out += bytes([state >> 24]) * 2
    assert o.prng_bytes(314159, 20, 2) == bytes(out)


def test_truncated_stored_block_hands_out_what_is_there(o):
    """Mode::CopyBlock (inflate.rs:1374-1394) copies min(length, room, input): a stored block whose end is missing still
    yields the bytes that arrived, and one that does not fit yields what fits -- checked against Python's zlib"""
    import zlib
    raw = o.gen_shard(5, 150000)
    co = zlib.compressobj(0, zlib.DEFLATED, -15)
    stream = co.compress(raw) + co.flush()
    for cut in (len(stream) - 1000, 70000, 65540, 10, 5):
        want = zlib.decompressobj(-15).decompress(stream[:cut])
        rc, got, used, msg = o.inflate(stream[:cut], len(raw), wrap=0)
        assert rc == -5 and got == want, (cut, rc, len(got), len(want))
    rc, got, used, msg = o.inflate(stream, 70000, wrap=0)   # room for one block and a bit
    assert rc == -5 and got == raw[:70000]
