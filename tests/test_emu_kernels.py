"""CPU tests of the HIP kernels themselves, compiled by g++ against the SIMT emulator
(tests/emu/, -DZMI_EMU).  Small inputs only (the emulator runs ~1 us per fiber switch); the same
kernels are tested at full size on the MI355X by test_gpu_parity.py."""
import ctypes as C
import json
import os
import zlib

import pytest

import oracle_lib
import parity_checks
import zmi_ctypes

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def eng():
    e = zmi_ctypes.Engine(zmi_ctypes.load_emu())
    yield e
    e.close()


@pytest.fixture(scope="module")
def o():
    return oracle_lib.load()


def test_deflate_kernels_roundtrip(eng, o):
    blobs = [b"", b"a", b"abcd", b"abcde" * 7, bytes(3000), b"abcd" * 1000, o.gen_shard(0, 1 << 13), o.gen_shard(3, 1 << 13),
             o.gen_shard(5, 1 << 12), o.gen_shard(7, 1 << 13)[2000:7000], o.prng_bytes(7, 5000, 1)]
    for level, wrap in ((6, 1), (1, 0), (9, 2), (0, 1)):
        outs, st = eng.deflate(blobs, level=level, wrap=wrap)
        assert all(s == 0 for s in st)
        for b, c in zip(blobs, outs):
            assert zlib.decompress(c, {0: -15, 1: 15, 2: 31}[wrap]) == b
            rc, back, _, msg = o.inflate(c, len(b), wrap)
            assert rc == 1 and back == b, msg


def test_long_matches_are_extended_by_one_lane_per_run(eng, o):
    """lz77.hip, short budgets: stop at 16 equal bytes, the leader of a run of equal distances extends, the others inherit"""
    def deflate(blobs, level):
        outs, st = eng.deflate(blobs, level=level, wrap=1)
        assert all(s == 0 for s in st)
        return outs
    assert parity_checks.long_match_checks(deflate, o) == 36


def test_deflate_multi_piece_streams(eng, o, monkeypatch):
    monkeypatch.setenv("ZMI_BLOCK_SPAN", "2048")
    d = o.gen_shard(1, 1 << 14)
    outs, st = eng.deflate([d, d[:5000], d[:2049]], level=6, wrap=2)
    assert st == [0, 0, 0]
    for b, c in zip([d, d[:5000], d[:2049]], outs):
        assert zlib.decompress(c, 31) == b


def strategy_token_rules(raw_stream, strategy):
    """what the strategy promises about the tokens (zlib-rs/src/deflate/algorithm/huff.rs, rle.rs; Strategy::Fixed in
    zng_tr_flush_block, deflate.rs:2316-2434): Z_HUFFMAN_ONLY no matches at all, Z_RLE distance 1 only, Z_FIXED no dynamic
    block.  Shared with the GPU test."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import deflate_dump
    blocks, out = deflate_dump.dump(raw_stream, 0)
    if strategy == 2:
        assert sum(b["matches"] for b in blocks) == 0
    if strategy == 3:
        assert max(b["maxdist"] for b in blocks) <= 1
    if strategy == 4:
        assert all(b["type"] in (0, 1) for b in blocks)
    return out


def test_deflate_strategies(eng, o):
    for cls in (0, 4):
        d = o.gen_shard(cls, 1 << 13) + bytes(300) + b"ab" * 100
        for strat in (1, 2, 3, 4):
            outs, st = eng.deflate([d], level=6, strategy=strat, wrap=1)
            assert st == [0] and zlib.decompress(outs[0]) == d
            assert strategy_token_rules(outs[0][2:-4], strat) == d


@pytest.mark.usefixtures("inf_selection")
def test_inflate_kernel_matches_oracle(eng, o):
    blobs = [b"", b"a", b"hello world", bytes(1000), b"abc" * 3000, o.gen_shard(0, 1 << 13), o.gen_shard(6, 1 << 13)]
    for wrap, wb in ((1, 15), (2, 31), (0, -15)):
        for level in (0, 1, 6, 9):
            streams = []
            for b in blobs:
                co = zlib.compressobj(level, zlib.DEFLATED, wb)
                streams.append(co.compress(b) + co.flush())
            outs, st = eng.inflate(streams, [len(b) + 4 for b in blobs], wrap)
            assert all(s == 0 for s in st), st
            assert outs == blobs
    # streams produced by the oracle's restatement of the reference (fixed + dynamic + stored blocks)
    streams = [o.deflate(b, 6, 1)[1] for b in blobs] + [o.deflate(blobs[5], 6, 1, 4)[1]]
    outs, st = eng.inflate(streams, [len(b) + 4 for b in blobs] + [len(blobs[5])], 1)
    assert all(s == 0 for s in st) and outs == blobs + [blobs[5]]


@pytest.mark.usefixtures("inf_selection")
def test_inflate_kernel_golden_bitstreams(eng, o):
    import parity_checks
    assert parity_checks.golden_bitstreams_exact(eng.inflate, o) >= 20
    assert parity_checks.golden_files_exact(eng.inflate, o) >= 10


@pytest.mark.usefixtures("inf_selection")
def test_inflate_kernel_corrupt_streams_report_the_oracles_code(eng, o):
    import parity_checks
    parity_checks.corrupt_streams_exact(eng.inflate, o, o.gen_shard(2, 1 << 13))


@pytest.mark.usefixtures("inf_selection")
def test_inflate_kernel_errors(eng, o):
    d = o.gen_shard(2, 1 << 13)
    good = zlib.compress(d, 6)
    bad = bytearray(good); bad[300] ^= 0x40
    badchk = bytearray(good); badchk[-2] ^= 1
    outs, st = eng.inflate([good, bytes(bad), bytes(badchk), good[:700], good, b"\x78\x9c\x07"], [len(d)] * 4 + [100, 10], 1)
    assert st[0] == 0 and outs[0] == d
    assert st[1] in (-3, -5) and st[2] == -3 and st[3] == -5 and st[4] == -5 and st[5] == -3


@pytest.mark.usefixtures("inf_selection")
def test_inflate_many_small_blocks(eng, o):
    """block headers, empty stored blocks and the 1 KiB input chunk boundary meet in every alignment"""
    d = o.gen_shard(0, 12000) + o.gen_shard(6, 9000)
    for flush in (zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH):
        for step in (37, 211):
            co = zlib.compressobj(6, zlib.DEFLATED, 15)
            s = b"".join(co.compress(d[i:i + step]) + co.flush(flush) for i in range(0, len(d), step)) + co.flush()
            assert zlib.decompress(s) == d
            outs, st = eng.inflate([s], [len(d)], 1)
            assert st == [0] and outs[0] == d


@pytest.mark.usefixtures("inf_selection")
def test_inflate_long_distances_and_runs(eng, o):
    r = o.prng_bytes(11, 32768, 1)
    blobs = [r + r + r[:700],                       # distance 32768, maximal lengths
             bytes(70000),                          # distance 1 runs across many 258-byte copies
             b"ab" * 20000 + b"xyz" * 9000,         # short periods
             r[:5000] + bytes(300) + r[:5000] + o.gen_shard(3, 1 << 15),
             # a hole-free stretch longer than the resolve ring between two regions with back-references
             o.gen_shard(0, 5000) + o.prng_bytes(5, 90000, 1) + o.gen_shard(0, 5000) + r[100:900]]
    for level in (1, 9):
        streams = [zlib.compress(b, level) for b in blobs]
        outs, st = eng.inflate(streams, [len(b) for b in blobs], 1)
        assert st == [0] * len(blobs)
        assert outs == blobs


def test_crc32_block_scheme_boundaries(eng, o):
    """gzip trailers (CRC-32) of shards around the 16 KiB block size of the checksum kernel: alone in the buffer (16-byte
    aligned: full blocks go through the interleaved scheme) and behind an odd-sized neighbour (unaligned: segment scheme)"""
    sizes = [16383, 16384, 16385, 2 * 16384, 3 * 16384 + 7, (1 << 17) + 63, 5 * 16384 - 1]
    for k, n in enumerate(sizes):
        d = o.gen_shard(k, n)
        outs, st = eng.deflate([d], level=1, wrap=2)
        assert st == [0] and zlib.decompress(outs[0], 31) == d, n
        outs, st = eng.deflate([b"x" * 3, d, d[:777], d], level=1, wrap=2)
        assert st == [0] * 4 and [zlib.decompress(c, 31) for c in outs] == [b"x" * 3, d, d[:777], d], n


def resolve_window_edge_streams(o):
    """after a long hole-free stretch the resolve pass restarts its ring one window in front of the
    batch.  The restart used to be derived from the LAST hole of the batch (+3), which with a batch spanning RES_SPAN
    (2048) and a first hole at p % 1024 in 1021..1023 landed up to 3 bytes above p - 32768: distances 32766..32768
    then copied bytes that were never staged.  Raw fixed-Huffman streams with exactly that geometry.  Shared with the
    GPU test.  Returns [(stream, expected)]."""
    import deflate_craft
    cases = []
    for pmod in (1021, 1022, 1023):
        for dist in (32766, 32767, 32768):
            lits = o.prng_bytes(100 + pmod + dist, 50000, 1)
            p = 40 * 1024 + pmod
            toks = list(lits[:10]) + [(4, 7)]
            k = 14
            toks += list(lits[k:p]); k = p
            toks += [(3, dist)]; k += 3
            toks += list(lits[k:p + 2048]); k = p + 2048
            toks += [(5, 32768)]; k += 5
            toks += list(lits[k:k + 500])
            cases.append((deflate_craft.fixed_block(toks), deflate_craft.expand(toks)))
    return cases


@pytest.mark.usefixtures("inf_selection")
def test_inflate_resolve_window_edge(eng, o):
    cases = resolve_window_edge_streams(o)
    for s, want in cases:
        assert zlib.decompress(s, -15) == want
    outs, st = eng.inflate([s for s, _ in cases], [len(w) for _, w in cases], 0)
    assert st == [0] * len(cases)
    assert outs == [w for _, w in cases]


def resolve_near_far_streams(o):
    """The resolve pass keeps RES_NEAR (2560) bytes of history in its LDS ring and reads sources further back from HBM,
    where they must already be final.  Raw fixed-Huffman streams around that boundary: distances RES_NEAR - 1 .. + 2
    and 32768, sources that are themselves the output of earlier (near and far) back-references, far and near holes
    mixed inside one batch and depending on each other, 258-byte copies, a far copy right behind a hole-free stretch
    longer than the ring.  Shared with the GPU test.  Returns [(stream, expected)]."""
    import deflate_craft
    cases = []
    for seed, lead in ((1, 3000), (2, 2563), (3, 40000)):
        lits = o.prng_bytes(900 + seed, 120000, 1)
        toks, n = [], 0

        def lit(k, toks=toks):
            nonlocal n
            toks += list(lits[n:n + k]); n += k

        def cp(length, dist, toks=toks):
            nonlocal n
            assert dist <= n
            toks.append((length, dist)); n += length

        lit(lead)
        for d in (2559, 2560, 2561, 2562):            # one batch: near and far next to each other
            cp(5, d); lit(2)
        cp(258, 2561); cp(258, 2560); cp(17, 258 + 2561)     # far source = the far copy just made
        lit(7)
        for rep in range(40):                          # dense holes: sources are earlier holes, near and far alternate
            cp(3 + rep % 9, 2550 + rep % 20); cp(4, 1 + rep % 7); lit(1)
        lit(9000)                                      # hole-free stretch longer than the ring
        cp(9, min(n, 32768)); cp(3, 2561); cp(258, min(n, 32768)); cp(6, 2560)
        lit(3)
        for rep in range(70):                          # > 64 holes: the batch boundary falls inside the run
            cp(3, 2561 + rep); cp(3, 3)
        lit(100)
        cases.append((deflate_craft.fixed_block(toks), deflate_craft.expand(toks)))
    return cases


@pytest.mark.usefixtures("inf_selection")
def test_inflate_resolve_near_far_boundary(eng, o):
    cases = resolve_near_far_streams(o)
    for s, want in cases:
        assert zlib.decompress(s, -15) == want
    # odd output offsets as well: the HBM reads of far sources round their address down to a word
    outs, st = eng.inflate([s for s, _ in cases], [len(w) for _, w in cases], 0)
    assert st == [0] * len(cases)
    assert outs == [w for _, w in cases]
    ioff, ooff, a, b_ = [], [], 1, 3
    for s, w in cases:
        ioff.append(a); a += len(s) + 5
        ooff.append(b_); b_ += len(w) + 1
    outs, st, guard = eng.inflate_dev([s for s, _ in cases], [len(w) for _, w in cases], ioff, ooff, 0, out_limit=1 << 20)
    assert st == [0] * len(cases) and outs == [w for _, w in cases]


@pytest.mark.usefixtures("inf_selection")
def test_inflate_unaligned_layout_and_scratch_limit(eng, o):
    blobs = [o.gen_shard(1, 5000), o.gen_shard(4, 3001), b"q" * 777 + o.gen_shard(6, 2000), o.gen_shard(2, 4097)]
    streams = [zlib.compress(b, 6) for b in blobs]
    caps = [len(b) for b in blobs]
    ioff, ooff, a, b_ = [], [], 3, 5
    for s, c in zip(streams, caps):
        ioff.append(a); a += len(s) + 7
        ooff.append(b_); b_ += c + 3          # odd offsets: the 4- and 16-byte aligned paths are not taken
    outs, st, guard = eng.inflate_dev(streams, caps, ioff, ooff, 1, out_limit=1 << 20)
    assert st == [0, 0, 0, 0] and outs == blobs
    assert all(g == b"\xee" for g in guard)   # nothing written past a stream's capacity
    # a context whose bitmap scratch covers only part of the batch: the rest reports Z_MEM_ERROR, never garbage
    big = [o.gen_shard(0, 1 << 17)] * 12
    sb = [zlib.compress(x, 1) for x in big]
    off_i = [sum(len(x) for x in sb[:i]) for i in range(len(sb))]
    off_o = [i << 17 for i in range(len(sb))]
    outs, st, _ = eng.inflate_dev(sb, [1 << 17] * len(sb), off_i, off_o, 1, out_limit=1 << 20)
    assert st[0] == 0 and outs[0] == big[0]
    assert -4 in st and all(x in (0, -4) for x in st)
    assert all(o_ == big[0] for o_, x in zip(outs, st) if x == 0)


def test_adaptive_block_splitting(eng, o, monkeypatch):
    """drifting statistics get blocks of their own (better ratio than one block per 64 KiB), stationary text does not
    split; the smallest sub-blocks on incompressible data (all stored) still fit the deflate bound"""
    walk, text = o.gen_shard(5, 1 << 16), o.gen_shard(0, 1 << 16)
    sizes = {}
    for tokens in ("1000000", "4096"):
        monkeypatch.setenv("ZMI_BLOCK_TOKENS", tokens)
        outs, st = eng.deflate([walk, text], level=6, wrap=1)
        assert st == [0, 0] and zlib.decompress(outs[0]) == walk and zlib.decompress(outs[1]) == text
        sizes[tokens] = [len(x) for x in outs]
    assert sizes["4096"][0] < sizes["1000000"][0] * 0.95          # the random walk: > 5 % smaller
    assert sizes["4096"][1] <= sizes["1000000"][1] * 1.002        # text: not worse (a split must pay for its header; the
                                                                  # sub-block boundaries also re-price short far matches: +-0.1 %)
    monkeypatch.setenv("ZMI_BLOCK_TOKENS", "64")
    noise = o.prng_bytes(3, 50000, 1)
    outs, st = eng.deflate([noise, noise[:777]], level=6, wrap=1)
    assert st == [0, 0] and zlib.decompress(outs[0]) == noise and zlib.decompress(outs[1]) == noise[:777]
    assert len(outs[0]) <= len(noise) + len(noise) // 8 + 64


@pytest.mark.usefixtures("inf_selection")
def test_resumable_inflate_from_block_checkpoints():
    """zmi_inflate_resume (include/zmi355.h): cut streams, restart at the reported block boundary with the output in
    front of it as history -- the device half of the streaming inflate (zlib-rs/src/inflate.rs:288-320)"""
    import resume_checks
    eng = zmi_ctypes.Engine(zmi_ctypes.load_emu())
    try:
        resume_checks.resume_chain_checks(eng, oracle_lib.load(), sizes=(60000, 40000, 20000, 20000), trials=2)
    finally:
        eng.close()


@pytest.mark.usefixtures("inf_selection")
def test_host_batch_pipeline_chunks(monkeypatch):
    """zmi_deflate_batch cuts a host batch into chunks that cycle through two device slots (copy-in / kernels /
    copy-out on three streams): same bytes as the one-chunk path, every shard in its own slot of the output"""
    import random
    o = oracle_lib.load()
    rnd = random.Random(11)
    shards = [o.gen_shard(i % 8, rnd.choice([0, 1, 15, 16, 17, 1000, 4096, 9999, 20000])) for i in range(37)]
    monkeypatch.delenv("ZMI_HOST_CHUNK", raising=False)
    eng = zmi_ctypes.Engine(zmi_ctypes.load_emu())
    try:
        one, st1 = eng.deflate(shards, level=6, wrap=1)
        for budget in ("30000", "20016", "1"):          # several shards per chunk ... one shard per chunk
            monkeypatch.setenv("ZMI_HOST_CHUNK", budget)
            many, st2 = eng.deflate(shards, level=6, wrap=1)
            assert st1 == st2 == [0] * len(shards)
            assert many == one
        assert [zlib.decompress(x) for x in one] == shards
        # and back: capacities that pack like the device slot (multiples of 16: one copy per chunk) and ragged ones
        monkeypatch.delenv("ZMI_HOST_CHUNK", raising=False)
        for caps in ([(len(x) + 15) & ~15 for x in shards], [len(x) + 3 for x in shards]):
            ref, rst = eng.inflate(one, caps, wrap=1)
            assert ref == shards and rst == [0] * len(shards)
            for budget in ("30000", "1"):
                monkeypatch.setenv("ZMI_HOST_CHUNK", budget)
                got, gst = eng.inflate(one, caps, wrap=1)
                assert got == shards and gst == rst
            monkeypatch.delenv("ZMI_HOST_CHUNK", raising=False)
        # how a chunk leaves the device follows what the chunks before it decoded: the whole region in one copy when most of the room
        # was used, range by range (decoded bytes only) when the streams were given far more room than they needed -- both ways, in
        # turns, on one context
        monkeypatch.setenv("ZMI_HOST_CHUNK", "30000")
        for caps in ([len(x) + 3 for x in shards], [4 * len(x) + 64 for x in shards], [4 * len(x) + 64 for x in shards],
                     [len(x) + 1 for x in shards], [len(x) + 1 for x in shards]):
            got, gst = eng.inflate(one, caps, wrap=1)
            assert got == shards and gst == [0] * len(shards)
        monkeypatch.delenv("ZMI_HOST_CHUNK", raising=False)
        # a stream that fails in the middle of a chunk keeps its own status; its neighbours are untouched
        hurt = list(one)
        hurt[5] = hurt[5][:len(hurt[5]) // 2]
        monkeypatch.setenv("ZMI_HOST_CHUNK", "30000")
        got, gst = eng.inflate(hurt, [len(x) + 16 for x in shards], wrap=1)
        assert gst[5] != 0 and all(v == 0 for i, v in enumerate(gst) if i != 5)
        assert all(got[i] == shards[i] for i in range(len(shards)) if i != 5)
    finally:
        eng.close()


def test_batch_api_argument_errors(eng):
    """the batch entry points refuse bad arguments with ZMI_E_ARG and a message, and an empty batch is a no-op"""
    import ctypes as C
    import numpy as np
    lib, E_ARG = eng.lib, -103
    data = np.frombuffer(b"hello hello hello hello" + bytes(9), dtype=np.uint8).copy()
    off, ln = np.zeros(1, np.uint64), np.array([23], np.uint32)
    stride = int(lib.zmi_deflate_bound(23, 1))
    assert stride % 16 == 0 and stride >= 23 + 11
    out, olen, st = np.zeros(stride, np.uint8), np.zeros(1, np.uint32), np.zeros(1, np.int32)
    u64p, u32p, i32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)

    def call(ctx=eng.ctx, n=1, level=6, strategy=0, wrap=1, stride=stride):
        return lib.zmi_deflate_batch(ctx, data.ctypes.data, off.ctypes.data_as(u64p), ln.ctypes.data_as(u32p), n, level, strategy,
                                     wrap, out.ctypes.data, stride, olen.ctypes.data_as(u32p), st.ctypes.data_as(i32p))
    assert call() == 0 and st[0] == 0 and zlib.decompress(bytes(out[:olen[0]])) == bytes(data[:23])
    assert call(level=-1) == 0          # Z_DEFAULT_COMPRESSION
    for kw in (dict(ctx=None), dict(level=10), dict(level=-2), dict(strategy=5), dict(strategy=-1), dict(wrap=3), dict(wrap=-1),
               dict(stride=stride - 16), dict(stride=stride + 8)):
        assert call(**kw) == E_ARG, kw
        assert lib.zmi_last_error()
    assert call(n=0) == 0
    comp = np.frombuffer(zlib.compress(bytes(data[:23])) + bytes(8), dtype=np.uint8).copy()
    clen, cap, ooff = np.array([len(comp) - 8], np.uint32), np.array([64], np.uint32), np.zeros(1, np.uint64)
    back = np.zeros(64, np.uint8)

    def icall(ctx=eng.ctx, n=1, wrap=1):
        return lib.zmi_inflate_batch(ctx, comp.ctypes.data, off.ctypes.data_as(u64p), clen.ctypes.data_as(u32p), n, wrap,
                                     back.ctypes.data, ooff.ctypes.data_as(u64p), cap.ctypes.data_as(u32p),
                                     olen.ctypes.data_as(u32p), st.ctypes.data_as(i32p))
    assert icall() == 0 and st[0] == 0 and bytes(back[:olen[0]]) == bytes(data[:23])
    assert icall(wrap=3) == 0 and st[0] == 0       # ZMI_WRAP_AUTO: zlib or gzip by the first bytes (inflate.rs:2298-2327, +32)
    assert icall(wrap=2) == 0 and st[0] == -3      # a zlib stream is not a gzip member: per-stream Z_DATA_ERROR, the call succeeds
    for kw in (dict(ctx=None), dict(wrap=4), dict(wrap=-1)):
        assert icall(**kw) == E_ARG, kw
    assert icall(n=0) == 0


def test_pack_slab_kernel(eng, o):
    """csrc/pack.hip: strided slots -> dense slab, every source / destination alignment, empty and multi-tile ranges"""
    rng_sizes = [0, 1, 2, 3, 4, 5, 15, 16, 17, 31, 33, 100, 4095, 4096, 4097, 8191, 8200, 12345, 0, 7]
    members = [o.prng_bytes(40 + i, n, 1) if n else b"" for i, n in enumerate(rng_sizes)]
    for stride, sb, db in ((12352, 0, 0), (12353, 3, 1), (12355, 13, 2), (12366, 16, 3)):
        slab, off = eng.pack_slab(members, stride, slot_base=sb, slab_base=db)
        assert slab == b"".join(members)
        assert off == [sum(rng_sizes[:i]) for i in range(len(rng_sizes) + 1)]
    few = [o.prng_bytes(9, 70001, 1), o.prng_bytes(8, 65536, 1)]          # few large ranges: several workgroups per range
    slab, off = eng.pack_slab(few, 70016)
    assert slab == b"".join(few) and off == [0, 70001, 70001 + 65536]


@pytest.mark.usefixtures("inf_selection")
def test_inflate_large_streams_fast_pass(eng, o):
    """the lane-serial fast pass of the decode kernel (inflate.hip inf_fast_pass) only runs on streams with >= 4 KiB left"""
    import parity_checks
    assert parity_checks.large_stream_checks(eng.inflate, o, lambda blobs, lvl, wrap: eng.deflate(blobs, level=lvl, wrap=wrap)) > 60


@pytest.mark.usefixtures("inf_selection")
def test_multi_wave_decode_up_to_512_streams(eng, o, monkeypatch):
    """the product gives every stream of a launch of up to 512 streams a workgroup of 16 waves (round 4; `inf_selection`
    "product"; "onewave" sends the same launches through the one-wave kernel)"""
    assert parity_checks.literal_group_checks(eng.inflate, o, size=1 << 13) > 60
    assert parity_checks.fixed_code_checks(eng.inflate, o, size=1 << 13) > 40
    assert parity_checks.golden_bitstreams_exact(eng.inflate, o) >= 20
    blobs = [o.gen_shard(i % 8, 3000 + 997 * i) for i in range(40)]
    comp, st = eng.deflate(blobs, level=6, wrap=1)
    assert st == [0] * 40
    back, st = eng.inflate(comp, [len(b) for b in blobs], wrap=1)
    assert st == [0] * 40 and back == blobs


@pytest.mark.usefixtures("inf_selection")
def test_fixed_code_streams_through_the_fast_pass(eng, o):
    """BTYPE 01 blocks: the fast pass finds its lanes' starts by walking every bit phase (no self-synchronisation to live on)"""
    assert parity_checks.fixed_code_checks(eng.inflate, o) > 40


@pytest.mark.usefixtures("inf_selection")
def test_literal_groups_in_the_lane_walk(eng, o):
    """up to four literals per iteration of the decode kernel's lane walk: literal-only and skewed-alphabet streams, corrupt variants"""
    assert parity_checks.literal_group_checks(eng.inflate, o, size=1 << 14) > 60


@pytest.mark.usefixtures("inf_selection")
def test_truncated_stored_blocks_match_the_oracle():
    e = zmi_ctypes.Engine(zmi_ctypes.load_emu())
    assert parity_checks.truncated_stored_checks(lambda streams, caps, wrap: e.inflate(streams, caps, wrap=wrap), oracle_lib.load()) == 8
    e.close()


@pytest.mark.usefixtures("inf_selection")
def test_split_inflate_equals_serial_inflate():
    """one stream decoded as segments cut at its flush points (zmi_inflate_split): the results of zmi_inflate_resume,
    whatever the proposed cuts are (true markers, data that looks like one, random offsets)"""
    e = zmi_ctypes.Engine(zmi_ctypes.load_emu())
    assert parity_checks.split_inflate_checks(e, oracle_lib.load()) == 12


@pytest.mark.usefixtures("inf_selection")
def test_block_scan_inflate_equals_serial_decode(monkeypatch):
    """zmi_inflate_blocks: restart points found by the device's scan for dynamic block headers (csrc/blockscan.hip)"""
    monkeypatch.setenv("ZMI_TUNING", "1")
    monkeypatch.setenv("ZMI_BLOCKS_MIN", "20000")
    monkeypatch.setenv("ZMI_BLOCKS_GAP", "2000")
    e = zmi_ctypes.Engine(zmi_ctypes.load_emu())
    try:
        assert parity_checks.blocks_inflate_checks(e, oracle_lib.load()) == 14
    finally:
        e.close()
    e.close()


@pytest.mark.usefixtures("inf_selection")
def test_jump_resolve_equals_serial_resolve():
    """few streams: back-references resolved by pointer jumping (resolve_jump.hip) -- byte for byte the serial pass's output"""
    e = zmi_ctypes.Engine(zmi_ctypes.load_emu())
    assert parity_checks.jump_resolve_checks(e, oracle_lib.load()) == 8
    e.close()


def test_cost_parse_against_the_lazy_rule(eng, o, monkeypatch):
    """the cost parse (csrc/parse.hip, levels 3-9) against the lazy rule on the same matches: valid streams either way, text and the
    XML-like class smaller, nothing more than 2 % larger; odd sizes around the chunk (4096) and strip (64) borders, and a shard whose
    pieces are no multiple of a chunk"""
    blobs = [parity_checks.tile(dict(parity_checks.real_fixtures())["lcet10.txt"], 1 << 17), o.gen_shard(3, 1 << 17), o.gen_shard(4, 100000),
             o.gen_shard(0, 4095), o.gen_shard(1, 4097), o.gen_shard(2, 65), o.gen_shard(6, 70001), o.gen_shard(5, 12345), b"", b"a" * 300]
    sizes = {}
    for cp in ("0", "1"):
        monkeypatch.setenv("ZMI_COST_PARSE", cp)
        for lvl in (3, 6, 9):
            comp, st = eng.deflate(blobs, level=lvl, wrap=1)
            assert st == [0] * len(blobs)
            for b, c in zip(blobs, comp):
                assert zlib.decompress(c) == b
            sizes[cp, lvl] = [len(c) for c in comp]
    for lvl in (3, 6, 9):
        assert sizes["1", lvl][0] < sizes["0", lvl][0] and sizes["1", lvl][1] < sizes["0", lvl][1], (lvl, sizes)
        for a, b in zip(sizes["1", lvl], sizes["0", lvl]):
            assert a <= b * 1.02 + 8, (lvl, sizes)


def test_levels_are_distinct_rungs(eng, o):
    """the level table (csrc/zmi_api.hip kLevels; the reference's has a row per level, deflate/algorithm/mod.rs:69-82): levels 2 ... 7 give
    strictly shrinking output on text and on the XML-like class -- until round 5 levels 4 and 5 were one configuration"""
    blobs = [parity_checks.tile(dict(parity_checks.real_fixtures())["lcet10.txt"], 1 << 17), o.gen_shard(0, 1 << 17), o.gen_shard(3, 1 << 17)]
    tot = {}
    for lvl in (2, 3, 4, 5, 6, 7):
        comp, st = eng.deflate(blobs, level=lvl, wrap=1)
        assert st == [0] * len(blobs)
        for b, c in zip(blobs, comp):
            assert zlib.decompress(c) == b
        tot[lvl] = sum(len(c) for c in comp)
    assert tot[2] > tot[3] > tot[4] > tot[5] > tot[6] > tot[7], tot


def test_cost_parse_distance_slot_covers_every_distance(eng):
    """parse.hip prices a distance through par_dq(): exponent and top mantissa bit of float(2 (dist - 1) + 1).  Every distance 1 ... 32768,
    under every value of the length bits next to it in the match word, must land in the slot of its RFC 1951 distance code."""
    lib = eng.lib
    for f in (lib.zmi_emu_par_dq, lib.zmi_emu_par_slot_of_code, lib.zmi_emu_par_dist_code):
        f.restype = C.c_uint32
        f.argtypes = [C.c_uint32]
    base = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
    def code(dist):
        c = 0
        while c + 1 < 30 and base[c + 1] <= dist:
            c += 1
        return c
    for dist in range(1, 32769):
        want = code(dist)
        assert lib.zmi_emu_par_dist_code(dist) == want
        slot = want if want else 30   # (zmi_emu_par_slot_of_code, checked below for every code)
        for length in ((0, 3, 255, 258, 511) if dist < 600 or dist % 251 == 0 else (0, 511)):   # (the length's top bit sits next to the distance)
            word = 0x41 | (length << 8) | ((dist - 1) << 17)
            assert lib.zmi_emu_par_dq(word) == slot, (dist, length)
    assert [lib.zmi_emu_par_slot_of_code(c) for c in range(30)] == [30] + list(range(1, 30))
