"""GPU parity tests (-m gpu): the HIP path through the C ABI against a conformant CPU inflater /
deflater on the same seeded inputs.

Deflate direction: compressed bytes need not match the reference (BASELINE.json north_star), the
bar is that a conformant inflater (system zlib and the oracle restatement) expands every stream
bit-exactly to the input and that the stream never exceeds compress_bound
(zlib-rs/src/deflate.rs:2975-2991).  Inflate direction: bit-exact against the CPU inflater,
including the error codes for corrupt / truncated input (test-libz-rs-sys/src/inflate.rs).
"""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHARD = 1 << 20


def _to_dev(e, blobs):
    import torch
    lens = np.array([len(b) for b in blobs], dtype=np.int32)
    # 16-byte aligned starts
    offs = np.zeros(len(blobs), dtype=np.int64)
    pos = 0
    for i, b in enumerate(blobs):
        offs[i] = pos
        pos += (len(b) + 15) & ~15
    buf = np.zeros(pos + 16, dtype=np.uint8)
    for o, b in zip(offs, blobs):
        buf[o:o + len(b)] = np.frombuffer(b, dtype=np.uint8)
    return (torch.from_numpy(buf).to(e.device), torch.from_numpy(offs).to(e.device), torch.from_numpy(lens).to(e.device),
            int(lens.max()) if len(blobs) else 0)


def _deflate(e, blobs, level=6, strategy=0, wrap=1):
    import torch
    d, off, ln, mx = _to_dev(e, blobs)
    out, olen, st = e.deflate_batch(d, off, ln, mx, level=level, strategy=strategy, wrap=wrap)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    olen = olen.cpu().numpy()
    st = st.cpu().numpy()
    return [bytes(out[i, :olen[i]]) for i in range(len(blobs))], st


def _inflate(e, streams, caps, wrap=1):
    import torch
    d, off, ln, _ = _to_dev(e, streams)
    caps = np.array(caps, dtype=np.int32)
    ooff = np.zeros(len(caps), dtype=np.int64)
    pos = 0
    for i, c in enumerate(caps):
        ooff[i] = pos
        pos += (int(c) + 15) & ~15
    out = torch.zeros(pos + 16, dtype=torch.uint8, device=e.device)
    olen, st = e.inflate_batch(d, off, ln, out, torch.from_numpy(ooff).to(e.device), torch.from_numpy(caps).to(e.device),
                               wrap=wrap)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    olen = olen.cpu().numpy()
    return [bytes(out[ooff[i]:ooff[i] + min(olen[i], caps[i])]) for i in range(len(caps))], st.cpu().numpy()


def _gen(e, n, shard=SHARD, first=0):
    import torch
    t = e.gen_shards(n, shard, first_shard=first)
    torch.cuda.synchronize()
    a = t.cpu().numpy()
    return [bytes(a[i * shard:(i + 1) * shard]) for i in range(n)]


def _dec(stream, wrap):
    wb = {0: -15, 1: 15, 2: 31}[wrap]
    return zlib.decompress(stream, wb)


def test_generator_matches_cpu_twin(engine):
    import oracle_lib
    o = oracle_lib.load()
    shards = _gen(engine, 16, 1 << 16)
    for i, s in enumerate(shards):
        assert s == o.gen_shard(i, 1 << 16), "shard %d differs from the CPU generator" % i


@pytest.mark.parametrize("level", [1, 6, 9])
def test_deflate_roundtrip_full_size(engine, level):
    shards = _gen(engine, 16)
    outs, st = _deflate(engine, shards, level=level)
    assert (st == 0).all()
    tot = 0
    for s, o in zip(shards, outs):
        assert len(o) <= engine.deflate_bound(len(s))
        assert zlib.decompress(o) == s
        tot += len(o)
    ratio = len(shards) * SHARD / tot
    assert ratio > 1.8, ratio


@pytest.mark.parametrize("wrap", [0, 1, 2])
def test_deflate_edge_sizes(engine, wrap):
    rng = np.random.default_rng(7)
    text = _gen(engine, 1, 1 << 16)[0]
    blobs = [b"", b"a", b"ab", b"abc", b"abcd", b"aaaaa", bytes(1), bytes(5), bytes(300), bytes(70000), text[:63], text[:64],
             text[:65], text[:1023], text[:1024], text[:1025], text[:4097], text[:33000], text,
             rng.integers(0, 256, 100000, dtype=np.uint8).tobytes(), b"abcdefgh" * 9000, text[:777] * 40]
    outs, st = _deflate(engine, blobs, level=6, wrap=wrap)
    assert (st == 0).all()
    for b, o in zip(blobs, outs):
        assert _dec(o, wrap) == b
        assert len(o) <= engine.deflate_bound(len(b), wrap)


@pytest.mark.parametrize("level", list(range(0, 10)))
def test_deflate_all_levels(engine, level):
    shards = _gen(engine, 8, 1 << 17)
    outs, st = _deflate(engine, shards, level=level)
    assert (st == 0).all()
    for s, o in zip(shards, outs):
        assert zlib.decompress(o) == s


@pytest.mark.parametrize("strategy", [1, 2, 3, 4])
def test_deflate_strategies(engine, strategy):
    shards = _gen(engine, 8, 1 << 16)
    outs, st = _deflate(engine, shards, level=6, strategy=strategy)
    assert (st == 0).all()
    for s, o in zip(shards, outs):
        assert zlib.decompress(o) == s
    from test_emu_kernels import strategy_token_rules     # pure-Python token walk: one shard is enough
    assert strategy_token_rules(bytes(outs[0])[2:-4], strategy) == bytes(shards[0])


def test_deflate_unaligned_offsets(engine):
    import torch
    base = _gen(engine, 1, 1 << 16)[0]
    blob = np.frombuffer(base, dtype=np.uint8)
    d = torch.from_numpy(blob.copy()).to(engine.device)
    offs = np.array([1, 3, 1001, 4099], dtype=np.int64)
    lens = np.array([5000, 777, 30000, 20001], dtype=np.int32)
    out, olen, st = engine.deflate_batch(d, torch.from_numpy(offs).to(engine.device), torch.from_numpy(lens).to(engine.device),
                                         int(lens.max()))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    olen = olen.cpu().numpy()
    assert (st.cpu().numpy() == 0).all()
    for i in range(4):
        assert zlib.decompress(bytes(out[i, :olen[i]])) == base[offs[i]:offs[i] + lens[i]]


@pytest.mark.usefixtures("inf_selection")
@pytest.mark.parametrize("wrap,level", [(1, 6), (2, 6), (0, 6), (1, 1), (1, 9), (2, 0)])
def test_inflate_matches_cpu(engine, wrap, level):
    shards = _gen(engine, 16)
    wb = {0: -15, 1: 15, 2: 31}[wrap]
    streams = []
    for s in shards:
        co = zlib.compressobj(level, zlib.DEFLATED, wb)
        streams.append(co.compress(s) + co.flush())
    outs, st = _inflate(engine, streams, [SHARD] * len(shards), wrap=wrap)
    assert (st == 0).all(), st
    for s, o in zip(shards, outs):
        assert o == s


@pytest.mark.usefixtures("inf_selection")
def test_inflate_of_gpu_deflate(engine):
    shards = _gen(engine, 16)
    for level in (1, 6, 9):
        outs, st = _deflate(engine, shards, level=level, wrap=2)
        assert (st == 0).all()
        back, st2 = _inflate(engine, outs, [SHARD] * len(shards), wrap=2)
        assert (st2 == 0).all()
        assert back == shards


@pytest.mark.usefixtures("inf_selection")
def test_inflate_errors(engine):
    s = _gen(engine, 1, 1 << 16)[0]
    good = zlib.compress(s, 6)
    bad_body = bytearray(good)
    bad_body[200] ^= 0x5A
    bad_check = bytearray(good)
    bad_check[-1] ^= 1
    bad_hdr = bytearray(good)
    bad_hdr[0] = 0x79
    streams = [good, bytes(bad_body), bytes(bad_check), bytes(bad_hdr), good[:1000], good, b"\x78\x9c\x07"]
    caps = [len(s)] * 5 + [100] + [100]
    outs, st = _inflate(engine, streams, caps, wrap=1)
    import oracle_lib
    o = oracle_lib.load(rebuild=False)
    assert st[0] == 0 and outs[0] == s
    rc1 = o.inflate(bytes(bad_body), len(s), 1)[0]
    assert st[1] == (0 if rc1 == 1 else rc1) and rc1 != 1     # corrupt body: exactly the oracle's code for this stream
    assert st[2] == -3                  # "incorrect data check"
    assert st[3] == -3                  # "incorrect header check"
    assert st[4] == -5                  # truncated input
    assert st[5] == -5                  # output buffer too small
    assert st[6] == -3                  # invalid block type (BTYPE=3)


@pytest.mark.usefixtures("inf_selection")
def test_inflate_unaligned_layout_and_patterns(engine):
    """odd input / output offsets (the 4- and 16-byte aligned paths are not taken), guard bytes behind every
    stream's capacity, long distances, self-overlapping runs, many tiny blocks, a hole-free stretch longer than
    the resolve ring between regions with back-references"""
    import torch
    rng = np.random.default_rng(7)
    r = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    text = _gen(engine, 1, 1 << 16)[0]
    co = zlib.compressobj(6, zlib.DEFLATED, 15)
    tiny = b"".join(co.compress(text[i:i + 97]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, 30000, 97)) + co.flush()
    blobs = [r + r + r[:700], bytes(70000), b"ab" * 20000 + b"xyz" * 9000,
             text[:5000] + rng.integers(0, 256, 90000, dtype=np.uint8).tobytes() + text[:5000] + r[100:900],
             text[:97 * len(range(0, 30000, 97))]]
    streams = [zlib.compress(b, 9) for b in blobs[:4]] + [tiny]
    caps = [len(b) for b in blobs]
    ioff, ooff, a, o = [], [], 3, 5
    for s_, c in zip(streams, caps):
        ioff.append(a); a += len(s_) + 7
        ooff.append(o); o += c + 3
    d = torch.zeros(a + 64, dtype=torch.uint8)
    for off_, s_ in zip(ioff, streams):
        d[off_:off_ + len(s_)] = torch.frombuffer(bytearray(s_), dtype=torch.uint8)
    d = d.to(engine.device)
    out = torch.full((o + 64,), 0xEE, dtype=torch.uint8, device=engine.device)
    olen, st = engine.inflate_batch(d, torch.tensor(ioff, dtype=torch.int64, device=engine.device),
                                    torch.tensor([len(s_) for s_ in streams], dtype=torch.int32, device=engine.device), out,
                                    torch.tensor(ooff, dtype=torch.int64, device=engine.device),
                                    torch.tensor(caps, dtype=torch.int32, device=engine.device), wrap=1)
    torch.cuda.synchronize()
    assert st.cpu().tolist() == [0] * len(blobs)
    h = out.cpu().numpy()
    for off_, c, b in zip(ooff, caps, blobs):
        assert bytes(h[off_:off_ + c]) == b
        assert bytes(h[off_ + c:off_ + c + 3]) == b"\xee\xee\xee"   # nothing written past the capacity


@pytest.mark.usefixtures("inf_selection")
def test_fuzz_roundtrip_mixed(engine):
    """ragged random batches both ways against system zlib: random sizes (0 .. 300 KiB), data classes, levels,
    strategies, wrappers and window sizes"""
    rng = np.random.default_rng(20260924)
    pool = b"".join(_gen(engine, 8, 1 << 18))
    for round_ in range(6):
        n = int(rng.integers(20, 60))
        shards = []
        for _ in range(n):
            ln = int(rng.choice([0, 1, 2, 3, 5, 63, 64, 65, 1000, 4096, 65535, 65536, 65537, int(rng.integers(0, 300000))]))
            at = int(rng.integers(0, len(pool) - ln + 1))
            shards.append(pool[at:at + ln])
        level, wrap, strategy = int(rng.integers(0, 10)), int(rng.integers(0, 3)), int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
        outs, st = _deflate(engine, shards, level=level, wrap=wrap, strategy=strategy)
        assert (st == 0).all(), (level, wrap, strategy)
        for s_, o in zip(shards, outs):
            assert _dec(o, wrap) == s_, (level, wrap, strategy, len(s_))
        back, st2 = _inflate(engine, outs, [len(s_) for s_ in shards], wrap=wrap)
        assert (st2 == 0).all() and back == shards
        # streams made by system zlib with random parameters, inflated on the GPU
        wbits = int(rng.integers(9, 16))
        zs = []
        for s_ in shards:
            co = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, {0: -wbits, 1: wbits, 2: 16 + wbits}[wrap], 8,
                                  int(rng.choice([0, 1, 2, 3, 4])))
            zs.append(co.compress(s_) + co.flush())
        back, st3 = _inflate(engine, zs, [len(s_) for s_ in shards], wrap=wrap)
        assert (st3 == 0).all() and back == shards, (wrap, wbits)


def test_checksums(engine):
    import torch
    blobs = [b"", b"a", b"abc", bytes(range(256)) * 300] + _gen(engine, 4, 1 << 16) + _gen(engine, 2)
    d, off, ln, _ = _to_dev(engine, blobs)
    a, c = engine.checksums(d, off, ln)
    torch.cuda.synchronize()
    a = a.cpu().numpy().astype(np.uint32)
    c = c.cpu().numpy().astype(np.uint32)
    for i, b in enumerate(blobs):
        assert int(a[i]) == zlib.adler32(b), i
        assert int(c[i]) == zlib.crc32(b), i


@pytest.mark.usefixtures("inf_selection")
def test_resumable_inflate_from_block_checkpoints():
    """zmi_inflate_resume through the C ABI on the MI355X: cut streams restart at the reported block boundary"""
    import oracle_lib
    import resume_checks
    import zmi_ctypes
    eng = zmi_ctypes.Engine(zmi_ctypes.load_product())
    try:
        resume_checks.resume_chain_checks(eng, oracle_lib.load(rebuild=False), trials=6)
    finally:
        eng.close()


@pytest.mark.usefixtures("inf_selection")
def test_host_batch_pipeline_on_gpu(monkeypatch):
    """zmi_deflate_batch / zmi_inflate_batch with host buffers: chunks cycle through two device slots on three HIP
    streams (copy-in / kernels / copy-out overlap); same bytes as the one-chunk path, bit-exact round trip"""
    import oracle_lib
    import zmi_ctypes
    o = oracle_lib.load(rebuild=False)
    shards = [o.gen_shard(i, 1 << 20) for i in range(40)] + [o.gen_shard(3, 12345), b"", o.gen_shard(5, 700001)]
    eng = zmi_ctypes.Engine(zmi_ctypes.load_product())
    try:
        monkeypatch.setenv("ZMI_HOST_PIPELINE", "0")
        one, st1 = eng.deflate(shards, level=6, wrap=2)
        monkeypatch.setenv("ZMI_HOST_PIPELINE", "1")
        monkeypatch.setenv("ZMI_HOST_CHUNK", str(6 << 20))          # 8 chunks
        for _ in range(2):
            many, st2 = eng.deflate(shards, level=6, wrap=2)
            assert st1 == st2 == [0] * len(shards) and many == one
        assert [zlib.decompress(x, 31) for x in one[::7]] == shards[::7]
        # (the copy-out of a chunk follows what the chunks before it decoded: one DMA copy of the region when the room was used,
        # decoded bytes range by range when it was not -- tight, loose, loose, tight, tight on one context)
        for caps in ([(len(x) + 15) & ~15 for x in shards], [len(x) + 1 for x in shards], [3 * len(x) + 4096 for x in shards],
                     [3 * len(x) + 4096 for x in shards], [len(x) + 1 for x in shards], [len(x) + 1 for x in shards]):
            got, gst = eng.inflate(one, caps, wrap=2)
            assert gst == [0] * len(shards) and got == shards
    finally:
        eng.close()


# ---- round 2: the reference's golden inflate vectors through the BATCH instantiation of the decode kernel (the one the
# benchmark times), with the oracle's exact return code per stream; crafted resolve-pass geometry; real files ----
def _inflate_fn(engine):
    return lambda streams, caps, wrap: _inflate(engine, streams, caps, wrap=wrap)


@pytest.mark.usefixtures("inf_selection")
def test_golden_inflate_bitstreams_through_the_batch_kernel(engine):
    import oracle_lib
    import parity_checks
    assert parity_checks.golden_bitstreams_exact(_inflate_fn(engine), oracle_lib.load(rebuild=False)) >= 20


@pytest.mark.usefixtures("inf_selection")
def test_golden_inflate_files_through_the_batch_kernel(engine):
    import oracle_lib
    import parity_checks
    assert parity_checks.golden_files_exact(_inflate_fn(engine), oracle_lib.load(rebuild=False)) >= 10


@pytest.mark.usefixtures("inf_selection")
def test_corrupt_streams_report_the_oracles_code(engine):
    import oracle_lib
    import parity_checks
    o = oracle_lib.load(rebuild=False)
    for cls in (0, 5):
        assert parity_checks.corrupt_streams_exact(_inflate_fn(engine), o, _gen(engine, 8, 1 << 16)[cls]) > 30


@pytest.mark.usefixtures("inf_selection")
def test_resolve_window_edge_on_gpu(engine):
    """ring restart of the resolve pass at p % 1024 in 1021..1023 with distances 32766..32768"""
    import oracle_lib
    from test_emu_kernels import resolve_window_edge_streams
    cases = resolve_window_edge_streams(oracle_lib.load(rebuild=False))
    outs, st = _inflate(engine, [c for c, _ in cases], [len(w) for _, w in cases], wrap=0)
    assert (st == 0).all()
    assert outs == [w for _, w in cases]


@pytest.mark.usefixtures("inf_selection")
def test_resolve_near_far_boundary_on_gpu(engine):
    """back-references around RES_NEAR (ring vs HBM source), sources that are earlier holes, mixed batches"""
    import oracle_lib
    from test_emu_kernels import resolve_near_far_streams
    cases = resolve_near_far_streams(oracle_lib.load(rebuild=False))
    outs, st = _inflate(engine, [c for c, _ in cases], [len(w) for _, w in cases], wrap=0)
    assert (st == 0).all()
    assert outs == [w for _, w in cases]


def test_real_fixtures_roundtrip_and_ratio_vs_oracle(engine):
    """SURVEY 8(d) non-synthetic cross-check: lcet10.txt, paper-100k.pdf, fireworks.jpg
    (test-libz-rs-sys/src/deflate.rs:1982-2003) tiled to 1 MiB, levels 1 / 6 / 9: a conformant inflater and the GPU
    inflater give the input back bit-exactly; the GPU's ratio is reported beside the oracle's (the reference's
    algorithm at the same level); level 6 may not be more than 1 % worse, level 9 not more than 1.5 % (round 5: the cost parse,
    csrc/parse.hip; the gate was 3 % while the parse was the three-deep lazy rule.  Level 9 measured: lcet10.txt 0.999 / 1.001,
    fireworks.jpg 0.999, paper-100k.pdf 0.990 / 0.994 -- the reference's level 9 also takes 3-byte matches, this engine's search does not)."""
    import json
    import os
    import oracle_lib
    import parity_checks
    o = oracle_lib.load(rebuild=False)
    files = parity_checks.real_fixtures()
    shards = [parity_checks.tile(raw) for _, raw in files] + [raw for _, raw in files]      # tiled and as they are
    names = [n + " (tiled to 1 MiB)" for n, _ in files] + [n for n, _ in files]
    table = {}
    for level in (1, 6, 9):
        outs, st = _deflate(engine, shards, level=level, wrap=2)
        assert (st == 0).all()
        for s_, c in zip(shards, outs):
            assert zlib.decompress(c, 31) == s_
            assert len(c) <= engine.deflate_bound(len(s_), 2)
        back, st2 = _inflate(engine, outs, [len(s_) for s_ in shards], wrap=2)
        assert (st2 == 0).all() and back == shards
        # the oracle's own gzip members of the same inputs, inflated on the GPU (configs[2] on real data)
        members = [o.deflate(s_, level, 2)[1] for s_ in shards]
        back, st3 = _inflate(engine, members, [len(s_) for s_ in shards], wrap=2)
        assert (st3 == 0).all() and back == shards
        for n, s_, c, m in zip(names, shards, outs, members):
            table.setdefault(n, {})["L%d" % level] = {"gpu_ratio": len(s_) / len(c), "oracle_ratio": len(s_) / len(m)}
    print(json.dumps(table, indent=1))
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(table, open(os.path.join("gpurun_out", "real_fixture_ratios.json"), "w"), indent=1)
    except OSError:
        pass
    for n, row in table.items():
        assert row["L6"]["gpu_ratio"] >= 0.99 * row["L6"]["oracle_ratio"], (n, row)
        assert row["L9"]["gpu_ratio"] >= 0.985 * row["L9"]["oracle_ratio"], (n, row)


def test_pack_slab_and_global_stitch_on_gpu(engine):
    """the batch's strided slots -> one dense slab (what a host write-out or a peer receives), then the
    scatter of two ranks' slabs into the globally ordered multi-member gzip file (round-robin ownership) -- all on the
    device (csrc/pack.hip); compared with b"".join(members) and read back by gzip"""
    import gzip
    import torch
    from zlib_rs_amd import dist as zd
    world, n_local, B = 2, 24, 1 << 18
    outs, tables, slabs = [], [], []
    for r in range(world):
        data = engine.gen_shards(n_local, B, first_shard=r, shard_step=world)
        off = torch.arange(n_local, dtype=torch.int64, device=engine.device) * B
        ln = torch.full((n_local,), B, dtype=torch.int32, device=engine.device)
        out, olen, st = engine.deflate_batch(data, off, ln, B, level=6, wrap=2)
        torch.cuda.synchronize()
        assert (st == 0).all()
        slab, so = engine.pack_slab(out, olen)
        torch.cuda.synchronize()
        host, hl = out.cpu().numpy(), olen.cpu().numpy()
        members = [bytes(host[i, :hl[i]]) for i in range(n_local)]
        assert bytes(slab[:int(so[-1])].cpu().numpy()) == b"".join(members)
        assert so.cpu().tolist() == [sum(int(x) for x in hl[:i]) for i in range(n_local + 1)]
        outs.append(members); tables.append(olen.cpu()); slabs.append(slab[:int(so[-1])])
    table = torch.stack(tables)
    stitched, total = zd.stitch_on_device(engine, slabs, table)
    torch.cuda.synchronize()
    blob = bytes(stitched[:total].cpu().numpy())
    assert blob == b"".join(outs[g % world][g // world] for g in range(world * n_local))
    want = b"".join(_gen(engine, world * n_local, B))
    assert gzip.decompress(blob) == want


def test_long_matches_are_extended_by_one_lane_per_run_on_gpu(engine):
    """lz77.hip, short budgets: a walk stops at 16 equal bytes; one lane per run of equal distances finds the real length"""
    import oracle_lib
    import parity_checks
    def deflate(blobs, level):
        outs, st = _deflate(engine, blobs, level=level, wrap=1)
        assert all(int(x) == 0 for x in st)
        return outs
    assert parity_checks.long_match_checks(deflate, oracle_lib.load(rebuild=False), scale=8) == 36


@pytest.mark.usefixtures("inf_selection")
def test_inflate_large_streams_fast_pass_on_gpu(engine):
    import oracle_lib
    import parity_checks
    o = oracle_lib.load(rebuild=False)
    n = parity_checks.large_stream_checks(_inflate_fn(engine), o, lambda blobs, lvl, wrap: _deflate(engine, blobs, level=lvl, wrap=wrap),
                                          size=1 << 19)
    assert n > 60


@pytest.mark.usefixtures("inf_selection")
def test_truncated_stored_blocks_match_the_oracle_on_gpu(engine):
    import oracle_lib
    import parity_checks
    assert parity_checks.truncated_stored_checks(_inflate_fn(engine), oracle_lib.load(rebuild=False)) == 8


@pytest.mark.usefixtures("inf_selection")
def test_fixed_code_streams_through_the_fast_pass_on_gpu(engine):
    """BTYPE 01 blocks (Z_FIXED, the reference's level 1): every class, corrupt / truncated variants with the oracle's codes"""
    import oracle_lib
    import parity_checks
    assert parity_checks.fixed_code_checks(_inflate_fn(engine), oracle_lib.load(rebuild=False), size=1 << 19) > 40


@pytest.mark.usefixtures("inf_selection")
def test_literal_groups_in_the_lane_walk_on_gpu(engine):
    """up to four literals per iteration of the decode kernel's lane walk (literal-only and skewed-alphabet streams, corrupt variants)"""
    import oracle_lib
    import parity_checks
    assert parity_checks.literal_group_checks(_inflate_fn(engine), oracle_lib.load(rebuild=False), size=1 << 19) > 60


@pytest.mark.usefixtures("inf_selection")
def test_multi_wave_decode_up_to_512_streams_on_gpu(engine, monkeypatch):
    """launches of up to 512 streams take the multi-wave kernel in the product (`inf_selection` "product"), under "onewave" the one-wave kernel"""
    import oracle_lib
    import parity_checks
    o = oracle_lib.load(rebuild=False)
    assert parity_checks.literal_group_checks(_inflate_fn(engine), o, size=1 << 18) > 60
    assert parity_checks.fixed_code_checks(_inflate_fn(engine), o, size=1 << 18) > 40
    assert parity_checks.golden_bitstreams_exact(_inflate_fn(engine), o) >= 20
    blobs = [o.gen_shard(i % 8, 300000 + 9973 * i) for i in range(300)]
    comp, st = _deflate(engine, blobs, level=6, wrap=1)
    assert all(int(x) == 0 for x in st)
    back, st = _inflate(engine, comp, [len(b) for b in blobs], wrap=1)
    assert all(int(x) == 0 for x in st) and back == blobs


@pytest.mark.usefixtures("inf_selection")
def test_split_inflate_equals_serial_inflate_on_gpu():
    """one stream decoded as segments cut at its flush points, on the whole chip (zmi_inflate_split): the results of the
    serial zmi_inflate_resume for true markers, false ones, history, corruption, short room"""
    import oracle_lib
    import parity_checks
    import zmi_ctypes
    eng = zmi_ctypes.Engine(zmi_ctypes.load_product())
    assert parity_checks.split_inflate_checks(eng, oracle_lib.load(rebuild=False), big=True) == 12
    eng.close()


@pytest.mark.usefixtures("inf_selection")
def test_block_scan_inflate_equals_serial_inflate_on_gpu():
    """one stream WITHOUT flush points: the device finds its dynamic block headers (csrc/blockscan.hip) and decodes the blocks
    side by side (zmi_inflate_blocks) -- the results of the serial zmi_inflate_resume for streams of the system zlib and of the
    oracle, stored / fixed blocks and decoy headers in between, history, truncation, corruption, short room, a start inside a byte"""
    import oracle_lib
    import parity_checks
    import zmi_ctypes
    eng = zmi_ctypes.Engine(zmi_ctypes.load_product())
    assert parity_checks.blocks_inflate_checks(eng, oracle_lib.load(rebuild=False), big=True) == 14
    eng.close()


@pytest.mark.usefixtures("inf_selection")
def test_uncompress_of_a_large_stream_takes_the_parallel_path_and_keeps_the_reference_codes():
    """uncompress() of streams of 256 KiB and more goes through zmi_inflate_blocks (tests/zlib_abi_harness.py::uncompress_large_checks)"""
    import ctypes as C
    import oracle_lib
    import zlib_abi_harness as H
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
    assert H.uncompress_large_checks(lib, oracle_lib.load(rebuild=False), 1 << 20) == 6


@pytest.mark.usefixtures("inf_selection")
def test_jump_resolve_equals_serial_resolve_on_gpu():
    """few streams: back-references resolved by pointer jumping on the whole chip (csrc/resolve_jump.hip) -- byte for byte
    the output of the one-wave-per-stream pass, incl. a 300 KB run of one byte, history and a corrupt stream"""
    import oracle_lib
    import parity_checks
    import zmi_ctypes
    eng = zmi_ctypes.Engine(zmi_ctypes.load_product())
    assert parity_checks.jump_resolve_checks(eng, oracle_lib.load(rebuild=False), big=True) == 8
    eng.close()


def test_cost_parse_against_the_lazy_rule_on_gpu(engine, monkeypatch):
    """levels 3-9 choose their tokens by price (csrc/parse.hip).  Against the lazy rule on the same matches (ZMI_COST_PARSE=0, a test
    override): valid streams either way, real text at least 1.5 % smaller, no data class more than 1.5 % larger (record-like data
    loses ~0.9 %: pricing from one's own statistics), the mix smaller."""
    import oracle_lib
    import parity_checks
    o = oracle_lib.load(rebuild=False)
    blobs = [parity_checks.tile(raw) for _, raw in parity_checks.real_fixtures()][:1] + [o.gen_shard(i, SHARD) for i in range(8)]
    sizes = {}
    for cp in ("0", "1"):
        monkeypatch.setenv("ZMI_COST_PARSE", cp)
        outs, st = _deflate(engine, blobs, level=6, wrap=1)
        assert (st == 0).all()
        for b, c in zip(blobs, outs):
            assert zlib.decompress(c) == b
        sizes[cp] = [len(c) for c in outs]
    monkeypatch.delenv("ZMI_COST_PARSE")
    assert sizes["1"][0] <= sizes["0"][0] * 0.985, ("lcet10.txt", sizes)
    for i in range(1, 9):
        assert sizes["1"][i] <= sizes["0"][i] * 1.015, (i, sizes)
    assert sum(sizes["1"][1:]) < sum(sizes["0"][1:])
