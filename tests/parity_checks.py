"""Parity checks shared by the emulator tests (CPU) and the -m gpu tests: the same seeded inputs go through the
batch entry points (`zmi_inflate_batch*` -- the instantiation the benchmark times) and through the CPU oracle, and
the per-stream return codes and bytes must be identical.

`inflate_fn(streams, caps, wrap) -> (outputs, statuses)`; statuses in zlib numbering with 0 = complete stream
(the oracle's zo_inflate returns Z_STREAM_END = 1 for that)."""
import base64
import json
import lzma
import os
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))


def _want(o, stream, cap, wrap):
    rc, out, _, msg = o.inflate(stream, cap, wrap)
    return (0 if rc == 1 else rc), out, msg


def golden_bitstreams_exact(inflate_fn, o):
    """the reference's hand-made inflate bitstreams (test-libz-rs-sys/src/inflate.rs:734-1030, extracted to
    tests/golden/inflate_vectors.json): every stream must end with exactly the code the oracle reports -- Z_OK with
    identical bytes, Z_DATA_ERROR where the reference's test expects it, Z_BUF_ERROR for the deliberately cut ones"""
    inf = json.load(open(os.path.join(HERE, "golden", "inflate_vectors.json")))
    streams = [bytes.fromhex(v["input"]) for v in inf["bitstreams"]]
    n = 0
    for wrap in (0, 3):
        idx = [i for i, v in enumerate(inf["bitstreams"]) if v["wrap"] == wrap]
        caps = [8 * len(streams[i]) + 64 for i in idx]
        outs, st = inflate_fn([streams[i] for i in idx], caps, wrap)
        for k, i in enumerate(idx):
            v = inf["bitstreams"][i]
            rc, want, msg = _want(o, streams[i], caps[k], wrap)
            assert int(st[k]) == rc, (v["source"], int(st[k]), rc, msg)
            if v["expect"] == "data_error":
                assert rc == -3, (v["source"], rc)
            if rc == 0:
                assert outs[k] == want, v["source"]
            n += 1
    return n


def golden_files_exact(inflate_fn, o):
    """the reference's binary inflate fixtures (test-libz-rs-sys/src/test-data: window-match-bug.zraw,
    op-len-edge-case.zraw, text.gz, issue-109.gz, compression-corpus/*.gz)"""
    inf = json.load(open(os.path.join(HERE, "golden", "inflate_vectors.json")))
    for wrap in (0, 2):
        files = [f for f in inf["files"] if f["wrap"] == wrap]
        raws = [base64.b64decode(f["data_b64"]) for f in files]
        outs, st = inflate_fn(raws, [f["out_len"] for f in files], wrap)
        for f, out, s, raw in zip(files, outs, st, raws):
            assert int(s) == 0, (f["source"], int(s))
            assert len(out) == f["out_len"] and zlib.crc32(out) == f["out_crc32"] and zlib.adler32(out) == f["out_adler32"], f["source"]
            assert out == o.inflate(raw, f["out_len"], wrap)[1]
        # one byte of room too few / input cut short: the oracle's code, whatever it is
        outs, st = inflate_fn(raws, [max(0, f["out_len"] - 1) for f in files], wrap)
        for f, s, raw in zip(files, st, raws):
            assert int(s) == _want(o, raw, max(0, f["out_len"] - 1), wrap)[0], f["source"]
        cut = [r[:len(r) * 2 // 3] for r in raws]
        outs, st = inflate_fn(cut, [f["out_len"] for f in files], wrap)
        for f, s, raw in zip(files, st, cut):
            assert int(s) == _want(o, raw, f["out_len"], wrap)[0], f["source"]
    return len(inf["files"])


def corrupt_streams_exact(inflate_fn, o, data):
    """bit flips, truncations, wrong trailers: the status of every stream equals the oracle's for that exact stream
    (replaces the `in (-3, -5)` of round 1)"""
    good = zlib.compress(data, 6)
    streams, caps = [good], [len(data)]
    for at in (2, 5, 40, 200, 777, len(good) // 2, len(good) - 6, len(good) - 1):
        for mask in (0x01, 0x5A, 0x80):
            b = bytearray(good)
            b[at] ^= mask
            streams.append(bytes(b)); caps.append(len(data))
    for cut in (0, 1, 2, 3, 10, 1000, len(good) - 5, len(good) - 4, len(good) - 1):
        streams.append(good[:cut]); caps.append(len(data))
    streams += [good, good, b"\x78\x9c\x07", b"\x79\x9c\x03\x00", b"\x78\x9c\x03\x00\x00\x00\x00\x01", b"\x78\x9c\x03\x00\x00\x00\x00\x00"]
    caps += [100, 0, 100, 100, 100, 100]
    outs, st = inflate_fn(streams, caps, 1)
    for i, (s_, c) in enumerate(zip(streams, caps)):
        rc, want, msg = _want(o, s_, c, 1)
        assert int(st[i]) == rc, (i, int(st[i]), rc, msg)
        if rc == 0:
            assert outs[i] == want
    return len(streams)


def real_fixtures():
    """[(name, bytes)] of lcet10.txt, paper-100k.pdf, fireworks.jpg (test-libz-rs-sys/src/deflate.rs:1982-2003)"""
    fx = os.path.join(HERE, "golden", "fixtures")
    out = []
    for m in json.load(open(os.path.join(fx, "manifest.json"))):
        raw = lzma.decompress(open(os.path.join(fx, m["name"] + ".xz"), "rb").read())
        assert len(raw) == m["bytes"] and zlib.crc32(raw) == m["crc32"], m["name"]
        out.append((m["name"], raw))
    return out


def tile(raw, size=1 << 20):
    return (raw * (size // len(raw) + 1))[:size]


def large_stream_checks(inflate_fn, o, deflate_fn=None, size=1 << 17):
    """Streams long enough for the decode kernel's lane-serial fast pass (it needs >= 4 KiB of input behind the current
    position; the small vectors above never get there): every data class, three zlib levels, three wrappers, this
    engine's own output (many blocks and byte-aligned pieces), and -- with the oracle's exact code per stream -- corrupt
    variants, every kind of capacity shortfall and inputs cut at many points around the 4 KiB switch-over."""
    blobs = [o.gen_shard(c, size) for c in range(8)] + [bytes(size + 999), o.prng_bytes(3, size - 7, 1), b"ab" * (size // 3)]
    for lvl in (1, 6, 9):
        for wrap, wb in ((1, 15), (2, 31), (0, -15)):
            streams = []
            for b in blobs:
                co = zlib.compressobj(lvl, zlib.DEFLATED, wb)
                streams.append(co.compress(b) + co.flush())
            outs, st = inflate_fn(streams, [len(b) for b in blobs], wrap)
            assert [int(x) for x in st] == [0] * len(blobs), (lvl, wrap, list(st))
            assert outs == blobs, (lvl, wrap)
    if deflate_fn is not None:
        outs, st = deflate_fn(blobs, 6, 2)
        back, st2 = inflate_fn(outs, [len(b) for b in blobs], 2)
        assert [int(x) for x in st2] == [0] * len(blobs) and back == blobs
    n = corrupt_streams_exact(inflate_fn, o, o.gen_shard(2, size)) + corrupt_streams_exact(inflate_fn, o, o.gen_shard(5, size // 2))
    d = o.gen_shard(0, size)
    good = zlib.compress(d, 6)
    caps = [len(d) - 1, len(d), len(d) - 5000, size // 2, 0, 1]
    outs, st = inflate_fn([good] * len(caps), caps, 1)
    for c, s_, ou in zip(caps, st, outs):
        rc, want, _ = _want(o, good, c, 1)
        assert int(s_) == rc, (c, int(s_), rc)
        if rc == 0:
            assert ou == want
    cuts = [len(good) - k for k in (1, 2, 3, 4, 5, 6, 100, 4000, 4097, 4127, 4128, 4129, 4200, 5000, 9000, 20000) if k < len(good)]
    outs, st = inflate_fn([good[:c] for c in cuts], [len(d)] * len(cuts), 1)
    for c, s_ in zip(cuts, st):
        assert int(s_) == _want(o, good[:c], len(d), 1)[0], c
    return n


def long_match_checks(deflate_fn, o, scale=1):
    """The match search's second half at the short budgets (lz77.hip, round 4): a walk stops at the first candidate equal in 16
    bytes, and the real length is found afterwards by ONE lane per run of neighbouring positions with the same distance -- the
    others take the leader's length minus their offset.  Inputs whose compressed size hangs on exactly that: runs (distance 1,
    capped at 258), records repeated at distances above and below 258, matches of 16..40 bytes placed across the 64-position
    claims, a match that ends with the shard.  Every stream must round-trip and stay within a few percent of zlib level 6 on the
    same bytes -- a quarter at level 1 -- (followers that kept 16 instead of their real length would double these sizes).  deflate_fn(blobs, level) ->
    list of zlib streams."""
    rnd = lambda seed, n: o.prng_bytes(seed, n, 1)
    blobs = [bytes(70000 * scale),
             rnd(1, 300) * (100 * scale),
             rnd(2, 40) * (400 * scale),
             rnd(3, 17) * (500 * scale),
             o.gen_shard(3, (1 << 15) * scale),
             o.gen_shard(4, (1 << 15) * scale)]
    # matches of 16..40 bytes at every alignment against the 64-position claims, the rest incompressible
    parts, seed = [], 10
    for k in range(16, 41):
        x = rnd(seed, k); seed += 1
        parts += [rnd(seed, 50 + k), x, rnd(seed + 1, 61 + 3 * k), x]; seed += 2
    blobs.append(b"".join(parts) * scale)
    # a long match that ends exactly where the shard ends, and one cut short by it
    body = rnd(90, 5000)
    blobs.append(body + rnd(91, 777) + body)
    blobs.append(body + rnd(92, 333) + body[:1234])
    n = 0
    for level in (1, 4, 6, 7):
        outs = deflate_fn(blobs, level)
        for b, c in zip(blobs, outs):
            assert zlib.decompress(c) == b, (level, len(b))
            ref = len(zlib.compress(b, 6))
            assert len(c) <= ref * (1.25 if level < 4 else 1.12) + 96 + len(b) // 200, (level, len(b), len(c), ref)   # (+0.5 % of the input: block and piece overheads)
            n += 1
    return n


def fixed_code_checks(inflate_fn, o, size=1 << 16):
    """Streams of FIXED-Huffman blocks (BTYPE 01: Z_FIXED, and what the reference's level 1 emits, deflate/algorithm/quick.rs:12-158)
    through the batch kernel's fast pass, whose lanes find their starts by walking every bit phase (inflate.hip inf_fixed_tracks):
    every data class at two levels in one launch of more than 16 streams (the one-wave-per-stream kernel) and one by one (the
    16-wave kernel), then corrupt and truncated variants with the oracle's exact code per stream, short capacities, and a mix
    of fixed and dynamic blocks in one stream."""
    blobs = [o.gen_shard(c, size + 1000 * c) for c in range(8)] + [bytes(size), o.prng_bytes(9, size // 2, 1), b"abcdefgh" * (size // 8)]
    streams, want = [], []
    for lvl in (1, 6):
        for b in blobs:
            co = zlib.compressobj(lvl, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
            streams.append(co.compress(b) + co.flush()); want.append(b)
    # fixed and dynamic blocks alternating in one stream (full flush between parts)
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    part1 = co.compress(blobs[0]) + co.flush(zlib.Z_FULL_FLUSH)
    mixed = bytearray(part1)
    rawd = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = rawd.compress(blobs[3]) + rawd.flush(zlib.Z_FULL_FLUSH)
    rawf = zlib.compressobj(1, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    tail = rawf.compress(blobs[6]) + rawf.flush()
    whole = blobs[0] + blobs[3] + blobs[6]
    mixed += body + tail + zlib.adler32(whole).to_bytes(4, "big")
    streams.append(bytes(mixed)); want.append(whole)
    outs, st = inflate_fn(streams, [len(w) for w in want], 1)
    assert [int(x) for x in st] == [0] * len(streams), list(st)
    assert outs == want
    for s_, w in list(zip(streams, want))[:4]:          # the 16-wave kernel (launches of a few streams)
        o1, s1 = inflate_fn([s_], [len(w)], 1)
        assert int(s1[0]) == 0 and o1[0] == w
    # corrupt / truncated / short of room: the oracle's code for the exact stream
    good, d = streams[2], want[2]
    bad, caps = [], []
    for at in (3, 100, len(good) // 3, len(good) // 2, len(good) - 9, len(good) - 2):
        for mask in (0x01, 0x40):
            b = bytearray(good); b[at] ^= mask
            bad.append(bytes(b)); caps.append(len(d))
    for cut in (5, 3000, 4200, 9000, len(good) - 4, len(good) - 1):
        bad.append(good[:cut]); caps.append(len(d))
    for cap in (len(d) - 1, len(d) // 2, 1):
        bad.append(good); caps.append(cap)
    bad += streams[:8]; caps += [len(w) for w in want[:8]]          # (filling the launch beyond 16 streams)
    outs, st = inflate_fn(bad, caps, 1)
    for i, (s_, c) in enumerate(zip(bad, caps)):
        rc, w, msg = _want(o, s_, c, 1)
        assert int(st[i]) == rc, (i, int(st[i]), rc, msg)
        if rc == 0:
            assert outs[i] == w
    return len(streams) + len(bad)


def literal_group_checks(inflate_fn, o, size=1 << 16):
    """The lane walk of the decode kernel takes up to four literals per iteration (inflate.hip inf_lane_decode, round 4): a literal
    is followed along while the next code is a first-level table entry that still starts inside the lane's sub-sequence.  Streams
    that are nothing but literals (Z_HUFFMAN_ONLY) and streams with literals of every code length -- a skewed alphabet gives
    codes from 1-2 bits up to 13-15, i.e. second-level entries in the middle of literal runs -- with dynamic codes and small
    blocks, in one launch of more than 16 streams and one by one; corrupt / truncated variants and short capacities must give
    the oracle's code and prefix for the exact stream."""
    skew = bytearray()
    x = 12345
    for i in range(size):   # geometric alphabet: byte k with probability ~2^-(k/6)
        x = (x * 1103515245 + 12345) & 0x7FFFFFFF
        k, r = 0, x
        while (r & 1) and k < 250:
            r >>= 1
            k += 6
        skew.append((k + (x >> 20) % 6) & 0xFF)
    blobs = [o.gen_shard(c, size + 777 * c) for c in (0, 4, 5, 6, 7)] + [bytes(skew), o.prng_bytes(5, size, 1), bytes(range(256)) * (size // 256)]
    streams, want = [], []
    for strat in (zlib.Z_HUFFMAN_ONLY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_RLE):
        for b in blobs:
            co = zlib.compressobj(6, zlib.DEFLATED, 15, 1, strat)   # memLevel 1: a block every ~500 symbols
            streams.append(co.compress(b) + co.flush()); want.append(b)
            co = zlib.compressobj(6, zlib.DEFLATED, 15, 9, strat)
            streams.append(co.compress(b) + co.flush()); want.append(b)
    outs, st = inflate_fn(streams, [len(w) for w in want], 1)
    assert [int(v) for v in st] == [0] * len(streams), list(st)
    assert outs == want
    for s_, w in list(zip(streams, want))[:6]:          # the 16-wave kernel (launches of a few streams)
        o1, s1 = inflate_fn([s_], [len(w)], 1)
        assert int(s1[0]) == 0 and o1[0] == w
    good, d = streams[11], want[11]                     # the skewed alphabet, large blocks, literals only
    bad, caps = [], []
    for at in (3, 200, len(good) // 3, len(good) // 2, len(good) - 9, len(good) - 2):
        for mask in (0x01, 0x40):
            b = bytearray(good); b[at] ^= mask
            bad.append(bytes(b)); caps.append(len(d))
    for cut in (5, 3000, 4200, 9000, len(good) - 4, len(good) - 1):
        bad.append(good[:cut]); caps.append(len(d))
    for cap in (len(d) - 1, len(d) - 2, len(d) - 3, len(d) // 2, 4097, 1):
        bad.append(good); caps.append(cap)
    outs, st = inflate_fn(bad, caps, 1)
    for i, (s_, c) in enumerate(zip(bad, caps)):
        rc, w, msg = _want(o, s_, c, 1)
        assert int(st[i]) == rc, (i, int(st[i]), rc, msg)
        if rc == 0:
            assert outs[i] == w
    return len(streams) + len(bad)


def jump_resolve_checks(e, o, big=False):
    """the resolve pass for few streams (csrc/resolve_jump.hip: pointer jumping over all output bytes) against the serial
    one-wave-per-stream pass, on the same compressed streams: runs that feed on themselves (dist < len), a 300 KB run of
    one byte (the deepest pointer chains there are), every data class, history (resumable decode with 32 KiB in front),
    and a corrupt stream (both passes must leave the same prefix and report the same code).  ZMI_INF_JUMP=1 / 0 force the
    pass (read under ZMI_TUNING)."""
    import zlib
    os.environ["ZMI_TUNING"] = "1"
    n = 400000 if big else 70000
    blobs = [b"a" * (300000 if big else 40000), b"ab" * 30000, b"abc" * 20000 + bytes(range(256)) * 40, o.gen_shard(0, n), o.gen_shard(3, n),
             o.gen_shard(5, n // 2), o.gen_shard(7, n)]
    comp = [zlib.compress(b, 6) for b in blobs]
    bad = bytearray(comp[3]); bad[len(bad) // 2] ^= 0x55
    comp.append(bytes(bad)); blobs.append(None)
    res = {}
    try:
        for mode in ("1", "0"):
            os.environ["ZMI_INF_JUMP"] = mode
            outs, st = e.inflate(comp, [len(b) if b is not None else n for b in blobs], wrap=1)
            one, st1 = e.inflate(comp[:1], [len(blobs[0])], wrap=1)
            # history: the second half of a raw stream decoded with the first half's output in front of it
            raw = zlib.compressobj(6, zlib.DEFLATED, -15)
            a = raw.compress(blobs[3][:n // 2]) + raw.flush(zlib.Z_SYNC_FLUSH)
            b2 = raw.compress(blobs[3][n // 2:]) + raw.flush()
            tail, s2, d2, used, ck = e.inflate_resume(b2, 0, blobs[3][:n // 2][-32768:], cap=n)
            res[mode] = (outs, st, one, st1, tail, s2)
    finally:
        os.environ.pop("ZMI_INF_JUMP", None)
    assert res["1"] == res["0"]
    outs, st, one, st1, tail, s2 = res["1"]
    for b, got, s_ in zip(blobs[:-1], outs, st):
        assert s_ == 0 and got == b
    assert st[-1] != 0 and one == [blobs[0]] and st1 == [0] and s2 == 0 and tail == blobs[3][n // 2:]
    return len(comp)


def split_inflate_checks(e, o, big=False):
    """zmi_inflate_split (one stream decoded as segments cut at its flush points, stitched by the pointer-jumping resolve)
    against zmi_inflate_resume on the same arguments: identical output, status, detail, in_used and resume -- for true
    markers, markers that are data (inside a stored block / proposed at random offsets), matches that reach across
    the cuts into earlier segments and into the history, a final block inside a segment, a truncated tail, a corrupt
    segment, too little room, and a segment that outgrows its decode region (a long run of one byte)."""
    import zlib
    import random
    rng = random.Random(11)
    n = 600000 if big else 90000
    piece = 65536 if big else 9000

    def flushed(data, step, level=6, finish=True, mode=zlib.Z_SYNC_FLUSH):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        out, cuts = b"", []
        for i in range(0, len(data), step):
            out += co.compress(data[i:i + step]) + co.flush(mode)
            cuts.append(len(out))
        if finish:
            out += co.flush()
        return out, cuts

    def markers(buf):
        seg, at = [0], 1
        while True:
            i = buf.find(b"\x00\x00\xff\xff", at)
            if i < 0 or i + 4 >= len(buf):
                break
            seg.append(i + 4)
            at = i + 4
        return seg

    cases = []
    text = o.gen_shard(0, n)
    mix = o.gen_shard(3, n // 2) + o.gen_shard(5, n // 2)
    # 1. the plain case: every marker true, matches reach across the cuts (sync flush keeps the window)
    comp, cuts = flushed(text, piece)
    cases.append(("sync", comp, markers(comp), b"", len(text) + 100))
    # 2. full flush points, no final block yet (need more input at the end), 3. tail cut in the middle of a block
    comp2, _ = flushed(mix, piece, finish=False, mode=zlib.Z_FULL_FLUSH)
    cases.append(("full-nofinish", comp2, markers(comp2), b"", len(mix) + 100))
    cases.append(("truncated", comp[:len(comp) - 777], markers(comp[:len(comp) - 777]), b"", len(text) + 100))
    # 4. markers that are data: a stored (level 0) part that holds the four bytes many times, between compressed parts
    fake = (b"\x00\x00\xff\xff" + bytes(range(97, 123)) * 40) * 30
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    a = co.compress(text[:piece * 2]) + co.flush(zlib.Z_SYNC_FLUSH)
    st0 = zlib.compressobj(0, zlib.DEFLATED, -15)
    stored = st0.compress(fake) + st0.flush(zlib.Z_SYNC_FLUSH)
    co2 = zlib.compressobj(6, zlib.DEFLATED, -15)
    b = co2.compress(mix[:piece * 3]) + co2.flush()
    comp4 = a + stored + b
    cases.append(("stored-fakes", comp4, markers(comp4), b"", piece * 5 + len(fake) + 100))
    # 5. proposals at random offsets (none of them a restart) + the true ones
    seg5 = sorted(set(markers(comp) + [rng.randrange(1, len(comp)) for _ in range(12)]))
    cases.append(("random-cuts", comp, seg5, b"", len(text) + 100))
    cases.append(("only-false", comp, [0] + sorted(set(rng.randrange(1, len(comp)) for _ in range(6))), b"", len(text) + 100))
    # 6. history in front: the second half of a stream, its matches reach into the first half's last 32 KiB
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    head = co.compress(text[:n // 3]) + co.flush(zlib.Z_SYNC_FLUSH)
    rest = b""
    for i in range(n // 3, n, piece):
        rest += co.compress(text[i:i + piece]) + co.flush(zlib.Z_SYNC_FLUSH)
    rest += co.flush()
    cases.append(("history", rest, markers(rest), text[:n // 3][-32768:], n))
    cases.append(("history-missing", rest, markers(rest), text[:n // 3][-2000:], n))   # distance too far back somewhere
    # 7. too little room (stops in the middle), 8. a run that outgrows its region, 9. a corrupt segment
    cases.append(("small-room", comp, markers(comp), b"", len(text) // 2 + 123))
    run = b"\x07" * (n * 2) + text[:piece]
    comp8, _ = flushed(run + text[:piece * 2], max(piece, len(run) // 2 + 10))
    cases.append(("long-run", comp8, markers(comp8), b"", len(run) + piece * 3 + 100))
    bad = bytearray(comp); bad[cuts[2] + 40] ^= 0x41
    cases.append(("corrupt", bytes(bad), markers(bytes(bad)), b"", len(text) + 100))
    # 10. a final block inside a segment, bytes (with a marker) behind the end of the stream
    comp10 = comp + b"\x00\x00\xff\xff" + b"trailing bytes" * 10
    cases.append(("behind-the-end", comp10, markers(comp10), b"", len(text) + 100))
    used_any = 0
    for name, stream, seg, hist, cap in cases:
        want = e.inflate_resume(stream, 0, hist, cap=cap)
        got = e.inflate_split(stream, seg, 0, hist, cap=cap)
        assert got[:5] == want, (name, want[1:], got[1:], len(want[0]), len(got[0]))
        used_any += got[5]
        if name == "sync":
            assert got[5] == len(seg) and got[1] == 0 and got[0] == text
        if name in ("only-false",):
            assert got[5] <= 1
    assert used_any > 0
    return len(cases)


def blocks_inflate_checks(e, o, big=False):
    """zmi_inflate_blocks (a stream WITHOUT flush points: the device scans every bit position for dynamic block headers, the
    stretches between the boundaries found are decoded side by side and stitched) against zmi_inflate_resume on the same
    arguments: identical output, status, detail, in_used and resume -- for streams of the system zlib at three levels and of the
    oracle (blocks of 16 383 symbols, cuts at any bit), stored and fixed blocks between dynamic ones, history in front and
    history missing, a truncated tail, a corrupt block, too little room, bytes behind the end, a start inside a byte, a stream
    made of fixed blocks only (nothing to find: the serial path) and decoy headers planted inside a stored block."""
    import os
    import zlib
    import random
    rng = random.Random(5)
    n = 1500000 if big else 200000

    def raw(data, level=6, strategy=0):
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        return co.compress(data) + co.flush()

    text = o.gen_shard(0, n)
    mix = o.gen_shard(3, n // 3) + o.gen_shard(4, n // 3) + o.gen_shard(6, n // 3)
    cases = []
    for lvl in (1, 6, 9):
        cases.append(("zlib-L%d" % lvl, raw(mix, lvl), b"", len(mix) + 100, 0, True))
    otext = text if big else text + o.gen_shard(1, n) + o.gen_shard(2, n)   # (blocks of 16 383 symbols: enough of them to cut)
    rc, oc = o.deflate(otext, 6, 0)
    cases.append(("oracle", oc, b"", len(otext) + 100, 0, True))
    comp = raw(text, 6)
    # stored and fixed blocks between dynamic ones (Z_FULL_FLUSH / Z_BLOCK keep the pieces one stream)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    part = co.compress(text[:n // 3]) + co.flush(zlib.Z_BLOCK if hasattr(zlib, "Z_BLOCK") else zlib.Z_SYNC_FLUSH)
    st0 = zlib.compressobj(0, zlib.DEFLATED, -15)
    # ... the stored part holds bytes that LOOK like block headers: real headers of another stream, at every bit offset
    decoy = b"".join(bytes([rng.randrange(256)]) + raw(o.gen_shard(1, 20000), 6)[:400] for _ in range(24))
    co2 = zlib.compressobj(6, zlib.DEFLATED, -15)
    mid = co.compress(b"")   # (nothing: the pieces below are separate streams spliced at sync points)
    # (the dynamic part in front must hold several blocks, so that cuts ARE taken before the decoys end the parallel part and the
    # rest is decoded serially from a cut that lies inside a byte: round 4 shipped that hand-over with the bit offset dropped)
    dtext = text if big else text + o.gen_shard(1, n) + o.gen_shard(2, n)
    a = zlib.compressobj(6, zlib.DEFLATED, -15)
    pa = a.compress(dtext) + a.flush(zlib.Z_SYNC_FLUSH)
    pb = st0.compress(decoy) + st0.flush(zlib.Z_SYNC_FLUSH)
    fx = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    pc = fx.compress(mix[:n // 6]) + fx.flush(zlib.Z_SYNC_FLUSH)
    pd = co2.compress(mix[n // 6:]) + co2.flush()
    spliced = pa + pb + pc + pd
    cases.append(("stored-fixed-decoys", spliced, b"", len(dtext) + len(decoy) + len(mix) + 100, 0, True))   # (the decoys are found, and refused)
    # history in front (the second part of a stream) and history missing
    hs = zlib.compressobj(6, zlib.DEFLATED, -15)
    head = hs.compress(text[:n // 4]) + hs.flush(zlib.Z_FULL_FLUSH if False else zlib.Z_SYNC_FLUSH)
    rest = hs.compress(text[n // 4:]) + hs.flush()
    cases.append(("history", rest, text[:n // 4][-32768:], n, 0, True))
    cases.append(("history-missing", rest, text[:n // 4][-1500:], n, 0, False))
    cases.append(("truncated", comp[:len(comp) - 1234], b"", len(text) + 100, 0, True))
    bad = bytearray(comp); bad[len(bad) // 2] ^= 0x10
    cases.append(("corrupt", bytes(bad), b"", len(text) + 100, 0, False))
    cases.append(("small-room", comp, b"", len(text) // 2 + 77, 0, True))
    cases.append(("behind-the-end", comp + b"trailing" * 50, b"", len(text) + 100, 0, True))
    # a start inside a byte: three filler bits in front (an empty fixed block is 10 bits: use a stored-free shift instead)
    shifted = bytearray(len(comp) + 1)
    carry = 0
    for i, b in enumerate(comp):
        v = (b << 3) | carry
        shifted[i] = v & 0xFF
        carry = v >> 8
    shifted[len(comp)] = carry
    cases.append(("start-bit-3", bytes(shifted), b"", len(text) + 100, 3, True))
    cases.append(("fixed-only", raw(text[:n // 2], 6, zlib.Z_FIXED), b"", n // 2 + 100, 0, None))
    cases.append(("random-bytes", os.urandom(1) + bytes(rng.randrange(256) for _ in range(70000)), b"", 100000, 0, None))
    used_any = 0
    for name, stream, hist, cap, bit, expect_parallel in cases:
        want = e.inflate_resume(stream, bit, hist, cap=cap)
        got = e.inflate_blocks(stream, bit, hist, cap=cap)
        assert got[:5] == want, (name, want[1:], got[1:], len(want[0]), len(got[0]))
        used_any += got[5]
        if expect_parallel and (big or name.startswith("zlib-L") or name == "oracle"):   # (the small streams of the emulator run hold few blocks)
            assert got[5] >= 3, (name, got[5])
        if name == "fixed-only":
            assert got[5] == 0
        if name.startswith("zlib-L") or name == "oracle":
            assert got[1] == 0 and got[0] == (otext if name == "oracle" else mix)
    assert used_any > 0
    return len(cases)


def truncated_stored_checks(inflate_fn, o):
    """a stored block that is cut off, or does not fit its room, through the batch kernel: the bytes and the code of the oracle
    (Mode::CopyBlock, inflate.rs:1374-1394: min(length, room, input) bytes are copied)"""
    import zlib
    raw = o.gen_shard(5, 150000)
    co = zlib.compressobj(0, zlib.DEFLATED, -15)
    stream = co.compress(raw) + co.flush()
    cases = [(stream[:cut], len(raw)) for cut in (len(stream) - 1000, 70000, 65540, 10, 5)] + [(stream, 70000), (stream, 65535), (stream, len(raw))]
    outs, st = inflate_fn([c for c, _ in cases], [cap for _, cap in cases], 0)
    for (c, cap), got, s_ in zip(cases, outs, st):
        rc, want, used, msg = o.inflate(c, cap, wrap=0)
        rc = 0 if rc == 1 else rc   # (the oracle says Z_STREAM_END, the batch status 0 for a complete stream)
        assert int(s_) == rc and bytes(got[:len(want)]) == want and len(got) >= len(want), (len(c), cap, int(s_), rc, len(got), len(want))
    return len(cases)
