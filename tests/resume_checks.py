"""Resumable raw-deflate decode (zmi_inflate_resume, include/zmi355.h) driven the way a streaming inflate drives it:
the stream arrives in pieces, every call restarts at the checkpoint of the previous one with the output in front of it
as history.  Shared by the emulator test and the MI355X test."""
import random
import zlib


def raw_stream(data, level=6, flushes=()):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    out, last = b"", 0
    for f in flushes:
        out += c.compress(data[last:f]) + c.flush(zlib.Z_SYNC_FLUSH)
        last = f
    return out + c.compress(data[last:]) + c.flush()


def run_chain(eng, stream, cuts, cap=1 << 16):
    """feed stream[:cut] for every cut, then all of it -> (output, bytes consumed, checks made)"""
    got, pend = b"", b""          # settled output / bytes decoded behind the checkpoint
    base, bit = 0, 0              # the checkpoint: absolute byte, bit
    for end in list(cuts) + [len(stream)]:
        while True:
            out, st, det, used, res = eng.inflate_resume(stream[base:end], bit, got[-32768:], cap)
            assert out[:len(pend)] == pend, "a restart reproduces what was decoded behind the checkpoint"
            if st == 0:
                assert res[3] == 1 and res[2] == len(out)
                return got + out, base + used
            assert st == -5 and det in (1, 2), (st, det)
            got += out[:res[2]]
            pend = out[res[2]:]
            base += res[0]
            bit = res[1]
            if det == 1:
                break                     # wants more input: next cut
            if res[2] == 0 and res[0] == 0:
                cap *= 2                  # one block larger than the room
    raise AssertionError("the stream did not end")


def resume_chain_checks(eng, o, sizes=(150000, 100000, 70000, 60000), trials=3, seed=5):
    rnd = random.Random(seed)
    data = o.gen_shard(0, sizes[0]) + o.gen_shard(4, sizes[1]) + bytes(rnd.randrange(256) for _ in range(sizes[2])) + o.gen_shard(3, sizes[3])
    for level, flushes in ((6, ()), (1, ()), (9, (1000, 50000, 50001, len(data) // 2)), (0, ())):
        s = raw_stream(data, level, flushes)
        for _ in range(trials):
            cuts = sorted(rnd.randrange(0, len(s)) for _ in range(rnd.randrange(1, 10)))
            out, used = run_chain(eng, s, cuts)
            assert out == data and used == len(s), (level, cuts, len(out), used, len(s))
    # a small stream in steps of a few bytes, and one with garbage behind its end
    small = o.gen_shard(3, 3000)
    s = raw_stream(small, 6, (100, 1500))
    out, used = run_chain(eng, s + b"\xff" * 9, list(range(1, len(s), 7)))
    assert out == small and used == len(s)
    # a start in the middle of a byte: three foreign bits in front (what inflatePrime / a mid-stream restart needs)
    # (a stream without stored blocks: their byte alignment would move with the shift)
    plain = raw_stream(small, 6)
    shifted = int.from_bytes(plain, "little") << 3 | 0b101
    sh = shifted.to_bytes(len(plain) + 1, "little")
    out, st, det, used, res = eng.inflate_resume(sh, 3, b"", 1 << 16)
    assert st == 0 and out == small and res[3] == 1
    # damaged data: the valid bytes in front of the damage still come out
    bad = bytearray(raw_stream(data[:60000], 6))
    bad[len(bad) // 2] ^= 0x10
    out, st, det, used, res = eng.inflate_resume(bytes(bad), 0, b"", 1 << 17)
    assert data[:60000].startswith(out[:1000]) and (st != 0 or out != data[:60000])   # (a flipped bit may still parse)
    # a history shorter than a distance the stream uses: "invalid distance too far back"
    tail = raw_stream(small + small, 6, (len(small),))
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    first = co.compress(small) + co.flush(zlib.Z_SYNC_FLUSH)
    second = co.compress(small) + co.flush()
    out, st, det, used, res = eng.inflate_resume(second, 0, small, 1 << 16)
    assert st == 0 and out == small
    out, st, det, used, res = eng.inflate_resume(second, 0, b"", 1 << 16)
    assert st == -3
    del tail, first
