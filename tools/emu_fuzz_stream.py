"""Mutation fuzz of the stream ABI's inflate() on the CPU emulator build, fed in 64-byte pieces like the reference's
fuzz/fuzz_targets/uncompress.rs; the system's zlib, fed the same pieces, is the judge: accepted streams give the same
bytes, rejected streams are rejected, and whatever was handed out before an error agrees byte for byte.
usage: python tools/emu_fuzz_stream.py SEED SECONDS   (test infrastructure; the product path needs an MI355X)"""
import ctypes as C
import os
import random
import sys
import time
import zlib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib
import zlib_abi_harness as H
import zmi_ctypes

zmi_ctypes.load_emu(False)
lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
o = oracle_lib.load()
rnd = random.Random(int(sys.argv[1]))
ver, zs = lib.zlibVersion(), C.sizeof(H.ZStream)
t0, rounds, accepted, rejected = time.time(), 0, 0, 0
while time.time() - t0 < float(sys.argv[2]):
    wbits = rnd.choice([15, 31, -15, 47])
    n = rnd.choice([0, 5, 300, 5000, rnd.randrange(60000)])
    d = o.gen_shard(rnd.randrange(8), n) if rnd.random() < 0.8 else bytes(rnd.randrange(3) for _ in range(n))
    co = zlib.compressobj(rnd.choice([0, 1, 6, 9]), zlib.DEFLATED, wbits if wbits != 47 else rnd.choice([15, 31]), 8, rnd.choice([0, 0, 2, 3, 4]))
    c = bytearray(co.compress(d[:n // 2]) + (co.flush(zlib.Z_SYNC_FLUSH) if rnd.random() < 0.5 else b"") + co.compress(d[n // 2:]) + co.flush())
    kind = rnd.random()
    if kind < 0.45 and c:
        for _ in range(rnd.choice([1, 1, 2, 8])):
            c[rnd.randrange(len(c))] ^= 1 << rnd.randrange(8)
    elif kind < 0.6 and c:
        del c[rnd.randrange(len(c)):]
    elif kind < 0.7:
        c = bytearray(rnd.getrandbits(8) for _ in range(rnd.randrange(1, 400)))
    c = bytes(c)
    # the judge
    dec, want, ok = zlib.decompressobj(wbits), bytearray(), True
    try:
        for at in range(0, len(c), 64):
            want += dec.decompress(c[at:at + 64])
            if dec.eof:
                break
        ok = dec.eof
    except zlib.error:
        ok = False
    # this library
    strm = H.ZStream()
    assert lib.inflateInit2_(C.byref(strm), wbits, ver, zs) == H.Z_OK
    src = C.create_string_buffer(c, len(c) or 1)
    room = rnd.choice([100, 4096, 100000])
    obuf = C.create_string_buffer(room)
    got, rc, at, spins = bytearray(), H.Z_OK, 0, 0
    while rc == H.Z_OK or rc == H.Z_BUF_ERROR:
        piece = min(64, len(c) - at)
        strm.next_in, strm.avail_in = C.addressof(src) + at, piece
        strm.next_out, strm.avail_out = C.addressof(obuf), room
        rc = lib.inflate(C.byref(strm), H.Z_NO_FLUSH)
        at += piece - strm.avail_in
        got += obuf.raw[:room - strm.avail_out]
        if piece == 0 and room == strm.avail_out:
            break                                   # input exhausted, nothing more comes out
        spins += 1
        assert spins < 200000, "no end in sight"
    assert lib.inflateEnd(C.byref(strm)) == H.Z_OK
    if ok:
        assert rc == H.Z_STREAM_END and bytes(got) == bytes(want), (rounds, rc, len(got), len(want))
        accepted += 1
    else:
        assert rc != H.Z_STREAM_END, (rounds, "zlib rejects, inflate() accepts", c[:40].hex())
        k = min(len(got), len(want))
        assert got[:k] == want[:k], (rounds, "bytes handed out before the error differ")
        rejected += 1
    rounds += 1
print("emu stream fuzz ok: %d rounds, %d accepted, %d rejected, seed %s" % (rounds, accepted, rejected, sys.argv[1]))
