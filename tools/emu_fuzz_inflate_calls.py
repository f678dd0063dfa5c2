"""Random call sequences on one inflate stream of the stream ABI (CPU emulator build): pieces and rooms of random size with
random flush arguments, inflateCopy (the copy carries on, the original is ended), inflateGetDictionary (= the last 32 KiB
handed out), inflateMark / inflateSyncPoint queries, inflateReset + replay in between; streams come from the system's zlib
(random level / wrapper / flush points / preset dictionary).
usage: python tools/emu_fuzz_inflate_calls.py SEED SECONDS   (test infrastructure; the product path needs an MI355X)"""
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
P = C.POINTER(H.ZStream)
lib.inflateCopy.argtypes = [P, P]
lib.inflateReset.argtypes = [P]
lib.inflateGetDictionary.argtypes = [P, C.c_char_p, C.POINTER(C.c_uint)]
lib.inflateMark.argtypes, lib.inflateMark.restype = [P], C.c_long
lib.inflateSyncPoint.argtypes = [P]
o = oracle_lib.load()
rnd = random.Random(int(sys.argv[1]))
ver, zs = lib.zlibVersion(), C.sizeof(H.ZStream)
t0, rounds, calls = time.time(), 0, 0
while time.time() - t0 < float(sys.argv[2]):
    wbits = rnd.choice([15, 31, -15, 47])
    n = rnd.choice([0, 1, 3000, rnd.randrange(80000), rnd.randrange(200000)])
    data = o.gen_shard(rnd.randrange(8), n)
    zdict = o.gen_shard(rnd.randrange(8), rnd.choice([20, 4000, 40000])) if wbits in (15, -15) and rnd.random() < 0.25 else None
    cw = wbits if wbits != 47 else rnd.choice([15, 31])
    co = zlib.compressobj(rnd.choice([0, 1, 6, 9]), zlib.DEFLATED, cw, 8, 0, zdict) if zdict else zlib.compressobj(rnd.choice([0, 1, 6, 9]), zlib.DEFLATED, cw)
    comp, at = b"", 0
    while at < n:
        k = rnd.randrange(1, n + 1)
        comp += co.compress(data[at:at + k])
        if rnd.random() < 0.3:
            comp += co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
        at += k
    comp += co.flush()
    strm = H.ZStream()
    assert lib.inflateInit2_(C.byref(strm), wbits, ver, zs) == H.Z_OK
    if zdict and wbits < 0:
        assert lib.inflateSetDictionary(C.byref(strm), zdict, len(zdict)) == H.Z_OK
    src = C.create_string_buffer(comp, len(comp) or 1)
    out, pos, rc, spins, replayed = bytearray(), 0, H.Z_OK, 0, False
    while rc != H.Z_STREAM_END:
        spins += 1
        assert spins < 400000, "no end in sight"
        op = rnd.random()
        if op < 0.05:
            twin = H.ZStream()
            assert lib.inflateCopy(C.byref(twin), C.byref(strm)) == H.Z_OK
            assert lib.inflateEnd(C.byref(strm)) == H.Z_OK
            strm = twin
            continue
        if op < 0.09 and not (zdict and wbits > 0):
            buf, ln = C.create_string_buffer(32768), C.c_uint(0)
            assert lib.inflateGetDictionary(C.byref(strm), buf, C.byref(ln)) == H.Z_OK
            hist = ((zdict or b"") + bytes(out))[-32768:]
            assert buf.raw[:ln.value] == hist[-ln.value:] if ln.value else True, (rounds, ln.value, len(out))
            assert ln.value == min(32768, len((zdict or b"") + bytes(out))) or ln.value <= len(hist)
            assert lib.inflateMark(C.byref(strm)) is not None and lib.inflateSyncPoint(C.byref(strm)) in (0, 1)
            continue
        if op < 0.10 and not replayed and not zdict:
            assert lib.inflateReset(C.byref(strm)) == H.Z_OK and strm.total_in == 0 and strm.total_out == 0
            out, pos, replayed = bytearray(), 0, True
            continue
        piece = min(len(comp) - pos, rnd.choice([1, 2, 9, 100, 5000, 1 << 20]))
        room = rnd.choice([1, 5, 300, 8192, 300000])
        obuf = C.create_string_buffer(room)
        strm.next_in, strm.avail_in = C.addressof(src) + pos, piece
        strm.next_out, strm.avail_out = C.addressof(obuf), room
        rc = lib.inflate(C.byref(strm), rnd.choice([0, 0, 2, 5]))
        calls += 1
        if rc == 2:   # Z_NEED_DICT
            assert zdict and strm.adler == zlib.adler32(zdict)
            pos += piece - strm.avail_in
            assert lib.inflateSetDictionary(C.byref(strm), zdict, len(zdict)) == H.Z_OK
            rc = H.Z_OK
            continue
        assert rc in (H.Z_OK, H.Z_STREAM_END, H.Z_BUF_ERROR), (rounds, rc, strm.msg)
        pos += piece - strm.avail_in
        out += obuf.raw[:room - strm.avail_out]
        if rc == H.Z_BUF_ERROR:
            assert piece == 0 and room == strm.avail_out
            assert pos < len(comp) or len(out) < n, "stuck with everything delivered"
    assert bytes(out) == data and strm.total_out == n and pos == len(comp), (rounds, len(out), n, pos, len(comp))
    assert lib.inflateEnd(C.byref(strm)) == H.Z_OK
    rounds += 1
print("emu inflate call fuzz ok: %d streams, %d inflate() calls, seed %s" % (rounds, calls, sys.argv[1]))
