"""Random call sequences on one deflate stream of the stream ABI (CPU emulator build): input pieces with every flush value,
output rooms from 1 byte up, deflateParams / deflatePending / deflateCopy / deflateTune in between, sometimes a preset
dictionary; the system's zlib must read the result back and the counters must add up.
usage: python tools/emu_fuzz_deflate_calls.py SEED SECONDS   (test infrastructure; the product path needs an MI355X)"""
import ctypes as C
import os
import random
import sys
import time
import zlib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.setdefault("ZMI_TUNING", "1")
os.environ.setdefault("ZMI_ABI_SEGMENT", "8192")
import oracle_lib
import zlib_abi_harness as H
import zmi_ctypes

zmi_ctypes.load_emu(False)
lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
P = C.POINTER(H.ZStream)
lib.deflateParams.argtypes = [P, C.c_int, C.c_int]
lib.deflateCopy.argtypes = [P, P]
lib.deflatePending.argtypes = [P, C.POINTER(C.c_uint), C.POINTER(C.c_int)]
lib.deflateTune.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int]
o = oracle_lib.load()
rnd = random.Random(int(sys.argv[1]))
ver, zs = lib.zlibVersion(), C.sizeof(H.ZStream)
t0, rounds, calls = time.time(), 0, 0
while time.time() - t0 < float(sys.argv[2]):
    w = rnd.choice([15, 15, 15, 12, 9])
    wbits = rnd.choice([w, w + 16, -w])
    n = rnd.choice([0, 1, 3000, rnd.randrange(60000), rnd.randrange(120000)])
    data = o.gen_shard(rnd.randrange(8), n) if rnd.random() < 0.85 else bytes(rnd.randrange(2) for _ in range(n))
    strm = H.ZStream()
    cfg = (rnd.randrange(10), wbits, rnd.randrange(1, 10), rnd.randrange(5))
    assert lib.deflateInit2_(C.byref(strm), cfg[0], 8, wbits, cfg[2], cfg[3], ver, zs) == H.Z_OK
    zdict = None
    if wbits < 16 and rnd.random() < 0.2:
        zdict = o.gen_shard(rnd.randrange(8), rnd.choice([10, 500, 40000]))
        assert lib.deflateSetDictionary(C.byref(strm), zdict, len(zdict)) == H.Z_OK
    src = C.create_string_buffer(data, n or 1)
    out, pos, log = bytearray(), 0, []

    def call(flush, take, room):
        global calls
        obuf = C.create_string_buffer(room)
        strm.next_in, strm.avail_in = C.addressof(src) + pos, take
        strm.next_out, strm.avail_out = C.addressof(obuf), room
        rc = lib.deflate(C.byref(strm), flush)
        calls += 1
        out.extend(obuf.raw[:room - strm.avail_out])
        log.append((flush, take, room, rc, take - strm.avail_in, room - strm.avail_out))
        return rc, take - strm.avail_in

    while pos < n:
        op = rnd.random()
        if op < 0.08:
            # deflateParams compresses what was supplied so far with the old parameters and writes it to next_out
            room = rnd.choice([1, 300, 200000])
            obuf = C.create_string_buffer(room)
            strm.next_in, strm.avail_in = C.addressof(src) + pos, 0
            strm.next_out, strm.avail_out = C.addressof(obuf), room
            rc = lib.deflateParams(C.byref(strm), rnd.randrange(-1, 10), rnd.randrange(5))
            out.extend(obuf.raw[:room - strm.avail_out])
            log.append(("params", rc, room - strm.avail_out))
            assert rc in (H.Z_OK, H.Z_BUF_ERROR), log[-5:]
            continue
        if op < 0.12:
            pend, bits = C.c_uint(0), C.c_int(0)
            assert lib.deflatePending(C.byref(strm), C.byref(pend), C.byref(bits)) == H.Z_OK and 0 <= bits.value < 8
            assert lib.deflateTune(C.byref(strm), 8, 16, 64, 128) == H.Z_OK
            continue
        if op < 0.16:
            twin = H.ZStream()
            assert lib.deflateCopy(C.byref(twin), C.byref(strm)) == H.Z_OK
            rc_end = lib.deflateEnd(C.byref(strm))
            assert rc_end in (H.Z_OK, H.Z_DATA_ERROR)
            strm = twin
            log.append(("copy",))
            continue
        take = min(n - pos, rnd.choice([1, 10, 1000, 8192, 8193, 50000]))
        flush = rnd.choice([0, 0, 0, 1, 2, 3, 5])
        rc, used = call(flush, take, rnd.choice([1, 7, 300, 5000, 200000]))
        assert rc in (H.Z_OK, H.Z_BUF_ERROR), (rc, log[-5:])
        pos += used
        if flush in (1, 2, 3) and used == take:
            # a flush is complete once a call leaves room: then everything fed so far must be readable
            spins = 0
            while strm.avail_out == 0:
                rc, _ = call(flush, 0, rnd.choice([1, 100, 100000]))
                spins += 1
                assert rc in (H.Z_OK, H.Z_BUF_ERROR) and spins < 300000, log[-5:]
            if flush != 1 or True:
                d = zlib.decompressobj(wbits, zdict=zdict) if zdict and wbits > 0 else zlib.decompressobj(wbits)
                if zdict and wbits < 0:
                    d = zlib.decompressobj(wbits, zdict=zdict)
                try:
                    got = d.decompress(bytes(out))
                except zlib.error as e:
                    raise AssertionError((rounds, wbits, n, str(e), "dict" if zdict else "", log[-12:]))
                assert got == data[:pos], (rounds, cfg, "dict" if zdict else "", "flush", flush, len(got), pos, log[-6:])
    spins = 0
    while True:
        rc, _ = call(4, 0, rnd.choice([1, 50, 4096, 300000]))
        spins += 1
        assert spins < 600000, "finish does not end"
        if rc == H.Z_STREAM_END:
            break
        assert rc == H.Z_OK, (rc, log[-5:])
    assert strm.total_in == n and strm.total_out == len(out) or any(e[0] == "copy" for e in log), (strm.total_in, n, strm.total_out, len(out))
    assert lib.deflateEnd(C.byref(strm)) == H.Z_OK
    d = zlib.decompressobj(wbits, zdict=zdict) if zdict else zlib.decompressobj(wbits)
    assert d.decompress(bytes(out)) == data and d.eof and not d.unused_data, (rounds, wbits, n, log[-8:])
    rounds += 1
print("emu deflate call fuzz ok: %d streams, %d deflate() calls, seed %s" % (rounds, calls, sys.argv[1]))
