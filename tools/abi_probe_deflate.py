"""Differential probe of the zlib stream ABI (deflate side) against the system's libz.so.1: the same call sequences on both
libraries, return codes / counters / outputs compared line by line.  zlib-rs follows zlib's observable behaviour here
except where noted in the output (deflatePrime accepts up to 32 bits: deflate.rs:566-579).
usage: python tools/abi_probe_deflate.py [--gpu]   (default: the CPU emulator build of the same sources; test infrastructure)"""
import os, sys, ctypes as C, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import zmi_ctypes, oracle_lib, zlib_abi_harness as H
from zlib_abi_harness import *
if "--gpu" in sys.argv:
    from zlib_rs_amd import _build
    lib = H.bind(C.CDLL(_build.ABI_LIB))
else:
    zmi_ctypes.load_emu(False)
    lib = H.bind(C.CDLL(os.path.join(zmi_ctypes.ROOT, "tests", "emu", "libzmi355_emu.so")))
sysz = H.bind(C.CDLL("libz.so.1"))
P = C.POINTER(ZStream)
for L in (lib, sysz):
    L.deflateParams.argtypes = [P, C.c_int, C.c_int]; L.deflatePrime.argtypes = [P, C.c_int, C.c_int]
    L.deflatePending.argtypes = [P, C.POINTER(C.c_uint), C.POINTER(C.c_int)]
    L.deflateCopy.argtypes = [P, P]; L.deflateSetHeader.argtypes = [P, C.POINTER(GzHeader)]
    L.deflateGetDictionary.argtypes = [P, C.c_char_p, C.POINTER(C.c_uint)]
data = oracle_lib.load(False).gen_shard(0, 50000)
def run(L):
    ver, zs = L.zlibVersion(), C.sizeof(ZStream); res = []
    src = C.create_string_buffer(data, len(data)); out = C.create_string_buffer(100000)
    def new(level=6, wb=15, strat=0):
        s = ZStream(); assert L.deflateInit2_(C.byref(s), level, 8, wb, 8, strat, ver, zs) == 0; return s
    # avail_out == 0
    s = new(); s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 100, C.addressof(out), 0
    res.append(("avail_out0", L.deflate(C.byref(s), 0), s.avail_in)); 
    res.append(("end_busy", L.deflateEnd(C.byref(s))))
    # invalid flush values
    s = new(); s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 100, C.addressof(out), 1000
    res.append(("flush6", L.deflate(C.byref(s), 6), L.deflate(C.byref(s), -1))); L.deflateEnd(C.byref(s))
    # no progress twice
    s = new(); s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 0, C.addressof(out), 1000
    res.append(("noprogress", L.deflate(C.byref(s), 0), L.deflate(C.byref(s), 0))); L.deflateEnd(C.byref(s))
    # next_in NULL with avail_in != 0 ; next_out NULL
    s = new(); s.next_in, s.avail_in, s.next_out, s.avail_out = None, 10, C.addressof(out), 1000
    res.append(("null_in", L.deflate(C.byref(s), 0))); L.deflateEnd(C.byref(s))
    s = new(); s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 10, None, 1000
    res.append(("null_out", L.deflate(C.byref(s), 0))); L.deflateEnd(C.byref(s))
    # after finish: more calls
    s = new(); s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 1000, C.addressof(out), 100000
    r1 = L.deflate(C.byref(s), 4); r2 = L.deflate(C.byref(s), 4); s.avail_in = 5; r3 = L.deflate(C.byref(s), 0); r4 = L.deflate(C.byref(s), 4)
    res.append(("after_finish", r1, r2, r3, r4)); res.append(("end_done", L.deflateEnd(C.byref(s))))
    # finish with small output: Z_OK until drained
    s = new(); s.next_in, s.avail_in = C.addressof(src), 50000
    codes = []; total = bytearray()
    for i in range(100000):
        s.next_out, s.avail_out = C.addressof(out), 4000
        r = L.deflate(C.byref(s), 4); total += out.raw[:4000 - s.avail_out]; codes.append(r)
        if r != 0: break
    res.append(("finish_small", sorted(set(codes[:-1])), codes[-1], zlib.decompress(bytes(total)) == data)); L.deflateEnd(C.byref(s))
    # flush kinds: each must make all input so far decodable (except Z_BLOCK which may leave bits)
    for fl in (1, 2, 3, 5):
        s = new(wb=-15); s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 20000, C.addressof(out), 100000
        r = L.deflate(C.byref(s), fl); n1 = 100000 - s.avail_out
        d = zlib.decompressobj(-15); got = d.decompress(out.raw[:n1])
        r_again = L.deflate(C.byref(s), fl)     # same flush, no new input
        tail4 = out.raw[n1 - 4:n1] == b"\0\0\xff\xff"
        res.append(("flush", fl, r, s.avail_in, len(got) == 20000 if fl != 5 else len(got) > 19000, tail4 if fl in (2, 3) else None, r_again)); L.deflateEnd(C.byref(s))
    # set dictionary after start, on gzip
    s = new(); s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 100, C.addressof(out), 1000
    L.deflate(C.byref(s), 0); res.append(("dict_late", L.deflateSetDictionary(C.byref(s), b"abc", 3))); L.deflateEnd(C.byref(s))
    s = new(wb=-15); s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 100, C.addressof(out), 1000
    L.deflate(C.byref(s), 2); res.append(("dict_raw_mid", L.deflateSetDictionary(C.byref(s), b"abc", 3))); L.deflateEnd(C.byref(s))
    # deflatePrime ranges
    s = new(wb=-15)
    res.append(("prime", L.deflatePrime(C.byref(s), 5, 31), L.deflatePrime(C.byref(s), 17, 0), L.deflatePrime(C.byref(s), -1, 0), L.deflatePrime(C.byref(s), 16, 0xABCD)))
    L.deflateEnd(C.byref(s))
    # set header on non-gzip
    s = new(); h = GzHeader(); res.append(("sethdr_zlib", L.deflateSetHeader(C.byref(s), C.byref(h)))); L.deflateEnd(C.byref(s))
    # deflateParams right after init, and with pending output in a too-small buffer
    s = new(level=1); res.append(("params_fresh", L.deflateParams(C.byref(s), 9, 1)))
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 50000, C.addressof(out), 100000
    L.deflate(C.byref(s), 0); s.avail_out = 0 if False else s.avail_out
    res.append(("params_mid", L.deflateParams(C.byref(s), 0, 0), s.avail_in)); r = L.deflate(C.byref(s), 4)
    res.append(("params_fin", r, zlib.decompress(out.raw[:100000 - s.avail_out]) == data)); L.deflateEnd(C.byref(s))
    # deflateCopy mid-stream: both finish to the same data
    s = new(); s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 25000, C.addressof(out), 100000
    L.deflate(C.byref(s), 0); n1 = 100000 - s.avail_out
    c = ZStream(); rc = L.deflateCopy(C.byref(c), C.byref(s)); o2 = C.create_string_buffer(100000)
    s.avail_in = 25000; r1 = L.deflate(C.byref(s), 4); a = out.raw[:100000 - s.avail_out]
    c.next_in, c.avail_in, c.next_out, c.avail_out = C.addressof(src) + 25000, 25000, C.addressof(o2), 100000
    r2 = L.deflate(C.byref(c), 4); b = out.raw[:n1] + o2.raw[:100000 - c.avail_out]
    res.append(("copy", rc, r1, r2, zlib.decompress(a) == data, zlib.decompress(b) == data)); L.deflateEnd(C.byref(s)); L.deflateEnd(C.byref(c))
    # level / strategy / wbits validation
    for args in ((6, 8, 15, 10, 0), (6, 8, 15, 0, 0), (6, 9, 15, 8, 0), (-2, 8, 15, 8, 0), (-1, 8, 15, 8, 0), (6, 8, 47, 8, 0), (6, 8, 8, 8, 0), (6, 8, -8, 8, 0), (6, 8, 24, 8, 0)):
        s = ZStream(); r = L.deflateInit2_(C.byref(s), args[0], args[1], args[2], args[3], args[4], ver, zs); res.append(("init", args, r))
        if r == 0: L.deflateEnd(C.byref(s))
    return res
a = run(lib); b = run(sysz)
for x, y in zip(a, b):
    print("OK  " if x == y else "DIFF", x, "" if x == y else y)
