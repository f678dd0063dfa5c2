"""Differential probe of the zlib stream ABI (inflate side) against the system's libz.so.1: the same call sequences on both
libraries, return codes / counters / outputs compared line by line.  zlib-rs follows zlib's observable behaviour here
except where noted in the output (deflatePrime accepts up to 32 bits: deflate.rs:566-579).
usage: python tools/abi_probe_inflate.py [--gpu]   (default: the CPU emulator build of the same sources; test infrastructure)"""
import os, sys, ctypes as C, zlib, gzip
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
sysz = C.CDLL("libz.so.1")
for L in (lib, sysz):
    L.inflateInit2_.argtypes = [C.POINTER(ZStream), C.c_int, C.c_char_p, C.c_int]
    L.inflate.argtypes = [C.POINTER(ZStream), C.c_int]; L.inflateEnd.argtypes = [C.POINTER(ZStream)]
    L.inflateGetHeader.argtypes = [C.POINTER(ZStream), C.POINTER(GzHeader)]
    L.inflateCopy.argtypes = [C.POINTER(ZStream), C.POINTER(ZStream)]
    L.inflateReset.argtypes = [C.POINTER(ZStream)]
    L.inflateValidate.argtypes = [C.POINTER(ZStream), C.c_int]
    L.zlibVersion.restype = C.c_char_p
def run(L, name):
    ver, zs = L.zlibVersion(), C.sizeof(ZStream)
    res = []
    # done state
    gz = bytes([31, 139, 8, 0, 0, 0, 0, 0, 0, 3, 203, 72, 205, 201, 201, 87, 40, 207, 47, 202, 73, 81, 200, 0, 179, 33, 36, 68, 4, 89, 28, 137, 13, 0, 181, 147, 9, 162, 53, 0, 0, 0])
    s = ZStream(); assert L.inflateInit2_(C.byref(s), 31, ver, zs) == 0
    src = C.create_string_buffer(gz, len(gz)); out = C.create_string_buffer(64)
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(gz), C.addressof(out), 64
    r1 = L.inflate(C.byref(s), 0)
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(gz), C.addressof(out), 64
    r2 = L.inflate(C.byref(s), 4)
    res.append(("done_state", r1, r2, s.avail_in, s.total_out)); L.inflateEnd(C.byref(s))
    # windowBits 0 = use the header's
    hello = zlib.compress(b"Hello World!\n", 6)
    s = ZStream(); rc = L.inflateInit2_(C.byref(s), 0, ver, zs)
    src = C.create_string_buffer(hello, len(hello))
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(hello), C.addressof(out), 64
    res.append(("wbits0", rc, L.inflate(C.byref(s), 4), out.raw[:13])); L.inflateEnd(C.byref(s))
    # inflateGetHeader on zlib-only stream / raw stream
    for wb in (15, -15, 31, 47):
        s = ZStream(); L.inflateInit2_(C.byref(s), wb, ver, zs); h = GzHeader()
        res.append(("gethdr", wb, L.inflateGetHeader(C.byref(s), C.byref(h)))); L.inflateEnd(C.byref(s))
    # header configured but stream is zlib (auto detect)
    s = ZStream(); L.inflateInit2_(C.byref(s), 47, ver, zs); h = GzHeader()
    L.inflateGetHeader(C.byref(s), C.byref(h))
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(hello), C.addressof(out), 64
    res.append(("hdr_no_gzip", L.inflate(C.byref(s), 4), h.done)); L.inflateEnd(C.byref(s))
    # gzip header fields with insufficient space
    import io
    extra = b"Scheduling and executing async tasks is a job handled by an async runtime, such as\0"
    name = b"tokio, async-std, and smol. You've probably used them at some point, either directly or\0"
    comment = b"indirectly. They, along with many frameworks that require async, do their best to hide\0"
    body = zlib.compressobj(6, zlib.DEFLATED, -15); raw = body.compress(b"Hello World\n") + body.flush()
    hdr = bytes([31, 139, 8, 2 | 4 | 8 | 16, 0, 0, 0, 0, 0, 3]) + len(extra).to_bytes(2, "little") + extra + name + comment
    hdr += (zlib.crc32(hdr) & 0xFFFF).to_bytes(2, "little")
    g = hdr + raw + zlib.crc32(b"Hello World\n").to_bytes(4, "little") + (12).to_bytes(4, "little")
    assert gzip.decompress(g) == b"Hello World\n"
    for chunk in (16, 1, 1000):
        s = ZStream(); L.inflateInit2_(C.byref(s), 31, ver, zs)
        eb, nb, cb = C.create_string_buffer(64), C.create_string_buffer(64), C.create_string_buffer(64)
        h = GzHeader(extra=C.addressof(eb), extra_max=64, name=C.addressof(nb), name_max=64, comment=C.addressof(cb), comm_max=64)
        L.inflateGetHeader(C.byref(s), C.byref(h))
        src2 = C.create_string_buffer(g, len(g)); pos = 0; outb = bytearray(); rc = 0
        while rc == 0 and pos <= len(g):
            k = min(chunk, len(g) - pos)
            s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src2) + pos, k, C.addressof(out), 64
            rc = L.inflate(C.byref(s), 0); pos += k - s.avail_in; outb += out.raw[:64 - s.avail_out]
            if k == 0: break
        res.append(("gzfields", chunk, rc, bytes(outb), h.done, h.extra_len, eb.raw[:8], nb.raw[60:64], cb.raw[60:64], h.hcrc, h.text, h.os))
        L.inflateEnd(C.byref(s))
    # inflateCopy after half input
    data = oracle_lib.load(False).gen_shard(0, 30000); gzd = gzip.compress(data, 9)
    s = ZStream(); L.inflateInit2_(C.byref(s), 31, ver, zs)
    src3 = C.create_string_buffer(gzd, len(gzd)); big = C.create_string_buffer(40000)
    half = len(gzd) // 2
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src3), half, C.addressof(big), 40000
    r1 = L.inflate(C.byref(s), 0); got1 = 40000 - s.avail_out
    c = ZStream(); rc = L.inflateCopy(C.byref(c), C.byref(s))
    outs = []
    for st in (s, c):
        b2 = C.create_string_buffer(40000)
        st.next_in, st.avail_in, st.next_out, st.avail_out = C.addressof(src3) + half - s.avail_in if st is s else C.addressof(src3) + half - c.avail_in, len(gzd) - half + st.avail_in, C.addressof(b2), 40000
        r = L.inflate(C.byref(st), 4); outs.append((r, big.raw[:got1] + b2.raw[:40000 - st.avail_out] == data)); L.inflateEnd(C.byref(st))
    res.append(("copy_half", r1, rc, outs))
    # inflateValidate toggles checksum
    bad = bytearray(hello); bad[-1] ^= 1
    for val in (1, 0):
        s = ZStream(); L.inflateInit2_(C.byref(s), 15, ver, zs); rv = L.inflateValidate(C.byref(s), val)
        srcb = C.create_string_buffer(bytes(bad), len(bad))
        s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(srcb), len(bad), C.addressof(out), 64
        res.append(("validate", val, rv, L.inflate(C.byref(s), 4))); L.inflateEnd(C.byref(s))
    return res
a = run(lib, "ours"); b = run(sysz, "sys")
for x, y in zip(a, b):
    print("OK  " if x == y else "DIFF", x, "" if x == y else y)
