"""ctypes view of the zlib stream ABI (include/zmi355_zlib.h) -- the calls a C program or the
reference's own examples (test-libz-rs-sys/examples/blogpost-compress.rs:43-122,
blogpost-uncompress.rs:6-44, libz-rs-sys-cdylib/zpipe.c) make."""
import ctypes as C
import os

Z_NO_FLUSH, Z_SYNC_FLUSH, Z_FULL_FLUSH, Z_FINISH = 0, 2, 3, 4
Z_OK, Z_STREAM_END, Z_DATA_ERROR, Z_BUF_ERROR, Z_STREAM_ERROR, Z_VERSION_ERROR = 0, 1, -3, -5, -2, -6
Z_MEM_ERROR = -4


class ZStream(C.Structure):
    _fields_ = [("next_in", C.c_void_p), ("avail_in", C.c_uint), ("total_in", C.c_ulong), ("next_out", C.c_void_p),
                ("avail_out", C.c_uint), ("total_out", C.c_ulong), ("msg", C.c_char_p), ("state", C.c_void_p),
                ("zalloc", C.c_void_p), ("zfree", C.c_void_p), ("opaque", C.c_void_p), ("data_type", C.c_int),
                ("adler", C.c_ulong), ("reserved", C.c_ulong)]


assert C.sizeof(ZStream) == 112  # zlib-rs/src/c_api.rs:54-71 on LP64


def bind(lib):
    lib.zlibVersion.restype = C.c_char_p
    lib.zError.restype = C.c_char_p
    lib.deflateInit2_.argtypes = [C.POINTER(ZStream), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
    lib.deflate.argtypes = [C.POINTER(ZStream), C.c_int]
    lib.deflateEnd.argtypes = [C.POINTER(ZStream)]
    lib.deflateReset.argtypes = [C.POINTER(ZStream)]
    lib.deflateBound.restype = C.c_ulong
    lib.deflateBound.argtypes = [C.POINTER(ZStream), C.c_ulong]
    lib.inflateInit2_.argtypes = [C.POINTER(ZStream), C.c_int, C.c_char_p, C.c_int]
    lib.inflate.argtypes = [C.POINTER(ZStream), C.c_int]
    lib.inflateEnd.argtypes = [C.POINTER(ZStream)]
    lib.compressBound.restype = C.c_ulong
    lib.compressBound.argtypes = [C.c_ulong]
    lib.compress2.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.c_ulong, C.c_int]
    lib.uncompress.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.c_ulong]
    lib.uncompress2.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.POINTER(C.c_ulong)]
    for f in ("adler32", "crc32"):
        getattr(lib, f).restype = C.c_ulong
        getattr(lib, f).argtypes = [C.c_ulong, C.c_void_p, C.c_uint]
    lib.deflateSetDictionary.argtypes = [C.POINTER(ZStream), C.c_char_p, C.c_uint]
    lib.inflateSetDictionary.argtypes = [C.POINTER(ZStream), C.c_char_p, C.c_uint]
    lib.crc32_combine.restype = C.c_ulong
    lib.crc32_combine.argtypes = [C.c_ulong, C.c_ulong, C.c_long]
    lib.adler32_combine.restype = C.c_ulong
    lib.adler32_combine.argtypes = [C.c_ulong, C.c_ulong, C.c_long]
    return lib


def deflate_stream(lib, data, level=6, wbits=15, chunk_in=None, chunk_out=4096, flush_every=None, strategy=0, mem_level=8):
    """the blogpost-compress.rs loop: feed input in chunks, drain output in chunks"""
    strm = ZStream()
    ver = lib.zlibVersion()
    assert lib.deflateInit2_(C.byref(strm), level, 8, wbits, mem_level, strategy, ver, C.sizeof(ZStream)) == Z_OK
    src = C.create_string_buffer(data, len(data) or 1)
    out = bytearray()
    obuf = C.create_string_buffer(chunk_out)
    pos, nchunk = 0, 0
    chunk_in = chunk_in or max(1, len(data))
    while True:
        n = min(chunk_in, len(data) - pos)
        strm.next_in = C.addressof(src) + pos
        strm.avail_in = n
        pos += n
        last = pos >= len(data)
        nchunk += 1
        flush = Z_FINISH if last else (Z_SYNC_FLUSH if flush_every and nchunk % flush_every == 0 else Z_NO_FLUSH)
        while True:
            strm.next_out = C.addressof(obuf)
            strm.avail_out = chunk_out
            rc = lib.deflate(C.byref(strm), flush)
            assert rc in (Z_OK, Z_STREAM_END, Z_BUF_ERROR), rc
            out += obuf.raw[:chunk_out - strm.avail_out]
            if rc == Z_STREAM_END or (strm.avail_out != 0 and flush != Z_FINISH) or rc == Z_BUF_ERROR:
                break
        if last:
            assert rc == Z_STREAM_END
            break
    assert strm.total_in == len(data) and strm.total_out == len(out)
    assert lib.deflateEnd(C.byref(strm)) == Z_OK
    return bytes(out)


def inflate_stream(lib, comp, wbits=15, chunk_in=1 << 30, chunk_out=8192):
    """the blogpost-uncompress.rs loop; returns (rc, output, unused input bytes)"""
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), wbits, lib.zlibVersion(), C.sizeof(ZStream)) == Z_OK
    src = C.create_string_buffer(comp, len(comp) or 1)
    out = bytearray()
    obuf = C.create_string_buffer(chunk_out)
    pos = 0
    rc = Z_OK
    while rc != Z_STREAM_END:
        n = min(chunk_in, len(comp) - pos)
        strm.next_in = C.addressof(src) + pos
        strm.avail_in = n
        strm.next_out = C.addressof(obuf)
        strm.avail_out = chunk_out
        flush = Z_FINISH if pos + n >= len(comp) else Z_NO_FLUSH
        rc = lib.inflate(C.byref(strm), flush)
        pos += n - strm.avail_in
        out += obuf.raw[:chunk_out - strm.avail_out]
        if rc not in (Z_OK, Z_STREAM_END, Z_BUF_ERROR):
            break
        if rc == Z_BUF_ERROR and n == 0 and strm.avail_out != 0:
            break
    lib.inflateEnd(C.byref(strm))
    return rc, bytes(out), len(comp) - pos


def run_abi_checks(lib, o, sizes=(0, 1, 100, 5000, 70000)):
    """shared body of the CPU (emulator) and GPU ABI tests"""
    import zlib
    assert lib.zlibVersion().startswith(b"1.")
    assert lib.zError(-3) == b"data error"
    strm = ZStream()
    assert lib.deflateInit2_(C.byref(strm), 6, 8, 15, 8, 0, b"2.0", C.sizeof(ZStream)) == Z_VERSION_ERROR
    assert lib.deflateInit2_(C.byref(strm), 6, 8, 15, 8, 0, lib.zlibVersion(), 100) == Z_VERSION_ERROR
    assert lib.deflateInit2_(C.byref(strm), 10, 8, 15, 8, 0, lib.zlibVersion(), C.sizeof(ZStream)) == Z_STREAM_ERROR
    assert lib.deflate(None, 0) == Z_STREAM_ERROR and lib.inflate(None, 0) == Z_STREAM_ERROR
    for n in sizes:
        d = (o.gen_shard(2, max(64, (n + 63) // 64 * 64)))[:n]
        # one-shot compress2 / uncompress (libz-rs-sys/src/lib.rs:1529, :499)
        cap = C.c_ulong(lib.compressBound(n))
        dst = C.create_string_buffer(cap.value)
        assert lib.compress2(dst, C.byref(cap), d, n, 6) == Z_OK
        comp = dst.raw[:cap.value]
        assert zlib.decompress(comp) == d
        ocap = C.c_ulong(n + 10)
        back = C.create_string_buffer(n + 10)
        assert lib.uncompress(back, C.byref(ocap), comp, len(comp)) == Z_OK and back.raw[:ocap.value] == d
        if n > 100:
            small = C.c_ulong(n // 2)
            assert lib.uncompress(back, C.byref(small), comp, len(comp)) == Z_BUF_ERROR
            assert lib.uncompress(back, C.byref(ocap), comp[:len(comp) // 2], len(comp) // 2) == Z_DATA_ERROR
        # streaming, all three wrappers, chunked input with sync flushes, small output buffers
        for wbits in (15, 31, -15):
            s1 = deflate_stream(lib, d, 6, wbits)
            assert zlib.decompress(s1, wbits) == d
            s2 = deflate_stream(lib, d, 1, wbits, chunk_in=1500, chunk_out=700, flush_every=3)
            assert zlib.decompress(s2, wbits) == d
            rc, out, unused = inflate_stream(lib, s1 + b"TRAILING", wbits, chunk_in=1 << 30)
            assert rc == Z_STREAM_END and out == d and unused == 8
            ref = zlib.compressobj(6, zlib.DEFLATED, wbits)
            s3 = ref.compress(d) + ref.flush()
            rc, out, unused = inflate_stream(lib, s3, wbits, chunk_in=997, chunk_out=333)
            assert rc == Z_STREAM_END and out == d and unused == 0
        # checksums (libz-rs-sys/src/lib.rs:150-412)
        assert lib.adler32(1, d, n) == zlib.adler32(d) and lib.crc32(0, d, n) == zlib.crc32(d)
        k = n // 3
        assert lib.crc32_combine(zlib.crc32(d[:k]), zlib.crc32(d[k:]), n - k) == zlib.crc32(d)
        assert lib.adler32_combine(zlib.adler32(d[:k]), zlib.adler32(d[k:]), n - k) == zlib.adler32(d)
    bad = bytearray(zlib.compress(o.gen_shard(0, 4096)))
    bad[50] ^= 0xFF
    rc, out, unused = inflate_stream(lib, bytes(bad), 15)
    assert rc == Z_DATA_ERROR


def dictionary_checks(lib, data, zdict, expect_gain=True):
    """deflateSetDictionary / inflateSetDictionary (libz-rs-sys/src/lib.rs:1689, :1121) against system zlib"""
    import zlib
    Z_NEED_DICT = 2
    for wbits in (15, -15, 10, -11):   # small windows: only the dictionary's last 2^windowBits - 262 bytes are reachable
        strm = ZStream()
        assert lib.deflateInit2_(C.byref(strm), 6, 8, wbits, 8, 0, lib.zlibVersion(), C.sizeof(ZStream)) == Z_OK
        assert lib.deflateSetDictionary(C.byref(strm), zdict, len(zdict)) == Z_OK
        if wbits > 0:
            assert strm.adler == zlib.adler32(zdict)
        src = C.create_string_buffer(data, len(data))
        cap = len(data) + 4096
        dst = C.create_string_buffer(cap)
        strm.next_in, strm.avail_in = C.addressof(src), len(data)
        strm.next_out, strm.avail_out = C.addressof(dst), cap
        assert lib.deflate(C.byref(strm), Z_FINISH) == Z_STREAM_END
        comp = dst.raw[:cap - strm.avail_out]
        assert lib.deflateEnd(C.byref(strm)) == Z_OK
        # system zlib decodes it with the same dictionary, and the dictionary was actually used
        d = zlib.decompressobj(wbits, zdict=zdict)
        assert d.decompress(comp) + d.flush() == data
        plain = deflate_stream(lib, data, level=6, wbits=wbits)
        if expect_gain and abs(wbits) == 15:
            assert len(comp) < len(plain), (len(comp), len(plain))
        # our inflate: Z_NEED_DICT for the wrapped stream, then the data; a wrong dictionary is refused
        ref = zlib.compressobj(6, zlib.DEFLATED, wbits, zdict=zdict)
        for stream in (comp, ref.compress(data) + ref.flush()):
            strm = ZStream()
            assert lib.inflateInit2_(C.byref(strm), wbits, lib.zlibVersion(), C.sizeof(ZStream)) == Z_OK
            csrc = C.create_string_buffer(stream, len(stream))
            out = C.create_string_buffer(len(data) + 16)
            strm.next_in, strm.avail_in = C.addressof(csrc), len(stream)
            strm.next_out, strm.avail_out = C.addressof(out), len(data) + 16
            if wbits > 0:
                assert lib.inflateSetDictionary(C.byref(strm), zdict, len(zdict)) == Z_STREAM_ERROR   # not asked for yet
                assert lib.inflate(C.byref(strm), Z_NO_FLUSH) == Z_NEED_DICT
                assert strm.adler == zlib.adler32(zdict)
                assert lib.inflateSetDictionary(C.byref(strm), b"wrong" + zdict, len(zdict) + 5) == Z_DATA_ERROR
            assert lib.inflateSetDictionary(C.byref(strm), zdict, len(zdict)) == Z_OK
            rc = lib.inflate(C.byref(strm), Z_FINISH)
            assert rc == Z_STREAM_END, rc
            assert out.raw[:len(data) + 16 - strm.avail_out] == data
            assert lib.inflateEnd(C.byref(strm)) == Z_OK
    # gzip streams take no dictionary (deflate.rs:507-509)
    strm = ZStream()
    assert lib.deflateInit2_(C.byref(strm), 6, 8, 31, 8, 0, lib.zlibVersion(), C.sizeof(ZStream)) == Z_OK
    assert lib.deflateSetDictionary(C.byref(strm), zdict, len(zdict)) == Z_STREAM_ERROR
    assert lib.deflateEnd(C.byref(strm)) == Z_OK


class GzHeader(C.Structure):
    """gz_header of include/zmi355_zlib.h (zlib-rs/src/c_api.rs:174-203)"""
    _fields_ = [("text", C.c_int), ("time", C.c_ulong), ("xflags", C.c_int), ("os", C.c_int), ("extra", C.c_void_p),
                ("extra_len", C.c_uint), ("extra_max", C.c_uint), ("name", C.c_void_p), ("name_max", C.c_uint),
                ("comment", C.c_void_p), ("comm_max", C.c_uint), ("hcrc", C.c_int), ("done", C.c_int)]


def header_copy_checks(lib, data):
    """deflateSetHeader / inflateGetHeader (lib.rs:1319, :1179), deflateCopy / inflateCopy (lib.rs:1837, :815),
    deflateGetDictionary / inflateGetDictionary (lib.rs:2332, :2287) against Python's gzip module"""
    import gzip
    import io
    import zlib
    for f in ("deflateSetHeader", "inflateGetHeader", "deflateCopy", "inflateCopy", "deflateGetDictionary", "inflateGetDictionary",
              "deflateResetKeep", "inflateResetKeep"):
        getattr(lib, f).restype = C.c_int
    lib.deflateSetHeader.argtypes = [C.POINTER(ZStream), C.POINTER(GzHeader)]
    lib.inflateGetHeader.argtypes = [C.POINTER(ZStream), C.POINTER(GzHeader)]
    lib.deflateCopy.argtypes = [C.POINTER(ZStream), C.POINTER(ZStream)]
    lib.inflateCopy.argtypes = [C.POINTER(ZStream), C.POINTER(ZStream)]
    lib.deflateGetDictionary.argtypes = [C.POINTER(ZStream), C.c_char_p, C.POINTER(C.c_uint)]
    lib.inflateGetDictionary.argtypes = [C.POINTER(ZStream), C.c_char_p, C.POINTER(C.c_uint)]
    ver, zs = lib.zlibVersion(), C.sizeof(ZStream)

    # ---- a gzip stream with name, comment, extra field, mtime and header CRC: Python's gzip module reads it
    name, comment, extra = C.create_string_buffer(b"shard-0001.bin"), C.create_string_buffer(b"made on an MI355X"), C.create_string_buffer(b"AB\x02\x00xy", 6)
    h = GzHeader(text=1, time=1234567890, os=3, extra=C.addressof(extra), extra_len=6, name=C.addressof(name),
                 comment=C.addressof(comment), hcrc=1)
    strm = ZStream()
    assert lib.deflateInit2_(C.byref(strm), 6, 8, 15, 8, 0, ver, zs) == Z_OK
    assert lib.deflateSetHeader(C.byref(strm), C.byref(h)) == Z_STREAM_ERROR      # only gzip streams have a header
    assert lib.deflateEnd(C.byref(strm)) == Z_OK
    strm = ZStream()
    assert lib.deflateInit2_(C.byref(strm), 6, 8, 31, 8, 0, ver, zs) == Z_OK
    assert lib.deflateSetHeader(C.byref(strm), C.byref(h)) == Z_OK
    src = C.create_string_buffer(data, len(data))
    cap = len(data) + 4096
    dst = C.create_string_buffer(cap)
    half = len(data) // 2
    strm.next_in, strm.avail_in = C.addressof(src), half
    strm.next_out, strm.avail_out = C.addressof(dst), cap
    assert lib.deflate(C.byref(strm), Z_NO_FLUSH) == Z_OK
    head = dst.raw[:cap - strm.avail_out]     # at least the gzip header: the first call writes it (deflate.rs:2543-2627)
    assert head[:4] == b"\x1f\x8b\x08\x1f"
    # deflateCopy in the middle of a stream: both copies finish it identically
    twin = ZStream()
    assert lib.deflateCopy(C.byref(twin), C.byref(strm)) == Z_OK
    dlen = C.c_uint(0)
    dbuf = C.create_string_buffer(32768)
    assert lib.deflateGetDictionary(C.byref(strm), dbuf, C.byref(dlen)) == Z_OK
    assert dbuf.raw[:dlen.value] == data[:half][-32768:]
    outs = []
    for st_ in (strm, twin):
        d2 = C.create_string_buffer(cap)
        st_.next_in, st_.avail_in = C.addressof(src) + half, len(data) - half
        st_.next_out, st_.avail_out = C.addressof(d2), cap
        assert lib.deflate(C.byref(st_), Z_FINISH) == Z_STREAM_END
        outs.append(d2.raw[:cap - st_.avail_out])
        assert lib.deflateEnd(C.byref(st_)) == Z_OK
    assert outs[0] == outs[1]
    comp = head + outs[0]
    assert comp[:4] == b"\x1f\x8b\x08\x1f" and comp[4:8] == (1234567890).to_bytes(4, "little")
    g = gzip.GzipFile(fileobj=io.BytesIO(comp))
    assert g.read() == data and g.mtime == 1234567890
    assert zlib.crc32(comp[:comp.index(b"made on an MI355X\x00") + 18]) & 0xFFFF == int.from_bytes(comp[comp.index(b"made on an MI355X\x00") + 18:][:2], "little")

    # ---- inflateGetHeader: our own stream and one written by Python (name + mtime)
    bio = io.BytesIO()
    with gzip.GzipFile(filename="from-python.txt", mode="wb", fileobj=bio, mtime=42) as gz:
        gz.write(data)
    for stream, want_name, want_time, want_comment in ((comp, b"shard-0001.bin", 1234567890, b"made on an MI355X"), (bio.getvalue(), b"from-python.txt", 42, None)):
        nbuf, cbuf, xbuf = C.create_string_buffer(64), C.create_string_buffer(64), C.create_string_buffer(16)
        hh = GzHeader(extra=C.addressof(xbuf), extra_max=16, name=C.addressof(nbuf), name_max=64, comment=C.addressof(cbuf), comm_max=64)
        strm = ZStream()
        assert lib.inflateInit2_(C.byref(strm), 31, ver, zs) == Z_OK
        assert lib.inflateGetHeader(C.byref(strm), C.byref(hh)) == Z_OK and hh.done == 0
        csrc = C.create_string_buffer(stream, len(stream))
        out = C.create_string_buffer(len(data) + 16)
        strm.next_in, strm.avail_in = C.addressof(csrc), len(stream)
        strm.next_out, strm.avail_out = C.addressof(out), len(data) + 16
        assert lib.inflate(C.byref(strm), Z_FINISH) == Z_STREAM_END
        assert out.raw[:len(data)] == data
        assert hh.done == 1 and hh.time == want_time and nbuf.value == want_name
        if want_comment is not None:
            assert cbuf.value == want_comment and hh.hcrc == 1 and hh.extra_len == 6 and xbuf.raw[:6] == b"AB\x02\x00xy" and hh.text == 1
        else:
            assert not hh.comment
        # inflateCopy of a finished stream and the window it reports
        twin = ZStream()
        assert lib.inflateCopy(C.byref(twin), C.byref(strm)) == Z_OK
        dlen = C.c_uint(0)
        assert lib.inflateGetDictionary(C.byref(twin), dbuf, C.byref(dlen)) == Z_OK
        assert dbuf.raw[:dlen.value] == data[-32768:]
        assert lib.inflateEnd(C.byref(twin)) == Z_OK and lib.inflateEnd(C.byref(strm)) == Z_OK
    # a zlib stream has no gzip header: Z_STREAM_ERROR (inflate.rs:2676-2678)
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), 15, ver, zs) == Z_OK
    assert lib.inflateGetHeader(C.byref(strm), C.byref(GzHeader())) == Z_STREAM_ERROR
    assert lib.inflateResetKeep(C.byref(strm)) == Z_OK and lib.inflateEnd(C.byref(strm)) == Z_OK


def progressive_inflate_checks(lib, data):
    """output is handed out as the input arrives (not only once the stream is complete), and the bytes in front of a
    corrupt spot are delivered before Z_DATA_ERROR, as the reference's streaming state machine does"""
    import zlib
    comp = zlib.compress(data, 6)
    ver, zs = lib.zlibVersion(), C.sizeof(ZStream)
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), 15, ver, zs) == Z_OK
    csrc = C.create_string_buffer(comp, len(comp))
    out = C.create_string_buffer(len(data) + 16)
    half = len(comp) // 2
    strm.next_in, strm.avail_in = C.addressof(csrc), half
    strm.next_out, strm.avail_out = C.addressof(out), len(data) + 16
    assert lib.inflate(C.byref(strm), Z_NO_FLUSH) == Z_OK
    got = strm.total_out
    assert 0 < got < len(data) and out.raw[:got] == data[:got]          # roughly half of the output is there already
    assert got > len(data) // 4
    strm.next_in, strm.avail_in = C.addressof(csrc) + half, len(comp) - half
    assert lib.inflate(C.byref(strm), Z_FINISH) == Z_STREAM_END
    assert strm.total_out == len(data) and out.raw[:len(data)] == data
    assert lib.inflateEnd(C.byref(strm)) == Z_OK
    # corrupt the second half: the first part still comes out, then the error
    bad = bytearray(comp)
    bad[(len(comp) * 3) // 4] ^= 0xFF
    bad = bytes(bad)
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), 15, ver, zs) == Z_OK
    bsrc = C.create_string_buffer(bad, len(bad))
    strm.next_in, strm.avail_in = C.addressof(bsrc), len(bad)
    small = C.create_string_buffer(4096)
    collected = bytearray()
    rc = Z_OK
    for _ in range(len(data) // 4096 + 8):
        strm.next_out, strm.avail_out = C.addressof(small), 4096
        rc = lib.inflate(C.byref(strm), Z_NO_FLUSH)
        collected += small.raw[:4096 - strm.avail_out]
        if rc != Z_OK:
            break
    assert rc == Z_DATA_ERROR, rc
    # everything decodable came out first; the part in front of the flipped byte is the original data
    assert len(collected) > len(data) // 2 and bytes(collected[:len(data) // 2]) == data[:len(data) // 2]
    assert lib.inflateEnd(C.byref(strm)) == Z_OK


IN_FUNC = C.CFUNCTYPE(C.c_uint, C.c_void_p, C.POINTER(C.c_void_p))
OUT_FUNC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint)


def _bind_streaming(lib):
    P = C.POINTER(ZStream)
    lib.inflateSync.argtypes = [P]
    lib.inflateSyncPoint.argtypes = [P]
    lib.inflatePrime.argtypes = [P, C.c_int, C.c_int]
    lib.inflateMark.restype = C.c_long
    lib.inflateMark.argtypes = [P]
    lib.inflateValidate.argtypes = [P, C.c_int]
    lib.inflateReset.argtypes = [P]
    lib.deflatePrime.argtypes = [P, C.c_int, C.c_int]
    lib.deflatePending.argtypes = [P, C.POINTER(C.c_uint), C.POINTER(C.c_int)]
    lib.inflateBackInit_.argtypes = [P, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
    lib.inflateBack.argtypes = [P, IN_FUNC, C.c_void_p, OUT_FUNC, C.c_void_p]
    lib.inflateBackEnd.argtypes = [P]


def _feed(lib, strm, buf_addr, n, out, flush=Z_NO_FLUSH, room=1 << 16):
    """one inflate() call over n input bytes, collecting what comes out (repeats while the room fills up)"""
    obuf = C.create_string_buffer(room)
    strm.next_in, strm.avail_in = buf_addr, n
    while True:
        strm.next_out, strm.avail_out = C.addressof(obuf), room
        rc = lib.inflate(C.byref(strm), flush)
        out += obuf.raw[:room - strm.avail_out]
        if rc != Z_OK or strm.avail_out != 0:
            return rc


def streaming_checks(lib, data, syslib=None):
    """the stream ABI when it is driven in pieces: resumable inflate (inflate.rs:288-320 keeps the state the device
    checkpoint replaces), sync / prime / mark / validate, inflateBack, deflatePrime / deflateUsed.
    syslib: the system's libz as an independent implementation of the same entry points."""
    import zlib
    _bind_streaming(lib)
    ver, zs = lib.zlibVersion(), C.sizeof(ZStream)
    assert len(data) >= 60000

    # -- 1. a packet protocol: every Z_SYNC_FLUSHed packet must come out completely in the call that delivers it
    co = zlib.compressobj(6, zlib.DEFLATED, 15)
    cuts = [0, 700, 701, 9000, 30000, len(data)]
    packets = []
    for a, b in zip(cuts, cuts[1:]):
        packets.append(co.compress(data[a:b]) + co.flush(zlib.Z_SYNC_FLUSH))
    tail = co.flush()
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), 15, ver, zs) == Z_OK
    got = bytearray()
    for i, pk in enumerate(packets):
        src = C.create_string_buffer(pk, len(pk))
        rc = _feed(lib, strm, C.addressof(src), len(pk), got)
        assert rc == Z_OK, rc
        assert bytes(got) == data[:cuts[i + 1]], (i, len(got), cuts[i + 1])
        assert strm.avail_in == 0 and strm.data_type & 128        # stopped right behind a block
    src = C.create_string_buffer(tail + b"XYZ", len(tail) + 3)
    assert _feed(lib, strm, C.addressof(src), len(tail) + 3, got) == Z_STREAM_END
    assert strm.avail_in == 3 and strm.total_in == sum(map(len, packets)) + len(tail) and strm.total_out == len(data)
    assert strm.adler == zlib.adler32(data)
    assert lib.inflateEnd(C.byref(strm)) == Z_OK

    # -- 2. in small pieces (a tiny stream one byte at a time), every wrapper; no Z_FINISH anywhere (the zpipe.c loop)
    small = data[:20000]
    for wbits, payload, step in ((15, data[:90], 1), (15, small, 501), (31, small, 97), (-15, small, 333), (47, small, 1000)):
        comp = zlib.compressobj(6, zlib.DEFLATED, 31 if wbits == 47 else wbits)
        blob = comp.compress(payload) + comp.flush()
        strm = ZStream()
        assert lib.inflateInit2_(C.byref(strm), wbits, ver, zs) == Z_OK
        src = C.create_string_buffer(blob, len(blob))
        got = bytearray()
        rc = Z_OK
        for at in range(0, len(blob), step):
            rc = _feed(lib, strm, C.addressof(src) + at, min(step, len(blob) - at), got)
            assert rc in (Z_OK, Z_STREAM_END), (wbits, at, rc)
        assert rc == Z_STREAM_END and bytes(got) == payload, (wbits, rc, len(got))
        assert lib.inflateEnd(C.byref(strm)) == Z_OK

    # -- 3. header and trailer errors carry the reference's messages (inflate.rs:1000-1062, 1790-1839)
    good = zlib.compress(small, 6)
    for blob, msg in ((b"\x78\x9d" + good[2:], b"incorrect header check"), (b"\x77\x9c"[:1] + b"\x85" + good[2:], b"unknown compression method"),
                      (good[:-1] + bytes([good[-1] ^ 1]), b"incorrect data check")):
        strm = ZStream()
        assert lib.inflateInit2_(C.byref(strm), 15, ver, zs) == Z_OK
        src = C.create_string_buffer(blob, len(blob))
        got = bytearray()
        rc = _feed(lib, strm, C.addressof(src), len(blob), got, Z_FINISH)
        assert rc == Z_DATA_ERROR and strm.msg == msg, (rc, strm.msg, msg)
        assert lib.inflateEnd(C.byref(strm)) == Z_OK
    gz = zlib.compressobj(6, zlib.DEFLATED, 31)
    gzblob = gz.compress(small) + gz.flush()
    badlen = gzblob[:-4] + bytes([gzblob[-4] ^ 1]) + gzblob[-3:]
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), 31, ver, zs) == Z_OK
    src = C.create_string_buffer(badlen, len(badlen))
    got = bytearray()
    assert _feed(lib, strm, C.addressof(src), len(badlen), got, Z_FINISH) == Z_DATA_ERROR and strm.msg == b"incorrect length check"
    # inflateValidate(0): the same stream passes when the check is switched off (inflate.rs:2595)
    assert lib.inflateReset(C.byref(strm)) == Z_OK and lib.inflateValidate(C.byref(strm), 0) == Z_OK
    got = bytearray()
    assert _feed(lib, strm, C.addressof(src), len(badlen), got, Z_FINISH) == Z_STREAM_END and bytes(got) == small
    assert lib.inflateEnd(C.byref(strm)) == Z_OK

    # -- 4. inflateSync: damage in front of a Z_FULL_FLUSH point, resume behind it (inflate.rs:2458-2535)
    co = zlib.compressobj(6, zlib.DEFLATED, 15)
    first = co.compress(data[:30000]) + co.flush(zlib.Z_FULL_FLUSH)
    second = co.compress(data[30000:60000]) + co.flush()
    hurt = bytearray(first)
    for i in range(40, 60):
        hurt[i] ^= 0x5A
    blob = bytes(hurt) + second
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), 15, ver, zs) == Z_OK
    src = C.create_string_buffer(blob, len(blob))
    got = bytearray()
    k = len(first) - 100                                     # the error shows up before the flush point arrives
    rc = _feed(lib, strm, C.addressof(src), k, got)
    assert rc == Z_DATA_ERROR, rc
    strm.next_in, strm.avail_in = C.addressof(src) + k, len(blob) - k
    assert lib.inflateSync(C.byref(strm)) == Z_OK
    assert strm.total_in == len(first), (strm.total_in, len(first))          # "where valid compressed data was found"
    rest = bytearray()
    rc = _feed(lib, strm, strm.next_in, strm.avail_in, rest, Z_FINISH)
    assert rc == Z_STREAM_END and bytes(rest) == data[30000:60000], (rc, len(rest))
    assert lib.inflateEnd(C.byref(strm)) == Z_OK
    # no marker in the input: Z_DATA_ERROR, everything consumed; with nothing to search: Z_BUF_ERROR
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), -15, ver, zs) == Z_OK
    junk = C.create_string_buffer(b"\x01\x02\x03\x00\x00\xff\x01" * 10, 70)
    assert lib.inflateSync(C.byref(strm)) == Z_BUF_ERROR
    strm.next_in, strm.avail_in = C.addressof(junk), 70
    assert lib.inflateSync(C.byref(strm)) == Z_DATA_ERROR and strm.avail_in == 0 and strm.total_in == 70
    assert lib.inflateEnd(C.byref(strm)) == Z_OK

    # -- 5. inflateSyncPoint (inflate.rs:2537): true while the LEN bytes of a flush marker are still missing
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    pk = co.compress(small) + co.flush(zlib.Z_SYNC_FLUSH)
    assert pk[-4:] == b"\x00\x00\xff\xff"
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), -15, ver, zs) == Z_OK
    assert lib.inflateMark(C.byref(strm)) in (-65536, -(1 << 63))        # nothing in flight
    src = C.create_string_buffer(pk, len(pk))
    got = bytearray()
    assert _feed(lib, strm, C.addressof(src), len(pk) - 4, got) == Z_OK and bytes(got) == small
    assert lib.inflateSyncPoint(C.byref(strm)) == 1
    assert _feed(lib, strm, C.addressof(src) + len(pk) - 4, 4, got) in (Z_OK, Z_BUF_ERROR)
    assert lib.inflateSyncPoint(C.byref(strm)) == 0
    assert lib.inflateMark(C.byref(strm)) == -65536
    assert lib.inflateEnd(C.byref(strm)) == Z_OK

    # -- 6. deflatePrime / inflatePrime (deflate.rs:566, inflate.rs:2160): 3 bits in front of a raw stream
    strm = ZStream()
    assert lib.deflateInit2_(C.byref(strm), 6, 8, -15, 8, 0, ver, zs) == Z_OK
    assert lib.deflatePrime(C.byref(strm), 3, 0b101) == Z_OK
    pend, bits = C.c_uint(9), C.c_int(9)
    assert lib.deflatePending(C.byref(strm), C.byref(pend), C.byref(bits)) == Z_OK and (pend.value, bits.value) == (0, 3)
    assert lib.deflatePrime(C.byref(strm), 33, 0) == Z_BUF_ERROR
    src = C.create_string_buffer(small, len(small))
    obuf = C.create_string_buffer(len(small) + 1000)
    strm.next_in, strm.avail_in = C.addressof(src), len(small)
    strm.next_out, strm.avail_out = C.addressof(obuf), len(small) + 1000
    assert lib.deflate(C.byref(strm), Z_FINISH) == Z_STREAM_END
    primed = obuf.raw[:strm.total_out]
    used = C.c_int(0)
    if hasattr(lib, "deflateUsed"):
        lib.deflateUsed.argtypes = [C.POINTER(ZStream), C.POINTER(C.c_int)]
        assert lib.deflateUsed(C.byref(strm), C.byref(used)) == Z_OK and 1 <= used.value <= 8
        assert primed[-1] >> used.value == 0 and (used.value == 1 or primed[-1] != 0 or True)
    assert lib.deflateEnd(C.byref(strm)) == Z_OK
    assert primed[0] & 7 == 0b101
    for L in [lib] + ([syslib] if syslib is not None else []):      # the system's zlib reads our primed stream too
        _bind_streaming(L) if L is lib else None
        if L is not lib:
            L.inflateInit2_.argtypes = [C.POINTER(ZStream), C.c_int, C.c_char_p, C.c_int]
            L.inflate.argtypes = [C.POINTER(ZStream), C.c_int]
            L.inflateEnd.argtypes = [C.POINTER(ZStream)]
            L.inflatePrime.argtypes = [C.POINTER(ZStream), C.c_int, C.c_int]
            L.zlibVersion.restype = C.c_char_p
        strm = ZStream()
        assert L.inflateInit2_(C.byref(strm), -15, L.zlibVersion(), zs) == Z_OK
        assert L.inflatePrime(C.byref(strm), 5, primed[0] >> 3) == Z_OK       # skip the three foreign bits
        psrc = C.create_string_buffer(primed, len(primed))
        got = bytearray()
        assert _feed(L, strm, C.addressof(psrc) + 1, len(primed) - 1, got, Z_FINISH) == Z_STREAM_END
        assert bytes(got) == small
        assert L.inflateEnd(C.byref(strm)) == Z_OK
    # an empty stream ends with the 10 bits of an empty static block: two bits of the last byte are in use
    if hasattr(lib, "deflateUsed"):
        strm = ZStream()
        assert lib.deflateInit2_(C.byref(strm), 6, 8, -15, 8, 0, ver, zs) == Z_OK
        assert lib.deflateUsed(C.byref(strm), C.byref(used)) == Z_OK and used.value == 0
        strm.next_in, strm.avail_in = None, 0
        strm.next_out, strm.avail_out = C.addressof(obuf), 100
        assert lib.deflate(C.byref(strm), Z_FINISH) == Z_STREAM_END and obuf.raw[:2] == b"\x03\x00"
        assert lib.deflateUsed(C.byref(strm), C.byref(used)) == Z_OK and used.value == 2
        assert lib.deflateEnd(C.byref(strm)) == Z_OK

    # -- 7. inflateBack (inflate/infback.rs): input pulled in pieces, output pushed one window at a time
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    blob = raw.compress(data[:60000]) + raw.flush() + b"TRAILING"
    holder = C.create_string_buffer(blob, len(blob))
    state = {"pos": 0, "out": bytearray(), "calls": 0}

    def pull(desc, bufp):
        n = min(1500, len(blob) - state["pos"])
        bufp[0] = C.addressof(holder) + state["pos"]
        state["pos"] += n
        return n

    def push(desc, buf, n):
        state["out"] += C.string_at(buf, n)
        state["calls"] += 1
        return 0

    window = C.create_string_buffer(1 << 15)
    strm = ZStream()
    assert lib.inflateBackInit_(C.byref(strm), 15, C.addressof(window), ver, zs) == Z_OK
    strm.next_in, strm.avail_in = None, 0
    rc = lib.inflateBack(C.byref(strm), IN_FUNC(pull), None, OUT_FUNC(push), None)
    assert rc == Z_STREAM_END, rc
    assert bytes(state["out"]) == data[:60000] and state["calls"] == 2        # 32768 + 27232
    left = C.string_at(strm.next_in, strm.avail_in) + blob[state["pos"]:]
    assert left == b"TRAILING", left
    # truncated input: Z_BUF_ERROR with next_in NULL; a failing out(): Z_BUF_ERROR with next_in set (infback.rs:705-722)
    state.update(pos=0, out=bytearray(), calls=0)
    full, blob = blob, blob[:4000]
    strm.next_in, strm.avail_in = None, 0
    rc = lib.inflateBack(C.byref(strm), IN_FUNC(pull), None, OUT_FUNC(push), None)
    assert rc == Z_BUF_ERROR and not strm.next_in
    state.update(pos=0, out=bytearray(), calls=0)
    blob = full
    strm.next_in, strm.avail_in = None, 0
    rc = lib.inflateBack(C.byref(strm), IN_FUNC(pull), None, OUT_FUNC(lambda d, b, n: 1), None)
    assert rc == Z_BUF_ERROR and strm.next_in
    assert lib.inflateBackEnd(C.byref(strm)) == Z_OK


def _bind_gz(lib):
    vp = C.c_void_p
    lib.gzopen.restype = vp
    lib.gzopen.argtypes = [C.c_char_p, C.c_char_p]
    lib.gzdopen.restype = vp
    lib.gzdopen.argtypes = [C.c_int, C.c_char_p]
    lib.gzbuffer.argtypes = [vp, C.c_uint]
    lib.gzread.argtypes = [vp, vp, C.c_uint]
    lib.gzwrite.argtypes = [vp, vp, C.c_uint]
    lib.gzfread.restype = C.c_size_t
    lib.gzfread.argtypes = [vp, C.c_size_t, C.c_size_t, vp]
    lib.gzfwrite.restype = C.c_size_t
    lib.gzfwrite.argtypes = [vp, C.c_size_t, C.c_size_t, vp]
    lib.gzputs.argtypes = [vp, C.c_char_p]
    lib.gzputc.argtypes = [vp, C.c_int]
    lib.gzgetc.argtypes = [vp]
    lib.gzungetc.argtypes = [C.c_int, vp]
    lib.gzgets.restype = vp
    lib.gzgets.argtypes = [vp, vp, C.c_int]
    lib.gzprintf.argtypes = [vp, C.c_char_p]
    lib.gzflush.argtypes = [vp, C.c_int]
    lib.gzsetparams.argtypes = [vp, C.c_int, C.c_int]
    lib.gzseek.restype = C.c_long
    lib.gzseek.argtypes = [vp, C.c_long, C.c_int]
    lib.gztell.restype = C.c_long
    lib.gztell.argtypes = [vp]
    lib.gzoffset.restype = C.c_long
    lib.gzoffset.argtypes = [vp]
    lib.gzrewind.argtypes = [vp]
    lib.gzeof.argtypes = [vp]
    lib.gzdirect.argtypes = [vp]
    lib.gzclose.argtypes = [vp]
    lib.gzclose_r.argtypes = [vp]
    lib.gzclose_w.argtypes = [vp]
    lib.gzerror.restype = C.c_char_p
    lib.gzerror.argtypes = [vp, C.POINTER(C.c_int)]
    lib.gzclearerr.argtypes = [vp]


def _gz_read_all(lib, path, chunk=50000, bufsize=None):
    f = lib.gzopen(path.encode(), b"rb")
    assert f
    if bufsize:
        assert lib.gzbuffer(f, bufsize) == 0
    buf = C.create_string_buffer(chunk)
    out = bytearray()
    while True:
        n = lib.gzread(f, buf, chunk)
        assert n >= 0, lib.gzerror(f, None)
        out += buf.raw[:n]
        if n < chunk:
            break
    assert lib.gzeof(f) == 1
    assert lib.gzclose(f) == Z_OK
    return bytes(out)


def gz_checks(lib, tmpdir, data, syslib=None):
    """the gz* file API (libz-rs-sys/src/gz.rs) against Python's gzip module and, when given, the system's libz:
    files written here are read there and the other way round; concatenated members, plain files, seeking, line and
    character I/O, append, error reporting"""
    import gzip
    import os
    _bind_gz(lib)
    if syslib is not None:
        _bind_gz(syslib)
    p = lambda name: os.path.join(str(tmpdir), name)
    text = b"".join(b"line %d of the test file\n" % i for i in range(3000))

    # -- write here (one big write, small writes, puts / putc / printf, a flush in between), read with Python and libz
    f = lib.gzopen(p("a.gz").encode(), b"wb6")
    assert f and lib.gzdirect(f) == 0
    src = C.create_string_buffer(data, len(data))
    assert lib.gzwrite(f, src, len(data)) == len(data)
    assert lib.gztell(f) == len(data)
    for i in range(0, 3000, 7):
        assert lib.gzwrite(f, C.addressof(src) + i, 7) == 7
    assert lib.gzflush(f, Z_SYNC_FLUSH) == Z_OK
    assert lib.gzputs(f, b"hello, ") == 7 and lib.gzputc(f, ord("w")) == ord("w")
    assert lib.gzprintf(f, b"orld %d %s\n", 42, b"ok") == 11
    assert lib.gzfwrite(src, 10, 5, f) == 5
    assert lib.gzread(f, src, 1) == -1                       # a write handle does not read
    assert lib.gzclose(f) == Z_OK
    small = b"".join(data[i:i + 7] for i in range(0, 3000, 7))
    expect = data + small + b"hello, world 42 ok\n" + data[:50]
    assert gzip.open(p("a.gz"), "rb").read() == expect
    assert _gz_read_all(lib, p("a.gz")) == expect
    assert _gz_read_all(lib, p("a.gz"), chunk=1000, bufsize=512) == expect     # far more output than the buffer holds
    if syslib is not None:
        assert _gz_read_all(syslib, p("a.gz")) == expect

    # -- read files made elsewhere: several members (two writers + append here), then junk that is not gzip
    with open(p("b.gz"), "wb") as fh:
        fh.write(gzip.compress(data[:40000], 6))
        fh.write(gzip.compress(text, 1))
    f = lib.gzopen(p("b.gz").encode(), b"ab9")               # append: a third member
    assert f and lib.gzwrite(f, src, 1234) == 1234 and lib.gzclose_w(f) == Z_OK
    with open(p("b.gz"), "ab") as fh:
        fh.write(b"\x00\x00 trailing junk that is no gzip member")
    whole = data[:40000] + text + data[:1234]
    assert _gz_read_all(lib, p("b.gz"), chunk=777) == whole
    assert _gz_read_all(lib, p("b.gz"), chunk=1 << 20) == whole       # large requests decode into the caller's buffer
    assert gzip.open(p("b.gz"), "rb").read(len(whole)) == whole

    # -- lines, characters, push-back, positions
    f = lib.gzopen(p("b.gz").encode(), b"r")
    assert lib.gzbuffer(f, 4096) == 0
    assert lib.gzseek(f, 40000, 0) == 40000 and lib.gztell(f) == 40000
    line = C.create_string_buffer(100)
    assert lib.gzgets(f, line, 100) and line.value == b"line 0 of the test file\n"
    assert lib.gzbuffer(f, 8192) == -1                       # too late
    assert lib.gzgets(f, line, 10) and line.value == b"line 1 of"      # len - 1 characters
    assert lib.gzgetc(f) == ord(" ")
    assert lib.gzungetc(ord("#"), f) == ord("#") and lib.gzungetc(ord("!"), f) == ord("!")
    assert lib.gzgets(f, line, 100) and line.value == b"!#the test file\n"
    at = lib.gztell(f)
    assert at == 40000 + 2 * len(b"line 0 of the test file\n")
    assert lib.gzseek(f, -20, 1) == at - 20                  # backwards: rewind and skip
    got = C.create_string_buffer(20)
    assert lib.gzread(f, got, 20) == 20 and got.raw == whole[at - 20:at]
    assert lib.gzseek(f, 0, 2) == -1                         # SEEK_END is not supported (gz.rs:2530)
    assert 0 < lib.gzoffset(f) <= os.path.getsize(p("b.gz"))
    assert lib.gzrewind(f) == 0 and lib.gztell(f) == 0 and lib.gzgetc(f) == data[0]
    assert lib.gzfread(got, 4, 5, f) == 5 and got.raw == whole[1:21]
    assert lib.gzeof(f) == 0
    assert lib.gzclose_r(f) == Z_OK

    # -- a plain file is passed through (and can really seek); "T" writes one
    f = lib.gzopen(p("plain.txt").encode(), b"wT")
    assert f and lib.gzdirect(f) == 1
    tsrc = C.create_string_buffer(text, len(text))
    assert lib.gzwrite(f, tsrc, len(text)) == len(text) and lib.gzclose(f) == Z_OK
    assert open(p("plain.txt"), "rb").read() == text
    f = lib.gzopen(p("plain.txt").encode(), b"r")
    assert lib.gzdirect(f) == 1
    assert lib.gzseek(f, 5000, 0) == 5000
    assert lib.gzread(f, got, 20) == 20 and got.raw == text[5000:5020]
    assert lib.gzclose(f) == Z_OK
    assert _gz_read_all(lib, p("plain.txt")) == text

    # -- writing with a seek (zeros), changed parameters, an fd handle; read back by Python
    fd = os.open(p("c.gz"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    f = lib.gzdopen(fd, b"w1")
    assert f and lib.gzwrite(f, src, 1000) == 1000
    assert lib.gzsetparams(f, 9, 1) == Z_OK
    assert lib.gzseek(f, -1, 1) == -1                        # forward only
    assert lib.gzseek(f, 500, 1) == 1500                     # 500 zero bytes
    assert lib.gzwrite(f, src, 1000) == 1000 and lib.gztell(f) == 2500
    assert lib.gzclose(f) == Z_OK
    assert gzip.open(p("c.gz"), "rb").read() == data[:1000] + bytes(500) + data[:1000]

    # -- long formatted output, lines longer than the buffer, push-back until the buffer is full
    f = lib.gzopen(p("d.gz").encode(), b"w")
    assert lib.gzbuffer(f, 64) == 0
    longarg = b"x" * 5000
    assert lib.gzprintf(f, b"[%s]\n", longarg) == 0             # does not fit the buffer: nothing written (gz.rs:2729-2810)
    assert lib.gzprintf(f, b"%s", b"y" * 64) == 0 and lib.gzprintf(f, b"%s", b"") == 0
    assert lib.gzputc(f, ord("[")) == ord("[") and lib.gzputs(f, longarg) == 5000 and lib.gzprintf(f, b"%c%c", ord("]"), 10) == 2
    assert lib.gzputs(f, b"") == 0 and lib.gzwrite(f, src, 0) == 0
    assert lib.gzclose(f) == Z_OK
    f = lib.gzopen(p("d.gz").encode(), b"r")
    assert lib.gzbuffer(f, 64) == 0
    linebuf = C.create_string_buffer(6000)
    assert lib.gzgets(f, linebuf, 6000) and linebuf.value == b"[" + longarg + b"]\n"     # a line of many buffers
    assert not lib.gzgets(f, linebuf, 6000) and lib.gzeof(f) == 1                        # nothing left
    assert lib.gzgetc(f) == -1
    pushed = 0
    while lib.gzungetc(ord("a") + pushed % 26, f) >= 0:                                # 2 x 64 bytes of room
        pushed += 1
        assert pushed <= 128
    assert pushed == 128 and lib.gzeof(f) == 0
    err = C.c_int(0)
    assert lib.gzerror(f, C.byref(err)).endswith(b"out of room to push characters") and err.value == Z_DATA_ERROR
    lib.gzclearerr(f)
    back = C.create_string_buffer(200)
    assert lib.gzread(f, back, 200) == 0           # a fatal error drops what was buffered (gz.rs:467-500), the file is at its end
    assert lib.gzrewind(f) == 0
    for i in range(100):                           # within the room: comes back in reverse order of the pushes
        assert lib.gzungetc(ord("a") + i % 26, f) >= 0
    assert lib.gzread(f, back, 101) == 101
    assert back.raw[:101] == bytes(ord("a") + i % 26 for i in reversed(range(100))) + b"["
    assert lib.gzclose(f) == Z_OK

    # -- an empty file written and read; a missing file; a bad mode; a truncated file
    f = lib.gzopen(p("empty.gz").encode(), b"w")
    assert lib.gzclose(f) == Z_OK and gzip.open(p("empty.gz"), "rb").read() == b""
    assert _gz_read_all(lib, p("empty.gz")) == b""
    assert not lib.gzopen(p("nope.gz").encode(), b"r") and not lib.gzopen(p("x.gz").encode(), b"r+")
    assert not lib.gzopen(p("x.gz").encode(), b"rT") and not lib.gzdopen(-1, b"r")
    blob = open(p("a.gz"), "rb").read()
    with open(p("cut.gz"), "wb") as fh:
        fh.write(blob[:len(blob) // 2])
    f = lib.gzopen(p("cut.gz").encode(), b"r")
    big = C.create_string_buffer(len(expect) + 10)
    n = lib.gzread(f, big, len(expect) + 10)
    assert 0 < n < len(expect) and big.raw[:n] == expect[:n]
    err = C.c_int(0)
    msg = lib.gzerror(f, C.byref(err))
    assert err.value == Z_BUF_ERROR and msg.endswith(b"unexpected end of file"), (err.value, msg)
    lib.gzclearerr(f)
    assert lib.gzerror(f, C.byref(err)) == b"" and err.value == Z_OK
    assert lib.gzclose(f) in (Z_OK, Z_BUF_ERROR)
    with open(p("bad.gz"), "wb") as fh:
        fh.write(blob[:100] + bytes(200) + blob[300:])
    f = lib.gzopen(p("bad.gz").encode(), b"r")
    # zeros in the middle of Huffman-coded data may decode to valid (wrong) symbols, and to more output than the file had:
    # read on until the stream ends or fails -- the trailer's CRC catches what the decoder cannot
    for _ in range(64):
        n = lib.gzread(f, big, len(expect) + 10)
        msg = lib.gzerror(f, C.byref(err))
        if n <= 0 or err.value != Z_OK:
            break
    assert n == -1 or err.value == Z_DATA_ERROR, (n, err.value, msg)
    assert lib.gzclose(f) == Z_OK


MESSAGES = {
    "invalid_stored_block_length": b"invalid stored block lengths", "invalid_block_type": b"invalid block type",
    "too_many_length_or_distance_symbols": b"too many length or distance symbols", "invalid_code_lengths_set": b"invalid code lengths set",
    "invalid_bit_length_repeat_1": b"invalid bit length repeat", "invalid_bit_length_repeat_2": b"invalid bit length repeat",
    "invalid_code_missing_end_of_block": b"invalid code -- missing end-of-block", "invalid_literal_lengths_set": b"invalid literal/lengths set",
    "invalid_distances_set": b"invalid distances set", "invalid_distance_too_far_back": b"invalid distance too far back",
    "incorrect_data_check": b"incorrect data check", "incorrect_length_check": b"incorrect length check",
}


def golden_inflate_checks(lib, vectors, steps=(0, 1, 3, 17), check_messages=True):
    """the reference's own inflate vectors (tests/golden/inflate_vectors.json: hand-made bitstreams with their
    expected error, test-libz-rs-sys/src/inflate.rs:734-1030, and its test-data files) through inflate(), whole and
    in steps -- the verdict must not depend on how the input arrives (inflate.rs:2376-2457)"""
    import base64
    import zlib
    ver, zs = lib.zlibVersion(), C.sizeof(ZStream)
    wb = {0: -15, 1: 15, 2: 31, 3: 47}
    n = 0
    for v in vectors["bitstreams"] + vectors["files"]:
        blob = bytes.fromhex(v["input"]) if "input" in v else base64.b64decode(v["data_b64"])
        expect_ok = v.get("expect", "ok") == "ok"
        want = None
        if expect_ok:
            d = zlib.decompressobj(wb[v["wrap"]])
            want = d.decompress(blob)
        for step in (steps if len(blob) <= 600 else (0, max(700, len(blob) // 5))):   # long fixtures only in large pieces
            strm = ZStream()
            assert lib.inflateInit2_(C.byref(strm), wb[v["wrap"]], ver, zs) == Z_OK
            src = C.create_string_buffer(blob, len(blob) or 1)
            got = bytearray()
            rc = Z_OK
            pieces = [(0, len(blob))] if not step else [(a, min(step, len(blob) - a)) for a in range(0, len(blob), step)]
            for a, ln in pieces:
                rc = _feed(lib, strm, C.addressof(src) + a, ln, got)
                if rc not in (Z_OK, Z_BUF_ERROR):
                    break
            if rc in (Z_OK, Z_BUF_ERROR):     # the input is used up: what Z_FINISH says now is the verdict
                rc = _feed(lib, strm, C.addressof(src), 0, got, Z_FINISH)
            if expect_ok:
                assert rc == Z_STREAM_END and bytes(got) == want, (v["source"], step, rc, len(got))
            elif v["expect"] == "data_error":
                assert rc == Z_DATA_ERROR, (v["source"], step, rc)
                # the vector's name in the reference's test file is its message (test-libz-rs-sys/src/inflate.rs:734-1030)
                name = v["source"].split(":")[-1]
                want_msg = MESSAGES.get(name)
                if want_msg is not None and check_messages:
                    assert strm.msg == want_msg, (name, step, strm.msg)
            else:
                assert rc in (Z_BUF_ERROR, Z_DATA_ERROR), (v["source"], step, rc)
            assert lib.inflateEnd(C.byref(strm)) == Z_OK
            n += 1
    return n


def random_streaming_roundtrips(lib, o, rounds, seed, max_len=60000):
    """randomised drive of inflate(): streams from the system's zlib (random level, wrapper, flush points, sometimes
    a preset-free raw stream), delivered in random pieces into output buffers of random size, with random flush
    arguments; the result must be the data, the end must be reported exactly once, trailing bytes of the last piece
    must come back"""
    import random
    import zlib
    rnd = random.Random(seed)
    ver, zs = lib.zlibVersion(), C.sizeof(ZStream)
    for r in range(rounds):
        n = rnd.choice([0, 1, 10, 300, 5000, rnd.randrange(max_len)])
        data = o.gen_shard(rnd.randrange(8), n) if rnd.random() < 0.8 else bytes(rnd.randrange(256) for _ in range(min(n, 3000)))
        wbits = rnd.choice([15, 31, -15])
        co = zlib.compressobj(rnd.choice([0, 1, 6, 9]), zlib.DEFLATED, wbits)
        comp, at = b"", 0
        while at < len(data):
            k = rnd.randrange(1, len(data) + 1)
            comp += co.compress(data[at:at + k])
            if rnd.random() < 0.4:
                comp += co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))
            at += k
        comp += co.flush()
        junk = bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 0, 5])))
        blob = comp + junk
        strm = ZStream()
        init_bits = wbits if wbits != 31 or rnd.random() < 0.5 else 47
        assert lib.inflateInit2_(C.byref(strm), init_bits, ver, zs) == Z_OK
        src = C.create_string_buffer(blob, len(blob) or 1)
        out = bytearray()
        pos, rc, calls = 0, Z_OK, 0
        while rc != Z_STREAM_END:
            calls += 1
            assert calls < 100000, "no end in sight"
            piece = min(len(blob) - pos, rnd.choice([1, 2, 7, 100, 4096, 1 << 20]))
            room = rnd.choice([1, 3, 64, 1000, 70000])
            obuf = C.create_string_buffer(room)
            strm.next_in, strm.avail_in = C.addressof(src) + pos, piece
            strm.next_out, strm.avail_out = C.addressof(obuf), room
            rc = lib.inflate(C.byref(strm), rnd.choice([Z_NO_FLUSH, Z_SYNC_FLUSH, Z_NO_FLUSH, 5]))
            assert rc in (Z_OK, Z_STREAM_END, Z_BUF_ERROR), (r, rc, strm.msg)
            pos += piece - strm.avail_in
            out += obuf.raw[:room - strm.avail_out]
            if rc == Z_BUF_ERROR:
                assert piece == 0 and room - strm.avail_out == 0          # only a call that could do nothing
                assert pos < len(blob) or len(out) < len(data), "stuck with everything delivered"
        assert bytes(out) == data, (r, len(out), len(data))
        assert strm.total_out == len(data)
        assert len(blob) - pos <= len(junk) and strm.total_in == pos
        assert lib.inflateEnd(C.byref(strm)) == Z_OK
    return rounds


def config_matrix_roundtrips(lib, o, rounds, seed, max_len=70000):
    """the property of test-libz-rs-sys/src/end_to_end.rs:5-85 over DeflateConfig::arbitrary (zlib-rs/src/deflate.rs:193-219:
    level 0-9, windowBits 9..15 / 25..31 / -15..-9, memLevel 1-9, the five strategies): whatever the configuration, an
    inflater that allocates ONLY the announced window (the system's zlib opened with the same windowBits) reads the data
    back, and so does this library's own inflate().  A back-reference farther than 2^windowBits - 262 would fail the first."""
    import random
    import zlib
    rnd = random.Random(seed)
    worst = 0
    for r in range(rounds):
        n = rnd.choice([0, 1, 600, 5000, rnd.randrange(max_len), rnd.randrange(max_len)])
        data = o.gen_shard(rnd.randrange(8), n) if rnd.random() < 0.85 else bytes(rnd.randrange(4) for _ in range(min(n, 4000)))
        w = rnd.randrange(9, 16)
        wbits = rnd.choice([w, w + 16, -w])
        level, strategy, mem_level = rnd.randrange(0, 10), rnd.randrange(0, 5), rnd.randrange(1, 10)
        comp = deflate_stream(lib, data, level=level, wbits=wbits, chunk_in=rnd.choice([None, None, 3000, 20000]),
                              chunk_out=rnd.choice([100, 4096, 200000]), flush_every=rnd.choice([None, 2]), strategy=strategy,
                              mem_level=mem_level)
        cfg = (r, n, level, wbits, mem_level, strategy)
        d = zlib.decompressobj(wbits)
        assert d.decompress(comp) == data and d.eof and not d.unused_data, cfg
        rc, out, unused = inflate_stream(lib, comp, wbits, chunk_in=rnd.choice([1 << 30, 1000]), chunk_out=rnd.choice([8192, 100000]))
        assert rc == Z_STREAM_END and out == data and unused == 0, cfg
        if wbits > 0 and wbits < 16:   # the zlib header announces the window (CINFO = windowBits - 8, deflate.rs:1572-1589)
            assert (comp[0] >> 4) + 8 == w and ((comp[0] << 8) | comp[1]) % 31 == 0, cfg
        if n:
            worst = max(worst, len(comp) - n)
    # windowBits 8 is accepted for the zlib wrapper only and means 9 (deflate.rs:293-301)
    data = o.gen_shard(1, 9000)
    comp = deflate_stream(lib, data, level=6, wbits=8)
    assert comp[0] == 0x18 and zlib.decompressobj(9).decompress(comp) == data
    strm = ZStream()
    for bad in (-8, 24, 7, 16, 32, -16):
        assert lib.deflateInit2_(C.byref(strm), 6, 8, bad, 8, 0, lib.zlibVersion(), C.sizeof(ZStream)) == Z_STREAM_ERROR, bad
    return worst


def misc_symbol_checks(lib, o):
    """the entry points nothing else drives: caller-supplied allocators (the fault-injecting zalloc of
    test-libz-rs-sys/src/inflate.rs:31-160 / zlib-rs/src/deflate.rs:3428-3444), deflateBound as a guarantee,
    deflateParams / deflateTune / *ResetKeep / inflateReset2, the _z one-shots, the combine operators, get_crc_table"""
    import random
    import zlib
    ver, zs = lib.zlibVersion(), C.sizeof(ZStream)
    P = C.POINTER(ZStream)
    # --- allocators: the state is obtained through zalloc and returned through zfree; a failing zalloc is Z_MEM_ERROR
    libc = C.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes, libc.free.argtypes = C.c_void_p, [C.c_size_t], [C.c_void_p]
    live, fail = {}, [False]
    peak, budget = [0], [None]   # most bytes live at once; allocations left before zalloc starts to fail (None: no limit)

    def za(opaque, items, size):
        if fail[0]:
            return None
        if budget[0] is not None:
            if budget[0] <= 0:
                return None
            budget[0] -= 1
        p = libc.malloc(items * size)
        live[p] = items * size
        peak[0] = max(peak[0], sum(live.values()))
        return p

    def zf(opaque, p):
        assert p in live, "zfree of a pointer zalloc never returned"
        del live[p]
        libc.free(p)
    za_c, zf_c = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint, C.c_uint)(za), C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)(zf)
    data = o.gen_shard(3, 30000)
    for failing in (True, False):
        fail[0] = failing
        s = ZStream()
        s.zalloc, s.zfree = C.cast(za_c, C.c_void_p), C.cast(zf_c, C.c_void_p)
        rc = lib.deflateInit2_(C.byref(s), 6, 8, 15, 8, 0, ver, zs)
        assert rc == (Z_MEM_ERROR if failing else Z_OK) and bool(s.state) == (not failing)
        if not failing:
            assert len(live) >= 1
            cap = lib.deflateBound(C.byref(s), len(data))
            src, dst = C.create_string_buffer(data, len(data)), C.create_string_buffer(cap)
            s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(data), C.addressof(dst), cap
            peak[0] = 0
            assert lib.deflate(C.byref(s), Z_FINISH) == Z_STREAM_END
            comp = dst.raw[:cap - s.avail_out]
            # the stream's buffers (the input it holds, the compressed bytes it queues) come from zalloc too, not only the state
            # (the reference's arena: zlib-rs/src/deflate.rs:252-439, allocate.rs:200-222)
            assert peak[0] >= len(data), peak[0]
            assert lib.deflateEnd(C.byref(s)) == Z_OK and not live
        s = ZStream()
        s.zalloc, s.zfree = C.cast(za_c, C.c_void_p), C.cast(zf_c, C.c_void_p)
        rc = lib.inflateInit2_(C.byref(s), 15, ver, zs)
        assert rc == (Z_MEM_ERROR if failing else Z_OK) and bool(s.state) == (not failing)
        if not failing:
            src, dst = C.create_string_buffer(comp, len(comp)), C.create_string_buffer(len(data))
            s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(comp), C.addressof(dst), len(data)
            peak[0] = 0
            assert lib.inflate(C.byref(s), Z_FINISH) == Z_STREAM_END and dst.raw == data
            assert peak[0] >= len(data), peak[0]
            assert lib.inflateEnd(C.byref(s)) == Z_OK and not live
            # a zalloc that starts to fail in the middle of the work: Z_MEM_ERROR, nothing leaked, nothing crashed
            # (not under the AddressSanitizer build of tools/emu_asan_check.sh: its preloaded runtime cannot intercept a C++ throw
            # from a library loaded later, and a failing zalloc surfaces as std::bad_alloc inside the library)
            for k in ([] if os.environ.get("ZMI_NO_ALLOC_FAULTS") else range(1, 6)):
                for direction in ("deflate", "inflate"):
                    s = ZStream()
                    s.zalloc, s.zfree = C.cast(za_c, C.c_void_p), C.cast(zf_c, C.c_void_p)
                    budget[0] = k
                    if direction == "deflate":
                        rc = lib.deflateInit2_(C.byref(s), 6, 8, 15, 8, 0, ver, zs)
                        if rc == Z_OK:
                            src, dst = C.create_string_buffer(data, len(data)), C.create_string_buffer(cap)
                            s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(data), C.addressof(dst), cap
                            rc = lib.deflate(C.byref(s), Z_FINISH)
                            assert rc in (Z_STREAM_END, Z_MEM_ERROR), rc
                            lib.deflateEnd(C.byref(s))
                    else:
                        rc = lib.inflateInit2_(C.byref(s), 15, ver, zs)
                        if rc == Z_OK:
                            src, dst = C.create_string_buffer(comp, len(comp)), C.create_string_buffer(len(data))
                            s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(comp), C.addressof(dst), len(data)
                            rc = lib.inflate(C.byref(s), Z_FINISH)
                            assert rc in (Z_STREAM_END, Z_MEM_ERROR), rc
                            lib.inflateEnd(C.byref(s))
                    budget[0] = None
                    assert not live, (direction, k, len(live))
    # --- deflateBound is a guarantee: one deflate(Z_FINISH) into that much room ends the stream, whatever the configuration
    rnd = random.Random(3)
    for r in range(60):
        n = rnd.choice([0, 1, 2, 5, 8, 9, 10, 100, 1000, 65535, 65536, 70000, rnd.randrange(150000)])
        raw = os.urandom(n) if r % 2 else bytes(rnd.randrange(200, 256) for _ in range(n))
        w = rnd.randrange(9, 16)
        wbits = rnd.choice([w, w + 16, -w])
        cfg = (n, rnd.randrange(10), wbits, rnd.randrange(1, 10), rnd.randrange(5))
        s = ZStream()
        assert lib.deflateInit2_(C.byref(s), cfg[1], 8, wbits, cfg[3], cfg[4], ver, zs) == Z_OK
        cap = lib.deflateBound(C.byref(s), n)
        src, dst = C.create_string_buffer(raw, n or 1), C.create_string_buffer(cap)
        s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), n, C.addressof(dst), cap
        assert lib.deflate(C.byref(s), Z_FINISH) == Z_STREAM_END, cfg
        assert zlib.decompressobj(wbits).decompress(dst.raw[:cap - s.avail_out]) == raw, cfg
        assert lib.deflateEnd(C.byref(s)) == Z_OK
    # ... and spelled out where the bound is tightest: incompressible input, levels 0 and 1, a window below 32 KiB --
    # the `n + n / 32 + ...` branch at level 0 -- in the geometry a single call gets (segments of 32 KiB, encoder pieces of 8 KiB,
    # a marker behind every piece)
    for lvl in (0, 1):
        for wbits in (9, 12, 14, -12, 9 + 16):
            for n in (8191, 8192, 8193, 32768, 65536 + 17, 300000):
                raw = os.urandom(n)
                s = ZStream()
                assert lib.deflateInit2_(C.byref(s), lvl, 8, wbits, 8, 0, ver, zs) == Z_OK
                cap = lib.deflateBound(C.byref(s), n)
                src, dst = C.create_string_buffer(raw, n), C.create_string_buffer(cap)
                s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), n, C.addressof(dst), cap
                assert lib.deflate(C.byref(s), Z_FINISH) == Z_STREAM_END, (lvl, wbits, n)
                assert zlib.decompressobj(wbits).decompress(dst.raw[:cap - s.avail_out]) == raw, (lvl, wbits, n)
                assert lib.deflateEnd(C.byref(s)) == Z_OK
    # --- deflateBound follows the stream's wrapper (deflate.rs:3193-3287; test-libz-rs-sys deflate.rs:620-675): the gzip header
    # fields handed in with deflateSetHeader count, a preset dictionary adds the 4-byte DICTID, no stream = the zlib wrapper
    lib.deflateSetHeader.argtypes = [P, C.POINTER(GzHeader)]
    n = len(data)
    comp_len = n + ((n + 7) >> 3) + ((n + 63) >> 6) + 5
    assert lib.deflateBound(None, n) == comp_len + 6
    for wb, lvl, want in ((15, 6, n + ((n + 7) >> 3) + 3 + 6), (-15, 6, n + ((n + 7) >> 3) + 3), (31, 6, n + ((n + 7) >> 3) + 3 + 18),
                          (12, 6, comp_len + 6), (-12, 0, n + (n >> 5) + (n >> 7) + (n >> 11) + 7), (28, 3, comp_len + 18)):
        s = ZStream()
        assert lib.deflateInit2_(C.byref(s), lvl, 8, wb, 8, 0, ver, zs) == Z_OK
        assert lib.deflateBound(C.byref(s), n) == want, (wb, lvl)
        if wb == 15:
            assert lib.deflateSetDictionary(C.byref(s), data[:100], 100) == Z_OK and lib.deflateBound(C.byref(s), n) == want + 4
        assert lib.deflateEnd(C.byref(s)) == Z_OK
    extra, name, comment = C.create_string_buffer(b"EXTRA-FIELD", 11), C.create_string_buffer(b"a-file-name.txt"), C.create_string_buffer(b"c" * 300)
    s = ZStream()
    assert lib.deflateInit2_(C.byref(s), 9, 8, 31, 3, 4, ver, zs) == Z_OK
    h = GzHeader(text=-1, time=1234567, os=7, extra=C.addressof(extra), extra_len=11, name=C.addressof(name), comment=C.addressof(comment), hcrc=-1)
    assert lib.deflateSetHeader(C.byref(s), C.byref(h)) == Z_OK
    cap = lib.deflateBound(C.byref(s), 5)
    assert cap == 5 + 1 + 1 + 3 + 18 + (2 + 11) + 16 + 301 + 2
    src, dst = C.create_string_buffer(b"hello", 5), C.create_string_buffer(cap)
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), 5, C.addressof(dst), cap
    assert lib.deflate(C.byref(s), Z_FINISH) == Z_STREAM_END
    gz = dst.raw[:cap - s.avail_out]
    assert gz[3] == 0x1F and gz[9] == 7          # FTEXT | FHCRC | FEXTRA | FNAME | FCOMMENT: text / hcrc are "!= 0" (deflate.rs test :678)
    assert zlib.decompress(gz, 31) == b"hello"
    assert lib.deflateEnd(C.byref(s)) == Z_OK
    # --- a one-shot into too little room is Z_BUF_ERROR at every size below the need (test_issue_455, test-libz-rs-sys deflate.rs:3078)
    lib.compress.argtypes = [C.c_void_p, C.POINTER(C.c_ulong), C.c_void_p, C.c_ulong]
    tiny = bytes([0, 0, 0, 0, 252, 0, 0, 62, 255, 255, 255, 42, 255, 255, 247, 255, 247, 255, 255, 255, 156, 70, 255, 255])
    dst, ln = C.create_string_buffer(64), C.c_ulong(64)
    assert lib.compress(dst, C.byref(ln), tiny, len(tiny)) == Z_OK and zlib.decompress(dst.raw[:ln.value]) == tiny
    need = ln.value
    for room in range(need + 1):
        ln = C.c_ulong(room)
        assert lib.compress(dst, C.byref(ln), tiny, len(tiny)) == (Z_OK if room >= need else Z_BUF_ERROR), room
    # --- deflateReset leaves nothing of the earlier stream behind (reset_deterministic, deflate.rs:3101-3190, issue 459)
    def one(strm, d):
        src, dst = C.create_string_buffer(d, len(d)), C.create_string_buffer(1024)
        strm.next_in, strm.avail_in, strm.next_out, strm.avail_out = C.addressof(src), len(d), C.addressof(dst), 1024
        assert lib.deflate(C.byref(strm), Z_FINISH) == Z_STREAM_END
        return dst.raw[:strm.total_out]
    da = b"\0AAAA\0AAAAAAAA"
    s = ZStream()
    assert lib.deflateInit2_(C.byref(s), 6, 8, 15, 8, 0, ver, zs) == Z_OK
    first = one(s, da)
    assert lib.deflateEnd(C.byref(s)) == Z_OK
    s = ZStream()
    assert lib.deflateInit2_(C.byref(s), 6, 8, 15, 8, 0, ver, zs) == Z_OK
    one(s, bytes([1]) * (len(da) + 1))
    assert lib.deflateReset(C.byref(s)) == Z_OK
    assert one(s, da) == first and zlib.decompress(first) == da
    assert lib.deflateEnd(C.byref(s)) == Z_OK
    # --- a fresh stream's first call without input is Z_OK (the wrapper's header goes out; old_flush starts at -2), the same
    # call again is Z_BUF_ERROR, a higher-ranked flush is work again, and the stream counts as started (deflate.rs:2505-2533, :728-743)
    for wb, hdr in ((15, 2), (31, 10), (-15, 0)):
        s = ZStream()
        assert lib.deflateInit2_(C.byref(s), 6, 8, wb, 8, 0, ver, zs) == Z_OK
        dst = C.create_string_buffer(100)
        s.next_in, s.avail_in, s.next_out, s.avail_out = None, 0, C.addressof(dst), 100
        assert lib.deflate(C.byref(s), Z_NO_FLUSH) == Z_OK and 100 - s.avail_out == hdr, wb
        assert lib.deflate(C.byref(s), Z_NO_FLUSH) == Z_BUF_ERROR
        assert lib.deflate(C.byref(s), Z_SYNC_FLUSH) == Z_OK and dst.raw[hdr:100 - s.avail_out][-4:] == b"\0\0\xff\xff"
        assert lib.deflate(C.byref(s), Z_SYNC_FLUSH) == Z_BUF_ERROR
        assert lib.deflateEnd(C.byref(s)) == Z_DATA_ERROR
    # --- short packets with a flush behind each (a protocol that flushes every message): every size from 1 byte up, incompressible
    # and text -- an 11-byte packet once overflowed its device slot (block + sync marker > compress_bound(11) = 16)
    for base_ in (os.urandom(80), b"abcdefghijklmnopqrstuvwxyz0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZ!?" * 2):
        s = ZStream()
        assert lib.deflateInit2_(C.byref(s), 6, 8, -15, 8, 0, ver, zs) == Z_OK
        rd = zlib.decompressobj(-15)
        for k in range(1, 70):
            pkt = base_[:k]
            src, dst = C.create_string_buffer(pkt, k), C.create_string_buffer(400)
            s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), k, C.addressof(dst), 400
            assert lib.deflate(C.byref(s), Z_SYNC_FLUSH if k % 3 else Z_FULL_FLUSH) == Z_OK, k
            assert rd.decompress(dst.raw[:400 - s.avail_out]) == pkt, k
        assert lib.deflateEnd(C.byref(s)) == Z_DATA_ERROR
    # --- deflateParams between two halves (lib.rs:1658, deflate.rs:441-497); deflateTune accepted; out-of-range refused
    lib.deflateParams.argtypes = [P, C.c_int, C.c_int]
    lib.deflateTune.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int]
    for (l0, s0), (l1, s1) in (((1, 0), (9, 0)), ((6, 0), (0, 0)), ((0, 0), (6, 0)), ((6, 0), (6, 3)), ((6, 2), (4, 1)), ((6, 0), (-1, 4))):
        s = ZStream()
        assert lib.deflateInit2_(C.byref(s), l0, 8, 15, 8, s0, ver, zs) == Z_OK
        assert lib.deflateTune(C.byref(s), 4, 8, 32, 64) == Z_OK
        assert lib.deflateParams(C.byref(s), 10, 0) == Z_STREAM_ERROR and lib.deflateParams(C.byref(s), 6, 5) == Z_STREAM_ERROR
        half = len(data) // 2
        src, dst = C.create_string_buffer(data, len(data)), C.create_string_buffer(len(data) + 1000)
        s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), half, C.addressof(dst), len(data) + 1000
        assert lib.deflate(C.byref(s), Z_NO_FLUSH) == Z_OK
        assert lib.deflateParams(C.byref(s), l1, s1) == Z_OK and s.avail_in == 0
        s.avail_in = len(data) - half
        assert lib.deflate(C.byref(s), Z_FINISH) == Z_STREAM_END
        assert zlib.decompress(dst.raw[:len(data) + 1000 - s.avail_out]) == data, (l0, s0, l1, s1)
        assert lib.deflateEnd(C.byref(s)) == Z_OK
    assert lib.deflateParams(None, 6, 0) == Z_STREAM_ERROR and lib.deflateTune(None, 1, 1, 1, 1) == Z_STREAM_ERROR
    # --- deflateResetKeep / inflateReset2: the stream object serves a second, differently wrapped, stream
    lib.deflateResetKeep.argtypes = [P]
    lib.inflateReset2.argtypes = [P, C.c_int]
    s = ZStream()
    assert lib.deflateInit2_(C.byref(s), 6, 8, 31, 8, 0, ver, zs) == Z_OK
    outs = []
    for part in (data[:1000], data[1000:9000]):
        src, dst = C.create_string_buffer(part, len(part)), C.create_string_buffer(len(part) + 100)
        s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(part), C.addressof(dst), len(part) + 100
        assert lib.deflate(C.byref(s), Z_FINISH) == Z_STREAM_END
        outs.append(dst.raw[:len(part) + 100 - s.avail_out])
        assert s.total_in == len(part)
        assert lib.deflateResetKeep(C.byref(s)) == Z_OK and s.total_in == 0 and s.total_out == 0
    assert lib.deflateEnd(C.byref(s)) == Z_OK
    assert zlib.decompress(outs[0], 31) == data[:1000] and zlib.decompress(outs[1], 31) == data[1000:9000]
    s = ZStream()
    assert lib.inflateInit2_(C.byref(s), 15, ver, zs) == Z_OK
    for wb, blob, want in ((31, outs[1], data[1000:9000]), (-15, zlib.compress(data, 6)[2:-4], data), (47, zlib.compress(data), data)):
        assert lib.inflateReset2(C.byref(s), wb) == Z_OK
        src, dst = C.create_string_buffer(blob, len(blob)), C.create_string_buffer(len(want))
        s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(blob), C.addressof(dst), len(want)
        assert lib.inflate(C.byref(s), Z_FINISH) == Z_STREAM_END and dst.raw == want, wb
    assert lib.inflateReset2(C.byref(s), 7) == Z_STREAM_ERROR and lib.inflateReset2(C.byref(s), 64) == Z_STREAM_ERROR
    lib.inflateUndermine.argtypes = [P, C.c_int]
    lib.inflateCodesUsed.argtypes, lib.inflateCodesUsed.restype = [P], C.c_ulong
    assert lib.inflateUndermine(C.byref(s), 1) in (Z_OK, Z_DATA_ERROR) and lib.inflateUndermine(None, 1) == Z_STREAM_ERROR
    # inflateCodesUsed (libz-rs-sys/src/lib.rs:1252, zlib-rs/src/inflate.rs:2372): the last stream above was a dynamic-block
    # stream -> the tables' entry count: at least the two root tables (2^9 + 2^8 here), never more than ENOUGH
    # (zlib-rs/src/lib.rs:88-102); zero again after a reset, still zero after a stored-only stream, (ulong)-1 for no stream
    used = lib.inflateCodesUsed(C.byref(s))
    assert 512 + 256 <= used <= 852 + 400 < 1332 + 592 + 1, used
    assert lib.inflateReset2(C.byref(s), 15) == Z_OK and lib.inflateCodesUsed(C.byref(s)) == 0
    blob0 = zlib.compress(data[:3000], 0)
    src, dst = C.create_string_buffer(blob0, len(blob0)), C.create_string_buffer(3000)
    s.next_in, s.avail_in, s.next_out, s.avail_out = C.addressof(src), len(blob0), C.addressof(dst), 3000
    assert lib.inflate(C.byref(s), Z_FINISH) == Z_STREAM_END and dst.raw == data[:3000] and lib.inflateCodesUsed(C.byref(s)) == 0
    assert lib.inflateCodesUsed(None) == C.c_ulong(-1).value
    assert lib.inflateEnd(C.byref(s)) == Z_OK
    # --- the _z one-shots (lib.rs:1379-1561, :433-583)
    zsz = C.c_size_t
    lib.compressBound_z.restype, lib.compressBound_z.argtypes = zsz, [zsz]
    lib.compress_z.argtypes = [C.c_void_p, C.POINTER(zsz), C.c_void_p, zsz]
    lib.compress2_z.argtypes = [C.c_void_p, C.POINTER(zsz), C.c_void_p, zsz, C.c_int]
    lib.uncompress_z.argtypes = [C.c_void_p, C.POINTER(zsz), C.c_void_p, zsz]
    lib.uncompress2_z.argtypes = [C.c_void_p, C.POINTER(zsz), C.c_void_p, C.POINTER(zsz)]
    for n, want in ((1024, 1161), (4096, 4617), (65536, 73737)):     # doctest of compress_bound, deflate.rs:2966-2968
        assert lib.compressBound_z(n) == want == lib.compressBound(n)
    cap = zsz(lib.compressBound_z(len(data)))
    dst = C.create_string_buffer(cap.value)
    assert lib.compress_z(dst, C.byref(cap), data, len(data)) == Z_OK and zlib.decompress(dst.raw[:cap.value]) == data
    cap9 = zsz(lib.compressBound_z(len(data)))
    dst9 = C.create_string_buffer(cap9.value)
    assert lib.compress2_z(dst9, C.byref(cap9), data, len(data), 9) == Z_OK and zlib.decompress(dst9.raw[:cap9.value]) == data
    small = zsz(10)
    assert lib.compress2_z(dst9, C.byref(small), data, len(data), 6) == Z_BUF_ERROR
    assert lib.compress2_z(dst9, C.byref(cap9), data, len(data), 11) == Z_STREAM_ERROR
    blob = dst.raw[:cap.value] + b"xyz"
    ocap, icap = zsz(len(data)), zsz(len(blob))
    back = C.create_string_buffer(len(data))
    assert lib.uncompress2_z(back, C.byref(ocap), blob, C.byref(icap)) == Z_OK and back.raw == data and icap.value == cap.value
    ocap = zsz(len(data))
    assert lib.uncompress_z(back, C.byref(ocap), blob, cap.value) == Z_OK and ocap.value == len(data)
    # --- uncompress edge cases (test-libz-rs-sys/src/inflate.rs:1337-1358; zlib-rs/src/inflate.rs:195-284): no input is a data
    # error; no room is a data error unless the stream is complete and empty; *destLen reports what was written either way
    hello = zlib.compress(b"Hello World!\n", 6)
    one = C.create_string_buffer(1)
    for room, blob, want_rc, want_len in ((0, b"", Z_DATA_ERROR, 0), (1, b"", Z_DATA_ERROR, 0), (0, hello, Z_DATA_ERROR, 0),
                                          (0, zlib.compress(b""), Z_OK, 0), (1, hello, Z_BUF_ERROR, 1), (1, hello[:-6], Z_BUF_ERROR, 1)):
        ocap = zsz(room)
        src = C.create_string_buffer(blob, len(blob) or 1)
        assert lib.uncompress_z(one, C.byref(ocap), src, len(blob)) == want_rc, (room, len(blob))
        assert ocap.value == want_len, (room, len(blob), ocap.value)
    big = C.create_string_buffer(64)
    ocap = zsz(64)
    assert lib.uncompress_z(big, C.byref(ocap), hello[:-6], len(hello) - 6) == Z_DATA_ERROR and big.raw[:ocap.value] == b"Hello World!\n"[:ocap.value]
    assert lib.uncompress_z(None, C.byref(ocap), hello, len(hello)) == Z_STREAM_ERROR
    assert lib.uncompress_z(big, None, hello, len(hello)) == Z_STREAM_ERROR
    # --- checksum variants and the combine operators (lib.rs:149-412; crc32/combine.rs)
    for f in ("adler32_z", "crc32_z"):
        getattr(lib, f).restype, getattr(lib, f).argtypes = C.c_ulong, [C.c_ulong, C.c_void_p, zsz]
    assert lib.crc32_z(0, bytes([1, 2, 3]), 3) == 1438416925                      # lib.rs:146,179
    assert lib.adler32_z(1, data, len(data)) == zlib.adler32(data) and lib.crc32_z(0, data, len(data)) == zlib.crc32(data)
    assert lib.adler32_z(7, None, 0) == 1 and lib.crc32_z(7, None, 0) == 0           # NULL buffer = the initial value
    for f, a in (("adler32_combine64", C.c_longlong), ("crc32_combine64", C.c_longlong)):
        getattr(lib, f).restype, getattr(lib, f).argtypes = C.c_ulong, [C.c_ulong, C.c_ulong, a]
    lib.crc32_combine_gen.restype, lib.crc32_combine_gen.argtypes = C.c_ulong, [C.c_long]
    lib.crc32_combine_gen64.restype, lib.crc32_combine_gen64.argtypes = C.c_ulong, [C.c_longlong]
    lib.crc32_combine_op.restype, lib.crc32_combine_op.argtypes = C.c_ulong, [C.c_ulong, C.c_ulong, C.c_ulong]
    for k in (0, 1, 777, len(data) - 1, len(data)):
        a, b = data[:k], data[k:]
        assert lib.adler32_combine64(zlib.adler32(a), zlib.adler32(b), len(b)) == zlib.adler32(data)
        assert lib.crc32_combine64(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(data)
        op = lib.crc32_combine_gen(len(b))
        assert op == lib.crc32_combine_gen64(len(b))
        assert lib.crc32_combine_op(zlib.crc32(a), zlib.crc32(b), op) == zlib.crc32(data)
    lib.get_crc_table.restype = C.POINTER(C.c_uint32)
    t = lib.get_crc_table()
    for i in (0, 1, 2, 128, 255):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0xEDB88320 if c & 1 else 0)
        assert t[i] == c
    lib.zlibCompileFlags.restype = C.c_ulong
    fl = lib.zlibCompileFlags()
    assert fl == (1 | 2 << 2 | 2 << 4 | 2 << 6)      # uInt 32 bit; uLong, pointers, z_off_t 64 bit; no feature bits (lib.rs:2219-2270)


def threaded_roundtrips(lib, o, threads=4, rounds=5):
    """different streams on different threads at the same time (SURVEY 8b threading: no shared stream state; the reference
    declares its streams Send + Sync, zlib-rs/src/deflate.rs:53-54).  ctypes releases the GIL inside every call, so the
    calls really overlap; each thread drives its own deflate and inflate streams in small pieces."""
    import threading
    import zlib
    errors = []

    def work(t):
        try:
            for r in range(rounds):
                data = o.gen_shard((t + r) % 8, 20000 + 7000 * t + 13 * r)
                wbits = (15, 31, -15)[(t + r) % 3]
                comp = deflate_stream(lib, data, level=(1, 6, 9)[r % 3], wbits=wbits, chunk_in=5000, chunk_out=3000, flush_every=2)
                assert zlib.decompressobj(wbits).decompress(comp) == data
                rc, out, unused = inflate_stream(lib, comp, wbits, chunk_in=777, chunk_out=4096)
                assert rc == Z_STREAM_END and out == data and unused == 0
                assert lib.crc32(0, data, len(data)) == zlib.crc32(data)
        except BaseException as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def handback_checks(lib, o, tmpdir, size=150000):
    """ADVICE r01 (medium x2): bytes behind the end of a stream come back to the caller exactly, also when the decode
    paused on the way (output queue limit, Z_NEED_DICT), and a gz reader drains a paused member without swallowing the
    next one.  Needs ZMI_ABI_QUEUE set small by the caller (environment is read per call)."""
    import gzip
    import zlib
    data = o.gen_shard(0, size) + o.gen_shard(3, size // 2)
    tail = bytes(range(25))
    co = zlib.compressobj(6, zlib.DEFLATED, 15)
    comp = b"".join(co.compress(data[i:i + 20000]) + co.flush(zlib.Z_FULL_FLUSH) for i in range(0, len(data), 20000)) + co.flush()
    # everything offered at once, small output rooms: the end is found many calls after the input was first seen
    for room in (4096, 50000):
        rc, out, unused = inflate_stream(lib, comp + tail, 15, chunk_in=1 << 30, chunk_out=room)
        assert rc == Z_STREAM_END and out == data and unused == len(tail), (room, rc, len(out), unused)
    # total_in follows what was really consumed
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), 15, lib.zlibVersion(), C.sizeof(ZStream)) == Z_OK
    src = C.create_string_buffer(comp + tail, len(comp) + len(tail))
    obuf = C.create_string_buffer(1 << 20)
    strm.next_in, strm.avail_in = C.addressof(src), len(comp) + len(tail)
    got = bytearray()
    for _ in range(10000):
        strm.next_out, strm.avail_out = C.addressof(obuf), 3000
        rc = lib.inflate(C.byref(strm), Z_NO_FLUSH)
        got += obuf.raw[:3000 - strm.avail_out]
        assert strm.next_in == C.addressof(src) + strm.total_in
        assert strm.total_in + strm.avail_in == len(comp) + len(tail)
        if rc != Z_OK:
            break
    assert rc == Z_STREAM_END and bytes(got) == data and strm.avail_in == len(tail) and strm.total_in == len(comp)
    lib.inflateEnd(C.byref(strm))

    # Z_NEED_DICT in the middle: the bytes behind the stream still come back
    zdict = data[:5000]
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, 0, zdict)
    dcomp = co.compress(data[3000:60000]) + co.flush()
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), 15, lib.zlibVersion(), C.sizeof(ZStream)) == Z_OK
    src = C.create_string_buffer(dcomp + tail, len(dcomp) + len(tail))
    strm.next_in, strm.avail_in = C.addressof(src), len(dcomp) + len(tail)
    strm.next_out, strm.avail_out = C.addressof(obuf), 1 << 20
    assert lib.inflate(C.byref(strm), Z_NO_FLUSH) == 2                      # Z_NEED_DICT
    assert strm.total_in == 6 and strm.avail_in == len(dcomp) + len(tail) - 6   # header + DICTID, nothing else
    assert lib.inflateSetDictionary(C.byref(strm), zdict, len(zdict)) == Z_OK
    got = bytearray()
    for _ in range(10000):
        strm.next_out, strm.avail_out = C.addressof(obuf), 1 << 20
        rc = lib.inflate(C.byref(strm), Z_NO_FLUSH)
        got += obuf.raw[:(1 << 20) - strm.avail_out]
        if rc != Z_OK:
            break
    assert rc == Z_STREAM_END and bytes(got) == data[3000:60000] and strm.avail_in == len(tail), (rc, strm.avail_in)
    lib.inflateEnd(C.byref(strm))

    # gz reader: two members, the first expands far beyond the queue limit; small gzread()s
    _bind_gz(lib)
    path = os.path.join(str(tmpdir), "two_members.gz")
    second = o.gen_shard(5, 30000)
    with open(path, "wb") as fh:
        fh.write(gzip.compress(bytes(size) + data, 6))
        fh.write(gzip.compress(second, 6))
    assert _gz_read_all(lib, path, chunk=1000) == bytes(size) + data + second
    assert _gz_read_all(lib, path, chunk=1 << 20) == bytes(size) + data + second


def multi_member_reader_checks(lib, o, member_bytes=(70000, 25000, 120000), piece=8192):
    """the loop of Python's gzip reader (_GzipReader: read 8 KiB, decompress, at the end of a member carry `unused_data` into the next
    one) on a file of three gzip members, through the DEFAULT mode of inflate(): every member ends in the call that delivers its last
    byte, what that call did not consume begins exactly at the next member's magic, nothing is swallowed or lost."""
    import gzip
    members = [o.gen_shard(i, n) for i, n in enumerate(member_bytes)]
    blob = b"".join(gzip.compress(m, 6) for m in members)
    src = C.create_string_buffer(blob, len(blob))
    obuf = C.create_string_buffer(1 << 18)
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), 31, lib.zlibVersion(), C.sizeof(ZStream)) == Z_OK
    pos, got, outs, ends = 0, bytearray(), [], 0
    pending = 0                                   # bytes of the last piece the finished member left behind (unused_data)
    while True:
        if pending == 0:
            if pos >= len(blob):
                break
            n = min(piece, len(blob) - pos)
            strm.next_in, strm.avail_in = C.addressof(src) + pos, n
            pos += n
        else:
            strm.next_in, strm.avail_in = C.addressof(src) + pos - pending, pending
            pending = 0
        while True:
            strm.next_out, strm.avail_out = C.addressof(obuf), len(obuf)
            rc = lib.inflate(C.byref(strm), Z_NO_FLUSH)
            got += obuf.raw[:len(obuf) - strm.avail_out]
            assert rc in (Z_OK, Z_STREAM_END, Z_BUF_ERROR), rc
            if rc == Z_STREAM_END:
                ends += 1
                outs.append(bytes(got))
                got = bytearray()
                pending = strm.avail_in
                at = pos - pending
                if at < len(blob):               # the unconsumed bytes start with the next member's header
                    assert blob[at:at + 2] == b"\x1f\x8b", (ends, at, blob[at:at + 4])
                assert lib.inflateReset(C.byref(strm)) == Z_OK
                break
            if strm.avail_in == 0 and strm.avail_out != 0:
                break
    assert lib.inflateEnd(C.byref(strm)) == Z_OK
    assert ends == len(members) and outs == members, (ends, [len(x) for x in outs])
    return ends


def block_stop_checks(lib, syslib, data):
    """inflate(Z_BLOCK) / inflate(Z_TREES): the calls of the reference's tests (test-libz-rs-sys/src/inflate.rs:640-676 runs every
    stream through a Z_TREES loop, :2036-2078 logs avail_in / avail_out / data_type after every Z_BLOCK call and compares the
    log with zlib-ng's).  Here the same log is compared with the system zlib's, call by call: return code, input left,
    output produced and data_type (unused bits | 64 last block | 128 at a block boundary | 256 behind a block header,
    zlib-rs/src/inflate.rs:1856-1873)."""
    import zlib
    Z_BLOCK, Z_TREES = 5, 6
    bind(syslib)
    streams = []
    for level, wbits in ((6, 15), (1, 31), (9, -15), (0, 15), (6, -15)):
        co = zlib.compressobj(level, zlib.DEFLATED, wbits)
        streams.append((wbits, co.compress(data[:30000]) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(data[30000:]) + co.flush()))
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_FIXED)
    streams.append((-15, co.compress(data[:5000]) + co.flush()))
    own = deflate_stream(lib, data, level=6, wbits=15)            # this library's own stream: many blocks, byte-aligned pieces
    streams.append((15, own))
    streams.append((15, zlib.compress(b"")))
    n_calls = 0
    for wbits, comp in streams:
        for flush in (Z_BLOCK, Z_TREES):
            logs = []
            for L in (syslib, lib):
                strm = ZStream()
                assert L.inflateInit2_(C.byref(strm), wbits, L.zlibVersion(), C.sizeof(ZStream)) == Z_OK
                src = C.create_string_buffer(comp + b"TAIL", len(comp) + 4)
                out = C.create_string_buffer(len(data) + 1024)
                strm.next_in, strm.avail_in = C.addressof(src), len(comp) + 4
                got = bytearray()
                log = []
                for _ in range(100000):
                    strm.next_out, strm.avail_out = C.addressof(out), len(out)
                    rc = L.inflate(C.byref(strm), flush)
                    got += out.raw[:len(out) - strm.avail_out]
                    log.append((rc, strm.avail_in, len(out) - strm.avail_out, strm.data_type))
                    # (a call that only moves from one stop to the next without input or output is Z_BUF_ERROR: not fatal)
                    if rc not in (Z_OK, Z_BUF_ERROR) or (rc == Z_BUF_ERROR and len(log) > 1 and log[-2][0] == Z_BUF_ERROR):
                        break
                assert rc == Z_STREAM_END and bytes(got) == data[:len(got)] and strm.avail_in == 4, (wbits, flush, rc, strm.avail_in)
                assert L.inflateEnd(C.byref(strm)) == Z_OK
                logs.append(log)
            assert logs[0] == logs[1], (wbits, flush, [(i, a, b) for i, (a, b) in enumerate(zip(*logs)) if a != b][:3], len(logs[0]), len(logs[1]))
            n_calls += len(logs[0])
    # a small output buffer: the stop is reported by the call that hands out the block's last byte
    wbits, comp = streams[0]
    strm = ZStream()
    assert lib.inflateInit2_(C.byref(strm), wbits, lib.zlibVersion(), C.sizeof(ZStream)) == Z_OK
    src = C.create_string_buffer(comp, len(comp))
    out = C.create_string_buffer(777)
    strm.next_in, strm.avail_in = C.addressof(src), len(comp)
    got, stops = bytearray(), 0
    for _ in range(100000):
        strm.next_out, strm.avail_out = C.addressof(out), len(out)
        rc = lib.inflate(C.byref(strm), Z_BLOCK)
        got += out.raw[:len(out) - strm.avail_out]
        stops += (strm.data_type & 128) != 0
        if rc not in (Z_OK, Z_BUF_ERROR):
            break
    assert rc == Z_STREAM_END and bytes(got) == data and stops >= 3, (rc, len(got), stops)
    assert lib.inflateEnd(C.byref(strm)) == Z_OK
    return n_calls


def flush_point_stream_checks(lib, o, seeds, big=False):
    """streams WITH flush points through inflate(): the stream ABI cuts what is buffered at the 00 00 FF FF markers and decodes
    the pieces side by side (zmi_inflate_split) -- the caller must see the serial decode's bytes and codes.  Random piece sizes
    and flush kinds (sync / full / none, so that some 'markers' sit inside stored blocks of data that holds the four bytes),
    raw / zlib / gzip wrappers, random feeding chunk and room sizes, a truncated and a corrupted variant of every stream."""
    import random
    import zlib
    n_cases = 0
    # a stored block whose end has not arrived: the bytes that are there come out (Mode::CopyBlock, inflate.rs:1374-1394)
    raw = o.gen_shard(5, 150000)
    co = zlib.compressobj(0, zlib.DEFLATED, -15)
    st_stream = co.compress(raw) + co.flush()
    for cut in (len(st_stream) - 1000, 70000, 65540, 10):
        rc, back, unused = inflate_stream(lib, st_stream[:cut], -15, chunk_in=1 << 30, chunk_out=1 << 20)
        want = zlib.decompressobj(-15).decompress(st_stream[:cut])
        assert rc in (Z_OK, Z_BUF_ERROR) and back == want, ("stored, cut", cut, rc, len(back), len(want))
    for seed in seeds:
        rng = random.Random(seed)
        total = rng.randrange(300000, 1200000 if big else 500000)
        parts = []
        while sum(len(x) for x in parts) < total:
            k = rng.randrange(8)
            ln = rng.randrange(2000, 300000 if big else 60000)
            if k == 7:
                parts.append((b"\x00\x00\xff\xff" + bytes(rng.randrange(256) for _ in range(40))) * (ln // 44))   # marker look-alikes
            else:
                parts.append(o.gen_shard(k, ln))
        data = b"".join(parts)
        wbits = rng.choice((-15, 15, 31))
        co = zlib.compressobj(rng.choice((0, 1, 6, 9)) if rng.random() < 0.3 else 6, zlib.DEFLATED, wbits)
        comp = b""
        pos = 0
        while pos < len(data):
            step = rng.randrange(4000, 200000 if big else 40000)
            comp += co.compress(data[pos:pos + step])
            pos += step
            r = rng.random()
            if r < 0.6:
                comp += co.flush(zlib.Z_SYNC_FLUSH)
            elif r < 0.8:
                comp += co.flush(zlib.Z_FULL_FLUSH)
        comp += co.flush()
        chunk_in = rng.choice((1 << 30, 1 << 22, 300000, 70001))
        chunk_out = rng.choice((1 << 22, 65536, 8192))
        rc, back, unused = inflate_stream(lib, comp, wbits, chunk_in=chunk_in, chunk_out=chunk_out)
        assert rc == Z_STREAM_END and back == data and unused == 0, (seed, rc, len(back), len(data), unused)
        # truncated: everything in front of the cut comes out, no error
        cut = rng.randrange(len(comp) // 2, len(comp) - 8)
        rc, back, unused = inflate_stream(lib, comp[:cut], wbits, chunk_in=chunk_in, chunk_out=chunk_out)
        want = zlib.decompressobj(wbits).decompress(comp[:cut])
        assert rc in (Z_OK, Z_BUF_ERROR) and back == want, (seed, "truncated", rc, len(back), len(want))
        # corrupted: the system zlib's verdict and its valid prefix
        bad = bytearray(comp)
        at = rng.randrange(len(comp) // 3, len(comp) - 8)
        bad[at] ^= 1 << rng.randrange(8)
        d = zlib.decompressobj(wbits)
        try:
            want = d.decompress(bytes(bad))
            failed = False
        except zlib.error:
            failed = True
        rc, back, unused = inflate_stream(lib, bytes(bad), wbits, chunk_in=chunk_in, chunk_out=chunk_out)
        if failed:
            assert rc == Z_DATA_ERROR, (seed, "corrupt", rc)
        else:
            assert back == want and rc in (Z_OK, Z_STREAM_END, Z_BUF_ERROR), (seed, "corrupt-but-valid", rc)
        n_cases += 3
    return n_cases


def uncompress_large_checks(lib, o, n_each):
    """uncompress() on the parallel-blocks path (zlib_abi.hip uncompress2_z -> zmi_inflate_blocks): what does not come back as a
    plain complete stream -- short room, a wrong Adler-32, corrupt data, truncation -- reports what the reference reports
    (zlib-rs/src/inflate.rs:202-284); shared by the emulator and the MI355X test"""
    import zlib
    data = b"".join(o.gen_shard(c, n_each) for c in (0, 3, 4, 6))
    z = zlib.compress(data, 6)

    def run(blob, room):
        dst = C.create_string_buffer(max(1, room))
        dl = C.c_ulong(room)
        rc = lib.uncompress(dst, C.byref(dl), blob, len(blob))
        return rc, dst.raw[:dl.value]

    rc, out = run(z, len(data))
    assert rc == Z_OK and out == data
    rc, out = run(z + b"trailing garbage", len(data) + 100)          # bytes behind the stream are not an error for uncompress
    assert rc == Z_OK and out == data
    rc, out = run(z, len(data) - 1)                                  # one byte short of room
    assert rc == Z_BUF_ERROR
    bad = bytearray(z); bad[-1] ^= 1                                  # wrong check value
    rc, out = run(bytes(bad), len(data))
    assert rc == Z_DATA_ERROR
    bad = bytearray(z); bad[len(z) // 2] ^= 0x20                      # corrupt data: whatever the system zlib makes of it
    try:
        zlib.decompress(bytes(bad)); ok = True
    except zlib.error:
        ok = False
    rc, out = run(bytes(bad), len(data))
    assert (rc == Z_OK) == ok and (ok or rc == Z_DATA_ERROR), rc
    rc, out = run(z[:len(z) * 2 // 3], len(data))                    # truncated
    assert rc == Z_DATA_ERROR
    return 6
