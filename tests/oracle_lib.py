"""Loader for the CPU oracle (oracle/zoracle.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_o = None


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        u8p = C.c_char_p
        lib.zo_adler32.restype = C.c_uint32
        lib.zo_adler32.argtypes = [C.c_uint32, u8p, C.c_size_t]
        lib.zo_crc32.restype = C.c_uint32
        lib.zo_crc32.argtypes = [C.c_uint32, u8p, C.c_size_t]
        lib.zo_adler32_combine.restype = C.c_uint32
        lib.zo_adler32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        lib.zo_crc32_combine.restype = C.c_uint32
        lib.zo_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        lib.zo_compress_bound.restype = C.c_uint64
        lib.zo_compress_bound.argtypes = [C.c_uint64, C.c_int]
        lib.zo_gen_shard.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.zo_prng_bytes.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t, C.c_uint32]
        lib.zo_inflate.restype = C.c_int
        lib.zo_inflate.argtypes = [u8p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_char_p)]
        if hasattr(lib, "zo_deflate"):
            lib.zo_deflate.restype = C.c_int
            lib.zo_deflate.argtypes = [u8p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_size_t)]
            lib.zo_deflate2.restype = C.c_int
            lib.zo_deflate2.argtypes = [u8p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_size_t)]

    def adler32(self, data, start=1):
        return self.lib.zo_adler32(start, data, len(data))

    def crc32(self, data, start=0):
        return self.lib.zo_crc32(start, data, len(data))

    def gen_shard(self, shard, nbytes=1 << 20, seed=0x5A4C4942):
        buf = C.create_string_buffer(nbytes)
        self.lib.zo_gen_shard(seed, shard, nbytes, buf)
        return buf.raw

    def prng_bytes(self, seed, n, step):
        buf = C.create_string_buffer(n)
        self.lib.zo_prng_bytes(seed, buf, n, step)
        return buf.raw

    def inflate(self, data, cap, wrap=1):
        """-> (rc, output bytes, input bytes used, message or None)"""
        out = C.create_string_buffer(max(cap, 1))
        olen, used, msg = C.c_size_t(0), C.c_size_t(0), C.c_char_p()
        rc = self.lib.zo_inflate(data, len(data), out, cap, wrap, C.byref(olen), C.byref(used), C.byref(msg))
        return rc, out.raw[:olen.value], used.value, (msg.value.decode() if msg.value else None)

    def deflate(self, data, level=6, wrap=1, strategy=0, mem_level=8, wbits=15):
        cap = int(self.lib.zo_compress_bound(len(data), wrap)) + 64 + 5 * (len(data) // 256 + 1)
        out = C.create_string_buffer(cap)
        olen = C.c_size_t(0)
        rc = self.lib.zo_deflate2(data, len(data), out, cap, level, wrap, strategy, mem_level, wbits, C.byref(olen))
        return rc, out.raw[:olen.value]


def load(rebuild=True):
    global _o
    if _o is None:
        d = os.path.join(ROOT, "oracle")
        so = os.path.join(d, "libzoracle.so")
        if rebuild and os.path.exists(os.path.join(d, "Makefile")) and _have_cc():
            subprocess.run(["make", "-s", "-C", d], check=True)
        _o = Oracle(C.CDLL(so))
    return _o


def _have_cc():
    from shutil import which
    return which("gcc") is not None
