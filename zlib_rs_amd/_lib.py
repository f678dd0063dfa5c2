"""ctypes binding of libzmi355.so (the C ABI in include/zmi355.h).  Fails loudly when the HIP
library is missing -- there is no CPU fallback in this package."""
import ctypes as C
import os

from ._build import LIB

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(this package has no CPU fallback)" % LIB)
    L = C.CDLL(LIB)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.zmi_version.restype = C.c_char_p
    L.zmi_last_error.restype = C.c_char_p
    L.zmi_ctx_create.argtypes = [C.POINTER(vp), i32]
    L.zmi_ctx_destroy.argtypes = [vp]
    L.zmi_ctx_set_scratch_limit.argtypes = [vp, u64]
    L.zmi_ctx_set_inflate_out_limit.argtypes = [vp, u64]
    L.zmi_deflate_bound.restype = u64
    L.zmi_deflate_bound.argtypes = [u64, i32]
    L.zmi_deflate_batch_dev.argtypes = [vp, vp, vp, vp, u32, u32, i32, i32, i32, vp, u64, vp, vp, vp]
    L.zmi_inflate_batch_dev.argtypes = [vp, vp, vp, vp, u32, i32, vp, vp, vp, vp, vp, vp]
    L.zmi_checksum_batch_dev.argtypes = [vp, vp, vp, vp, u32, i32, vp, vp, vp]
    L.zmi_gen_shards_dev.argtypes = [vp, vp, u64, u32, u32, u32, vp]
    L.zmi_gen_shards_strided_dev.argtypes = [vp, vp, u64, u32, u32, u32, u32, vp]
    L.zmi_scan_sizes_dev.argtypes = [vp, vp, u32, vp, vp]
    L.zmi_copy_ranges_dev.argtypes = [vp, vp, vp, u64, vp, u32, u32, vp, vp, u64, vp]
    L.zmi_pack_slab_dev.argtypes = [vp, vp, u64, vp, u32, vp, u64, vp, vp]
    L.zmi_deflate_batch.argtypes = [vp, vp, vp, vp, u32, i32, i32, i32, vp, u64, vp, vp]
    L.zmi_inflate_batch.argtypes = [vp, vp, vp, vp, u32, i32, vp, vp, vp, vp, vp]
    # the multi-GPU stitch (csrc/exchange.hip); RCCL itself is loaded by the library on first use
    L.zmi_comm_unique_id.argtypes = [vp]
    L.zmi_comm_create.argtypes = [C.POINTER(vp), vp, i32, i32, vp]
    L.zmi_comm_destroy.argtypes = [vp]
    L.zmi_comm_abort.argtypes = [vp]
    L.zmi_exchange_sizes.argtypes = [vp, vp, u32, vp, vp]
    L.zmi_stitch_plan_dev.argtypes = [vp, vp, u32, u32, vp, vp, vp, vp, vp]
    L.zmi_exchange_slabs.argtypes = [vp, vp, vp, vp, u64, i32, vp]
    L.zmi_exchange_slabs_round.argtypes = [vp, vp, vp, u64, u64, vp, i32, vp]
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: rc=%d (%s)" % (what, rc, lib().zmi_last_error().decode()))
