"""Build the HIP shared library (libzmi355.so) in-tree with hipcc for gfx950."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzmi355.so")
ABI_LIB = os.path.join(HERE, "libz_mi355.so")
SOURCES = ["gen.hip", "checksum.hip", "lz77.hip", "parse.hip", "encode.hip", "encode_cp.hip", "inflate.hip", "resolve_jump.hip", "pack.hip", "blockscan.hip", "exchange.hip", "zmi_api.hip"]
ABI_SOURCES = ["zlib_abi.hip", "gz_api.hip", "host_sums.cpp"]


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(ABI_LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", f)
                                                                 for f in os.listdir(os.path.join(HERE, "..", "include"))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> zlib_rs_amd/libzmi355.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-atomic-optimizer-strategy=None: the kernels' LDS atomics are issued by one lane (the claim counter of lz77.hip) or go
    # to per-lane addresses; the optimizer's wave reduction in front of every atomic in divergent code (ballot, mbcnt, a multiply,
    # a readfirstlane) was ten instructions per claim in a kernel that is bound by its instruction count
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
           "-o", LIB] + srcs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    # the zlib-named symbols (deflate, inflate, crc32 ...) live in their own library so that merely
    # loading the engine never interposes the system libz of the process
    # -Bsymbolic-functions: calls between the zlib-named entry points (inflateResetKeep -> inflateReset, ...) must stay
    # inside this library even when the process has the system libz loaded in front of it
    # version script: the zlib version nodes (libz-rs-sys/include/zlib.map) and nothing exported but the entry points
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wl,-Bsymbolic-functions",
           "-Wl,--version-script=" + os.path.join(CSRC, "libz_mi355.map"), "-Wl,-soname,libz_mi355.so", "-o", ABI_LIB] + \
          [os.path.join(CSRC, s) for s in ABI_SOURCES] + ["-L" + HERE, "-lzmi355", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB
