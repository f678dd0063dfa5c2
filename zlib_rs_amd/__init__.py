"""zlib_rs_amd -- MI355X-native DEFLATE engine behind the zlib C ABI of trifectatechfoundation/zlib-rs.

Only what the deflate/inflate hot path needs lives here:
  csrc/        hand-written HIP kernels for gfx950 + the C ABI (libzmi355.so)
  engine.py    device-resident batch API (torch tensors in HBM, HIP stream from torch)
  dist.py      multi-GPU plumbing: shard ownership, size-table all-gather, slab exchange
"""
from ._build import build  # noqa: F401


def _engine(*a, **k):
    from .engine import Engine
    return Engine(*a, **k)
