"""The C-ABI stitch on the MI355X with the REAL RCCL (one rank: what a one-GPU box can run; world sizes 2 and 3 run on the CPU
against the mock, tests/test_exchange_mock.py): communicator through zmi_comm_unique_id / zmi_comm_create, the size-table
all-gather, the plan kernel against numpy, the exchange entry points with no peer, and the scatter of the packed slab into the
stitched output, read back by gzip."""
import ctypes as C
import gzip

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c_abi_stitch_one_rank_on_gpu():
    import torch
    from zlib_rs_amd.engine import Engine, uniform_layout, WRAP_GZIP
    e = Engine(0)
    n, B = 96, 1 << 16
    data = e.gen_shards(n, B)
    off, ln = uniform_layout(n, B, e.device)
    out, olen, st = e.deflate_batch(data, off, ln, B, level=6, wrap=WRAP_GZIP)
    torch.cuda.synchronize()
    assert int((st != 0).sum().item()) == 0
    comm = e.comm_create(1, 0, e.comm_unique_id())          # ncclGetUniqueId + ncclCommInitRank on the context's device
    assert e.L.zmi_comm_world(comm) == 1 and e.L.zmi_comm_rank(comm) == 0
    table = e.exchange_sizes(comm, olen, 1)                  # ncclAllGather
    torch.cuda.synchronize()
    assert torch.equal(table[0], olen)
    goff, soff, totals = e.stitch_plan(table)
    sizes = olen.cpu().numpy().astype(np.uint64)
    want = np.cumsum(sizes) - sizes
    assert (goff[0].cpu().numpy().astype(np.uint64) == want).all() and (soff[0, :-1].cpu().numpy().astype(np.uint64) == want).all()
    assert totals == [int(sizes.sum()), int(sizes.sum())] and int(soff[0, -1].item()) == totals[0]
    slab = torch.empty(totals[0] + 16, dtype=torch.uint8, device=e.device)
    e.copy_ranges(out, None, out.stride(0), olen, out.stride(0), slab, soff[0])
    # the exchange itself has no peer at world 1: both forms must come back at once and leave the slab alone
    stage = [None]
    e.exchange_round(comm, slab, totals, 0, 1 << 20, stage, -1)
    tb = (C.c_uint64 * 1)(totals[0])
    ptrs = (C.c_void_p * 1)(None)
    assert e.L.zmi_exchange_slabs(comm, slab.data_ptr(), tb, ptrs, 1 << 20, -1, torch.cuda.current_stream().cuda_stream) == 0
    stitched = torch.zeros(totals[1] + 16, dtype=torch.uint8, device=e.device)
    e.copy_ranges(slab, soff[0, :-1].contiguous(), 0, olen, int(olen.max().item()), stitched, goff[0].contiguous())
    torch.cuda.synchronize()
    assert gzip.decompress(bytes(stitched[:totals[1]].cpu().numpy())) == bytes(data.cpu().numpy())
    e.comm_destroy(comm)
    e.close()
