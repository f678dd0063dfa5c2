"""CPU test of the N>1 path: two processes over gloo exercise the shard assignment, the size-table
all-gather, the offset algebra and the stitch (multi-member gzip + single-stream crc32_combine)."""
import gzip
import os
import sys
import zlib

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from zlib_rs_amd import dist as zd
    o = oracle_lib.load(rebuild=False)
    mine = zd.shards_of_rank(n_total, rank, world)
    shards = [o.gen_shard(g, 1 << 13) for g in mine]
    # stand-in for the GPU deflate of this rank: the oracle's gzip members (same framing)
    members = [o.deflate(s, 6, 2)[1] for s in shards]
    sizes = torch.tensor([len(m) for m in members], dtype=torch.int32)
    table = zd.exchange_sizes(sizes)
    offs, total = zd.stitch_offsets(table)
    # every rank derives the same offsets; check own members land where the table says
    assert table.shape == (world, len(mine))
    assert int(table[rank].sum()) == sum(len(m) for m in members)
    ordered = zd.gather_members_to_root(members, 0)
    tmax = zd.max_over_ranks(0.1 * (rank + 1), torch.device("cpu"))
    assert abs(tmax - 0.1 * world) < 1e-9
    if rank == 0:
        blob = b"".join(ordered)
        assert len(blob) == total
        for g in range(n_total):
            r, j = g % world, g // world
            assert blob[int(offs[r, j]):int(offs[r, j]) + int(table[r, j])] == ordered[g]
        want = b"".join(o.gen_shard(g, 1 << 13) for g in range(n_total))
        assert gzip.decompress(blob) == want  # multi-member gzip reader sees one file
        # single-stream stitch algebra: crc of the whole from the members' crcs (deflate.rs:4149-4221)
        crc = 0
        for g in range(n_total):
            part = o.gen_shard(g, 1 << 13)
            crc = o.lib.zo_crc32_combine(crc, zlib.crc32(part), len(part))
        assert crc == zlib.crc32(want)
        q.put("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_stitch():
    import oracle_lib
    oracle_lib.load()  # build once in the parent
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 10, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"
