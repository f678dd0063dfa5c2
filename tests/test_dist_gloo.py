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
    # ---- the slab exchange itself (the code that runs over RCCL on the GPUs), on CPU tensors: dense slab per rank,
    # direct grouped send/recv in several rounds, all-gather and gather-to-root; the scatter by global offsets is done
    # here in numpy (on the GPU it is csrc/pack.hip, tested in test_emu_kernels.py / test_gpu_parity.py)
    import numpy as np
    slab = torch.from_numpy(np.frombuffer(b"".join(members), dtype=np.uint8).copy())
    slab_bytes = [int(x) for x in table.to(torch.int64).sum(1)]
    so = zd.slab_offsets(table)
    assert int(so[rank, -1]) == slab.numel() == slab_bytes[rank]
    for mode, chunk in (("allgather", 1 << 30), ("allgather", 1000), ("gather", 4096)):
        slabs = zd.exchange_slabs(slab, slab_bytes, mode=mode, root=0, chunk_bytes=chunk)
        if mode == "gather" and rank != 0:
            assert slabs is None
            continue
        stitched = np.zeros(total, dtype=np.uint8)
        for r in range(world):
            assert slabs[r].numel() == slab_bytes[r]
            sr = slabs[r].numpy()
            for j in range(table.shape[1]):
                a, n = int(so[r, j]), int(table[r, j])
                stitched[int(offs[r, j]):int(offs[r, j]) + n] = sr[a:a + n]
        want = b"".join(o.gen_shard(g, 1 << 13) for g in range(n_total))
        assert gzip.decompress(stitched.tobytes()) == want, mode      # every rank holds the whole file after the all-gather
    # the bounded-memory form (what bench.py runs at N > 1): chunks arrive in reused staging and are consumed at once
    for mode in ("allgather", "gather"):
        stitched = np.zeros(total, dtype=np.uint8)
        mine_so = so[rank]
        for j in range(table.shape[1]):          # own shards do not travel
            a, n = int(mine_so[j]), int(table[rank, j])
            stitched[int(offs[rank, j]):int(offs[rank, j]) + n] = slab.numpy()[a:a + n]
        peer_bytes = {}

        def consume(peer, lo, view, peer_bytes=peer_bytes):
            peer_bytes.setdefault(peer, bytearray())
            assert len(peer_bytes[peer]) == lo
            peer_bytes[peer] += view.numpy().tobytes()

        got = zd.exchange_slabs_streaming(slab, slab_bytes, chunk_bytes=777, consume=consume, mode=mode, root=0)
        if mode == "gather" and rank != 0:
            assert got == 0
            continue
        assert got == sum(slab_bytes) - slab_bytes[rank]
        for r, bts in peer_bytes.items():
            sr = np.frombuffer(bytes(bts), dtype=np.uint8)
            for j in range(table.shape[1]):
                a, n = int(so[r, j]), int(table[r, j])
                stitched[int(offs[r, j]):int(offs[r, j]) + n] = sr[a:a + n]
        assert gzip.decompress(stitched.tobytes()) == b"".join(o.gen_shard(g, 1 << 13) for g in range(n_total)), mode
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
