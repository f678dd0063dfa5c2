"""Multi-GPU plumbing of the shard-parallel deflate job (one process per GPU, torch.distributed).

Ownership: round-robin -- rank r compresses the shards g with g % world == r (BASELINE.json configs[4]); its local
shard j is global shard j*world + r.  The compression itself has no collective.  The exchange that `north_star` names
("RCCL all-gather over xGMI to reassemble the compressed blocks") is the stitch:

  1. every rank packs its compress_bound-strided slots into one dense slab (Engine.pack_slab, csrc/pack.hip),
  2. the u32 size table is all-gathered (fixed size: 4 B per shard),
  3. the slabs -- different sizes -- travel point to point: RCCL has no all-gather-v, and xGMI is a full mesh of
     point-to-point links (7 x ~153 GB/s per GPU), so every rank posts one send and one receive per peer in ONE
     group (dist.batch_isend_irecv -> ncclGroupStart/End): 7 concurrent transfers, each on its own link.  A ring
     would push all 7 slabs through one link and is never used.  gather-to-root (only `root` receives) is the cheaper
     variant when one consumer writes the file,
  4. the receiver scatters every slab into the globally ordered output (Engine.copy_ranges with the global offsets):
     concatenated gzip / zlib members in shard order, the format libz-rs-sys/src/gz.rs:1464-1506 reads back; the
     semantic contract is the append loop of the reference's parallel-deflate recipe, zlib-rs/src/deflate.rs:4145-4221.

Transfers go in rounds of `chunk_bytes` per peer, so a caller that scatters round by round needs one chunk of staging
per peer beside its working set (`exchange_slabs` keeps whole slabs: what the tests and a gather-to-root of a few
tens of GiB want; see DESIGN.md section 5 for the sizes at 8 x 64 GiB).

Works with backend "nccl" (= RCCL over xGMI on MI355X) and with "gloo" (CPU tensors, tests/test_dist_gloo.py).
"""
import torch
import torch.distributed as dist


def shards_of_rank(n_total, rank, world):
    """global shard ids owned by `rank` under round-robin assignment"""
    return list(range(rank, n_total, world))


def exchange_sizes(local_sizes):
    """all-gather the per-shard compressed sizes (int32/int64 tensor of equal length on every rank).
    Returns a [world, n_local] tensor on every rank."""
    world = dist.get_world_size()
    out = [torch.empty_like(local_sizes) for _ in range(world)]
    dist.all_gather(out, local_sizes)
    return torch.stack(out)


def stitch_offsets(size_table):
    """size_table[r, j] = compressed size of global shard j*world + r.  Returns (offsets [world, n_local]
    in global shard order, total bytes)."""
    world, n_local = size_table.shape
    flat = size_table.t().reshape(-1).to(torch.int64)  # global order: shard g = j*world + r
    excl = torch.cumsum(flat, 0) - flat
    return excl.reshape(n_local, world).t().contiguous(), int(flat.sum().item())


def slab_offsets(size_table):
    """offsets of every shard inside its owner's dense slab: [world, n_local + 1] (last column = slab size)"""
    t = size_table.to(torch.int64)
    z = torch.zeros((t.shape[0], 1), dtype=torch.int64, device=t.device)
    return torch.cat([z, torch.cumsum(t, 1)], 1)


def exchange_slabs(slab, slab_bytes, mode="allgather", root=0, chunk_bytes=1 << 30):
    """Variable-size slab exchange.  `slab`: this rank's dense uint8 slab; `slab_bytes`: list of every rank's slab size
    (from the size table).  mode "allgather": every rank returns [slab_0, ..., slab_{world-1}] (its own entry is a view
    of `slab`); mode "gather": only `root` does, the others return None.  Direct sends -- one group of point-to-point
    operations per round, never a ring."""
    world, rank = dist.get_world_size(), dist.get_rank()
    assert mode in ("allgather", "gather")
    slab_bytes = [int(b) for b in slab_bytes]
    receives = mode == "allgather" or rank == root
    bufs = None
    if receives:
        bufs = [slab[:slab_bytes[r]] if r == rank else torch.empty(slab_bytes[r], dtype=torch.uint8, device=slab.device)
                for r in range(world)]
    biggest = max(slab_bytes) if slab_bytes else 0
    rounds = (biggest + chunk_bytes - 1) // chunk_bytes
    mine = slab_bytes[rank]
    dests = [p for p in range(world) if p != rank] if mode == "allgather" else ([root] if rank != root else [])
    for c in range(rounds):
        lo = c * chunk_bytes
        ops = []
        if lo < mine:
            piece = slab[lo:min(mine, lo + chunk_bytes)]
            ops += [dist.P2POp(dist.isend, piece, p) for p in dests]
        if receives:
            for p in range(world):
                if p != rank and lo < slab_bytes[p]:
                    ops.append(dist.P2POp(dist.irecv, bufs[p][lo:min(slab_bytes[p], lo + chunk_bytes)], p))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
    return bufs


def exchange_slabs_streaming(slab, slab_bytes, chunk_bytes=1 << 30, consume=None, mode="allgather", root=0):
    """The same exchange with bounded memory: per round every peer's next `chunk_bytes` land in a staging buffer that is
    reused in the following round; `consume(peer, byte_offset_in_peer_slab, view)` sees every received chunk before its
    buffer is overwritten (scatter it, write it out, checksum it).  Returns the number of bytes this rank received.
    This is the form that fits beside a 64 GiB-per-GPU working set (8 slabs of ~29 GiB do not)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    assert mode in ("allgather", "gather")
    slab_bytes = [int(b) for b in slab_bytes]
    receives = mode == "allgather" or rank == root
    biggest = max(slab_bytes) if slab_bytes else 0
    rounds = (biggest + chunk_bytes - 1) // chunk_bytes
    stage = {}
    if receives:
        for p in range(world):
            if p != rank and slab_bytes[p]:
                stage[p] = torch.empty(min(chunk_bytes, slab_bytes[p]), dtype=torch.uint8, device=slab.device)
    mine = slab_bytes[rank]
    dests = [p for p in range(world) if p != rank] if mode == "allgather" else ([root] if rank != root else [])
    got = 0
    for c in range(rounds):
        lo = c * chunk_bytes
        ops, views = [], []
        if lo < mine:
            piece = slab[lo:min(mine, lo + chunk_bytes)]
            ops += [dist.P2POp(dist.isend, piece, p) for p in dests]
        if receives:
            for p in range(world):
                if p != rank and lo < slab_bytes[p]:
                    v = stage[p][:min(slab_bytes[p], lo + chunk_bytes) - lo]
                    ops.append(dist.P2POp(dist.irecv, v, p))
                    views.append((p, v))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for p, v in views:
            got += int(v.numel())
            if consume is not None:
                consume(p, lo, v)
    return got


def stitch_on_device(engine, slabs, size_table, out=None):
    """Scatter the per-rank slabs into the globally ordered output on the GPU (csrc/pack.hip through Engine.copy_ranges).
    Returns (out uint8, total bytes).  Device tensors only -- there is no host path."""
    offs, total = stitch_offsets(size_table)
    so = slab_offsets(size_table)
    if out is None:
        out = torch.empty(total + 16, dtype=torch.uint8, device=engine.device)
    max_len = int(size_table.max().item()) if size_table.numel() else 0
    for r, sl in enumerate(slabs):
        engine.copy_ranges(sl, so[r, :-1].contiguous().to(engine.device), 0, size_table[r].to(torch.int32).contiguous().to(engine.device),
                           max_len, out, offs[r].contiguous().to(engine.device))
    return out, total


def max_over_ranks(seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_members_to_root(local_members, root=0):
    """Small-job helper (tests, host-side stitching): every rank contributes its compressed members
    (list of bytes, local order); rank `root` returns them in global shard order, others None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    gathered = [None] * world if rank == root else None
    dist.gather_object(local_members, gathered, dst=root)
    if rank != root:
        return None
    n_local = max(len(g) for g in gathered)
    out = []
    for j in range(n_local):
        for r in range(world):
            if j < len(gathered[r]):
                out.append(gathered[r][j])
    return out
