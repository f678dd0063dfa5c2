"""Multi-GPU plumbing of the shard-parallel deflate job (one process per GPU, torch.distributed).

Shards are independent deflate streams, so the data path has NO collective: rank r compresses the
shards g with g % world == r (round-robin, BASELINE.json configs[4]).  What the ranks do exchange is
the table of compressed sizes -- 4 bytes per shard, one fixed-size all-gather -- from which every
rank (or the host that writes the result) derives the byte offset of every shard in the stitched
output.  The stitch itself is a concatenation of complete gzip/zlib members in global shard order
(multi-member gzip, the format libz-rs-sys/src/gz.rs:931-932,1464-1506 reads back), or a single
stream via the crc32_combine algebra of zlib-rs/src/deflate.rs:4149-4221.

Works with backend "nccl" (= RCCL over xGMI on MI355X) and with "gloo" (CPU tests).
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
