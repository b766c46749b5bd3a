"""Read sharding across the GPUs of one node: reads are independent units (src/mers.rs:77-102 has no
cross-read state) and the index is read-only and replicated, so rank r maps a contiguous slice of the batch and the
results are concatenated in rank order = input order (the reference writes PAF in input order, src/closures.rs:117-123).
No data-path collective; `gather_hits` is the only exchange and exists for hosts that want one merged result array.
"""
import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous slice [lo, hi) of n reads for `rank` of `world`: sizes differ by at most one, order preserved."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_reads(bases, offsets, world, rank):
    """This rank's reads as (bases_view, offsets rebased to 0, lo)."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    lo, hi = shard_bounds(offsets.size - 1, world, rank)
    o = offsets[lo:hi + 1]
    return bases[int(o[0]):int(o[-1])], (o - o[0]).astype(np.uint64), lo


def gather_hits(local_hits, group=None):
    """All ranks' result records concatenated in rank order (== input order).  Uses torch.distributed (RCCL or gloo)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rec = local_hits.dtype
    mine = torch.from_numpy(np.ascontiguousarray(local_hits).view(np.uint8).reshape(-1).copy())
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mine.numel()], dtype=torch.int64), group=group)
    mx = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(mx, dtype=torch.uint8)
    pad[:mine.numel()] = mine
    bufs = [torch.zeros(mx, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    parts = [bufs[r][:int(sizes[r].item())].numpy().view(rec) for r in range(world)]
    return np.concatenate(parts) if parts else np.zeros(0, dtype=rec)
