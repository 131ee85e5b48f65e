"""Satellite sharding for multi-GPU runs: independent satellites, contiguous blocks per rank, no
data-path collective.  The only collectives are the timing reduction of bench.py and
the optional final trajectory gather."""


def shard_block(S_total, world, rank):
    """(first, count) of the contiguous block of satellites owned by `rank`; blocks differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(S_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def gather_trajectories(local, group=None):
    """All-gather per-rank result tensors (count_r, ...) along the satellite axis (RCCL over xGMI when the
    process group is nccl; gloo on CPU).  Ranks may hold different counts."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
