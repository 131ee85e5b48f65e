"""Satellite sharding for multi-GPU runs: independent satellites, contiguous blocks per rank, no
data-path collective.  Two ways to use several devices:
 * one process per GPU (bench.py under torch.distributed.run): every rank takes shard_block(S, world, rank); the only
   collectives are the timing reduction and the optional final gather_trajectories;
 * one process, several devices (the drop-in API: ConstellationMPC / mpc_step_batch / mpc_update_batch with devices=[...]):
   sharded_call below -- one host thread and one mpcx context (own stream, own staging pools) per device, each solving its
   contiguous block; the blocks' results are joined on the host.  This replaces the reference's serial loop over the
   constellation (simulator.py:41,58)."""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def shard_block(S_total, world, rank):
    """(first, count) of the contiguous block of satellites owned by `rank`; blocks differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(S_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def device_contexts(devices):
    """[(device, slot)] for a device list: a device named several times gets one context slot per mention (slot 0, 1, ...),
    so that devices=[0, 0] are two contexts (two streams) on one GPU -- how the multi-device path is rehearsed on a one-GPU box."""
    seen = {}
    out = []
    for d in devices:
        d = int(d)
        out.append((d, seen.get(d, 0)))
        seen[d] = seen.get(d, 0) + 1
    return out


_pools = {}
_pools_lock = threading.Lock()


def _pool(n):
    """worker threads are kept: a thread that has used a context keeps using it (a context is not thread-safe, include/mpcx.h;
    here every call names its (device, slot) explicitly and no two calls of one sharded_call share one)"""
    with _pools_lock:
        if n not in _pools:
            _pools[n] = ThreadPoolExecutor(max_workers=n, thread_name_prefix="mpcx-dev")
        return _pools[n]


def sharded_call(fn, devices, batched, *args, **kw):
    """fn(*block_of_each_batched_array, *args, device=d, slot=s, **kw) for the contiguous block of every device, concurrently
    (ctypes releases the GIL inside the library call: the devices really run side by side); returns the list of results in
    device order.  batched: arrays with the satellite axis first (None entries are passed through).  Devices that get no
    satellite (more devices than satellites) are skipped."""
    ctxs = device_contexts(devices)
    S = next(a for a in batched if a is not None).shape[0]
    jobs = []
    for rank, (dev, slot) in enumerate(ctxs):
        first, count = shard_block(S, len(ctxs), rank)
        if count == 0:
            continue
        blk = [None if a is None else a[first:first + count] for a in batched]
        jobs.append((blk, dev, slot))
    if len(jobs) == 1:
        blk, dev, slot = jobs[0]
        return [fn(*blk, *args, device=dev, slot=slot, **kw)]
    futs = [_pool(len(jobs)).submit(fn, *blk, *args, device=dev, slot=slot, **kw) for blk, dev, slot in jobs]
    return [f.result() for f in futs]


def join_results(parts, cls=None):
    """one result object from the blocks' results: every ndarray attribute concatenated along the satellite axis (axis 0, or
    axis 1 for the per-iteration records (n_scp, S) of an update); attributes that are None everywhere stay None"""
    first = parts[0]
    if len(parts) == 1:
        return first
    out = first.__class__.__new__(first.__class__)
    for k, v in first.__dict__.items():
        vals = [getattr(p, k) for p in parts]
        if all(x is None for x in vals):
            setattr(out, k, None)
        elif isinstance(v, np.ndarray):
            ax = 1 if (v.ndim == 2 and k in ("status", "iters") and getattr(first, "Ks", None) is not None) else 0
            setattr(out, k, np.concatenate(vals, axis=ax))
        else:
            setattr(out, k, v)
    return out


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
