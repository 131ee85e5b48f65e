"""Satellite sharding for multi-GPU runs: independent satellites, contiguous blocks per rank, no
data-path collective.  Two ways to use several devices:
 * one process per GPU (bench.py under torch.distributed.run): every rank takes shard_block(S, world, rank); the only
   collectives are the timing reduction and the optional final gather_trajectories;
 * one process, several devices (the drop-in API: ConstellationMPC / mpc_step_batch / mpc_update_batch with devices=[...]):
   sharded_call below -- one host thread and one mpcx context (own stream, own staging pools) per device, each solving its
   contiguous block and writing its results IN PLACE into its slice of ONE result set allocated for the whole constellation
   (the satellite axis is outermost everywhere, so a block's slice of a C-ordered array is itself contiguous: the library's
   copy-out is the only pass over the results).  This replaces the reference's serial loop over the constellation
   (simulator.py:41,58).  (Until round 4 every block allocated its own arrays and the blocks were np.concatenate'd afterwards on
   one host thread: 267 MB at 65 536 x 30, the largest term of an 8-device step.)"""
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


last_call = {}      # timing of the last multi-block sharded_call: per block (t_start, t_end) around the library call, and the wall span


def sharded_call(fn, devices, batched, outs, *args, **kw):
    """fn(*block_of_each_batched_array, *args, device=d, slot=s, out=views, **kw) for the contiguous block of every device,
    concurrently (ctypes releases the GIL inside the library call: the devices really run side by side).
    batched: input arrays with the satellite axis first (None entries are passed through).
    outs: the whole constellation's result set, dict name -> array (satellite axis 0) or (array, 1) for the per-iteration
    records (n_scp, S) of an update; None values are skipped.  Every block gets VIEWS of its satellites' part (`out=`) and
    writes its results there: nothing is joined afterwards.  Returns the list of fn's return values in device order (whatever
    the wrappers want to hand back beside the arrays).  Devices that get no satellite (more devices than satellites) are skipped."""
    import time
    ctxs = device_contexts(devices)
    S = next(a for a in batched if a is not None).shape[0]
    jobs = []
    for rank, (dev, slot) in enumerate(ctxs):
        first, count = shard_block(S, len(ctxs), rank)
        if count == 0:
            continue
        blk = [None if a is None else a[first:first + count] for a in batched]
        jobs.append((blk, dev, slot, block_views(outs, first, count)))
    spans = [None] * len(jobs)

    def run(i):
        blk, dev, slot, out = jobs[i]
        t0 = time.perf_counter()
        r = fn(*blk, *args, device=dev, slot=slot, out=out, **kw)
        spans[i] = (t0, time.perf_counter())
        return r
    t_in = time.perf_counter()
    if len(jobs) == 1:
        res = [run(0)]
    else:
        futs = [_pool(len(jobs)).submit(run, i) for i in range(len(jobs))]
        res = [f.result() for f in futs]
    last_call.update(blocks=spans, wall=(t_in, time.perf_counter()))
    return res


def block_views(outs, first, count):
    """the part of every whole-constellation result array that belongs to satellites first .. first + count - 1"""
    views = {}
    for name, a in (outs or {}).items():
        if a is None:
            continue
        arr, axis = a if isinstance(a, tuple) else (a, 0)
        views[name] = arr[first:first + count] if axis == 0 else arr[:, first:first + count]
    return views


class OutArrays:
    """The result arrays of one wrapper call: the caller's (`out`: views of a whole-constellation result set, sharded_call) where
    given, new ones otherwise.  A view that is not C-contiguous (a block's columns of an (n_scp, S) record) is filled through
    a small contiguous temporary copied over in finish()."""

    def __init__(self, out=None):
        self.out = out or {}
        self.late = []

    def get(self, name, shape, dtype=np.float64, make=None):
        a = self.out.get(name)
        if a is None:
            return make() if make is not None else np.empty(shape, dtype=dtype)
        if a.shape != tuple(shape) or a.dtype != np.dtype(dtype):
            raise ValueError(f"out[{name!r}]: expected {tuple(shape)} {np.dtype(dtype)}, got {a.shape} {a.dtype}")
        if a.flags.c_contiguous:
            return a
        tmp = np.empty(shape, dtype=dtype)
        self.late.append((a, tmp))
        return tmp

    def finish(self):
        for a, tmp in self.late:
            a[...] = tmp


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
