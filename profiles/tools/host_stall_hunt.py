"""profiling helper: what makes ONE host-pointer call in many take 40-50 ms (BENCH_r03: 53 ms among 1.5 ms calls; the second
warm-up call of bench.py at 4096 satellites)?  MPCX_HOST_TRACE puts the time into the wait for the first download's event
while the kernels themselves run as fast as ever (rocprofv3 trace): something else occupies the device or its copy engines.
Suspects tried here between bursts of small host-pointer calls: releasing device memory (torch.cuda.empty_cache after a
multi-GB tensor: the driver unmaps / clears VRAM), growing the context's workspace (hipFree + hipMalloc), fresh page-locked
staging (hipHostMalloc).  Run it with and without HSA_ENABLE_SDMA=0 (copies by shader kernels instead of the SDMA engines).
usage: MPCX_HOST_TRACE=5 python profiles/tools/host_stall_hunt.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from test_full_size_gpu import workload
from mpconstellation_amd import mpc_step_batch, _ffi

xbar, ubar, consts, r_des = workload(4096, 30, first=0, count=64)
tf = np.ones(64)
big = workload(4096, 30, first=0, count=4096)

def burst(label, n=40):
    ms = []
    for _ in range(n):
        t0 = time.perf_counter(); mpc_step_batch(xbar, ubar, tf, consts, r_des); ms.append((time.perf_counter() - t0) * 1e3)
    ms = np.array(ms)
    print(f"{label:60s} median {np.median(ms):.3f} max {ms.max():.3f} ms at call {int(ms.argmax())}; calls > 3 ms: {[(i, round(v, 1)) for i, v in enumerate(ms) if v > 3]}", flush=True)

print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
burst("first burst (context, pools, workspace created)")
burst("steady state")
for gb in (1, 4, 16):
    t = torch.empty(gb << 27, dtype=torch.float64, device="cuda"); t.fill_(1.0); torch.cuda.synchronize()
    del t; torch.cuda.empty_cache()
    burst(f"after releasing a {gb} GB torch tensor to the driver (empty_cache)")
t = torch.empty(4 << 27, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
burst("after ALLOCATING a 4 GB torch tensor (kept)")
t0 = time.perf_counter(); mpc_step_batch(big[0], big[1], np.ones(4096), big[2], big[3]); print(f"one 4096-satellite call (grows workspace and pools): {(time.perf_counter() - t0) * 1e3:.1f} ms")
burst("after the workspace / pool growth")
t0 = time.perf_counter(); mpc_step_batch(big[0], big[1], np.ones(4096), big[2], big[3]); print(f"second 4096-satellite call: {(time.perf_counter() - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); mpc_step_batch(big[0], big[1], np.ones(4096), big[2], big[3]); print(f"third 4096-satellite call: {(time.perf_counter() - t0) * 1e3:.1f} ms")
p = [_ffi.pinned_copy(a) for a in big]
burst("after page-locking 17 MB of caller arrays (hipHostMalloc)")
del p
burst("after freeing them (hipHostFree)")
