"""profiling helper: the time-parallel kernel (MPCX_SOLVE_TIME_PARALLEL, flags = 64) against the default kernels on the benchmark
constellation -- statuses, iteration counts, distance of the solutions, solve time."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mpconstellation_amd import mpc_step_batch
from test_full_size_gpu import workload

for S, K in [(4, 30), (64, 30), (64, 60), (37, 17), (16, 9), (5, 5), (256, 30)]:
    xbar, ubar, consts, r_des = workload(4096, K, first=0, count=S)
    tf = np.ones(S)
    a = mpc_step_batch(xbar, ubar, tf, consts, r_des)
    b = mpc_step_batch(xbar, ubar, tf, consts, r_des, flags=64)
    t = []
    for fl in (0, 64):
        for _ in range(2): mpc_step_batch(xbar, ubar, tf, consts, r_des, flags=fl)
        t0 = time.perf_counter()
        for _ in range(5): mpc_step_batch(xbar, ubar, tf, consts, r_des, flags=fl)
        t.append((time.perf_counter() - t0) / 5 * 1e3)
    print(f"S {S:4d} K {K:3d}: status default {np.bincount(a.status, minlength=1).tolist()} tp {np.bincount(b.status, minlength=1).tolist()}  iters {a.iters.mean():.2f}/{a.iters.max()} vs {b.iters.mean():.2f}/{b.iters.max()} same {int((a.iters == b.iters).sum())}/{S}"
          f"  |dX| {np.abs(a.X - b.X).max():.2e} |dU| {np.abs(a.U - b.U).max():.2e} |dtf| {np.abs(a.tf - b.tf).max():.2e}   call {t[0]:.3f} -> {t[1]:.3f} ms", flush=True)
