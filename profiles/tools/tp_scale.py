"""profiling helper: call time of the default kernels and of the time-parallel one against the batch size"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mpconstellation_amd import mpc_step_batch
from test_full_size_gpu import workload
K = int(os.environ.get("K", "30"))
for S in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 32, 48, 64, 96, 128]:
    xbar, ubar, consts, r_des = workload(4096, K, first=int(os.environ.get("FIRST", "0")), count=S)
    tf = np.ones(S)
    t = []
    for fl in (0, 64):
        for _ in range(3): r = mpc_step_batch(xbar, ubar, tf, consts, r_des, flags=fl)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); r = mpc_step_batch(xbar, ubar, tf, consts, r_des, flags=fl); ts.append((time.perf_counter() - t0) * 1e3)
        t.append(np.median(ts))
    print(f"S {S:4d} K {K}: default {t[0]:.3f} ms   time-parallel {t[1]:.3f} ms   iters max {r.iters.max()} at {int(r.iters.argmax())}  tp iters {r.iters.tolist() if S <= 16 else None}", flush=True)
