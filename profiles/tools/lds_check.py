"""profiling helper: the LDS-resident small-batch kernel (solve_lds.hip: batches of at most one satellite per compute unit, K <= 30)
against the two-wave kernel with the workspace in global memory (flag MPCX_SOLVE_NO_LDS = 32) and the one-wave kernel (16):
results must be bit-identical; durations of the host-pointer call"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_full_size_gpu import workload
from mpconstellation_amd import mpc_step_batch
F = ("X", "U", "NU", "tf", "iters", "status", "kkt", "n_regularised")
for S, K, opts in ((1, 30, {}), (64, 30, {}), (256, 30, {}), (64, 20, {}), (64, 30, {"eps_r": 1e-6, "eps_vr": 1e-16, "tf_max": 1.0}), (64, 30, {"u_lim": [0, 0.3]}), (3, 3, {}),
                   (64, 31, {}), (300, 30, {})):
    xbar, ubar, consts, r_des = workload(4096, K, first=0, count=S)
    tf = np.ones(S)
    a = mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, regularised=True)                # default: LDS-resident where it applies
    b = mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, flags=32, regularised=True)      # two waves, global workspace
    c = mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, flags=16, regularised=True)      # one wave
    same = all(np.array_equal(getattr(a, f), getattr(b, f), equal_nan=True) and np.array_equal(getattr(a, f), getattr(c, f), equal_nan=True) for f in F)
    def t(flags):
        for _ in range(2): mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, flags=flags)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, flags=flags); ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts)
    print(f"S {S:4d} K {K:3d} {str(opts):40s} bit-identical {same}  status {sorted(set(a.status.tolist()))} iters max {a.iters.max()}  "
          f"host-pointer call (best of 7): default {t(0):.3f} ms, two waves / global workspace {t(32):.3f} ms, one wave {t(16):.3f} ms", flush=True)
