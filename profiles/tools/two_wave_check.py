"""profiling helper: the two-wave small-batch kernel against the one-wave kernel (flag MPCX_SOLVE_ONE_WAVE = 16): results must
be bit-identical; solve_kernel durations by HIP events"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_full_size_gpu import workload
from mpconstellation_amd import mpc_step_batch
for S, K, opts in ((1, 30, {}), (64, 30, {}), (64, 100, {}), (512, 30, {}), (256, 30, {"eps_r": 1e-6, "eps_vr": 1e-16}), (256, 30, {"u_lim": [0, 0.3]}), (3, 3, {})):
    xbar, ubar, consts, r_des = workload(4096, K, first=0, count=S)
    tf = np.ones(S)
    a = mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, flags=16, regularised=True)
    b = mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, regularised=True)
    same = all(np.array_equal(getattr(a, f), getattr(b, f), equal_nan=True) for f in ("X", "U", "NU", "tf", "iters", "status", "kkt", "n_regularised"))
    def t(flags):
        mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, flags=flags)
        t0 = time.perf_counter()
        for _ in range(5): mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, flags=flags)
        return (time.perf_counter() - t0) / 5 * 1e3
    print(f"S {S:4d} K {K:3d} {str(opts):40s} bit-identical {same}  status {sorted(set(b.status.tolist()))} iters max {b.iters.max()} reg max {b.n_regularised.max()}  "
          f"host-pointer call: one wave {t(16):.3f} ms, two waves {t(0):.3f} ms", flush=True)
