"""profiling helper: three quick cases of the time-parallel kernel against the default ones (correctness counters + call time); MPCX_LIB selects another build"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mpconstellation_amd import mpc_step_batch
from test_full_size_gpu import workload
for S, K in [(8, 30), (64, 30), (64, 60)]:
    xbar, ubar, consts, r_des = workload(4096, K, first=0, count=S)
    tf = np.ones(S)
    a = mpc_step_batch(xbar, ubar, tf, consts, r_des)
    t = []
    for fl in (0, 64):
        for _ in range(3): b = mpc_step_batch(xbar, ubar, tf, consts, r_des, flags=fl)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); b = mpc_step_batch(xbar, ubar, tf, consts, r_des, flags=fl); ts.append((time.perf_counter() - t0) * 1e3)
        t.append(np.median(ts))
    print(f"{os.environ.get('MPCX_LIB', 'default lib')[-14:]} S {S} K {K}: ok {int((b.status == 0).sum())}/{S} iters same {int((a.iters == b.iters).sum())}/{S} |dX| {np.abs(a.X - b.X).max():.1e}  default {t[0]:.3f} tp {t[1]:.3f} ms", flush=True)
