"""profiling helper: iteration-count histogram of the benchmark constellation at K nodes, the slowest satellites with
their regularisation records.  usage: python profiles/tools/iters_hist.py K [S]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_full_size_gpu import workload
from mpconstellation_amd import mpc_step_batch
K = int(sys.argv[1]); S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
xbar, ubar, consts, r_des = workload(4096, K, first=0, count=S)
r = mpc_step_batch(xbar, ubar, np.ones(S), consts, r_des, regularised=True)
print(f"K {K} S {S}: status", {int(k): int(v) for k, v in zip(*np.unique(r.status, return_counts=True))}, "iters mean %.2f max %d" % (r.iters.mean(), r.iters.max()))
print("histogram (iterations: satellites)", {int(i): int(n) for i, n in enumerate(np.bincount(r.iters)) if n})
print("regularised iterations: satellites", {int(i): int(n) for i, n in enumerate(np.bincount(r.n_regularised)) if n})
top = np.argsort(-r.iters)[:16]
print("slowest (satellite, iterations, regularised, first regularised):", [(int(i), int(r.iters[i]), int(r.n_regularised[i]), int(r.first_regularised[i])) for i in top])
for n in (0, 1, 2, 3):
    m = r.n_regularised == n
    if m.any(): print(f"  satellites with {n} regularised iterations: {int(m.sum())}, mean iterations {r.iters[m].mean():.2f}")
m = r.n_regularised >= 4
if m.any(): print(f"  satellites with >= 4 regularised iterations: {int(m.sum())}, mean iterations {r.iters[m].mean():.2f}")
