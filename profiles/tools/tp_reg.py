import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mpconstellation_amd import mpc_step_batch
from test_full_size_gpu import workload
xbar, ubar, consts, r_des = workload(4096, 30, first=24, count=8)
for fl in (0, 64):
    r = mpc_step_batch(xbar, ubar, np.ones(8), consts, r_des, flags=fl, regularised=True)
    print("flags", fl, "iters", r.iters.tolist(), "n_regularised", r.n_regularised.tolist(), "first", r.first_regularised.tolist(), "kkt", ["%.1e" % v for v in r.kkt])
