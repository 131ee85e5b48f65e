"""profiling helper: the u_max 0.3 edge case (thrust limit below the reference thrust) with a given build; prints the
statuses, the slowest satellites and their regularisation records.  usage: python profiles/tools/umax_probe.py lib.so [u_max]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
from test_full_size_gpu import workload
from mpconstellation_amd import mpc_step_batch
umax = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
xbar, ubar, consts, r_des = workload(4096, 30, first=0, count=256)
kw = dict(regularised=True) if hasattr(_ffi.load(), "mpcx_solve_regularised") else {}
try:
    r = mpc_step_batch(xbar, ubar, np.ones(256), consts, r_des, options={"u_lim": [0, umax]}, **kw)
except Exception as e:
    kw = {}; r = mpc_step_batch(xbar, ubar, np.ones(256), consts, r_des, options={"u_lim": [0, umax]})
print(os.path.basename(sys.argv[1]), "status", dict(zip(*np.unique(r.status, return_counts=True))), "iters mean %.1f max %d" % (r.iters.mean(), r.iters.max()))
top = np.argsort(-r.iters)[:8]
print("  slowest:", [(int(i), int(r.iters[i]), int(r.status[i]), "%.1e" % r.kkt[i]) + ((int(r.n_regularised[i]),) if kw else ()) for i in top])
np.save("/tmp/umax_%s.npy" % os.path.basename(sys.argv[1]), r.iters)
