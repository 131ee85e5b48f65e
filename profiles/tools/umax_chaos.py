"""profiling helper: how sensitive the u_max 0.3 edge case is to rounding-level changes of its inputs.  The same 256
problems with ubar scaled by (1 + j * 2^-50), j = 0..7: statuses and iteration statistics per perturbation.
usage: python profiles/tools/umax_chaos.py lib.so"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
from test_full_size_gpu import workload
from mpconstellation_amd import mpc_step_batch
xbar, ubar, consts, r_des = workload(4096, 30, first=0, count=256)
base = None
for j in range(8):
    r = mpc_step_batch(xbar, ubar * (1.0 + j * 2.0 ** -50), np.ones(256), consts, r_des, options={"u_lim": [0, 0.3]})
    if base is None: base = r.iters.copy()
    print(os.path.basename(sys.argv[1]), "perturbation", j, "status", {int(k): int(v) for k, v in zip(*np.unique(r.status, return_counts=True))},
          "iters mean %.1f max %d" % (r.iters.mean(), r.iters.max()), "| satellites whose iteration count differs from j=0:", int((r.iters != base).sum()),
          "max |diff|", int(np.abs(r.iters - base).max()))
