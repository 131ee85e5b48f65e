"""profiling helper: per-iteration log (-DMPCX_ITER_LOG build) of one satellite of the benchmark constellation under given options
usage: python profiles/tools/iter_log_sat.py <satellite index> '<options dict>' [K]"""
import os, sys, subprocess, ast
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpconstellation_amd import build as b
lib = "/tmp/libmpcx_iterlog.so"
subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DMPCX_ITER_LOG", "-o", lib] + b.sources())
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = lib
from test_full_size_gpu import workload
from mpconstellation_amd import mpc_step_batch
sat = int(sys.argv[1]); opts = ast.literal_eval(sys.argv[2]) if len(sys.argv) > 2 else {}
K = int(sys.argv[3]) if len(sys.argv) > 3 else 30
xbar, ubar, consts, r_des = workload(4096, K, first=sat, count=1)
r = mpc_step_batch(xbar, ubar, np.ones(1), consts, r_des, options=opts)
print("status", r.status[0], "iters", r.iters[0], "kkt", r.kkt[0])
lg = r.X[0].ravel(); lu = r.U[0].ravel()
for i in range(min(int(r.iters[0]) + 1, lg.size // 5)):
    extra = f" first trial: alpha {lu[3*i]:.4f} |F|/|F0| {lu[3*i+1]:.4f} nbhd margin {lu[3*i+2]:.2e}" if 3 * i + 2 < lu.size else ""
    print(f"it {i:3d} mu {lg[5*i]:.1e} E0 {lg[5*i+1]:.3e} alpha {lg[5*i+2]:.4f} delta_w {lg[5*i+3]:.1e} fails {int(lg[5*i+4]):06d}{extra}")
