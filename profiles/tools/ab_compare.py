"""profiling helper: two builds of libmpcx.so must give bit-identical results (refactors of solve_kernel that are meant to keep
every iteration path).  usage: python profiles/tools/ab_compare.py libA.so libB.so"""
import os, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
code = '''
import sys, os
sys.path.insert(0, "%s"); sys.path.insert(0, os.path.join("%s", "tests"))
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = "%s"
import numpy as np
from test_full_size_gpu import workload
from mpconstellation_amd import mpc_step_batch
out = {}
for name, (S, K, opts) in {"k30": (1024, 30, {}), "k100": (256, 100, {}), "mpc": (512, 30, {"eps_r": 1e-6, "eps_vr": 1e-16, "r_des": 1.05}),
                           "umax": (256, 30, {"u_lim": [0, 0.3]})}.items():
    xbar, ubar, consts, r_des = workload(4096, K, first=0, count=S)
    o = dict(opts); rd = np.full(S, o.pop("r_des")) if "r_des" in o else r_des
    r = mpc_step_batch(xbar, ubar, np.ones(S), consts, rd, options=o)
    for f in ("X", "U", "NU", "tf", "iters", "status", "kkt"): out[name + "_" + f] = getattr(r, f)
np.savez("%s", **out)
'''
files = []
for lib in sys.argv[1:3]:
    f = tempfile.mktemp(suffix=".npz"); files.append(f)
    subprocess.check_call([sys.executable, "-c", code % (ROOT, ROOT, os.path.abspath(lib), f)])
import numpy as np
a, b = np.load(files[0]), np.load(files[1])
bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
print("bit-identical" if not bad else f"DIFFERENT: {bad}")
for k in bad:
    d = np.abs(a[k].astype(float) - b[k].astype(float)); print(k, "max abs diff", np.nanmax(d), "entries", int((d > 0).sum()))
sys.exit(1 if bad else 0)
