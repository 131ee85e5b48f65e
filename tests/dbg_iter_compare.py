"""debug helper (not a test): per-iteration (mu, E0, step, delta_w) of the oracle next to the device's iteration log
(-DMPCX_ITER_LOG build)"""
import os, sys, subprocess, io, contextlib
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..")
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import oracle_lib as O, nlp_ipm as N
from mpconstellation_amd import build as b
lib = "/tmp/libmpcx_iterlog.so"
subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DMPCX_ITER_LOG", "-o", lib] + b.sources())
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = lib
from mpconstellation_amd import solve_batch
name = sys.argv[1]
d = np.load(os.path.join(HERE, "golden", f"disc_{name}.npz"))
x, u, tf, cst = d["x"], d["u"], float(d["tf"]), d["const"]
r_des = np.linalg.norm(x[:3, -1])
stage = {k: d[k] for k in ("A", "Bp", "Bn", "Sigma", "xi")}
P = N.MpcProblem(x, u, tf, cst[0], stage, O.constraint_terms(x, u, cst[0]), {"r_des": r_des})
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    ro = N.solve(P, verbose=True)
print(buf.getvalue())
r = solve_batch(d["A"][None], d["Bp"][None], d["Bn"][None], d["Sigma"][None], d["xi"][None], x[None], u[None], [tf], cst[None], [r_des])
print("gpu status", r.status[0], "iters", r.iters[0], "kkt", r.kkt[0])
lg = r.X[0].ravel(); lu = r.U[0].ravel()
for i in range(min(int(r.iters[0]) + 1, lg.size // 5)):
    print(f"it {i:3d} mu {lg[5*i]:.1e} E0 {lg[5*i+1]:.2e} alpha {lg[5*i+2]:.4f} delta_w {lg[5*i+3]:.1e} fails {int(lg[5*i+4]):06d}" + (f" | first trial a {lu[3*i]:.4f} r/r0 {lu[3*i+1]:.6f} prodmin/thr {lu[3*i+2]:.4e}" if 3*i+2 < lu.size else ""))
