"""profiling helper: per-iteration log of solve_kernel from a -DMPCX_ITER_LOG build"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from mpconstellation_amd import build as b
lib = "/tmp/libmpcx_iterlog.so"
LOG_IT = os.environ.get("LOG_IT")
subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DMPCX_ITER_LOG"] + ([f"-DMPCX_LOG_IT={LOG_IT}"] if LOG_IT else []) + ["-o", lib] + b.sources())
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = lib
from mpconstellation_amd import solve_batch
G = os.path.join(ROOT, "tests", "golden")
name = sys.argv[1] if len(sys.argv) > 1 else "tan_K20_tf2"
d = np.load(os.path.join(G, f"disc_{name}.npz"))
x, u, tf, cst = d["x"], d["u"], float(d["tf"]), d["const"]
r = solve_batch(d["A"][None], d["Bp"][None], d["Bn"][None], d["Sigma"][None], d["xi"][None], x[None], u[None], [tf], cst[None],
                [np.linalg.norm(x[:3, -1])])
print("status", r.status[0], "iters", r.iters[0], "kkt", r.kkt[0])
lg = r.X[0].ravel()
for i in range(min(int(r.iters[0]) + 1, lg.size // 5)):
    print(f"it {i:3d} mu {lg[5*i]:.1e} E0 {lg[5*i+1]:.3e} alpha {lg[5*i+2]:.4f} delta_w {lg[5*i+3]:.1e} fails(finite|border|factor) {int(lg[5*i+4]):06d}")
if LOG_IT:
    v = r.NU[0].ravel(); n = 0
    def take(k):
        global n
        o = v[n:n + k]; n += k; return o
    np.set_printoptions(precision=6, linewidth=200)
    print("sol", take(7)); print("tw", take(5)); print("twin", take(5)); print("siglam", take(8))
    print("xK\n", take(56).reshape(8, 7)); print("gtf_rhs rvt_rhs", take(2)); print("gterm", take(5)); print("Wtf gam delta_w", take(3))
