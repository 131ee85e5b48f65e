"""debug helper (not a test): GPU solver vs numpy oracle at increasing iteration caps"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_lib as O, nlp_ipm as N
from mpconstellation_amd import solve_batch, mpc_step_batch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NREF = int(os.environ.get("NREF", "1"))
name = sys.argv[1] if len(sys.argv) > 1 else "tan_K30_tf1"
d = np.load(os.path.join(G, f"disc_{name}.npz"))
x, u, tf, cst = d["x"], d["u"], float(d["tf"]), d["const"]
stage = {k: d[k] for k in ("A", "Bp", "Bn", "Sigma", "xi")}
terms = O.constraint_terms(x, u, cst[0])
r_des = np.linalg.norm(x[:3, -1])
P = N.MpcProblem(x, u, tf, cst[0], stage, terms, {"r_des": r_des})
for mi in [int(a) for a in sys.argv[2:]] or [0, 1, 2, 5, 200]:
    ro = N.solve(P, max_iter=mi, n_refine=NREF)
    t = time.time()
    rg = solve_batch(d["A"][None], d["Bp"][None], d["Bn"][None], d["Sigma"][None], d["xi"][None], x[None], u[None],
                     [tf], cst[None], [r_des], max_iter=mi, n_refine=NREF)
    dt = time.time() - t
    print(f"max_iter {mi:3d}: oracle st {ro['status']} it {ro['iters']} tf {ro['tf']:.12f} kkt {ro['kkt']:.3e} | gpu st {rg.status[0]} it {rg.iters[0]} tf {rg.tf[0]:.12f} kkt {rg.kkt[0]:.3e} | dX {np.abs(rg.X[0]-ro['X']).max():.2e} dU {np.abs(rg.U[0]-ro['U']).max():.2e} dNU {np.abs(rg.NU[0]-ro['NU']).max():.2e}  ({dt*1e3:.1f} ms)")
rf = mpc_step_batch(x[None], u[None], [tf], cst[None], [r_des])
print("fused step: st", rf.status[0], "it", rf.iters[0], "tf", rf.tf[0], "dX vs oracle", np.abs(rf.X[0] - ro["X"]).max())
