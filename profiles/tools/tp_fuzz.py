"""profiling helper: random batches (horizon, final time, target radius, reference thrust, thrust limit, window widths) on the
time-parallel kernel against the default kernels: statuses and iteration counts"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mpconstellation_amd import mpc_step_batch, _ffi
from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
from mpconstellation_amd.simulator import propagate_batch
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = dict(n=0, a_ok=0, b_ok=0, status_differs=0, same=0, within1=0, dead=0)
for it in range(n_batches):
    S = int(rng.integers(1, 129)); K = int(rng.integers(24, 101)); tf = float(rng.uniform(0.4, 2.2)); thrust = float(rng.choice([0.1, 0.3, 0.5, 1.0]))
    first = int(rng.integers(0, 4096 - S))
    y0, consts = normalize_batch(constellation_states(4096, first=first, count=S))
    xbar, st, _ = propagate_batch(y0, np.full(S, tf), consts, (_ffi.CTRL_TANGENTIAL, np.array([thrust]), 0, None), K)
    ubar = np.ascontiguousarray(tangential_thrust(xbar, thrust))
    r_end = np.linalg.norm(xbar[:, :3, -1], axis=1)
    r_des = r_end * rng.uniform(0.97, 1.08, S)
    opts = {}
    if rng.random() < 0.3: opts.update(eps_r=1e-6, eps_vr=1e-16, tf_max=tf)
    if rng.random() < 0.2: opts["u_lim"] = [0.0, float(rng.uniform(0.4, 2.0))]
    kw = dict(options=opts, linear_vt=bool(rng.random() < 0.2))
    a = mpc_step_batch(xbar, ubar, np.full(S, tf), consts, r_des, **kw)
    b = mpc_step_batch(xbar, ubar, np.full(S, tf), consts, r_des, flags=64, **kw)
    oka, okb = np.isin(a.status, (0, 7)), np.isin(b.status, (0, 7))
    tot["n"] += S; tot["a_ok"] += int(oka.sum()); tot["b_ok"] += int(okb.sum()); tot["status_differs"] += int((oka != okb).sum())
    tot["same"] += int((a.iters == b.iters).sum()); tot["within1"] += int((np.abs(a.iters - b.iters) <= 1).sum()); tot["dead"] += int((b.kkt == -1.0).sum())
    print(f"S {S:3d} K {K:3d} tf {tf:.2f} thrust {thrust} opts {sorted(opts)} linvt {kw['linear_vt']}: ok {int(oka.sum())}/{int(okb.sum())} of {S}  iters {a.iters.mean():.1f}/{a.iters.max()} vs {b.iters.mean():.1f}/{b.iters.max()}"
          f"  same {(a.iters == b.iters).mean():.2f}  status codes tp {np.unique(b.status).tolist()}", flush=True)
print(tot)
