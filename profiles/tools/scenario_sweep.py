"""profiling helper: convergence statistics of the device solver over scenario variations"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mpconstellation_amd import mpc_step_batch, _ffi
from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
from mpconstellation_amd.simulator import propagate_batch
S = 1024
for K in (30, 60):
    for thrust in (0.1, 0.5, 1.5):
        for tf in (0.5, 1.0, 2.0):
            y0, consts = normalize_batch(constellation_states(4096, first=0, count=S))
            xbar, st, _ = propagate_batch(y0, np.full(S, tf), consts, (_ffi.CTRL_TANGENTIAL, np.array([thrust]), 0, None), K)
            ubar = np.ascontiguousarray(tangential_thrust(xbar, thrust))
            r_des = np.linalg.norm(xbar[:, :3, -1], axis=1)
            res = mpc_step_batch(xbar, ubar, np.full(S, tf), consts, r_des)
            u, c = np.unique(res.status, return_counts=True)
            print(f"K {K} thrust {thrust} tf {tf}: status {dict(zip(u.tolist(), c.tolist()))} iters mean {res.iters.mean():.1f} max {res.iters.max()} kkt max {res.kkt.max():.2e} rollout bad {int((st != 0).sum())}")
