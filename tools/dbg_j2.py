"""diagnostic (not a test): convergence with the J2 term in the rollout and in the linearisation"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from mpconstellation_amd import mpc_step_batch, _ffi
from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
from mpconstellation_amd.simulator import propagate_batch
S=1024; K=30
y0, consts = normalize_batch(constellation_states(4096, first=0, count=S))
for j2 in (False, True):
    xbar, st, _ = propagate_batch(y0, np.ones(S), consts, (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), K, include_J2=j2)
    ubar = np.ascontiguousarray(tangential_thrust(xbar, 0.5))
    res = mpc_step_batch(xbar, ubar, np.ones(S), consts, np.linalg.norm(xbar[:, :3, -1], axis=1), include_J2=j2)
    u, c = np.unique(res.status, return_counts=True)
    print("J2", j2, dict(zip(u.tolist(), c.tolist())), "iters mean %.1f max %d" % (res.iters.mean(), res.iters.max()), "kkt max %.2e" % res.kkt.max())
