"""profiling helper: the OptimalController option set (eps_r 1e-6, eps_vr 1e-16, tf_max = horizon; control.py:192-197) on the
time-parallel kernel against the default kernels, 128 satellites per line (the time-parallel kernel's batch limit)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mpconstellation_amd import mpc_step_batch, _ffi
from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
from mpconstellation_amd.simulator import propagate_batch
S = 128
for first in (0, 1024):
    y0, consts = normalize_batch(constellation_states(4096, first=first, count=S))
    for K, tf in ((30, 2.0), (60, 2.0), (30, 1.0)):
        for r_des in (1.05, 1.2, 1.5):
            xbar, st, _ = propagate_batch(y0, np.full(S, tf), consts, (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), K)
            ubar = np.ascontiguousarray(tangential_thrust(xbar, 0.5))
            opts = {"eps_r": 0.000001, "eps_vr": 0.0000000000000001, "tf_max": tf}
            a = mpc_step_batch(xbar, ubar, np.full(S, tf), consts, np.full(S, r_des), options=opts)
            b = mpc_step_batch(xbar, ubar, np.full(S, tf), consts, np.full(S, r_des), options=opts, flags=64)
            sa = dict(zip(*[v.tolist() for v in np.unique(a.status, return_counts=True)])); sb = dict(zip(*[v.tolist() for v in np.unique(b.status, return_counts=True)]))
            both = (a.status == 0) & (b.status == 0)
            print(f"first {first:4d} K {K} tf {tf} r_des {r_des}: status default {sa} time-parallel {sb}  iters {a.iters.mean():.1f}/{a.iters.max()} vs {b.iters.mean():.1f}/{b.iters.max()}"
                  f" same {(a.iters == b.iters).mean():.2f}  |dX| {np.abs(a.X - b.X)[both].max():.1e} |dtf| {np.abs(a.tf - b.tf)[both].max():.1e}", flush=True)
