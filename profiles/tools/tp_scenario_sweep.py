"""profiling helper: the scenario variations of scenario_sweep.py (reference thrust, final time, horizon) on the time-parallel
kernel against the default kernels, in batches of 128 satellites (the time-parallel kernel's limit), 1024 satellites per line"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mpconstellation_amd import mpc_step_batch, _ffi
from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
from mpconstellation_amd.simulator import propagate_batch
S, B = 1024, 128
tot = [0, 0, 0]
for K in (30, 60):
    for thrust in (0.1, 0.5, 1.5):
        for tf in (0.5, 1.0, 2.0):
            y0, consts = normalize_batch(constellation_states(4096, first=0, count=S))
            xbar, st, _ = propagate_batch(y0, np.full(S, tf), consts, (_ffi.CTRL_TANGENTIAL, np.array([thrust]), 0, None), K)
            ubar = np.ascontiguousarray(tangential_thrust(xbar, thrust))
            r_des = np.linalg.norm(xbar[:, :3, -1], axis=1)
            sa, sb, ia, ib, dx = [], [], [], [], 0.0
            for b0 in range(0, S, B):
                sl = slice(b0, b0 + B)
                a = mpc_step_batch(xbar[sl], ubar[sl], np.full(B, tf), consts[sl], r_des[sl])
                b = mpc_step_batch(xbar[sl], ubar[sl], np.full(B, tf), consts[sl], r_des[sl], flags=64)
                sa.append(a.status); sb.append(b.status); ia.append(a.iters); ib.append(b.iters)
                both = (a.status == 0) & (b.status == 0)
                if both.any(): dx = max(dx, float(np.abs(a.X - b.X)[both].max()))
            sa, sb, ia, ib = map(np.concatenate, (sa, sb, ia, ib))
            tot[0] += S; tot[1] += int((sa == 0).sum()); tot[2] += int((sb == 0).sum())
            print(f"K {K} thrust {thrust} tf {tf}: status 0 default {int((sa == 0).sum())} time-parallel {int((sb == 0).sum())} of {S}  iters {ia.mean():.2f}/{ia.max()} vs {ib.mean():.2f}/{ib.max()}"
                  f"  same {(ia == ib).mean():.3f}  within one {(np.abs(ia - ib) <= 1).mean():.3f}  |dX| {dx:.1e}", flush=True)
print(f"total {tot[0]}: status 0 default {tot[1]}, time-parallel {tot[2]}")
