"""diagnostic (not a test): IPM iteration histogram for a constellation"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mpconstellation_amd import mpc_step_batch, _ffi
from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
from mpconstellation_amd.simulator import propagate_batch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
y0, consts = normalize_batch(constellation_states(S))
xbar, st, _ = propagate_batch(y0, np.ones(S), consts, (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), K)
ubar = tangential_thrust(xbar, 0.5)
r_des = np.linalg.norm(xbar[:, :3, -1], axis=1)
t = time.time(); res = mpc_step_batch(xbar, ubar, np.ones(S), consts, r_des); dt = time.time() - t
it = res.iters
print("S", S, "K", K, "time", dt, "status counts", {int(k): int((res.status == k).sum()) for k in np.unique(res.status)})
print("iters: mean", it.mean(), "p50", np.percentile(it, 50), "p90", np.percentile(it, 90), "p99", np.percentile(it, 99), "max", it.max())
print("hist", np.histogram(it, bins=[0, 30, 40, 50, 60, 80, 100, 150, 201])[0])
worst = np.argsort(it)[-8:]
print("worst sats", worst, it[worst])
np.save("gpurun_out/iters_%d_%d.npy" % (S, K), it)
