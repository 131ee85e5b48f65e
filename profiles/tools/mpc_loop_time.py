"""profiling helper: where a ConstellationMPC.run_segment spends its time"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from mpconstellation_amd import Satellite, ConstellationMPC
from mpconstellation_amd import constellation_mpc as cm
from mpconstellation_amd.constellation import constellation_states
for S in (64, 1024):
    st = constellation_states(S)
    sats = [Satellite(st[i, :3], st[i, 3:6], st[i, 6]) for i in range(S)]
    mpc = ConstellationMPC(sats, base_res=30, tf_horizon=1, tf_interval=1, r_des=1.05, sim_base_res=30, include_drag=False, include_J2=False)
    acc = {"propagate": 0.0, "mpc_step": 0.0}
    op, om = cm.propagate_batch, cm.mpc_step_batch
    def tp(*a, **k):
        t = time.perf_counter(); r = op(*a, **k); acc["propagate"] += time.perf_counter() - t; return r
    def tm(*a, **k):
        t = time.perf_counter(); r = om(*a, **k); acc["mpc_step"] += time.perf_counter() - t; return r
    cm.propagate_batch, cm.mpc_step_batch = tp, tm
    mpc.run_segment(tf=1)                       # warm-up (library load, first allocations)
    acc = {"propagate": 0.0, "mpc_step": 0.0}
    mpc.horizon = 1
    t0 = time.perf_counter(); mpc.run_segment(tf=1); dt = time.perf_counter() - t0
    cm.propagate_batch, cm.mpc_step_batch = op, om
    print(f"S {S}: run_segment {dt*1e3:.1f} ms = propagate calls {acc['propagate']*1e3:.1f} + mpc_step calls {acc['mpc_step']*1e3:.1f} + host {1e3*(dt-acc['propagate']-acc['mpc_step']):.1f}; status counts {dict(zip(*np.unique(mpc.last_status, return_counts=True)))}")
