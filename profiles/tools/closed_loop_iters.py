"""profiling helper: interior-point iteration counts of every solve of the closed MPC loop (bench.py's closed_loop leg)
usage: python profiles/tools/closed_loop_iters.py [S] [segments]      (MPCX_LIB=other.so for an A/B)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
from mpconstellation_amd import _ffi
if os.environ.get("MPCX_LIB"): _ffi.LIB_PATH = os.path.abspath(os.environ["MPCX_LIB"])
from mpconstellation_amd import Satellite, ConstellationMPC
import mpconstellation_amd.constellation_mpc as M
from mpconstellation_amd.constellation import constellation_states
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
segs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
orig = M.mpc_step_batch
def spy(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); dt = time.perf_counter() - t0
    Ks = k.get("Ks")
    print(f"  solve: K {a[0].shape[2]}{'' if Ks is None else f' (ragged {Ks.min()}..{Ks.max()})'} tf_bar {a[2].min():.3f}..{a[2].max():.3f} tf_max {k['options']['tf_max']}"
          f" iters mean {r.iters.mean():.2f} max {r.iters.max()} status!=0 {(r.status != 0).sum()} {dt * 1e3:.1f} ms", flush=True)
    return r
M.mpc_step_batch = spy
st = constellation_states(S)
mpc = ConstellationMPC([Satellite(s[:3].copy(), s[3:6].copy(), float(s[6])) for s in st], base_res=30, tf_horizon=2, tf_interval=1, r_des=1.5, sim_base_res=100)
for i in range(segs):
    print("segment", i); mpc.run_segment(1)
