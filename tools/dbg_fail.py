"""diagnostic (not a test): find a satellite the device solver fails on in an off-benchmark scenario and log its iterations"""
import sys, os, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from mpconstellation_amd import build as b
LOG = os.environ.get("ITERLOG") == "1"
if LOG:
    lib = "/tmp/libmpcx_iterlog.so"
    subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DMPCX_ITER_LOG", "-o", lib] + b.sources())
    from mpconstellation_amd import _ffi
    _ffi.LIB_PATH = lib
from mpconstellation_amd import mpc_step_batch, _ffi
from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
from mpconstellation_amd.simulator import propagate_batch
K, thrust, tf = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
sats = [int(a) for a in sys.argv[4:]] or list(range(0, 1024, 8))
xs = []; 
y0 = []; cs = []
for s in sats:
    y, c = normalize_batch(constellation_states(4096, first=s, count=1)); y0.append(y[0]); cs.append(c[0])
y0 = np.array(y0); consts = np.array(cs); S = len(sats)
xbar, st, _ = propagate_batch(y0, np.full(S, tf), consts, (_ffi.CTRL_TANGENTIAL, np.array([thrust]), 0, None), K)
ubar = np.ascontiguousarray(tangential_thrust(xbar, thrust))
r_des = np.linalg.norm(xbar[:, :3, -1], axis=1)
res = mpc_step_batch(xbar, ubar, np.full(S, tf), consts, r_des)
bad = [sats[i] for i in range(S) if res.status[i] != 0]
print("failing satellites:", bad[:40], "of", S)
if LOG:
    for i in range(min(S, 2)):
        print("sat", sats[i], "status", res.status[i], "iters", res.iters[i], "kkt", res.kkt[i])
        lg = res.X[i].ravel()
        for it in range(min(int(res.iters[i]) + 1, lg.size // 5)):
            print(f"  it {it:3d} mu {lg[5*it]:.1e} E0 {lg[5*it+1]:.3e} alpha {lg[5*it+2]:.4f} delta_w {lg[5*it+3]:.1e} fails {int(lg[5*it+4]):06d}")
