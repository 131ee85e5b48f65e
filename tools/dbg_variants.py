"""diagnostic (not a test): solver-constant variants.  Usage: dbg_variants.py name=kMuInit:0.3,kAlphaFloor:0.5 ...
Each variant is compiled from a patched copy of solve.hip into /tmp and summarised by iterations on the benchmark
workload and by convergence over the scenario sweeps of dbg_robust.py / dbg_robust_mpc.py (compact)."""
import os, re, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mpconstellation_amd import build as b
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, "{root}")
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = "{lib}"
from mpconstellation_amd import mpc_step_batch
from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
from mpconstellation_amd.simulator import propagate_batch
S = 1024
y0, consts = normalize_batch(constellation_states(4096, first=0, count=S))
def ref(K, thrust, tf):
    xbar, st, _ = propagate_batch(y0, np.full(S, tf), consts, (_ffi.CTRL_TANGENTIAL, np.array([thrust]), 0, None), K)
    return xbar, np.ascontiguousarray(tangential_thrust(xbar, thrust))
xbar, ubar = ref(30, 0.5, 1.0)
r = mpc_step_batch(xbar, ubar, np.ones(S), consts, np.linalg.norm(xbar[:, :3, -1], axis=1))
out = ["bench it %.2f max %d ok %d" % (r.iters.mean(), r.iters.max(), (r.status == 0).sum())]
bad = 0; its = []; mx = 0
for K in (30, 60):
    for thrust in (0.1, 0.5, 1.5):
        for tf in (0.5, 1.0, 2.0):
            xbar, ubar = ref(K, thrust, tf)
            r = mpc_step_batch(xbar, ubar, np.full(S, tf), consts, np.linalg.norm(xbar[:, :3, -1], axis=1))
            bad += int((r.status != 0).sum()); its.append(r.iters.mean()); mx = max(mx, int(r.iters.max()))
out.append("sweep bad %d it %.1f..%.1f max %d" % (bad, min(its), max(its), mx))
ok = acc = fail = 0; its = []
for K, tf in ((30, 2.0), (60, 2.0), (30, 1.0)):
    for r_des in (1.05, 1.2):
        xbar, ubar = ref(K, 0.5, tf)
        r = mpc_step_batch(xbar, ubar, np.full(S, tf), consts, np.full(S, r_des), options={{"eps_r": 1e-6, "eps_vr": 1e-16, "tf_max": tf}})
        ok += int((r.status == 0).sum()); acc += int((r.status == 7).sum()); fail += int(((r.status != 0) & (r.status != 7)).sum()); its.append(r.iters.mean())
out.append("mpc-options ok %d acceptable %d failed %d it %.1f..%.1f" % (ok, acc, fail, min(its), max(its)))
print("{name}: " + " | ".join(out), flush=True)
'''
src = open(os.path.join(ROOT, "mpconstellation_amd", "csrc", "solve.hip")).read()
for spec in sys.argv[1:]:
    name, _, kv = spec.partition("=")
    t = src
    for item in filter(None, kv.split(",")):
        k, v = item.split(":")
        t, n = re.subn(rf"\b{k} = [0-9.e+-]+", f"{k} = {v}", t, count=1)
        assert n == 1, k
    path = f"/tmp/solve_{name}.hip"
    open(path, "w").write(t)
    lib = f"/tmp/libmpcx_{name}.so"
    srcs = [p for p in b.sources() if not p.endswith("solve.hip")] + [path]
    subprocess.check_call([b.HIPCC] + b.FLAGS + ["-I", os.path.join(ROOT, "mpconstellation_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-o", lib] + srcs)
    subprocess.check_call([sys.executable, "-c", CHILD.format(root=ROOT, lib=lib, name=spec)])
