"""profiling helper: per-phase cycle shares of solve_kernel from a -DMPCX_PHASE_TIMING build"""
import os, sys, subprocess, ctypes as C
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from mpconstellation_amd import build as b
lib = "/tmp/libmpcx_timing.so"
import glob
srcs = sorted(glob.glob(os.path.join(os.environ["SRC"], "*.hip"))) if os.environ.get("SRC") else b.sources()      # (SRC: another source tree, for an A/B)
subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DMPCX_PHASE_TIMING", "-o", lib] + srcs)
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = lib
if os.environ.get("SRC"):          # (an older source tree lacks the entry points added since: bind what it has)
    _have = C.CDLL(lib)
    _ffi._SIGS = {k: v for k, v in _ffi._SIGS.items() if hasattr(_have, k)}
from mpconstellation_amd import solve_batch
G = os.path.join(ROOT, "tests", "golden")
d = np.load(os.path.join(G, "disc_tan_K30_tf1.npz"))
x, u, tf, cst = d["x"], d["u"], float(d["tf"]), d["const"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rep = lambda a: np.repeat(a[None], S, axis=0)
import time
args = (rep(d["A"]), rep(d["Bp"]), rep(d["Bn"]), rep(d["Sigma"]), rep(d["xi"]), rep(x), rep(u), [tf] * S, rep(cst), [np.linalg.norm(x[:3, -1])] * S)
import ast
opts = ast.literal_eval(os.environ.get("OPTS", "{}"))          # e.g. OptimalController's set: OPTS='{"eps_r": 1e-6, "eps_vr": 1e-16, "tf_max": 1.0}'
if "r_des" in opts: args = args[:-1] + ([opts.pop("r_des")] * S,)
flags = int(os.environ.get("FLAGS", "0"))        # 16: the one-wave kernel for small batches too
if os.environ.get("K"):          # another horizon: the benchmark constellation's first S satellites through the fused step
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_full_size_gpu import workload
    from mpconstellation_amd import mpc_step_batch
    Kk = int(os.environ["K"])
    xb, ub, cs, rd = workload(4096, Kk, first=0, count=S)
    x = xb[0]
    solve_batch = lambda *a, **kw: mpc_step_batch(xb, ub, np.ones(S), cs, rd, **kw)
    args = ()
r = solve_batch(*args, options=opts, flags=flags)
t0 = time.perf_counter(); r = solve_batch(*args, options=opts, flags=flags); wall = time.perf_counter() - t0
print(f"S {S}: host-pointer solve_batch wall {wall*1e3:.3f} ms (copies of {S*30*17*8/1e6:.1f} MB results included)")
names = ["eval_res(E0,Emu,r0)", "newton_blocks", "riccati_factor", "sweep_bwd 8ch", "sweep_fwd 8ch+border", "reduced_residual",
         "start-up: start point", "border_solve+comb fwd", "finish_direction", "start-up: stage transposition", "line-search evals", "start-up: terms (lane 0)"]
t = r.NU[0].ravel()[:24].reshape(12, 2)
tot = t[:, 0].sum()
rt, mt = r.NU[0].ravel()[40:42]
print("iters", r.iters[0], "status", r.status[0], "total s_memtime ticks in phases", tot,
      f"| satellite 0 lived {rt/100:.1f} us (s_memrealtime, 100 MHz) = {mt:.0f} s_memtime ticks -> s_memtime at {mt/(rt/100):.1f} MHz")
for n, (cyc, cnt) in zip(names, t):
    if cnt: print(f"{n:28s} {cyc/tot*100:6.2f}%  calls {int(cnt):5d}  per call {cyc/cnt:10.0f}")
fn = ["fetch issue", "P1-3 Bh,WxBp,LDL,subst + bwd sweep of k+1", "(mark only)", "P4 Pt,G,Minv", "P5", "P6", "P7-9 inv3,P_k,Kg", "(mark only)", "sweep inputs", "stash+sync",
      "fwd: fetch", "fwd: u,x", "fwd: yhat,nu,lam", "fwd: store", "fwd: stash", "-"]
f = r.NU[0].ravel()[24:40]
nn = max(1, int(t[2, 1])) * x.shape[1]
print("-- inside the recursions (cycles per node per call) --")
for n, c in zip(fn, f):
    if c: print(f"   {n:24s} {c/nn:8.0f}")
if flags & 64:
    w = r.NU[0].ravel()[48:64]
    print("-- time-parallel: cycles of the last factorisation command per workgroup (segment: factorisation, sweeps, exchange record + release) --")
    for j in range(3): print(f"   segment {j}: {w[4*j]:8.0f} {w[4*j+1]:8.0f} {w[4*j+2]:8.0f}")
    print(f"   last segment (first workgroup): factorisation {w[12]:8.0f}  sweeps {w[13]:8.0f}  then waited {w[14]:8.0f} for the others")
