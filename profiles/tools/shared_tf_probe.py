"""profiling helper: the monolithic shared-tf solve (MPCX_SOLVE_SHARED_TF) against the satellites' own-tf solves and the
decomposition: iteration counts, regularisation records, wall time.  usage: python profiles/tools/shared_tf_probe.py [S]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_full_size_gpu import workload
from mpconstellation_amd import mpc_step_batch, Discretizer, solve_shared_tf
for S in ([int(a) for a in sys.argv[1:]] or [2, 8, 64, 512]):
    xbar, ubar, consts, r_des = workload(4096, 30, first=0, count=S)
    tf = np.ones(S)
    own = mpc_step_batch(xbar, ubar, tf, consts, r_des, regularised=True)
    mpc_step_batch(xbar, ubar, tf, consts, r_des, shared_tf=True)
    t0 = time.perf_counter(); sh = mpc_step_batch(xbar, ubar, tf, consts, r_des, shared_tf=True, regularised=True); t1 = time.perf_counter()
    print(f"S {S:4d}: own tf in [{own.tf.min():.6f}, {own.tf.max():.6f}] iters mean {own.iters.mean():.1f} max {own.iters.max()} | shared tf {sh.tf[0]:.8f} status {sorted(set(sh.status.tolist()))} "
          f"iters {sh.iters[0]} regularised {sh.n_regularised[0]} (first {sh.first_regularised[0]}) kkt {sh.kkt[0]:.2e}  {1e3*(t1-t0):.2f} ms per call", flush=True)
    if S <= 8:
        A, Bp, Bn, Sig, xi, st = Discretizer(None).discretize_batch(xbar, ubar, tf, consts)
        t0 = time.perf_counter(); deco, ev = solve_shared_tf(A, Bp, Bn, Sig, xi, xbar, ubar, tf, consts, r_des, monolithic=False); t1 = time.perf_counter()
        print(f"        decomposition: tf {deco.tf[0]:.8f} ({len(ev)} inner batched solves, {1e3*(t1-t0):.1f} ms)  |dtf| {abs(deco.tf[0]-sh.tf[0]):.2e}")
