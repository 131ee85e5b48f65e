"""checking helper (GPU box): random satellites, horizons and option sets, device (mpc_step_batch through the C ABI) against the
CPU oracle: status, iteration count, regularised iterations, |dX|, |dtf|.  The oracle is the checker here as in tests/.
usage: python tests/tools/fuzz_vs_oracle.py [n_problems] [seed]      (lives under tests/: it uses oracle/ as the checker)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import multiprocessing as mp
from mpconstellation_amd import _ffi
if os.environ.get("MPCX_LIB"): _ffi.LIB_PATH = os.path.abspath(os.environ["MPCX_LIB"])       # e.g. a -DMPCX_ITER_LOG build
ONLY = int(os.environ["FUZZ_ONLY"]) if os.environ.get("FUZZ_ONLY") else None                 # one problem, with both iteration logs


def make(args):
    import oracle_lib as O
    from mpconstellation_amd import _ffi
    from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
    idx, K, tf, thrust = args
    y0, consts = normalize_batch(constellation_states(4096, first=int(idx), count=1))
    ctrl = O.make_ctrl(_ffi.CTRL_TANGENTIAL, thrust=(thrust, 0, 0))
    x, rc, _ = O.propagate(y0[0], tf, consts[0], ctrl, K)
    assert rc == 0
    return x, np.ascontiguousarray(tangential_thrust(x[None], thrust))[0], consts[0]


def oracle(job):
    import oracle_lib as O, nlp_ipm as N
    x, u, cst, tf, opts = job
    P = N.MpcProblem(x, u, tf, cst[0], O.discretize(x, u, tf, cst), O.constraint_terms(x, u, cst[0]), opts)
    r = N.solve(P, verbose=ONLY is not None)
    return r["status"], r["iters"], r["n_regularised"], r["X"], r["tf"], bool(r["iterate"].clean) if r["iterate"] is not None else False


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    from mpconstellation_amd import mpc_step_batch
    specs = []
    for i in range(n):
        K = int(rng.choice([8, 12, 20, 30, 45]))
        tf = float(rng.choice([0.5, 1.0, 1.0, 2.0]))
        thrust = float(rng.choice([0.1, 0.5, 0.5, 1.0]))
        specs.append((int(rng.integers(0, 4096)), K, tf, thrust))
    with mp.Pool(min(16, os.cpu_count())) as pool:
        data = pool.map(make, specs)
        jobs = []
        for (x, u, cst), (idx, K, tf, thrust) in zip(data, specs):
            rK = float(np.linalg.norm(x[:3, -1]))
            o = {"r_des": rK * float(rng.choice([1.0, 1.0, 1.0, 1.002, 0.99, 1.1]))}
            if rng.random() < 0.3: o["eps_r"] = float(rng.choice([1e-6, 1e-3, 0.05]))
            if rng.random() < 0.2: o["eps_vr"] = 1e-16
            if rng.random() < 0.3: o["tf_max"] = tf * float(rng.choice([1.0, 1.05, 2.0]))
            if rng.random() < 0.15: o["u_lim"] = [0, thrust * float(rng.choice([0.8, 1.5]))]
            if rng.random() < 0.2: o["w_tr"] = float(rng.choice([0.02, 0.2]))
            jobs.append((x, u, cst, tf, o))
        if ONLY is not None: jobs = [jobs[ONLY]]
        t0 = time.time(); ref = pool.map(oracle, jobs); t_or = time.time() - t0
    bad = 0; n_clean = 0
    for j, (job, r) in enumerate(zip(jobs, ref)):
        x, u, cst, tf, o = job
        res = mpc_step_batch(x[None], u[None], [tf], cst[None], [o["r_des"]], options={k: v for k, v in o.items() if k != "r_des"}, regularised=True)
        st, it, nr = int(res.status[0]), int(res.iters[0]), int(res.n_regularised[0])
        if ONLY is not None and os.environ.get("MPCX_LIB"):      # iteration log of a -DMPCX_ITER_LOG build (it overwrites X and U)
            lg = res.X[0].ravel(); lu = res.U[0].ravel()
            for i in range(min(it + 1, lg.size // 5)):
                print(f"device it {i:3d} mu {lg[5*i]:.2e} E0 {lg[5*i+1]:.3e} alpha {lg[5*i+2]:.4f} delta_w {lg[5*i+3]:.1e} first trial alpha {lu[3*i]:.4f} |F|/|F0| {lu[3*i+1]:.4f}")
        both_ok = st == 0 and r[0] == 0
        dx = float(np.abs(res.X[0] - r[3]).max()) if both_ok else float("nan"); dtf = abs(float(res.tf[0]) - r[4]) if both_ok else float("nan")
        n_clean += r[5]
        flag = ""
        if st != r[0]: flag += " STATUS"
        if both_ok and nr == 0 and r[2] == 0 and abs(it - r[1]) > 1: flag += " ITERS"
        if both_ok and (dx > 5e-6 or dtf > 5e-6): flag += " SOLUTION"
        if flag: bad += 1
        print(f"{j:3d} K {x.shape[1]:2d} tf {tf} {o}  clean {int(r[5])} | device st {st} it {it} reg {nr} | oracle st {r[0]} it {r[1]} reg {r[2]} | dX {dx:.1e} dtf {dtf:.1e}{flag}", flush=True)
    print(f"{n} problems ({n_clean} clean starts), oracle {t_or:.0f} s, flagged {bad}")
