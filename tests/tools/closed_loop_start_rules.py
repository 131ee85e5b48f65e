"""Experiments on the interior-point iteration count of the CLOSED LOOP's solves, on the CPU
oracle (lives under tests/: it uses oracle/).  The reference's test_mpc configuration (base_res 30, horizon 2, r_des 1.5,
OptimalController's option set eps_r 1e-6 / eps_vr 1e-16 / tf_max = horizon) poses four solves per satellite and two segments;
`gen` builds them for 32 satellites of the benchmark constellation with the oracle chain (rollouts, discretisation, solve,
re-rollout, truth flight) and stores them; `run` solves all 128 with variants of the start / barrier rules (solve_x: a copy
of nlp_ipm.solve with hooks) and prints mean / max iterations per solve kind and the distance of the solutions from the
base rule's.
usage: python tests/tools/closed_loop_start_rules.py gen [cache.pkl]
       python tests/tools/closed_loop_start_rules.py run "{'base': {}, 'sigma .05': {'sigma': 0.05}}" [cache.pkl]
rule keys: sup_always, sup_after_full, sup_when_feas, sup_end, sigma, mu0, push, push_viol, zcap, soc_ball, tau_min"""
import os, pickle, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import numpy as np
from multiprocessing import Pool
import oracle_lib as O, nlp_ipm as N
from nlp_ipm import *
from mpconstellation_amd.constellation import constellation_states, normalize_batch
CACHE = sys.argv[3] if len(sys.argv) > 3 else (sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "gen" else "/tmp/closed_loop_problems.pkl")

def chain(i):
    st = constellation_states(4096, first=i, count=1)
    y0, cst = normalize_batch(st); y0 = y0[0]; cst = cst[0]
    base_res, horizon, interval, r_des = 30, 2.0, 1.0, 1.5
    probs = []
    for seg in range(2):
        K = int(base_res * horizon)
        ctrl = O.make_ctrl(2, thrust=(0.5, 0.0, 0.0))
        x = O.propagate(y0, horizon, cst, ctrl, K)[0]; t = np.linspace(0, 1, K)
        tf_u = horizon
        for it in range(2):
            u_bar = O.extract_uk(x, t, ctrl)
            d = O.discretize(x, u_bar, tf_u, cst)
            opts = {"r_des": r_des, "eps_r": 0.000001, "eps_vr": 0.0000000000000001, "tf_max": horizon}
            P = N.MpcProblem(x, u_bar, tf_u, cst[0], d, O.constraint_terms(x, u_bar, cst[0]), opts)
            r = N.solve(P)
            probs.append(dict(sat=i, seg=seg, it=it, x=x, u=u_bar, tf=tf_u, mu=cst[0], d={k: d[k] for k in ("A","Bp","Bn","Sigma","xi")},
                              terms=O.constraint_terms(x, u_bar, cst[0]), opts=opts, base_iters=r["iters"], base_status=r["status"],
                              sol=dict(X=r["X"], U=r["U"], tf=r["tf"])))
            tf_u = r["tf"]
            ctrl = O.make_ctrl(3, useq=np.ascontiguousarray(r["U"]), end_tau=1.0)
            Kn = int(base_res * tf_u)
            if it == 0:
                x = O.propagate(y0, tf_u, cst, ctrl, Kn)[0]; t = np.linspace(0, 1, Kn)
        # fly the segment under the truth model (drag + J2) with the plan (end_tau = tf_u / interval)
        ctrl = O.make_ctrl(3, useq=np.ascontiguousarray(r["U"]), end_tau=tf_u / interval)
        y = O.propagate(y0, interval, cst, ctrl, 100, flags=3)[0]
        y0 = y[:, -1].copy()     # (same scale kept: the product re-normalises per segment? see note)
        horizon -= interval
    return probs


def initial_iterate_x(P, prm, rule):
    K = P.K; it = Iterate()
    it.X = P.xbar.copy(); it.U = P.ubar.copy(); it.tf = P.tfbar
    it.NU = np.zeros((7, K - 1)); it.T = np.zeros((7, K - 1))
    it.lam = np.zeros((7, K - 1)); it.lam_vt = 0.0
    g = P.ineq(it.X, it.U, it.NU, it.T, it.tf)
    bnd = {"u": P.b_u, "rmax": P.b_rmax, "rmin": P.b_rmin, "term": P.bT, "rfmax": P.b_rfmax, "tp": 0.0, "tn": 0.0, "tf": P.b_tf}
    push = rule.get("push", prm["bound_push"])
    it.s = {k: np.maximum(-v, push * np.maximum(1.0, np.abs(bnd[k]))) for k, v in g.items()}
    if rule.get("push_viol"):
        for k, v in g.items(): it.s[k] = np.maximum(it.s[k], rule["push_viol"] * np.maximum(v, 0.0))
    it.clean = False; it.mu0 = rule.get("mu0", prm["mu_init"])
    mu0 = it.mu0
    it.z = {k: mu0 / it.s[k] for k in g}
    if rule.get("zcap"):
        for k in g: it.z[k] = np.minimum(it.z[k], rule["zcap"])
    zl = P.w_nu / 2.0
    it.T = np.abs(it.NU) + mu0 / zl
    for k, sg in (("tp", 1.0), ("tn", -1.0)):
        it.s[k] = it.T - sg * it.NU; it.z[k] = np.full_like(it.s[k], zl)
    return it

def candidate_x(P, it, d, a, mu_clip, prm, rule, tau):
    n = step(it, d, a)
    if rule.get("soc_ball"):
        # exact slack update of the quadratic (ball) constraints: g(w + a dw) = g + a dg + a^2 q, q = |d|^2 >= 0
        q = {"u": (d["U"] ** 2).sum(0), "rmax": (d["X"][:3, 1:] ** 2).sum(0), "rfmax": np.array([(d["X"][:3, P.K - 1] ** 2).sum()])}
        th = rule["soc_ball"]
        for k in q:
            sc = n.s[k] - a * a * q[k]
            n.s[k] = np.where(sc >= th * n.s[k], sc, n.s[k])
    g = P.ineq(n.X, n.U, n.NU, n.T, n.tf)
    for k in n.s:
        n.s[k] = np.maximum(n.s[k], -g[k])
        n.z[k] = np.minimum(n.z[k], prm["kappa_sigma"] * mu_clip / n.s[k])
    return n

def solve_x(P, tol=1e-8, max_iter=200, acceptable_tol=1e-6, acceptable_iter=15, n_refine=1, verbose=False, rule=None, dense=False):
    rule = rule or {}
    prm = dict(FAST)
    it = initial_iterate_x(P, prm, rule)
    mu = it.mu0
    n_acc = 0; status = ST_MAXITER; k_it = 0
    mono = False; n_small = 0
    dw_last = 0.0; n_reg = 0; first_reg = -1
    a_prev = 0.0; a_prev2 = 0.0; fast = False
    log = []
    for k_it in range(max_iter + 1):
        E0, dd, pp, cc = optimality_error(P, it, 0.0)
        if not np.isfinite(E0): status = ST_NUMERIC; break
        if E0 <= tol: status = ST_OK; break
        n_acc = n_acc + 1 if E0 <= acceptable_tol else 0
        if n_acc >= acceptable_iter: status = ST_ACCEPTABLE; break
        if k_it == max_iter: status = ST_ACCEPTABLE if E0 <= acceptable_tol else ST_MAXITER; break
        mu_cur = sum((it.s[k] * it.z[k]).sum() for k in it.s) / sum(v.size for v in it.s.values())
        if not mono and n_small >= FB_N:
            mono = True
            mu = max(tol / 10, min(MU_INIT, FB_BOOST * mu_cur))
        if not mono:
            sup = it.clean
            if rule.get("sup_after_full") and a_prev >= rule["sup_after_full"]: sup = True
            if rule.get("sup_always"): sup = True
            if rule.get("sup_end") and mu_cur <= rule["sup_end"] and a_prev >= 0.9: sup = True
            if rule.get("sup_when_feas") and max(dd, pp) <= rule["sup_when_feas"] * mu_cur: sup = True
            sig = rule.get("sigma", SIGMA)
            mu_t = min(sig * mu_cur, mu_cur * np.sqrt(mu_cur)) if sup else sig * mu_cur
            if rule.get("sup2") and sup: mu_t = min(sig * mu_cur, mu_cur ** rule["sup2"])
            mu = max(mu_t, tol / 10, MU_ERR * E0)
        else:
            while mu > tol / 10 and optimality_error(P, it, mu)[0] <= 10.0 * mu:
                mu = max(tol / 10, min(0.2 * mu, mu ** 1.5))
        d = None; dw = 0.0
        while True:
            try:
                d = newton_direction_dense(P, it, mu, dw) if dense else newton_direction(P, it, mu, dw, n_refine if dw == 0 else 0)
                if all(np.isfinite(d[k]).all() for k in ("X", "U", "NU")) and np.isfinite(d["tf"]): break
                d = None
            except np.linalg.LinAlgError:
                d = None
            if dw == 0.0: dw = DW_FIRST if dw_last == 0.0 else max(DW_MIN, dw_last / 3.0)
            else: dw *= 100.0 if dw_last == 0.0 else 8.0
            if dw > DW_MAX: break
        if d is None: status = ST_NUMERIC; break
        if dw > 0.0:
            dw_last = dw; n_reg += 1
            if first_reg < 0: first_reg = k_it
        tau = max(rule.get("tau_min", 0.99), 1 - mu)
        a = 1.0; lim = None
        for nm, v, dv in (("s", it.s, d["s"]), ("z", it.z, d["z"])):
            for k in v:
                neg = dv[k] < 0
                if neg.any():
                    c = (-tau * v[k][neg] / dv[k][neg]).min()
                    if c < a: a = c; lim = nm + ":" + k
        amax = a
        r0 = residual_norm(P, it, mu)
        mu_clip = max(mu, mu_cur)
        n = None; nrej = 0
        for ls in range(30):
            if 0.5 * a < ALPHA_FLOOR: n = None; break
            n = candidate_x(P, it, d, a, mu_clip, prm, rule, tau)
            prod = np.concatenate([(n.s[k] * n.z[k]).ravel() for k in n.s])
            if residual_norm(P, n, mu) <= (1 - 1e-4 * a) * r0 and prod.min() >= GAMMA_NBHD * min(mu, prod.mean()):
                break
            a *= 0.5; nrej += 1
        n_small = n_small + 1 if a < FB_ALPHA else 0
        log.append((k_it, E0, dd, pp, cc, mu, mu_cur, amax, a, lim, dw))
        if verbose: print(f"it {k_it:3d} E0 {E0:.2e} (d {dd:.1e} p {pp:.1e} c {cc:.1e}) mu {mu:.2e} mucur {mu_cur:.2e} amax {amax:.3f} [{lim}] a {a:.3f} rej {nrej} dw {dw:.0e} tf {it.tf:.6f}")
        a_prev2 = a_prev; a_prev = a
        it = n if n is not None else candidate_x(P, it, d, a, mu_clip, prm, rule, tau)
    K = P.K
    return dict(X=it.X, U=it.U, tf=it.tf, status=status, iters=k_it, n_regularised=n_reg, log=log)

probs = None
def run_one(args):
    i, rule = args
    q = probs[i]
    P = N.MpcProblem(q["x"], q["u"], q["tf"], q["mu"], q["d"], q["terms"], q["opts"])
    r = solve_x(P, rule=rule)
    return (q["seg"], q["it"], r["iters"], r["status"], float(np.abs(r["X"] - q["sol"]["X"]).max()), abs(r["tf"] - q["sol"]["tf"]), r["n_regularised"])


if __name__ == "__main__":
    if sys.argv[1] == "gen":
        idx = [int(a) for a in np.linspace(0, 4095, 32)]
        with Pool(min(8, os.cpu_count())) as p: out = p.map(chain, idx)
        allp = [q for c in out for q in c]
        pickle.dump(allp, open(CACHE, "wb"))
        for k in range(4):
            its = [q["base_iters"] for q in allp if (q["seg"], q["it"]) == (k // 2, k % 2)]
            print("segment", k // 2, "SCP iteration", k % 2, "nodes", sorted(set(q["x"].shape[1] for q in allp if (q["seg"], q["it"]) == (k // 2, k % 2))), "iterations mean", np.mean(its), "max", max(its))
    else:
        probs = pickle.load(open(CACHE, "rb"))
        rules = eval(sys.argv[2])
        with Pool(min(8, os.cpu_count())) as pool:
            for name, rule in rules.items():
                t0 = time.time()
                out = pool.map(run_one, [(i, rule) for i in range(len(probs))])
                line = f"{name:28s}"
                for g in ((0, 0), (0, 1), (1, 0), (1, 1)):
                    its = [o[2] for o in out if (o[0], o[1]) == g]
                    line += f" | {np.mean(its):5.2f} max {max(its):2d}"
                its = [o[2] for o in out]
                line += f" || all {np.mean(its):5.2f} max {max(its)} not converged {sum(1 for o in out if o[3] != 0)} dX {max(o[4] for o in out):.1e} dtf {max(o[5] for o in out):.1e} regularised {sum(o[6] for o in out)} ({time.time() - t0:.0f}s)"
                print(line, flush=True)
