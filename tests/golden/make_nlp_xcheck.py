#!/usr/bin/env python3
"""Independent solutions of the reference's SCP subproblem (the NLP of optimizer.py:254-603) for the parity tests of
the solve half.  NOT produced by the reference's own solver (pyomo / ipopt are not installed and cannot be) and NOT by
this repo's oracle or kernel: the NLP is written here a second time, directly from optimizer.py --

  variables   x[7,K] u[3,K] nu[7,K] t[7,K] tf                                   :267-270, 287   (24K+1 unknowns)
  objective   tf + w_nu sum(t) + w_tr (|x-xbar|^2 + |u-ubar|^2 + (tf-tfbar)^2)  :300-325
  equalities  x_0 = xbar_0 :344-345 ; dynamics rows :327-342 ; (v.t)^2 = vt_des^2 |t|^2, t = (r x v) x r  :492-517
  inequal.    m_K >= min_mass :351 ; |u_k|^2 <= u_max^2 :379 ; rbar_hat_k . r_k >= r_min (k < K-1) :384 ;
              |r_k|^2 <= r_max^2 :393 ; rf_hat . r_K >= r_des - eps_r :398 ; |r_K|^2 <= (r_des + eps_r)^2 :403 ;
              |Vr lin| <= eps_vr :406-416 ; |Vn lin| <= eps_vn :436-446 ; -t <= nu <= t (all k) :579-585 ; 0 <= tf <= tf_max :588
  variant "linvt": the quartic equality replaced by the linearised pair the reference keeps commented out at :575-576
              (max_tan_vel_rule / min_tan_vel_rule, :471-489) -- then every constraint is linear or convex quadratic and
              the objective strictly convex in (x, u, tf): the minimiser is unique, any correct solver must find it

-- with no eliminations and no reformulation (bounds as written, except where stated below), and handed to scipy.optimize.minimize
(method='trust-constr': Byrd-Hribar-Nocedal trust-region interior point, scipy's own code) with exact sparse Jacobians
and Hessians.  The dynamics matrices A_k, B_k+-, Sigma_k, xi_k and the constraint terms are the REFERENCE's own
(tests/golden/disc_*.npz, written by make_golden.py from the imported reference), so nothing of this repo's
discretisation enters either.  The tangential gradient is taken by complex-step differentiation of the polynomial
exactly as the reference writes it.

Usage: python tests/golden/make_nlp_xcheck.py [case ...]   (all cases: ~10 min on 8 cores; writes xcheck_<case>.npz)
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
from scipy.optimize import Bounds, LinearConstraint, NonlinearConstraint, minimize

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULTS = dict(min_mass=0.1, u_lim=[0, 5], r_lim=[0.99, 5], r_des=1, eps_r=0.01, eps_vr=0.00001, eps_vn=0.00001,
                eps_vt=0.00001, tf_max=5, w_nu=1000, w_tr=0.002)                                  # optimizer.py:178-188
MPC = dict(eps_r=0.000001, eps_vr=0.0000000000000001, eps_vt=0.01)                                # control.py:192-197

# case -> (golden disc fixture, option overrides, variant, start)
CASES = {
    # BASELINE configs[0]: Hubble, tangential 0.5, tf = 2, base_res = 10 -> K = 20, r_des = |rbar_K| (test_optimizer.py:30-55)
    "tan_K20_tf2": ("tan_K20_tf2", {}, "exact", "ref"),
    "tan_K30_tf1": ("tan_K30_tf1", {}, "exact", "ref"),                # the N = 30 horizon of configs[1..2]
    "tan_K30_tf1_zero": ("tan_K30_tf1", {}, "exact", "zero"),          # ipopt's start: every variable 0 (:267-270)
    "const_K30_tf1": ("const_K30_tf1", {}, "exact", "ref"),            # constant thrust (test_discretizer.py:59): nu != 0 at the optimum
    "tan_K20_tf2_linvt": ("tan_K20_tf2", {}, "linvt", "ref"),
    "tan_K30_tf1_linvt": ("tan_K30_tf1", {}, "linvt", "ref"),
    "const_K30_tf1_linvt": ("const_K30_tf1", {}, "linvt", "ref"),
    # OptimalController's option set (control.py:192-197, tf_max = horizon) raising to r_des = 1.05 / 1.2
    "tan_K30_tf1_mpc105": ("tan_K30_tf1", {**MPC, "r_des": 1.05, "tf_max": 1}, "exact", "ref"),
    "tan_K60_tf2_mpc12": ("tan_K60_tf2", {**MPC, "r_des": 1.2, "tf_max": 2}, "exact", "ref"),
    # the reference's own test_mpc (test_simulator.py:79-98): base_res 30, tf_horizon 2 -> K = 60, default r_des = 1.5
    # (control.py:147) -- out of reach of the thrust limit within tf_max: the optimum needs virtual control
    "tan_K60_tf2_mpc15": ("tan_K60_tf2", {**MPC, "r_des": 1.5, "tf_max": 2}, "exact", "ref"),
    # two satellites in ONE Optimizer: they share the final-time variable (optimizer.py:287); tangential and constant-thrust
    # references, r_des of the first for both (one option set)
    "shared_tf_K30": (("tan_K30_tf1", "const_K30_tf1"), {}, "exact", "ref"),
    "shared_tf_K30_linvt": (("tan_K30_tf1", "const_K30_tf1"), {}, "linvt", "ref"),
    # (its second segment, K = 30 with tf_max = 1, ends with tf on its bound and a saturated thrust arc: trust-constr
    #  does not reach its own tolerances on it within 12 000 iterations, so there is no fixture for it)
}


def vt_poly(z, vt_des):
    """optimizer.py:492-517 as written (works for complex z: complex-step differentiation)"""
    r1, r2, r3, v1, v2, v3 = z
    h1 = (r2 * v3) - (r3 * v2); h2 = (r3 * v1) - (r1 * v3); h3 = (r1 * v2) - (r2 * v1)
    t1 = (h2 * r3) - (h3 * r2); t2 = (h3 * r1) - (h1 * r3); t3 = (h1 * r2) - (h2 * r1)
    norm_t2 = t1 ** 2 + t2 ** 2 + t3 ** 2
    vt_act = v1 * t1 + v2 * t2 + v3 * t3
    return vt_act ** 2 - (vt_des ** 2) * norm_t2


def vt_grad(z, vt_des):
    g = np.zeros(6)
    for i in range(6):
        zc = z.astype(complex); zc[i] += 1e-30j
        g[i] = vt_poly(zc, vt_des).imag / 1e-30
    return g


def vt_hess(z, vt_des):
    H = np.zeros((6, 6)); e = 1e-5
    for i in range(6):
        zp = z.copy(); zp[i] += e; zm = z.copy(); zm[i] -= e
        H[i] = (vt_grad(zp, vt_des) - vt_grad(zm, vt_des)) / (2 * e)
    return 0.5 * (H + H.T)


def solve_case(case):
    fixtures, over, variant, start = CASES[case]
    fixtures = [fixtures] if isinstance(fixtures, str) else list(fixtures)
    S = len(fixtures)
    ds = [np.load(os.path.join(HERE, f"disc_{f}.npz")) for f in fixtures]
    K = ds[0]["x"].shape[1]; tfb = float(ds[0]["tf"]); cst = ds[0]["const"]
    assert all(d["x"].shape[1] == K and float(d["tf"]) == tfb and np.array_equal(d["const"], cst) for d in ds)
    # several satellites in one Optimizer: ONE tf variable, one option set, one constant set (optimizer.py:287, 234, 29)
    o = {**DEFAULTS, "r_des": float(np.linalg.norm(ds[0]["x"][:3, -1])), **over}
    # OptimalController asks for a radial-velocity window of +-1e-16 (control.py:195), narrower than fp64 resolves around
    # values of O(1): the two one-sided rows then have no interior and the problem as written has no interior-point
    # solution.  ipopt widens every inequality bound b by bound_relax_factor * max(1, |b|) = 1e-8 before it starts (its
    # documented default); for such option sets the fixture is the solution of that relaxed problem (window_relax in the
    # file says so); fixtures with the reference's default options are unrelaxed
    relax = 1e-8 if min(o["eps_vr"], o["eps_vn"]) < 1e-8 else 0.0
    vt_des = np.sqrt(cst[0] / o["r_des"])                                                       # :283
    nb = 24 * K; n = S * nb + 1; itf = S * nb
    ix = lambda s, i, k: s * nb + i * K + k
    iu = lambda s, i, k: s * nb + 7 * K + i * K + k
    inu = lambda s, i, k: s * nb + 10 * K + i * K + k
    it = lambda s, i, k: s * nb + 17 * K + i * K + k
    wb = np.zeros(n); dq = np.zeros(n); lin = np.zeros(n)
    for s, d in enumerate(ds):
        wb[s * nb:s * nb + 7 * K] = d["x"].ravel(); wb[s * nb + 7 * K:s * nb + 10 * K] = d["u"].ravel()
        dq[s * nb:s * nb + 10 * K] = 2 * o["w_tr"]; lin[s * nb + 17 * K:s * nb + 24 * K] = o["w_nu"]
    wb[itf] = tfb; dq[itf] = 2 * o["w_tr"] * S; lin[itf] = 1.0          # J_trust_s carries (tf - tf_bar)^2 for every s (:322)
    f = lambda w: lin @ w + 0.5 * dq @ (w - wb) ** 2
    gf = lambda w: lin + dq * (w - wb)
    H0 = sp.diags(dq).tocsr()

    rows, cols, vals, rhs = [], [], [], []
    def put(r, c, v): rows.append(r); cols.append(c); vals.append(v)
    # ---- linear equalities: initial state, dynamics ----
    m = 0
    for s, d in enumerate(ds):
        A, Bp, Bn, Sig, xi, xb = d["A"], d["Bp"], d["Bn"], d["Sigma"], d["xi"], d["x"]
        for i in range(7):
            put(m, ix(s, i, 0), 1.0); rhs.append(xb[i, 0]); m += 1
        for k in range(K - 1):
            for i in range(7):
                put(m, ix(s, i, k + 1), 1.0)
                for j in range(7): put(m, ix(s, j, k), -A[k, i, j])
                for j in range(3): put(m, iu(s, j, k), -Bn[k, i, j]); put(m, iu(s, j, k + 1), -Bp[k, i, j])
                put(m, itf, -Sig[i, k]); put(m, inu(s, i, k), -1.0)
                rhs.append(xi[i, k]); m += 1
    Ce = sp.csr_matrix((vals, (rows, cols)), shape=(m, n)); de = np.array(rhs)

    # ---- linear inequalities  G w <= h ----
    rows, cols, vals, rhs = [], [], [], []
    m = 0
    for s, d in enumerate(ds):
        ct = {k[3:]: d[k] for k in d.files if k.startswith("ct_")}
        put(m, ix(s, 6, K - 1), -1.0); rhs.append(-o["min_mass"]); m += 1
        for k in range(K - 1):
            for i in range(3): put(m, ix(s, i, k), -ct["rbar_hat"][i, k])
            rhs.append(-o["r_lim"][0]); m += 1
        for i in range(3): put(m, ix(s, i, K - 1), -ct["rf_hat"][i])
        rhs.append(-(o["r_des"] - o["eps_r"])); m += 1
        for V, D, Db, eps in (("Vr", "DrVr_DvVr", "DrVr_DvVr_bar", "eps_vr"), ("Vn", "DrVn_DvVn", "DrVn_DvVn_bar", "eps_vn")):
            c0 = float(ct[V]) - float(ct[Db])
            for sgn in (1.0, -1.0):
                for i in range(6): put(m, ix(s, i, K - 1), sgn * ct[D][i])
                rhs.append(o[eps] - sgn * c0); m += 1
        if variant == "linvt":
            # :471-489: |Vt_lin(x_K) - Vc_lin(x_K)| <= eps_vt
            a = np.array(ct["DrVt_DvVt"], dtype=float).copy(); a[:3] -= ct["DrVc"]
            c0 = float(ct["Vt"]) - float(ct["DrVt_DvVt_bar"]) - float(ct["Vc"]) + float(ct["DrVc_rbar"])
            for sgn in (-1.0, 1.0):                         # max_tan_vel_rule (sgn -1), min_tan_vel_rule (sgn +1)
                for i in range(6): put(m, ix(s, i, K - 1), sgn * a[i])
                rhs.append(o["eps_vt"] - sgn * c0); m += 1
        for k in range(K):
            for i in range(7):
                put(m, inu(s, i, k), 1.0); put(m, it(s, i, k), -1.0); rhs.append(0.0); m += 1
                put(m, inu(s, i, k), -1.0); put(m, it(s, i, k), -1.0); rhs.append(0.0); m += 1
    Gl = sp.csr_matrix((vals, (rows, cols)), shape=(m, n)); hl = np.array(rhs)
    hl = hl + relax * np.maximum(1.0, np.abs(hl))

    # ---- quadratic balls ----
    qidx = []; qb = []
    for s in range(S):
        qidx += [[iu(s, i, k) for i in range(3)] for k in range(K)] + [[ix(s, i, k) for i in range(3)] for k in range(K)] \
            + [[ix(s, i, K - 1) for i in range(3)]]
        qb += [o["u_lim"][1] ** 2] * K + [o["r_lim"][1] ** 2] * K + [(o["r_des"] + o["eps_r"]) ** 2]
    qidx = np.array(qidx); qb = np.array(qb)
    qb = qb + relax * np.maximum(1.0, np.abs(qb))
    nq = len(qb)
    qfun = lambda w: (w[qidx] ** 2).sum(1) - qb
    def qjac(w):
        return sp.csr_matrix((2 * w[qidx].ravel(), (np.repeat(np.arange(nq), 3), qidx.ravel())), shape=(nq, n))
    def qhess(w, v):
        dd = np.zeros(n); np.add.at(dd, qidx.ravel(), np.repeat(2 * v, 3))
        return sp.diags(dd).tocsr()

    cons = [LinearConstraint(Ce, de, de), LinearConstraint(Gl, -np.inf, hl),
            NonlinearConstraint(qfun, -np.inf, 0.0, jac=qjac, hess=qhess)]
    iKs = [np.array([ix(s, i, K - 1) for i in range(6)]) for s in range(S)]
    if variant == "exact":
        # scaled by 1/|h|^2|r|^2 at the reference point only to give the solver an O(1) row (a constant factor)
        scs = [1.0 / max(1e-12, abs((np.cross(d["x"][:3, -1], d["x"][3:6, -1]) ** 2).sum() * (d["x"][:3, -1] ** 2).sum())) for d in ds]
        vfun = lambda w: np.array([scs[s] * vt_poly(w[iKs[s]], vt_des) for s in range(S)])
        def vjac(w):
            g = np.concatenate([scs[s] * vt_grad(w[iKs[s]], vt_des) for s in range(S)])
            return sp.csr_matrix((g, (np.repeat(np.arange(S), 6), np.concatenate(iKs))), shape=(S, n))
        def vhess(w, v):
            H = sp.csr_matrix((n, n))
            for s in range(S):
                Hs = scs[s] * v[s] * vt_hess(w[iKs[s]], vt_des)
                H = H + sp.csr_matrix((Hs.ravel(), (np.repeat(iKs[s], 6), np.tile(iKs[s], 6))), shape=(n, n))
            return H
        cons.append(NonlinearConstraint(vfun, 0.0, 0.0, jac=vjac, hess=vhess))
    lb = np.full(n, -np.inf); ubd = np.full(n, np.inf)
    lb[itf] = 0.0 - relax; ubd[itf] = o["tf_max"] + relax * max(1.0, abs(o["tf_max"]))

    if start == "ref":
        w0 = wb.copy()
    else:
        w0 = np.zeros(n); w0[itf] = 1e-3                 # pyomo Vars without initial values: ipopt starts from 0
    for s in range(S): w0[s * nb + 17 * K:s * nb + 24 * K] = 1e-3
    t0 = time.time()
    res = minimize(f, w0, jac=gf, hess=lambda w: H0, constraints=cons, bounds=Bounds(lb, ubd), method="trust-constr",
                   options={"maxiter": 12000, "gtol": 1e-11, "xtol": 1e-14, "barrier_tol": 1e-12, "sparse_jacobian": True,
                            # (the relaxed windows are 2e-8 wide: a first barrier parameter of 0.1 sends trust-constr far
                            # from them and it does not come back within the iteration limit; 1e-3 does)
                            "initial_barrier_parameter": 1e-3 if relax else 0.1,
                            "initial_barrier_tolerance": 1e-3 if relax else 0.1, "initial_tr_radius": 1.0})
    w = res.x
    blk = lambda a, b, r: np.stack([w[s * nb + a * K:s * nb + b * K].reshape(r, K) for s in range(S)])
    X, U, NU, T = blk(0, 7, 7), blk(7, 10, 3), blk(10, 17, 7), blk(17, 24, 7)
    if S == 1: X, U, NU, T = X[0], U[0], NU[0], T[0]
    ceq = np.abs(Ce @ w - de).max()
    cvt = max(abs(vt_poly(w[iKs[s]], vt_des)) for s in range(S)) if variant == "exact" else 0.0
    gin = max((Gl @ w - hl).max(), qfun(w).max(), lb[itf] - w[itf], w[itf] - ubd[itf])
    print(f"{case}: status {res.status} nit {res.nit} time {time.time() - t0:.1f}s f {res.fun:.10f} tf {w[itf]:.9f} "
          f"|nu|_1 {np.abs(NU).sum():.3e} ceq {ceq:.1e} vt {cvt:.1e} gin {gin:.1e} optimality {res.optimality:.1e}", flush=True)
    np.savez_compressed(os.path.join(HERE, f"xcheck_{case}.npz"), fixture=np.array(fixtures[0] if S == 1 else fixtures),
                        variant=np.array(variant), start=np.array(start), window_relax=np.float64(relax),
                        option_keys=np.array(sorted(over)), option_vals=np.array([float(over[k]) for k in sorted(over)]),
                        r_des=np.float64(o["r_des"]), X=X, U=U, NU=NU, T=T, tf_opt=np.float64(w[itf]), fun=np.float64(res.fun),
                        ceq_max=np.float64(ceq), vt_abs=np.float64(cvt), gin_max=np.float64(gin), status=np.int64(res.status),
                        nit=np.int64(res.nit), optimality=np.float64(res.optimality))
    return res


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    if len(names) > 1:
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=min(6, len(names))) as pool:
            list(pool.map(solve_case, names))
    else:
        solve_case(names[0])
