#!/usr/bin/env python3
"""Independent cross-check vector for the solve half (NOT from the reference: pyomo/ipopt are not
installed).  The full NLP of optimizer.py:254-603 -- all 24K+1 variables, the polynomial tangential
constraint exactly as written at :492-517, no bound relaxation, no eliminations -- is handed to
scipy.optimize.minimize(method='trust-constr') (scipy's own interior-point / trust-region code) started
from the reference trajectory.  Its answer is stored; tests/test_oracle_solver.py requires the oracle
to agree with it.  Runs in ~2-3 minutes."""
import os
import sys
import time

import numpy as np
from scipy.optimize import NonlinearConstraint, minimize

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle_lib as O   # noqa: E402

OPT = dict(min_mass=0.1, u_lim=[0, 5], r_lim=[0.99, 5], eps_r=0.01, eps_vr=1e-5, eps_vn=1e-5, tf_max=5, w_nu=1000, w_tr=0.002)


def build(K):
    c = np.load(os.path.join(HERE, "constants_hubble.npz"))
    cst, y0 = c["const"], c["x_norm"]
    tan = O.make_ctrl(O.CTRL_TANGENTIAL, (0.5, 0, 0))
    x, rc, _ = O.propagate(y0, 1.0, cst, tan, K)
    u = O.extract_uk(x, np.linspace(0, 1, K), tan)
    d = O.discretize(x, u, 1.0, cst)
    ct = O.constraint_terms(x, u, cst[0])
    return x, u, cst, d, ct


def main(K=10):
    x, u, cst, d, ct = build(K)
    r_des = float(np.linalg.norm(x[:3, -1])); vt_des = np.sqrt(cst[0] / r_des); tfbar = 1.0
    n = 24 * K + 1
    ix = lambda i, k: i * K + k
    iu = lambda i, k: 7 * K + i * K + k
    inu = lambda i, k: 10 * K + i * K + k
    it = lambda i, k: 17 * K + i * K + k
    itf = 24 * K
    o = OPT

    def f(w):
        X = w[:7 * K].reshape(7, K); U = w[7 * K:10 * K].reshape(3, K)
        return w[itf] + o["w_nu"] * w[17 * K:24 * K].sum() + o["w_tr"] * (((X - x) ** 2).sum() + ((U - u) ** 2).sum() + (w[itf] - tfbar) ** 2)

    def gf(w):
        g = np.zeros(n)
        g[:7 * K] = 2 * o["w_tr"] * (w[:7 * K] - x.ravel()); g[7 * K:10 * K] = 2 * o["w_tr"] * (w[7 * K:10 * K] - u.ravel())
        g[17 * K:24 * K] = o["w_nu"]; g[itf] = 1 + 2 * o["w_tr"] * (w[itf] - tfbar)
        return g

    rows = []; rhs = []
    for i in range(7):
        r = np.zeros(n); r[ix(i, 0)] = 1; rows.append(r); rhs.append(x[i, 0])
    for k in range(K - 1):
        for i in range(7):
            r = np.zeros(n); r[ix(i, k + 1)] = 1
            for j in range(7): r[ix(j, k)] -= d["A"][k, i, j]
            for j in range(3): r[iu(j, k)] -= d["Bn"][k, i, j]; r[iu(j, k + 1)] -= d["Bp"][k, i, j]
            r[itf] -= d["Sigma"][i, k]; r[inu(i, k)] -= 1
            rows.append(r); rhs.append(d["xi"][i, k])
    Ce = np.array(rows); de = np.array(rhs)

    def vt(w):
        r = np.array([w[ix(i, K - 1)] for i in range(3)]); v = np.array([w[ix(3 + i, K - 1)] for i in range(3)])
        h = np.cross(r, v); t = np.cross(h, r)
        return (v @ t) ** 2 - vt_des ** 2 * (t @ t)

    def ceq(w): return np.concatenate([Ce @ w - de, [vt(w)]])

    def jeq(w):
        e = 1e-7; g = np.zeros(n)
        for i in range(6):
            wp = w.copy(); wp[ix(i, K - 1)] += e; wm = w.copy(); wm[ix(i, K - 1)] -= e
            g[ix(i, K - 1)] = (vt(wp) - vt(wm)) / (2 * e)
        return np.vstack([Ce, g])

    rows = []; rhs = []
    def add(r, b): rows.append(r); rhs.append(b)
    r = np.zeros(n); r[ix(6, K - 1)] = -1; add(r, -o["min_mass"])
    for k in range(K - 1):
        r = np.zeros(n)
        for i in range(3): r[ix(i, k)] = -ct["rbar_hat"][i, k]
        add(r, -o["r_lim"][0])
    r = np.zeros(n)
    for i in range(3): r[ix(i, K - 1)] = -ct["rf_hat"][i]
    add(r, -(r_des - o["eps_r"]))
    for V, D, Db, eps in (("Vr", "DrVr_DvVr", "DrVr_DvVr_bar", "eps_vr"), ("Vn", "DrVn_DvVn", "DrVn_DvVn_bar", "eps_vn")):
        g = np.zeros(n)
        for i in range(6): g[ix(i, K - 1)] = ct[D][i]
        c0 = ct[V] - ct[Db]
        add(g.copy(), o[eps] - c0); add(-g, o[eps] + c0)
    for k in range(K):
        for i in range(7):
            r = np.zeros(n); r[inu(i, k)] = 1; r[it(i, k)] = -1; add(r, 0.0)
            r = np.zeros(n); r[inu(i, k)] = -1; r[it(i, k)] = -1; add(r, 0.0)
    r = np.zeros(n); r[itf] = -1; add(r, 0.0)
    r = np.zeros(n); r[itf] = 1; add(r, o["tf_max"])
    Gl = np.array(rows); hl = np.array(rhs)
    qidx = [[iu(i, k) for i in range(3)] for k in range(K)] + [[ix(i, k) for i in range(3)] for k in range(K)] + [[ix(i, K - 1) for i in range(3)]]
    qb = [o["u_lim"][1] ** 2] * K + [o["r_lim"][1] ** 2] * K + [(r_des + o["eps_r"]) ** 2]
    qidx = np.array(qidx); qb = np.array(qb)

    def gin(w): return np.concatenate([Gl @ w - hl, (w[qidx] ** 2).sum(1) - qb])

    def jin(w):
        J = np.zeros((len(qb), n))
        for m, idx in enumerate(qidx): J[m, idx] = 2 * w[idx]
        return np.vstack([Gl, J])

    w0 = np.zeros(n); w0[:7 * K] = x.ravel(); w0[7 * K:10 * K] = u.ravel(); w0[itf] = tfbar; w0[17 * K:24 * K] = 1e-3
    from scipy.optimize import BFGS
    nc = [NonlinearConstraint(ceq, 0, 0, jac=jeq, hess=BFGS()), NonlinearConstraint(gin, -np.inf, 0, jac=jin, hess=BFGS())]
    H0 = np.zeros((n, n)); dd = np.zeros(n); dd[:10 * K] = 2 * o["w_tr"]; dd[itf] = 2 * o["w_tr"]; H0[np.diag_indices(n)] = dd
    t0 = time.time()
    res = minimize(f, w0, jac=gf, hess=lambda w: H0, constraints=nc, method="trust-constr",
                   options={"maxiter": 5000, "gtol": 1e-10, "xtol": 1e-13, "barrier_tol": 1e-11})
    print("trust-constr status", res.status, "nit", res.nit, "time", time.time() - t0, "f", res.fun, "tf", res.x[itf],
          "ceq", np.abs(ceq(res.x)).max(), "gin", gin(res.x).max())
    np.savez_compressed(os.path.join(HERE, f"solve_xcheck_K{K}.npz"), x=x, u=u, const=cst, tf=np.float64(tfbar),
                        r_des=np.float64(r_des), X=res.x[:7 * K].reshape(7, K), U=res.x[7 * K:10 * K].reshape(3, K),
                        NU=res.x[10 * K:17 * K].reshape(7, K), tf_opt=np.float64(res.x[itf]), fun=np.float64(res.fun),
                        ceq_max=np.float64(np.abs(ceq(res.x)).max()), gin_max=np.float64(gin(res.x).max()),
                        status=np.int64(res.status), nit=np.int64(res.nit))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
