"""The solve-half oracle (oracle/nlp_ipm.py) checked without ipopt: the tangential-velocity reformulation is
the same constraint, its solutions are KKT points of the restated NLP (independent dense assembly), its
structured linear algebra equals a dense LAPACK solve, and it agrees with scipy's trust-constr run on the
un-eliminated polynomial NLP (fixture tests/golden/solve_xcheck_K10.npz, made by make_solver_xcheck.py)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import oracle_lib as O
import nlp_ipm as N


def problem(golden_dir, name, **opts):
    d = np.load(os.path.join(golden_dir, f"disc_{name}.npz"))
    x, u, tf, cst = d["x"], d["u"], float(d["tf"]), d["const"]
    stage = {k: d[k] for k in ("A", "Bp", "Bn", "Sigma", "xi")}            # the reference's own discretisation
    return N.MpcProblem(x, u, tf, cst[0], stage, O.constraint_terms(x, u, cst[0]),
                        {"r_des": float(np.linalg.norm(x[:3, -1])), **opts})


def test_vt_reformulation_is_the_same_constraint():
    rng = np.random.default_rng(3)
    for _ in range(50):
        r = rng.normal(size=3); v = rng.normal(size=3) * 6; vd = rng.uniform(4, 8)
        c, g, H = N.vt_reduced(r, v, vd)
        h = np.cross(r, v)
        # (v.t)^2 - vd^2 |t|^2 = |h|^2 |r|^2 * c~   (optimizer.py:492-517 vs the reduced form)
        assert abs(N.vt_poly(r, v, vd) - (h @ h) * (r @ r) * c) < 1e-9 * max(1.0, abs(N.vt_poly(r, v, vd)))
        z = np.concatenate([r, v]); e = 1e-6
        gn = np.array([(N.vt_reduced((z + e * np.eye(6)[i])[:3], (z + e * np.eye(6)[i])[3:], vd)[0]
                        - N.vt_reduced((z - e * np.eye(6)[i])[:3], (z - e * np.eye(6)[i])[3:], vd)[0]) / (2 * e) for i in range(6)])
        assert np.abs(g - gn).max() < 1e-6 * max(1.0, np.abs(g).max())
        Hn = np.array([(N.vt_reduced((z + e * np.eye(6)[i])[:3], (z + e * np.eye(6)[i])[3:], vd)[1]
                        - N.vt_reduced((z - e * np.eye(6)[i])[:3], (z - e * np.eye(6)[i])[3:], vd)[1]) / (2 * e) for i in range(6)])
        assert np.abs(H - Hn).max() < 1e-5 * max(1.0, np.abs(H).max()) and np.abs(H - H.T).max() < 1e-12


@pytest.mark.parametrize("name", ["tan_K20_tf2", "tan_K30_tf1"])
def test_solution_is_a_kkt_point(golden_dir, name):
    P = problem(golden_dir, name)
    r = N.solve(P)
    assert r["status"] == N.ST_OK and r["kkt"] <= 1e-8
    it = r["iterate"]
    # primal feasibility of the restated NLP (relaxed bounds are 1e-8 wide)
    assert np.abs(P.dyn_residual(r["X"], r["U"], r["NU"][:, :-1], r["tf"])).max() < 1e-8
    g = P.ineq(it.X, it.U, it.NU, it.T, it.tf)
    assert max(v.max() for v in g.values()) < 1e-8
    h = np.cross(r["X"][:3, -1], r["X"][3:6, -1])
    assert abs(N.vt_poly(r["X"][:3, -1], r["X"][3:6, -1], P.vt_des)) < 1e-5          # the polynomial as the reference writes it
    # dual feasibility / complementarity, multipliers non-negative
    dual, prim, comp = N.residual_vectors(P, it, 0.0)
    assert max(np.abs(v).max() for v in dual) < 1e-7
    assert all((z >= 0).all() for z in it.z.values())
    assert max((it.s[k] * it.z[k]).max() for k in it.s) < 1e-7
    # second order: the reduced KKT matrix at the solution has the inertia of a strict local minimiser
    d = N.newton_direction_dense(P, it, 1e-9)
    assert d["inertia_ok"]
    # objective does not exceed the reference trajectory's (which is feasible up to its linearisation defect)
    assert r["tf"] < P.tfbar


def test_structured_solve_equals_dense(golden_dir):
    P = problem(golden_dir, "tan_K30_tf1")
    # unrefined (barrier weights below REFINE_TW) the recursion agrees with LAPACK on the dense matrix to 1e-7 .. 1e-6
    # relative (observed 1.2e-7 at worst on these iterates)
    for cap in (2, 5, 9, 25):
        it = N.solve(P, max_iter=cap)["iterate"]
        a = N.newton_direction(P, it, 1e-3, n_refine=1); b = N.newton_direction_dense(P, it, 1e-3)
        for k in ("X", "U", "NU", "T", "lam"):
            assert np.abs(a[k] - b[k]).max() <= 1e-6 * max(1.0, np.abs(b[k]).max()), (cap, k)
        assert abs(a["tf"] - b["tf"]) <= 1e-6 * max(1.0, abs(b["tf"]))
    full = N.solve(P); dense = N.solve(P, dense=True)
    assert full["status"] == dense["status"] == 0
    # same iteration path, two linear solvers: rounding-level differences of the directions, amplified by the flat objective
    assert np.abs(full["X"] - dense["X"]).max() < 1e-8 and abs(full["tf"] - dense["tf"]) < 1e-8


def test_against_scipy_trust_constr(golden_dir):
    f = np.load(os.path.join(golden_dir, "solve_xcheck_K10.npz"))
    assert f["ceq_max"] < 1e-8 and f["gin_max"] < 1e-8
    x, u, cst = f["x"], f["u"], f["const"]
    P = N.MpcProblem(x, u, float(f["tf"]), cst[0], O.discretize(x, u, float(f["tf"]), cst), O.constraint_terms(x, u, cst[0]),
                     {"r_des": float(f["r_des"])})
    r = N.solve(P)
    assert r["status"] == 0
    # two different algorithms, two formulations of the tangential constraint, tol 1e-8 each: 1e-5 on a flat objective
    assert abs(r["tf"] - float(f["tf_opt"])) < 1e-6
    # u is held only by the trust-region weight 2 w_tr = 0.004, so a KKT error of 1e-8 leaves it loose to ~1e-5
    assert np.abs(r["X"] - f["X"]).max() < 1e-5 and np.abs(r["U"] - f["U"]).max() < 5e-5
    assert abs(r["objective"] - float(f["fun"])) < 5e-6      # includes w_nu * sum(t) with t at the final barrier level


def test_status_codes(golden_dir):
    P = problem(golden_dir, "tan_K20_tf2")
    r = N.solve(P, max_iter=4)
    assert r["status"] == N.ST_MAXITER and r["iters"] == 4
    # a tolerance fp64 does not reach: three consecutive iterates at the acceptable level end the run
    r = N.solve(P, tol=1e-12, acceptable_tol=1e-6, acceptable_iter=3, max_iter=120)
    assert r["status"] == N.ST_ACCEPTABLE and r["kkt"] <= 1e-6
