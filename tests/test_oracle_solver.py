"""The solve-half oracle (oracle/nlp_ipm.py) checked without ipopt: the tangential-velocity reformulation is
the same constraint, its solutions are KKT points of the restated NLP (independent dense assembly), its
structured linear algebra equals a dense LAPACK solve, and it agrees with scipy's trust-constr run on the
un-eliminated polynomial NLP: fixtures tests/golden/xcheck_*.npz (make_nlp_xcheck.py: an independent transcription of
optimizer.py on the reference's own A/B matrices; default and MPC option sets, the non-zero-virtual-control optimum, the
convex linearised-vt variant, ipopt's zero start) and the older solve_xcheck_K10.npz (make_solver_xcheck.py).  Start
points and the frozen ipopt-default parameter set are checked to give the same solution."""
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
    r = N.solve(P, tol=1e-15, acceptable_tol=1e-6, acceptable_iter=3, max_iter=120)
    assert r["status"] == N.ST_ACCEPTABLE and r["kkt"] <= 1e-6


def test_empty_constraint_set_is_reported_at_once(golden_dir):
    """The virtual control makes every x_1..x_K reachable, so the NLP is infeasible only when the constraint set itself
    is empty; the solver says so (ST_INFEASIBLE) before iterating and hands the reference back.  The terminal-radius case
    is confirmed independently: the smallest violation of its three constraints over r_K (SLSQP on the epigraph form)."""
    from scipy.optimize import minimize
    base = problem(golden_dir, "tan_K30_tf1")
    rK = float(np.linalg.norm(base.xbar[:3, -1]))
    for opts in ({"r_lim": [1.01, 5]}, {"r_des": 5.5}, {"r_lim": [0.99, 1.2], "r_des": 1.5}, {"r_lim": [2.0, 1.5]},
                 {"eps_vr": -1e-3}, {"eps_vn": -1e-3}, {"tf_max": -1.0}):
        P = problem(golden_dir, "tan_K30_tf1", **opts)
        r = N.solve(P)
        assert r["status"] == N.ST_INFEASIBLE and r["iters"] == 0 and r["kkt"] > 0, opts
        assert np.array_equal(r["X"], P.xbar) and np.array_equal(r["U"], P.ubar) and not r["NU"].any() and r["tf"] == P.tfbar
    P = problem(golden_dir, "tan_K30_tf1", r_des=5.5)
    rh = -P.aT[0, :3]
    cons = [{"type": "ineq", "fun": lambda w: w[3] - (-(rh @ w[:3]) - P.bT[0])},
            {"type": "ineq", "fun": lambda w: w[3] - (w[:3] @ w[:3] - P.b_rmax)},
            {"type": "ineq", "fun": lambda w: w[3] - (w[:3] @ w[:3] - P.b_rfmax)}]
    m = minimize(lambda w: w[3], np.r_[5.0 * rh, 1.0], constraints=cons, method="SLSQP", options={"ftol": 1e-12})
    assert m.success and m.fun > 0.1                         # no r_K satisfies all three
    # hard but feasible option sets are NOT flagged: a target far above the orbit, tf_max below the reference time,
    # a thrust limit below the reference thrust, the start exactly on the r_min plane
    for opts in ({"r_des": 3.0}, {"tf_max": 0.5}, {"u_lim": [0, 0.05]}, {"r_lim": [1.0, 5]}, {"r_des": rK, "eps_r": 0.0},
                 {"eps_vr": 0.0, "eps_vn": 0.0}):
        assert problem(golden_dir, "tan_K30_tf1", **opts).structural_violation() <= 0.0, opts


# ---- independent fixtures: scipy trust-constr on a second transcription of optimizer.py (tests/golden/make_nlp_xcheck.py) ----
XCHECK = ["tan_K20_tf2", "tan_K30_tf1", "tan_K30_tf1_zero", "const_K30_tf1", "tan_K20_tf2_linvt", "tan_K30_tf1_linvt",
          "const_K30_tf1_linvt", "tan_K30_tf1_mpc105", "tan_K60_tf2_mpc12", "tan_K60_tf2_mpc15"]
# two solvers, tol 1e-8 each, on an objective that is flat in u (only the trust-region weight 2 w_tr = 0.004 holds it):
# x to 1e-5, tf to 1e-6; u to 5e-4; the convex variant (unique minimiser) much tighter
TOL_X, TOL_U, TOL_TF = 1e-5, 5e-4, 1e-6


def xcheck_problem(golden_dir, case):
    f = np.load(os.path.join(golden_dir, f"xcheck_{case}.npz"))
    d = np.load(os.path.join(golden_dir, f"disc_{str(f['fixture'])}.npz"))
    x, u, tf, cst = d["x"], d["u"], float(d["tf"]), d["const"]
    stage = {k: d[k] for k in ("A", "Bp", "Bn", "Sigma", "xi")}
    opts = {"r_des": float(f["r_des"]), **{str(k): float(v) for k, v in zip(f["option_keys"], f["option_vals"])}}
    return N.MpcProblem(x, u, tf, cst[0], stage, O.constraint_terms(x, u, cst[0]), opts, variant=str(f["variant"])), f


@pytest.mark.parametrize("case", XCHECK)
def test_oracle_vs_independent_nlp_solution(golden_dir, case):
    P, f = xcheck_problem(golden_dir, case)
    assert f["status"] in (1, 2) and f["ceq_max"] < 1e-9 and f["gin_max"] < 1e-9 and f["optimality"] < 5e-8
    r = N.solve(P)
    assert r["status"] == N.ST_OK and r["n_regularised"] == 0
    convex = P.variant == "linvt"
    assert np.abs(r["X"] - f["X"]).max() < (1e-6 if convex else TOL_X)
    assert np.abs(r["U"] - f["U"]).max() < (1e-5 if convex else TOL_U)
    assert np.abs(r["NU"] - f["NU"]).max() < 1e-6
    assert abs(r["tf"] - float(f["tf_opt"])) < (5e-8 if convex else TOL_TF)


def test_virtual_control_optimum(golden_dir):
    """The constant-thrust reference of test_discretizer.py:59 cannot meet the terminal window with thrust alone: the
    optimum keeps a virtual control on one position component of the last interval.  Both solvers find it."""
    P, f = xcheck_problem(golden_dir, "const_K30_tf1")
    r = N.solve(P)
    assert np.abs(f["NU"]).sum() > 0.07 and abs(np.abs(r["NU"]).sum() - np.abs(f["NU"]).sum()) < 1e-6
    k, i = np.unravel_index(np.abs(r["NU"].T).argmax(), (P.K, 7))
    assert k == P.K - 2 and np.abs(r["NU"]).sum() - abs(r["NU"][i, k]) < 1e-6          # a single entry carries it


@pytest.mark.parametrize("case", ["tan_K20_tf2", "tan_K30_tf1", "const_K30_tf1", "tan_K30_tf1_linvt", "tan_K60_tf2_mpc12"])
def test_frozen_ipopt_default_mode_ends_at_the_same_point(golden_dir, case):
    """FAST (adaptive mu, push 1e-4, kappa_Sigma 100 one-sided, mu-based multipliers, centred L1 pairs) against ipopt's
    defaults for the same knobs (monotone mu from 0.1, push 1e-2, kappa_Sigma 1e10, multipliers 1): the same KKT point."""
    P, f = xcheck_problem(golden_dir, case)
    a = N.solve(P); b = N.solve(P, mode="ipopt_default", max_iter=400)
    assert a["status"] == b["status"] == N.ST_OK
    assert np.abs(a["X"] - b["X"]).max() < 5e-6 and abs(a["tf"] - b["tf"]) < 1e-6
    assert np.abs(b["X"] - f["X"]).max() < TOL_X and abs(b["tf"] - float(f["tf_opt"])) < TOL_TF


@pytest.mark.parametrize("case", ["tan_K20_tf2", "tan_K30_tf1_linvt", "const_K30_tf1"])
def test_start_point_does_not_change_the_solution(golden_dir, case):
    """ipopt starts from zeros (pyomo Vars without values, optimizer.py:267-270, 287), oracle and device from the
    reference trajectory.  Evidence that both land on the same local solution: (i) the independent solver does
    (fixture tan_K30_tf1_zero against tan_K30_tf1, test above and here); (ii) the oracle reaches its own solution from
    8 random starts (20 % noise on every state, unit noise on the thrust, tf scaled by 0.5 .. 1.5) and, in the convex
    variant (where the reduced tangential form plays no role and x = 0 is admissible), from all-zeros."""
    P, f = xcheck_problem(golden_dir, case)
    base = N.solve(P)
    rng = np.random.default_rng(5)
    for _ in range(8):
        st = dict(X=P.xbar * (1 + 0.2 * rng.standard_normal(P.xbar.shape)), U=P.ubar + rng.standard_normal(P.ubar.shape),
                  tf=P.tfbar * rng.uniform(0.5, 1.5))
        r = N.solve(P, start=st, max_iter=400)
        assert r["status"] == N.ST_OK
        assert np.abs(r["X"] - base["X"]).max() < 5e-6 and abs(r["tf"] - base["tf"]) < 1e-6
    if P.variant == "linvt":
        for mode in ("fast", "ipopt_default"):
            r = N.solve(P, start="zero", mode=mode, max_iter=400)
            assert r["status"] == N.ST_OK
            assert np.abs(r["X"] - base["X"]).max() < 5e-6 and abs(r["tf"] - base["tf"]) < 1e-6


def test_independent_solver_start_points(golden_dir):
    a = np.load(os.path.join(golden_dir, "xcheck_tan_K30_tf1.npz")); b = np.load(os.path.join(golden_dir, "xcheck_tan_K30_tf1_zero.npz"))
    assert str(b["start"]) == "zero" and np.abs(a["X"] - b["X"]).max() < 1e-6 and abs(float(a["tf_opt"]) - float(b["tf_opt"])) < 1e-7


def test_zero_like_start_exact_variant(golden_dir):
    """In the exact variant the device's reduced tangential form is undefined at r = 0 (the reference's polynomial has a
    zero gradient there); from 1e-3 x_bar with zero thrust and tf = 0 the frozen ipopt-default mode still ends at the
    reference-start solution."""
    P, f = xcheck_problem(golden_dir, "tan_K30_tf1")
    base = N.solve(P)
    r = N.solve(P, start=dict(X=1e-3 * P.xbar, U=np.zeros_like(P.ubar), tf=0.0), mode="ipopt_default", max_iter=600)
    assert r["status"] == N.ST_OK and np.abs(r["X"] - base["X"]).max() < 5e-6 and abs(r["tf"] - base["tf"]) < 1e-6


# ---- several satellites sharing one tf (optimizer.py:287) ----
def shared_problems(golden_dir, case):
    f = np.load(os.path.join(golden_dir, f"xcheck_{case}.npz"))
    Ps = []
    for name in f["fixture"]:
        d = np.load(os.path.join(golden_dir, f"disc_{str(name)}.npz"))
        x, u, tf, cst = d["x"], d["u"], float(d["tf"]), d["const"]
        Ps.append(N.MpcProblem(x, u, tf, cst[0], {k: d[k] for k in ("A", "Bp", "Bn", "Sigma", "xi")},
                               O.constraint_terms(x, u, cst[0]), {"r_des": float(f["r_des"])}, variant=str(f["variant"])))
    return Ps, f


@pytest.mark.parametrize("case", ["shared_tf_K30", "shared_tf_K30_linvt"])
def test_shared_tf_vs_independent_monolithic_solution(golden_dir, case):
    """The fixture is the MONOLITHIC two-satellite NLP (one tf variable, 2 x 24K + 1 unknowns) solved by trust-constr;
    the oracle solves it by the decomposition the device uses (inner problems at fixed tf + scalar root search)."""
    Ps, f = shared_problems(golden_dir, case)
    assert f["X"].shape == (2, 7, 30) and f["optimality"] < 1e-8
    tf, out, ev = N.solve_shared_tf(Ps, 5.0)
    # (the convex case's optimum sits on a kink of the value function -- a virtual-control component switching on -- where
    #  the tf row jumps through zero: the bracketing search then closes in by halving, ~35 inner solves instead of ~8)
    assert abs(tf - float(f["tf_opt"])) < 1e-6 and len(ev) <= 45
    if case == "shared_tf_K30":
        assert abs(1.0 + sum(o["g_tf"] for o in out)) < 1e-6      # the tf stationarity row of the monolithic problem
    for s, o in enumerate(out):
        assert o["status"] == N.ST_OK
        assert np.abs(o["X"] - f["X"][s]).max() < 1e-5 and np.abs(o["NU"] - f["NU"][s]).max() < 1e-6
    # the coupling matters: solved separately the two satellites want different final times
    sep = []
    for P in Ps:
        sep.append(N.solve(P)["tf"])
    assert abs(sep[0] - sep[1]) > 1e-3 and min(sep) - 1e-6 < tf < max(sep) + 1e-6


def test_fixed_tf_inner_problem(golden_dir):
    """at the free-tf optimum the tf stationarity row reads 1 + g = 0, and fixing tf there reproduces the solution"""
    P = problem(golden_dir, "tan_K30_tf1")
    base = N.solve(P)
    Q = problem(golden_dir, "tan_K30_tf1"); Q.fixed_tf = base["tf"]
    r = N.solve(Q)
    assert r["status"] == N.ST_OK and abs(1.0 + r["g_tf"]) < 2e-5 and r["tf"] == base["tf"]
    assert np.abs(r["X"] - base["X"]).max() < 5e-6
