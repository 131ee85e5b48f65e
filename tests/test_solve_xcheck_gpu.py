"""The HIP solver (through the C ABI) against INDEPENDENT solutions of the reference's NLP: tests/golden/xcheck_*.npz,
computed by scipy's trust-constr from a second transcription of optimizer.py on the reference's own A/B matrices
(tests/golden/make_nlp_xcheck.py) -- not by this repo's oracle, which is only compared alongside.  Covers
BASELINE configs[0]'s workload (Hubble, tangential 0.5, tf 2, K = 20), the N = 30 horizon, ipopt's zero start,
an optimum that needs virtual control, OptimalController's option set and the convex linearised-vt variant."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import oracle_lib as O
import nlp_ipm as N

pytestmark = pytest.mark.gpu
XCHECK = ["tan_K20_tf2", "tan_K30_tf1", "tan_K30_tf1_zero", "const_K30_tf1", "tan_K20_tf2_linvt", "tan_K30_tf1_linvt",
          "const_K30_tf1_linvt", "tan_K30_tf1_mpc105", "tan_K60_tf2_mpc12", "tan_K60_tf2_mpc15"]
# two solvers at tol 1e-8 on an objective that is flat in u (held by 2 w_tr = 0.004 only): x 1e-5, tf 1e-6, u 5e-4;
# the convex variant has a unique minimiser: x 1e-6, tf 5e-8, u 1e-5
TOL = {"exact": (1e-5, 5e-4, 1e-6), "linvt": (1e-6, 1e-5, 5e-8)}


def load(golden_dir, case):
    f = np.load(os.path.join(golden_dir, f"xcheck_{case}.npz"))
    d = np.load(os.path.join(golden_dir, f"disc_{str(f['fixture'])}.npz"))
    opts = {str(k): float(v) for k, v in zip(f["option_keys"], f["option_vals"])}
    return f, d, opts


def device_solve(d, opts, r_des, variant, **kw):
    from mpconstellation_amd import solve_batch
    x, u = d["x"], d["u"]
    return solve_batch(d["A"][None], d["Bp"][None], d["Bn"][None], d["Sigma"][None], d["xi"][None], x[None], u[None],
                       [float(d["tf"])], d["const"][None], [r_des], options=opts, linear_vt=(variant == "linvt"), **kw)


@pytest.mark.parametrize("case", XCHECK)
def test_device_vs_independent_nlp_solution(golden_dir, case):
    f, d, opts = load(golden_dir, case)
    variant = str(f["variant"])
    res = device_solve(d, opts, float(f["r_des"]), variant, regularised=True)
    assert res.status[0] == 0 and res.kkt[0] <= 1e-8
    tx, tu, ttf = TOL[variant]
    assert np.abs(res.X[0] - f["X"]).max() < tx
    assert np.abs(res.U[0] - f["U"]).max() < tu
    assert np.abs(res.NU[0] - f["NU"]).max() < 1e-6
    assert abs(res.tf[0] - float(f["tf_opt"])) < ttf
    # ... and against the CPU oracle (same algorithm): solver-tolerance agreement, same iteration count give or take
    # the decisions rounding can flip
    x, u, cst = d["x"], d["u"], d["const"]
    P = N.MpcProblem(x, u, float(d["tf"]), cst[0], {k: d[k] for k in ("A", "Bp", "Bn", "Sigma", "xi")},
                     O.constraint_terms(x, u, cst[0]), {"r_des": float(f["r_des"]), **opts}, variant=variant)
    ref = N.solve(P)
    assert ref["status"] == 0
    assert np.abs(res.X[0] - ref["X"]).max() < 5e-6 and abs(res.tf[0] - ref["tf"]) < 5e-6
    # the device's own count of regularised iterations tells a rounding flip of a breakdown decision (both sides
    # regularise, at possibly different iterations) from a path divergence: without regularisation on either side the
    # iteration counts agree to the last-iterate decision and the solutions to 1e-8
    clean = ref["n_regularised"] == 0 and int(res.n_regularised[0]) == 0
    assert abs(int(res.iters[0]) - ref["iters"]) <= (1 if clean else 10)
    if clean and int(res.iters[0]) == ref["iters"]:
        assert np.abs(res.X[0] - ref["X"]).max() < 1e-8 and abs(res.tf[0] - ref["tf"]) < 1e-8


def test_virtual_control_optimum_on_device(golden_dir):
    """test_discretizer.py:59's constant-thrust reference: the optimum keeps |nu|_1 = 0.074 on one position component of
    the last interval (round 1's solver stalled on it)."""
    f, d, opts = load(golden_dir, "const_K30_tf1")
    res = device_solve(d, opts, float(f["r_des"]), "exact")
    assert res.status[0] == 0
    assert abs(np.abs(res.NU[0]).sum() - np.abs(f["NU"]).sum()) < 1e-6 and np.abs(f["NU"]).sum() > 0.07
    assert res.iters[0] <= 60


def test_convex_variant_batch_is_order_independent(golden_dir):
    """linear-vt flag on a batch: every satellite converges, results do not depend on the batch composition."""
    from mpconstellation_amd import solve_batch
    c64 = np.load(os.path.join(golden_dir, "constellation64.npz"))
    idx = list(c64["idx"])
    pack = lambda key: np.stack([c64[f"{key}_{i}"] for i in idx])
    x, u, cs = pack("x"), pack("u"), pack("const")
    r_des = np.linalg.norm(x[:, :3, -1], axis=1)
    args = [pack(k) for k in ("A", "Bp", "Bn", "Sigma", "xi")]
    a = solve_batch(*args, x, u, np.ones(len(idx)), cs, r_des, linear_vt=True)
    assert (a.status == 0).all() and a.kkt.max() <= 1e-8
    rev = slice(None, None, -1)
    b = solve_batch(*[v[rev] for v in args], x[rev], u[rev], np.ones(len(idx)), cs[rev], r_des[rev], linear_vt=True)
    assert np.array_equal(a.X, b.X[rev]) and np.array_equal(a.tf, b.tf[rev])
    e = solve_batch(*args, x, u, np.ones(len(idx)), cs, r_des)                   # the exact variant differs (another constraint)
    assert (e.status == 0).all() and np.abs(e.tf - a.tf).max() > 1e-6


# ---- several satellites sharing one tf (optimizer.py:287): device = batched inner solves at fixed tf + host root search ----
def shared_inputs(golden_dir, case):
    f = np.load(os.path.join(golden_dir, f"xcheck_{case}.npz"))
    ds = [np.load(os.path.join(golden_dir, f"disc_{str(n)}.npz")) for n in f["fixture"]]
    st = lambda k: np.stack([d[k] for d in ds])
    return f, ds, [st(k) for k in ("A", "Bp", "Bn", "Sigma", "xi")], st("x"), st("u"), st("const")


@pytest.mark.parametrize("case", ["shared_tf_K30", "shared_tf_K30_linvt"])
@pytest.mark.parametrize("monolithic", [True, False])
def test_shared_tf_device_vs_monolithic_fixture(golden_dir, case, monolithic):
    """Two satellites in one Optimizer (optimizer.py:287: one tf) against the monolithic 2 (24 K) + 1 variable NLP solved by
    scipy trust-constr (tests/golden/make_nlp_xcheck.py).  monolithic=True: ONE device solve of that NLP
    (MPCX_SOLVE_SHARED_TF, a cooperative launch); False: the decomposition -- batched fixed-tf solves inside a scalar root
    search on the host -- which is also what the CPU oracle runs."""
    from mpconstellation_amd import solve_shared_tf
    f, ds, mats, x, u, cs = shared_inputs(golden_dir, case)
    res, ev = solve_shared_tf(*mats, x, u, np.ones(2), cs, np.full(2, float(f["r_des"])), linear_vt=(str(f["variant"]) == "linvt"),
                              monolithic=monolithic)
    assert (res.status == 0).all() and ev.converged and len(ev) <= (0 if monolithic else 45)
    assert res.tf[0] == res.tf[1] and abs(res.tf[0] - float(f["tf_opt"])) < 1e-6
    if monolithic:
        assert res.iters[0] == res.iters[1] <= 80 and res.kkt[0] == res.kkt[1] <= 1e-8          # one problem, one iteration count
    elif case == "shared_tf_K30":                 # (the convex case's optimum sits on a kink of the value function)
        assert abs(1.0 + res.g_tf.sum()) < 1e-6
    assert np.abs(res.X - f["X"]).max() < 1e-5 and np.abs(res.NU - f["NU"]).max() < 1e-6
    # the oracle runs the decomposition: same tf to the root tolerance
    Ps = [N.MpcProblem(d["x"], d["u"], 1.0, d["const"][0], {k: d[k] for k in ("A", "Bp", "Bn", "Sigma", "xi")},
                       O.constraint_terms(d["x"], d["u"], d["const"][0]), {"r_des": float(f["r_des"])}, variant=str(f["variant"])) for d in ds]
    tf_o, out, _ = N.solve_shared_tf(Ps, 5.0)
    assert abs(res.tf[0] - tf_o) < 1e-6
    for s in range(2):      # (the convex case's optimum sits on a kink of the value function: the decomposition's root is the less accurate side)
        assert np.abs(res.X[s] - out[s]["X"]).max() < (1e-5 if case.endswith("linvt") else 5e-6)


def test_shared_tf_monolithic_batch_properties():
    """The monolithic shared-tf solve on 64 satellites of the constellation (one tf for all of them): it converges, every
    satellite reports the same tf / iteration count, the result satisfies each satellite's constraints, the shared tf lies
    between the smallest of the satellites' own optimal final times and tf_bar (the trust-region terms w_tr (tf - tf_bar)^2
    of all 64 satellites pull on the one variable, the objective's tf only once), two runs are bit-identical (the
    launch-wide reductions fold in a fixed order), and the decomposition (fixed-tf batches + host root search) finds
    the same tf.  One satellite with an empty constraint set makes the whole problem infeasible."""
    from mpconstellation_amd import mpc_step_batch, Discretizer, solve_shared_tf
    from test_full_size_gpu import workload
    S, K = 64, 30
    xbar, ubar, consts, r_des = workload(4096, K, first=0, count=S)
    tf = np.ones(S)
    own = mpc_step_batch(xbar, ubar, tf, consts, r_des)
    sh = mpc_step_batch(xbar, ubar, tf, consts, r_des, shared_tf=True)
    assert (own.status == 0).all() and (sh.status == 0).all() and sh.kkt.max() <= 1e-8
    assert len(set(sh.tf.tolist())) == 1 and len(set(sh.iters.tolist())) == 1 and sh.iters[0] <= 60
    assert own.tf.min() - 1e-6 <= sh.tf[0] <= 1.0 and sh.tf[0] > own.tf.mean()
    rn = np.linalg.norm(sh.X[:, :3, :], axis=1)
    assert np.abs(rn[:, -1] - r_des).max() <= 0.01 + 1e-6 and np.linalg.norm(sh.U, axis=1).max() <= 5 + 1e-6
    assert np.abs(sh.X[:, :, 0] - xbar[:, :, 0]).max() == 0.0
    again = mpc_step_batch(xbar, ubar, tf, consts, r_des, shared_tf=True)
    assert np.array_equal(sh.X, again.X) and np.array_equal(sh.tf, again.tf) and np.array_equal(sh.iters, again.iters)
    A, Bp, Bn, Sig, xi, st = Discretizer(None).discretize_batch(xbar[:8], ubar[:8], tf[:8], consts[:8])
    mono, ev1 = solve_shared_tf(A, Bp, Bn, Sig, xi, xbar[:8], ubar[:8], tf[:8], consts[:8], r_des[:8])
    deco, ev2 = solve_shared_tf(A, Bp, Bn, Sig, xi, xbar[:8], ubar[:8], tf[:8], consts[:8], r_des[:8], monolithic=False)
    assert ev1.converged and ev2.converged and len(ev1) == 0 and len(ev2) >= 3
    assert abs(mono.tf[0] - deco.tf[0]) < 1e-6 and np.abs(mono.X - deco.X).max() < 1e-5
    rd = r_des.copy(); rd[5] = 7.0
    bad = mpc_step_batch(xbar, ubar, tf, consts, rd, shared_tf=True)
    assert (bad.status == 8).all() and (bad.iters == 0).all()


def test_fixed_tf_flag_vs_oracle(golden_dir):
    """MPCX_SOLVE_FIXED_TF: tf held at the value passed in, g_s returned in its place; same iterations as the oracle's
    fixed-tf solve"""
    from mpconstellation_amd import solve_batch
    f, ds, mats, x, u, cs = shared_inputs(golden_dir, "shared_tf_K30")
    held = np.array([0.97, 1.02])
    res = solve_batch(*mats, x, u, np.ones(2), cs, np.full(2, float(f["r_des"])), fixed_tf=held)
    assert (res.status == 0).all() and np.array_equal(res.tf, held)
    for s, d in enumerate(ds):
        P = N.MpcProblem(d["x"], d["u"], 1.0, d["const"][0], {k: d[k] for k in ("A", "Bp", "Bn", "Sigma", "xi")},
                         O.constraint_terms(d["x"], d["u"], d["const"][0]), {"r_des": float(f["r_des"])}, fixed_tf=held[s])
        r = N.solve(P)
        assert r["status"] == 0 and abs(int(res.iters[s]) - r["iters"]) <= 3
        assert np.abs(res.X[s] - r["X"]).max() < 5e-6 and abs(res.g_tf[s] - r["g_tf"]) < 1e-5 * max(1.0, abs(r["g_tf"]))


def test_optimizer_class_shares_tf_by_default(golden_dir):
    """the reference's multi-satellite Optimizer (one tf, get_solved_tf ignores s, optimizer.py:199-203)"""
    from mpconstellation_amd import Optimizer, Discretizer, Simulator, SatelliteScale, Satellite
    f, ds, mats, x, u, cs = shared_inputs(golden_dir, "shared_tf_K30")
    sat = Satellite(np.array([5371.4806, -4133.1393, 1399.9594]) * 1000, np.array([4.6921, 4.9848, -3.2752]) * 1000, 12200)
    scale = SatelliteScale(sat=sat)
    assert np.array_equal(scale.get_normalized_constants().as_vector(), cs[0])
    d = Discretizer(scale.get_normalized_constants())
    opt = Optimizer([x[0], x[1]], [u[0], u[1]], [None, None], 1.0, d, Simulator.satellite_dynamics, scale, verbose=False)
    opt.solve_OPT(input_options={"r_des": float(f["r_des"])})
    assert opt.get_solved_tf(0) == opt.get_solved_tf(1)
    # device discretisation instead of the fixture's (reference) matrices: the same answer at the solver tolerance
    assert abs(opt.get_solved_tf(0) - float(f["tf_opt"])) < 5e-6
    assert np.abs(opt.get_solved_trajectory(1) - f["X"][1]).max() < 1e-5
