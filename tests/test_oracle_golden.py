"""The CPU oracle (oracle/*.c) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  This pins the discretize half of the hot path
(SURVEY.md §8a D1-D8, S2, U1).  fp64; tolerances are rounding-level."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O

RTOL = 1e-12  # relative to the largest magnitude of the compared array


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_scale_constants(golden_dir):
    c = np.load(os.path.join(golden_dir, "constants_hubble.npz"))
    sc, cst = O.scale(c["state"])
    assert relerr(sc, c["scale"]) < 1e-15
    assert relerr(cst, c["const"]) < 1e-15
    assert abs(cst[0] - 4 * np.pi ** 2) < 1e-12  # MU = 4 pi^2 in designer units


@pytest.mark.parametrize("tag,flags", [("nj2", 0), ("j2", O.FLAG_J2)])
def test_pointwise_functions(golden_dir, tag, flags):
    p = np.load(os.path.join(golden_dir, "pointwise.npz"))
    cst = p["const"]
    for i in range(p["x"].shape[0]):
        x, u, tf = p["x"][i], p["u"][i], float(p["tf"][i])
        f, rc = O.dynamics(x, u, tf, cst, flags)
        assert rc == 0
        assert relerr(f, p["f_" + tag][i]) < RTOL
        assert relerr(O.A_func(x, u, tf, cst, flags), p["A_" + tag][i]) < RTOL
        assert relerr(O.B_func(x, u, tf, cst), p["B_" + tag][i]) < RTOL
        assert relerr(O.xi_func(x, u, tf, cst, flags), p["xi_" + tag][i]) < RTOL
        s, _ = O.dynamics(x, u, 1.0, cst, flags)
        assert relerr(s, p["Sigma_" + tag][i]) < RTOL
        fd, _ = O.dynamics(x, u, tf, cst, O.FLAG_DRAG | O.FLAG_J2)
        assert relerr(fd, p["f_drag_j2"][i]) < RTOL


def test_mass_guard():
    # simulator.py:135-136 raises on m <= 0
    _, rc = O.dynamics(np.array([1, 0, 0, 0, 6, 0, 0.0]), np.zeros(3), 1.0,
                       np.array([39.0, .9, 1e-3, 46.0, .08, 1e-12, 7e6, 3e-17]), 0)
    assert rc == 1


@pytest.mark.parametrize("K", [2, 3, 20, 30, 100])
def test_foh(golden_dir, K):
    fo = np.load(os.path.join(golden_dir, "foh.npz"))
    u = fo[f"u_{K}"]
    for t, v in zip(fo[f"tau_{K}"], fo[f"val_{K}"]):
        o, rc = O.u_foh(t, u)
        assert rc == 0
        assert np.array_equal(o, v)  # bit-exact: same index and same blend arithmetic


DISC_FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "disc_*.npz")))


@pytest.mark.parametrize("fn", DISC_FILES, ids=[os.path.basename(f)[5:-4] for f in DISC_FILES])
def test_discretize(fn):
    d = np.load(fn)
    flags = O.FLAG_J2 if "J2" in fn else 0
    o = O.discretize(d["x"], d["u"], float(d["tf"]), d["const"], flags, dump_nodes=True)
    assert o["status"] == 0
    for k in ("A", "Bp", "Bn", "Sigma", "xi"):
        assert o[k].shape == d[k].shape
        assert relerr(o[k], d[k]) < RTOL, k
    if "node_counts" in d:
        # the adaptive RK45 step sequence of scipy is reproduced node for node
        assert np.array_equal(o["node_counts"], d["node_counts"])
        assert np.array_equal(o["node_nfev"], d["node_nfev"])
        assert np.abs(o["node_t"] - d["node_t"]).max() < 1e-15
        assert np.abs(o["node_y"] - d["node_y"]).max() < 1e-12


def test_constraint_terms(golden_dir):
    for name in ("tan_K20_tf2", "tan_K30_tf1", "tan_K60_tf2", "tan_K100_tf1", "const_K30_tf1"):
        d = np.load(os.path.join(golden_dir, f"disc_{name}.npz"))
        ct = O.constraint_terms(d["x"], d["u"], d["const"][0])
        for k, v in ct.items():
            g = d["ct_" + k]
            assert np.allclose(v, g, rtol=0, atol=1e-13, equal_nan=True), (name, k)


def test_propagate(golden_dir):
    p = np.load(os.path.join(golden_dir, "propagate.npz"))
    cst, y0 = p["const"], p["y0"]
    tan = O.make_ctrl(O.CTRL_TANGENTIAL, (0.5, 0, 0))
    y, rc, ns = O.propagate(y0, 1.0, cst, tan, 30, 0)
    assert rc == 0 and np.abs(y - p["x_tan_plain"]).max() < 1e-12
    y, rc, ns = O.propagate(y0, 1.0, cst, tan, 30, O.FLAG_DRAG | O.FLAG_J2)
    assert rc == 0 and np.abs(y - p["x_tan_dragj2"]).max() < 1e-12
    seq = O.make_ctrl(O.CTRL_SEQUENCE, useq=p["useq"], end_tau=1.0)
    y, rc, ns = O.propagate(y0, 0.8, cst, seq, 32, 0)
    assert rc == 0 and np.abs(y - p["x_seq_full"]).max() < 1e-12
    seq = O.make_ctrl(O.CTRL_SEQUENCE, useq=p["useq"], end_tau=0.6)
    y, rc, ns = O.propagate(y0, 1.0, cst, seq, 40, 0)
    assert rc == 0 and np.abs(y - p["x_seq_tail"]).max() < 1e-12


def test_extract_uk_and_constellation(golden_dir):
    c64 = np.load(os.path.join(golden_dir, "constellation64.npz"))
    tan = O.make_ctrl(O.CTRL_TANGENTIAL, (0.5, 0, 0))
    for i in c64["idx"]:
        st = c64[f"state_{i}"]
        sc, cs = O.scale(st)
        assert relerr(cs, c64[f"const_{i}"]) < 1e-15
        y0 = np.concatenate([st[:3] / sc[0], st[3:6] / sc[2], [st[6] / sc[4]]])
        y, rc, _ = O.propagate(y0, 1.0, cs, tan, 30, 0)
        assert rc == 0 and np.abs(y - c64[f"x_{i}"]).max() < 1e-12
        u = O.extract_uk(c64[f"x_{i}"], c64[f"t_{i}"], tan)
        assert np.abs(u - c64[f"u_{i}"]).max() < 1e-14
        o = O.discretize(c64[f"x_{i}"], c64[f"u_{i}"], 1.0, cs)
        for k in ("A", "Bp", "Bn", "Sigma", "xi"):
            assert relerr(o[k], c64[f"{k}_{i}"]) < RTOL


@pytest.mark.parametrize("steps", [101, 11])
def test_uniform_steps_mode(golden_dir, steps):
    """use_uniform_steps / integrator_steps: the oracle's dense-output quadrature nodes against the reference's arrays"""
    g = np.load(os.path.join(golden_dir, "uniform_steps_K12_tf1.npz"))
    o = O.discretize(g["x"], g["u"], float(g["tf"]), g["const"], uniform_steps=steps)
    assert o["status"] == 0
    for k, key in (("A", "A"), ("Bp", "Bp"), ("Bn", "Bn"), ("Sigma", "Sigma"), ("xi", "xi")):
        ref = g[f"{key}_{steps}"]
        assert np.abs(o[k] - ref).max() <= 2e-15 * np.abs(ref).max(), key


def test_rk23_mode(golden_dir):
    """Discretizer.ivp_solver = 'RK23' (linearize_discretize.py:40,105): the oracle's stepper with scipy's Bogacki-Shampine
    tableau against arrays the reference produced with that setting -- step nodes node for node, the five arrays, and the
    uniform-step mode on the RK23 dense output."""
    g = np.load(os.path.join(golden_dir, "rk23_discretize.npz"))
    for name in g["cases"]:
        x, u, tf = g[f"x_{name}"], g[f"u_{name}"], float(g[f"tf_{name}"])
        o = O.discretize(x, u, tf, g["const"], O.FLAG_RK23, dump_nodes=True)
        assert o["status"] == 0
        assert np.array_equal(o["node_counts"], g[f"node_counts_{name}"]) and np.array_equal(o["node_nfev"], g[f"node_nfev_{name}"])
        # (the same steps; their lengths agree to 1e-11 only: RK23's error estimate is a difference of nearly equal stage
        #  values, so rounding-level differences of the right-hand side reach the step-size factor at 1e-9 relative)
        assert np.abs(o["node_t"] - g[f"node_t_{name}"]).max() < 1e-11 and np.abs(o["node_y"] - g[f"node_y_{name}"]).max() < 1e-10
        for k in ("A", "Bp", "Bn", "Sigma", "xi"):
            assert relerr(o[k], g[f"{k}_{name}"]) < RTOL, (name, k)
        # (not the RK45 result: the quadrature nodes differ)
        assert relerr(O.discretize(x, u, tf, g["const"])["Bp"], g[f"Bp_{name}"]) > 1e-6
    name = "tan_K30_tf1"
    o = O.discretize(g[f"x_{name}"], g[f"u_{name}"], 1.0, g["const"], O.FLAG_RK23, uniform_steps=int(g["uni_steps"]))
    for k in ("A", "Bp", "Bn", "Sigma", "xi"):
        assert relerr(o[k], g["uni_" + k]) < RTOL, k


def test_scipy_zoh_mode(golden_dir):
    """Discretizer(use_scipy_ZOH=True) (linearize_discretize.py:327-329): the reference then evaluates the thrust hold with
    scipy.interpolate.interp1d(kind='linear') instead of u_FOH -- the same piecewise-linear function in another rounding.  The
    reference's own two paths differ by <= 1e-16 relative on these inputs (the foh_* arrays of the same file, asserted here), so
    the FOH formula IS the evaluation of that mode to the last digit or two: the oracle against arrays the reference produced
    with the flag set -- same accepted step nodes, the five arrays to the tolerance every other mode is held to."""
    g = np.load(os.path.join(golden_dir, "scipy_zoh_discretize.npz"))
    for name in g["cases"]:
        x, u, tf = g[f"x_{name}"], g[f"u_{name}"], float(g[f"tf_{name}"])
        o = O.discretize(x, u, tf, g["const"], dump_nodes=True)
        assert o["status"] == 0 and np.array_equal(o["node_counts"], g[f"node_counts_{name}"])
        assert np.abs(o["node_t"] - g[f"node_t_{name}"]).max() < 1e-13
        for k in ("A", "Bp", "Bn", "Sigma", "xi"):
            assert relerr(g[f"foh_{k}_{name}"], g[f"{k}_{name}"]) < 1e-15, (name, k)      # the reference against itself
            assert relerr(o[k], g[f"{k}_{name}"]) < RTOL, (name, k)
