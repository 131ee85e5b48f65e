"""HIP discretizer (through the C ABI) against the golden vectors of the reference and against
the CPU oracle on seeded inputs.  fp64; tolerance 1e-10 relative to each array's magnitude
(the only differences are fused multiply-adds, pow and summation order)."""
import glob
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
RTOL = 1e-10
KEYS = ("A", "Bp", "Bn", "Sigma", "xi")


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


class _Const:
    def __init__(self, v):
        self.v = np.asarray(v, dtype=np.float64)

    def as_vector(self):
        return self.v


def satellite_dynamics(*a, **k):  # token accepted by Discretizer.discretize
    raise RuntimeError("host dynamics are never called")


DISC_FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "disc_*.npz")))


@pytest.mark.parametrize("fn", DISC_FILES, ids=[os.path.basename(f)[5:-4] for f in DISC_FILES])
def test_discretize_vs_reference_golden(fn):
    from mpconstellation_amd import Discretizer
    d = np.load(fn)
    disc = Discretizer(_Const(d["const"]), include_J2=("J2" in fn))
    out = disc.discretize(satellite_dynamics, d["x"], d["u"], float(d["tf"]))
    for k, o in zip(KEYS, out):
        assert o.shape == d[k].shape
        assert relerr(o, d[k]) < RTOL, k


def test_constellation_batch_vs_golden_and_oracle(golden_dir):
    from mpconstellation_amd import Discretizer
    c64 = np.load(os.path.join(golden_dir, "constellation64.npz"))
    idx = list(c64["idx"])
    x = np.stack([c64[f"x_{i}"] for i in idx]); u = np.stack([c64[f"u_{i}"] for i in idx])
    cs = np.stack([c64[f"const_{i}"] for i in idx])
    disc = Discretizer(_Const(cs[0]))
    A, Bp, Bn, Sig, xi, st = disc.discretize_batch(x, u, np.ones(len(idx)), cs)
    assert (st == 0).all()
    for n, i in enumerate(idx):
        for k, o in zip(KEYS, (A[n], Bp[n], Bn[n], Sig[n], xi[n])):
            assert relerr(o, c64[f"{k}_{i}"]) < RTOL, (i, k)


def test_random_batch_vs_oracle():
    """Seeded ragged-ish batch: different tf, thrust tables and constants per satellite; S*(K-1)
    not a multiple of the 8 groups per wave."""
    from mpconstellation_amd import Discretizer
    rng = np.random.default_rng(99)
    S, K = 5, 12
    cst = np.array([39.47841760435743, 0.92, 1.08262668E-3, 46.5, 0.0873, 1e-12, 6.9e6, 3.7e-17])
    tan = O.make_ctrl(O.CTRL_TANGENTIAL, (0.5, 0, 0))
    xs, us, tfs, css = [], [], [], []
    for s in range(S):
        v = 2 * np.pi * (1 + 0.1 * rng.random())
        y0 = np.array([1, 0, 0, 0, v * np.cos(0.3 * s), v * np.sin(0.3 * s), 1.0])
        tf = rng.uniform(0.5, 1.5)
        c = cst.copy(); c[3] *= rng.uniform(0.8, 1.2)
        x, rc, _ = O.propagate(y0, tf, c, tan, K)
        assert rc == 0
        xs.append(x); us.append(rng.normal(size=(3, K)) * 0.5); tfs.append(tf); css.append(c)
    us[2][:] = 0.0   # zero-thrust satellite: ||u|| <= eps branch
    disc = Discretizer(_Const(cst), include_J2=True)
    A, Bp, Bn, Sig, xi, st = disc.discretize_batch(np.stack(xs), np.stack(us), np.array(tfs), np.stack(css))
    assert (st == 0).all()
    for s in range(S):
        o = O.discretize(xs[s], us[s], tfs[s], css[s], O.FLAG_J2)
        assert o["status"] == 0
        for k, g in zip(KEYS, (A[s], Bp[s], Bn[s], Sig[s], xi[s])):
            assert relerr(g, o[k]) < RTOL, (s, k)


def test_long_intervals_exercise_the_pivoting():
    """Intervals of a third to a full orbit: Phi is far from the identity, its first column's largest entry is off the
    diagonal, so the quadrature solve's partial pivoting (lu_solve_cols: pivot row chosen by the column's owner, rows
    swapped in every lane) really exchanges rows -- on the benchmark's short intervals it never does.  Against the CPU
    oracle, which inverts Phi by its own pivoted LU (np.linalg.inv of the reference)."""
    from mpconstellation_amd import Discretizer
    cst = np.array([39.47841760435743, 0.92, 1.08262668E-3, 46.5, 0.0873, 1e-12, 6.9e6, 3.7e-17])
    tan = O.make_ctrl(O.CTRL_TANGENTIAL, (0.3, 0, 0))
    y0 = np.array([1, 0, 0, 0, 2 * np.pi, 0.4, 1.0])
    disc = Discretizer(_Const(cst))
    swaps = 0
    for K, tf in ((3, 2.0), (4, 1.0), (3, 0.7), (5, 3.0)):
        x, rc, _ = O.propagate(y0, tf, cst, tan, K)
        assert rc == 0
        u = np.full((3, K), 0.2)
        A, Bp, Bn, Sig, xi = disc.discretize(satellite_dynamics, x, u, tf)
        o = O.discretize(x, u, tf, cst)
        assert o["status"] == 0
        for k, g in zip(KEYS, (A, Bp, Bn, Sig, xi)):
            assert relerr(g, o[k]) < 1e-8, (K, tf, k)          # (Phi's condition number is 1e3 .. 2e4 here)
        swaps += int((np.abs(o["A"][:, 1:6, 0]).max(axis=1) > np.abs(o["A"][:, 0, 0])).sum())
    assert swaps >= 3


def test_status_codes():
    from mpconstellation_amd import Discretizer
    cst = np.array([39.47841760435743, 0.92, 1.08262668E-3, 46.5, 0.0873, 1e-12, 6.9e6, 3.7e-17])
    K = 4
    x = np.tile(np.array([1, 0, 0, 0, 6.28, 0, 1.0])[:, None], (1, K))
    xbad = x.copy(); xbad[6, 2] = -1.0       # non-positive mass on one node
    u = np.zeros((3, K))
    disc = Discretizer(_Const(cst))
    *_, st = disc.discretize_batch(np.stack([x, xbad]), np.stack([u, u]), np.ones(2), np.stack([cst, cst]))
    assert st[0] == 0 and st[1] == 1
    with pytest.raises(Exception):
        disc.discretize(satellite_dynamics, xbad, u, 1.0)
    with pytest.raises(NotImplementedError):
        disc.discretize(lambda *a: None, x, u, 1.0)


@pytest.mark.parametrize("steps", [101, 11])
def test_uniform_steps_mode_vs_reference(golden_dir, steps):
    """Discretizer.use_uniform_steps / integrator_steps (linearize_discretize.py:27-30, 50-53, 104-109): quadrature over
    uniform points of the RK45 dense output.  Golden arrays from the reference itself (make_golden.py uniform)."""
    from mpconstellation_amd import Discretizer, Simulator
    from mpconstellation_amd.constants import Constants
    g = np.load(os.path.join(golden_dir, "uniform_steps_K12_tf1.npz"))
    d = Discretizer(Constants(*g["const"]))
    d.use_uniform_steps = True; d.integrator_steps = steps
    A, Bp, Bn, Sig, xi = d.discretize(Simulator.satellite_dynamics, g["x"], g["u"], float(g["tf"]))
    for got, key in ((A, "A"), (Bp, "Bp"), (Bn, "Bn"), (Sig, "Sigma"), (xi, "xi")):
        ref = g[f"{key}_{steps}"]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max(), key
    # it is a different quadrature: the default mode's result differs from it
    d.use_uniform_steps = False
    A0, Bp0, *_ = d.discretize(Simulator.satellite_dynamics, g["x"], g["u"], float(g["tf"]))
    assert np.abs(Bp0 - Bp).max() > 1e-9


def test_rk23_solver_vs_reference(golden_dir):
    """Discretizer.ivp_solver = 'RK23' (linearize_discretize.py:40,105: the attribute is solve_ivp's `method`): the kernel's
    Bogacki-Shampine instantiation against arrays the reference produced with that setting (make_golden.py rk23) -- adaptive
    quadrature nodes on two references, the uniform-step mode on the RK23 dense output, the fused step with the flag, and the
    methods that are not implemented raise."""
    from mpconstellation_amd import Discretizer, Simulator, mpc_step_batch
    from mpconstellation_amd.constants import Constants
    g = np.load(os.path.join(golden_dir, "rk23_discretize.npz"))
    d = Discretizer(Constants(*g["const"]))
    d.ivp_solver = 'RK23'
    for name in g["cases"]:
        x, u, tf = g[f"x_{name}"], g[f"u_{name}"], float(g[f"tf_{name}"])
        out = d.discretize(Simulator.satellite_dynamics, x, u, tf)
        for got, key in zip(out, ("A", "Bp", "Bn", "Sigma", "xi")):
            ref = g[f"{key}_{name}"]
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max(), (name, key)
    name = "tan_K30_tf1"
    x, u = g[f"x_{name}"], g[f"u_{name}"]
    d.use_uniform_steps = True; d.integrator_steps = int(g["uni_steps"])
    for got, key in zip(d.discretize(Simulator.satellite_dynamics, x, u, 1.0), ("A", "Bp", "Bn", "Sigma", "xi")):
        assert np.abs(got - g["uni_" + key]).max() <= 1e-10 * np.abs(g["uni_" + key]).max(), key
    # the default method gives a different quadrature (other step nodes) -- and so a slightly different solve
    d.use_uniform_steps = False
    d45 = Discretizer(Constants(*g["const"]))
    assert np.abs(d45.discretize(Simulator.satellite_dynamics, x, u, 1.0)[1] - g[f"Bp_{name}"]).max() > 1e-7
    r_des = np.linalg.norm(x[:3, -1])
    a = mpc_step_batch(x[None], u[None], [1.0], g["const"][None], [r_des], rk23=True)
    b = mpc_step_batch(x[None], u[None], [1.0], g["const"][None], [r_des])
    assert a.status[0] == 0 and b.status[0] == 0 and 0 < np.abs(a.X - b.X).max() < 1e-2       # (B, xi carry ~1e-3 of quadrature error either way)
    for m in ('DOP853', 'Radau', 'BDF', 'LSODA'):
        d.ivp_solver = m
        with pytest.raises(NotImplementedError):
            d.discretize(Simulator.satellite_dynamics, x, u, 1.0)


def test_scipy_zoh_mode_vs_reference(golden_dir):
    """Discretizer(use_scipy_ZOH=True) (linearize_discretize.py:327-329: interp1d(kind='linear') in place of u_FOH): the device
    evaluates the hold by the FOH formula in that mode too -- the same function; the reference's two evaluations differ by
    <= 1e-16 relative (tests/test_oracle_golden.py::test_scipy_zoh_mode) -- and is held to arrays the reference produced WITH
    the flag (make_golden.py zoh) at the tolerance of every other mode, 1e-10 relative (observed ~1e-13)."""
    from mpconstellation_amd import Discretizer, Simulator
    from mpconstellation_amd.constants import Constants
    g = np.load(os.path.join(golden_dir, "scipy_zoh_discretize.npz"))
    d = Discretizer(Constants(*g["const"]), use_scipy_ZOH=True)
    worst = 0.0
    for name in g["cases"]:
        x, u, tf = g[f"x_{name}"], g[f"u_{name}"], float(g[f"tf_{name}"])
        for got, key in zip(d.discretize(Simulator.satellite_dynamics, x, u, tf), ("A", "Bp", "Bn", "Sigma", "xi")):
            ref = g[f"{key}_{name}"]
            err = np.abs(got - ref).max() / np.abs(ref).max()
            worst = max(worst, err)
            assert got.shape == ref.shape and err <= 1e-10, (name, key, err)
    print(f"use_scipy_ZOH: worst relative difference to the reference's arrays {worst:.2e}")
