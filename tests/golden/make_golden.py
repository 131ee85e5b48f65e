#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference's own numpy/scipy code.

Runs ONLY in the build container (needs /root/reference); the GPU box never
runs this.  It imports the reference modules unmodified and records inputs and
expected outputs of the discretize half of the hot path (SURVEY.md §8a rows
D1-D8, S2, U1) as small .npz files next to this script.

The reference's module chain simulator -> control -> optimizer imports pyomo at
module top (optimizer.py:2-3).  pyomo/ipopt are not installed here, so empty
placeholder modules are registered in sys.modules to let the *discretize half*
import; nothing numerical is stubbed.  The solve half (optimizer.py:254-613)
cannot run here and has no golden vectors ("parity unpinned").

Only data is written: arrays of inputs and outputs.  No reference source text.
"""
import os
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
for _name in ["pyomo", "pyomo.environ", "pyomo.core", "pyomo.core.base",
              "pyomo.core.base.expression"]:
    sys.modules[_name] = types.ModuleType(_name)
sys.modules["pyomo.core.base.expression"].ScalarExpression = object
sys.path.insert(0, "/root/reference")

import numpy as np
from scipy import integrate

from simulator import Simulator            # noqa: E402
from satellite import Satellite            # noqa: E402
from satellite_scale import SatelliteScale  # noqa: E402
from linearize_discretize import Discretizer  # noqa: E402
from optimizer import Optimizer            # noqa: E402
from control import (ConstantTangentialThrustController,  # noqa: E402
                     ConstantThrustController, SequenceController)

HERE = os.path.dirname(os.path.abspath(__file__))
R_HUBBLE = np.array([5371.4806, -4133.1393, 1399.9594]) * 1000.0
V_HUBBLE = np.array([4.6921, 4.9848, -3.2752]) * 1000.0
M_HUBBLE = 12200.0
F = Simulator.satellite_dynamics
CONST_KEYS = ["MU", "R_E", "J2", "G0", "ISP", "S", "R0", "RHO"]


def const_vec(const):
    return np.array([getattr(const, k) for k in CONST_KEYS], dtype=np.float64)


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def constellation_state(i, S, U):
    """SURVEY.md §8(d) config 2/3/5 generator for satellite i of S."""
    R = rot_x(np.pi * ((i * 0.61803) % 1.0) / 3.0) @ rot_z(2.0 * np.pi * i / S)
    return R @ R_HUBBLE, R @ (V_HUBBLE * (1.0 + 0.1 * U[i])), M_HUBBLE


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: " + ", ".join(f"{k}{np.shape(v)}" for k, v in arrs.items()))


def gen_constants_and_pointwise():
    sat = Satellite(R_HUBBLE, V_HUBBLE, M_HUBBLE)
    scale = SatelliteScale(sat=sat)
    const = scale.get_normalized_constants()
    x_dim = sat.get_state_vector()
    save("constants_hubble.npz", state=x_dim, const=const_vec(const),
         const_keys=np.array(CONST_KEYS),
         scale=np.array([scale._r0, scale._s0, scale._v0, scale._a0, scale._m0,
                         scale._T0, scale._mu0]),
         x_norm=scale.normalize_state(x_dim),
         x_redim=scale.redim_state(scale.normalize_state(x_dim)))

    rng = np.random.default_rng(12345)
    n = 128
    xs = np.zeros((n, 7)); us = np.zeros((n, 3)); tfs = np.zeros(n)
    for i in range(n):
        r = rng.normal(size=3); r *= rng.uniform(0.9, 2.5) / np.linalg.norm(r)
        v = rng.normal(size=3); v *= rng.uniform(3.0, 8.0) / np.linalg.norm(v)
        xs[i] = np.concatenate([r, v, [rng.uniform(0.3, 1.2)]])
        us[i] = rng.normal(size=3) * rng.uniform(0.0, 2.0)
        tfs[i] = rng.uniform(0.1, 3.0)
    us[::16] = 0.0  # exercise the ||u|| <= eps branch of B_func
    out = {}
    for j2 in (False, True):
        d = Discretizer(const, include_drag=False, include_J2=j2)
        fs = np.zeros((n, 7)); As = np.zeros((n, 7, 7)); Bs = np.zeros((n, 7, 3))
        xis = np.zeros((n, 7)); sigs = np.zeros((n, 7))
        for i in range(n):
            u_i = us[i]
            ufun = lambda y, tau, u_i=u_i: u_i
            fs[i] = F(0.3, xs[i], ufun, tfs[i], const, include_drag=False, include_J2=j2)
            As[i] = d.A_func(xs[i], us[i], tfs[i])
            Bs[i] = d.B_func(xs[i], us[i], tfs[i])
            xis[i] = d.xi_func(F, xs[i], us[i], tfs[i])
            sigs[i] = d.Sigma_func(F, xs[i], ufun, 0.3)
        tag = "j2" if j2 else "nj2"
        out.update({f"f_{tag}": fs, f"A_{tag}": As, f"B_{tag}": Bs,
                    f"xi_{tag}": xis, f"Sigma_{tag}": sigs})
    # full-physics truth dynamics (drag + J2, simulator.py:150-158)
    fdrag = np.zeros((n, 7))
    for i in range(n):
        u_i = us[i]
        fdrag[i] = F(0.3, xs[i], lambda y, tau, u_i=u_i: u_i, tfs[i], const,
                     include_drag=True, include_J2=True)
    save("pointwise.npz", const=const_vec(const), x=xs, u=us, tf=tfs, f_drag_j2=fdrag, **out)

    # FOH at node/edge taus (linearize_discretize.py:294-315)
    for K in (2, 3, 30, 100):
        pass
    foh = {}
    for K in (2, 3, 20, 30, 100):
        u = rng.normal(size=(3, K))
        d = Discretizer(const)
        tau_nodes = np.linspace(0, 1, K)
        taus = np.concatenate([tau_nodes,
                               np.nextafter(tau_nodes[1:], 0.0),
                               np.nextafter(tau_nodes[:-1], 1.0),
                               rng.uniform(0, 1, size=64)])
        vals = np.array([d.u_FOH(t, u) for t in taus])
        foh[f"u_{K}"] = u; foh[f"tau_{K}"] = taus; foh[f"val_{K}"] = vals
    save("foh.npz", **foh)


def reference_case(sat, scale, controller, tf, base_res):
    sim = Simulator(sats=[sat], controller=controller, scale=scale, base_res=base_res,
                    include_drag=False, include_J2=False)
    sim.run(tf=tf)
    x = sim.sim_data[sat.id]
    t = sim.sim_time[sat.id]
    u = Discretizer.extract_uk(x, t, controller)
    return x, t, u


def constraint_terms(x, u, tf, d, scale):
    opt = Optimizer([x], [u], [np.zeros_like(x)], tf, d, F, scale, verbose=False)
    terms = opt.get_constraint_terms()
    return {f"ct_{k}": np.asarray(v[0]) for k, v in terms.items()}


def rk_nodes(d, x, u, tf):
    """Accepted RK45 nodes per interval, exactly the call get_matrices makes
    (linearize_discretize.py:34-41)."""
    K = x.shape[1]
    tau = np.linspace(0, 1, K)
    d._Discretizer__tau = tau
    d._Discretizer__u = u
    counts = np.zeros(K - 1, dtype=np.int64); nfev = np.zeros(K - 1, dtype=np.int64)
    ts = []; ys = []
    for k in range(K - 1):
        y0 = np.concatenate([np.eye(7).flatten(), x[:, k]])
        sol = integrate.solve_ivp(d.dPhi_gen(), [tau[k], tau[k + 1]], y0,
                                  args=(F, d.u_func, tf), max_step=d.ivp_max_step,
                                  method=d.ivp_solver, t_eval=None)
        counts[k] = sol.t.size; nfev[k] = sol.nfev
        ts.append(sol.t); ys.append(sol.y.T)
    return counts, nfev, np.concatenate(ts), np.concatenate(ys, axis=0)


def gen_discretize_cases():
    sat = Satellite(R_HUBBLE, V_HUBBLE, M_HUBBLE)
    scale = SatelliteScale(sat=sat)
    const = scale.get_normalized_constants()
    cases = [
        # name, controller, tf, base_res  -> K = int(base_res*tf)
        ("tan_K20_tf2", ConstantTangentialThrustController([sat], 0.5), 2, 10),
        ("tan_K30_tf1", ConstantTangentialThrustController([sat], 0.5), 1, 30),
        ("tan_K60_tf2", ConstantTangentialThrustController([sat], 0.5), 2, 30),
        ("tan_K100_tf1", ConstantTangentialThrustController([sat], 0.5), 1, 100),
        ("const_K30_tf1", ConstantThrustController([sat], np.array([0.44, 0.7, 1.0])), 1, 30),
    ]
    for name, ctrl, tf, base_res in cases:
        x, t, u = reference_case(sat, scale, ctrl, tf, base_res)
        d = Discretizer(const, include_drag=False, include_J2=False)
        A, Bp, Bn, Sig, xi = d.discretize(F, x, u, tf)
        extra = constraint_terms(x, u, tf, d, scale)
        if name in ("tan_K30_tf1", "tan_K20_tf2"):
            counts, nfev, nt, ny = rk_nodes(d, x, u, tf)
            extra.update(node_counts=counts, node_nfev=nfev, node_t=nt, node_y=ny)
        save(f"disc_{name}.npz", const=const_vec(const), x=x, t=t, u=u, tf=np.float64(tf),
             A=A, Bp=Bp, Bn=Bn, Sigma=Sig, xi=xi, **extra)

    # J2 in the linearisation (A_func J2 branch, linearize_discretize.py:149-158)
    ctrl = ConstantTangentialThrustController([sat], 0.5)
    x, t, u = reference_case(sat, scale, ctrl, 1, 30)
    d = Discretizer(const, include_drag=False, include_J2=True)
    A, Bp, Bn, Sig, xi = d.discretize(F, x, u, 1)
    save("disc_tanJ2_K30_tf1.npz", const=const_vec(const), x=x, t=t, u=u, tf=np.float64(1),
         A=A, Bp=Bp, Bn=Bn, Sigma=Sig, xi=xi)

    # K=2 / K=3 smoke cases of test_discretizer.py:30-54 and :57-85 (incl. the
    # un-normalised state of the K=2 case, which the reference feeds as-is)
    T_init = np.array([0.44, 0.7, 1.0])
    d = Discretizer(const)
    x2 = np.column_stack([sat.get_state_vector()] * 2)
    u2 = np.column_stack([T_init] * 2)
    A, Bp, Bn, Sig, xi = d.discretize(F, x2, u2, 1)
    save("disc_K2_dimensional.npz", const=const_vec(const), x=x2, u=u2, tf=np.float64(1),
         A=A, Bp=Bp, Bn=Bn, Sigma=Sig, xi=xi)
    xn = scale.normalize_state(sat.get_state_vector())
    x3 = np.column_stack([xn] * 3)
    u3 = np.column_stack([T_init] * 3)
    A, Bp, Bn, Sig, xi = d.discretize(F, x3, u3, 0.1)
    save("disc_K3_tf0p1.npz", const=const_vec(const), x=x3, u=u3, tf=np.float64(0.1),
         A=A, Bp=Bp, Bn=Bn, Sigma=Sig, xi=xi)

    # zero-thrust reference (B_func ||u||<=eps branch inside discretize)
    xz, tz, uz = reference_case(sat, scale, ConstantThrustController([sat], np.zeros(3)), 1, 20)
    A, Bp, Bn, Sig, xi = d.discretize(F, xz, uz, 1)
    save("disc_zero_K20_tf1.npz", const=const_vec(const), x=xz, t=tz, u=uz, tf=np.float64(1),
         A=A, Bp=Bp, Bn=Bn, Sigma=Sig, xi=xi)


def gen_constellation_cases():
    """A few satellites of the S=64 benchmark constellation."""
    S = 64
    U = np.random.default_rng(20260101).random(S)
    idx = [0, 1, 17, 63]
    out = {"U": U, "idx": np.array(idx)}
    for i in idx:
        r, v, m = constellation_state(i, S, U)
        sat = Satellite(r, v, m)
        scale = SatelliteScale(sat=sat)
        const = scale.get_normalized_constants()
        ctrl = ConstantTangentialThrustController([sat], 0.5)
        x, t, u = reference_case(sat, scale, ctrl, 1, 30)
        d = Discretizer(const, include_drag=False, include_J2=False)
        A, Bp, Bn, Sig, xi = d.discretize(F, x, u, 1)
        out.update({f"state_{i}": sat.get_state_vector(), f"const_{i}": const_vec(const),
                    f"x_{i}": x, f"t_{i}": t, f"u_{i}": u, f"A_{i}": A, f"Bp_{i}": Bp,
                    f"Bn_{i}": Bn, f"Sigma_{i}": Sig, f"xi_{i}": xi})
    save("constellation64.npz", **out)


def gen_propagation_cases():
    """Nonlinear rollouts (simulator.py:164-189) incl. drag/J2 truth model and
    FOH sequence playback (control.py:86-143)."""
    sat = Satellite(R_HUBBLE, V_HUBBLE, M_HUBBLE)
    scale = SatelliteScale(sat=sat)
    out = {}
    for name, drag, j2 in (("plain", False, False), ("dragj2", True, True)):
        ctrl = ConstantTangentialThrustController([sat], 0.5)
        sim = Simulator(sats=[sat], controller=ctrl, scale=scale, base_res=30,
                        include_drag=drag, include_J2=j2)
        sim.run(tf=1)
        out[f"x_tan_{name}"] = sim.sim_data[sat.id]; out[f"t_tan_{name}"] = sim.sim_time[sat.id]
    rng = np.random.default_rng(7)
    useq = rng.normal(size=(3, 12)) * 0.5
    for name, tf_u, tf_sim in (("full", 0.8, 0.8), ("tail", 0.6, 1.0)):
        ctrl = SequenceController(u=useq, tf_u=tf_u, tf_sim=tf_sim)
        sim = Simulator(sats=[sat], controller=ctrl, scale=scale, base_res=40,
                        include_drag=False, include_J2=False)
        sim.run(tf=tf_sim)
        out[f"x_seq_{name}"] = sim.sim_data[sat.id]; out[f"t_seq_{name}"] = sim.sim_time[sat.id]
    out["useq"] = useq
    out["const"] = const_vec(scale.get_normalized_constants())
    out["y0"] = scale.normalize_state(sat.get_state_vector())
    save("propagate.npz", **out)


def gen_uniform_steps_cases():
    """Discretizer.use_uniform_steps = True (linearize_discretize.py:27-30, 50-53): solve_ivp evaluates the dense output
    at integrator_steps uniform points per interval and the quadrature runs over those instead of the accepted nodes."""
    sat = Satellite(R_HUBBLE, V_HUBBLE, M_HUBBLE)
    scale = SatelliteScale(sat=sat)
    const = scale.get_normalized_constants()
    out = {}
    ctrl = ConstantTangentialThrustController([sat], 0.5)
    x, t, u = reference_case(sat, scale, ctrl, 1, 12)
    for steps in (101, 11):
        d = Discretizer(const, include_drag=False, include_J2=False)
        d.use_uniform_steps = True; d.integrator_steps = steps
        A, Bp, Bn, Sig, xi = d.discretize(F, x, u, 1)
        out.update({f"A_{steps}": A, f"Bp_{steps}": Bp, f"Bn_{steps}": Bn, f"Sigma_{steps}": Sig, f"xi_{steps}": xi})
    save("uniform_steps_K12_tf1.npz", const=const_vec(const), x=x, t=t, u=u, tf=np.float64(1), steps=np.array([101, 11]), **out)


def gen_rk23_cases():
    """Discretizer.ivp_solver = 'RK23' (linearize_discretize.py:40,105: the attribute goes to solve_ivp's `method`): the
    default adaptive quadrature nodes and the uniform-step mode, with the accepted step nodes of the adaptive run."""
    sat = Satellite(R_HUBBLE, V_HUBBLE, M_HUBBLE)
    scale = SatelliteScale(sat=sat)
    const = scale.get_normalized_constants()
    out = {}
    for name, ctrl, tf, base_res in (("tan_K30_tf1", ConstantTangentialThrustController([sat], 0.5), 1, 30),
                                     ("const_K20_tf2", ConstantThrustController([sat], np.array([0.44, 0.7, 1.0])), 2, 10)):
        x, t, u = reference_case(sat, scale, ctrl, tf, base_res)
        d = Discretizer(const, include_drag=False, include_J2=False)
        d.ivp_solver = 'RK23'
        A, Bp, Bn, Sig, xi = d.discretize(F, x, u, tf)
        counts, nfev, nt, ny = rk_nodes(d, x, u, tf)
        out.update({f"x_{name}": x, f"t_{name}": t, f"u_{name}": u, f"tf_{name}": np.float64(tf), f"A_{name}": A, f"Bp_{name}": Bp,
                    f"Bn_{name}": Bn, f"Sigma_{name}": Sig, f"xi_{name}": xi, f"node_counts_{name}": counts, f"node_nfev_{name}": nfev,
                    f"node_t_{name}": nt, f"node_y_{name}": ny})
        if name == "tan_K30_tf1":
            d.use_uniform_steps = True; d.integrator_steps = 21
            A, Bp, Bn, Sig, xi = d.discretize(F, x, u, tf)
            out.update({"uni_steps": np.int64(21), "uni_A": A, "uni_Bp": Bp, "uni_Bn": Bn, "uni_Sigma": Sig, "uni_xi": xi})
    save("rk23_discretize.npz", const=const_vec(const), cases=np.array(["tan_K30_tf1", "const_K20_tf2"]), **out)


def gen_scipy_zoh_case():
    """Discretizer(use_scipy_ZOH=True) (linearize_discretize.py:327-329): u_func evaluates scipy.interpolate.interp1d(tau, u,
    kind='linear') instead of u_FOH -- the same piecewise-linear hold, other rounding (slope form, searchsorted's interval at the
    nodes).  Outputs and accepted step nodes, beside the default path's on the same inputs."""
    sat = Satellite(R_HUBBLE, V_HUBBLE, M_HUBBLE)
    scale = SatelliteScale(sat=sat)
    const = scale.get_normalized_constants()
    out = {}
    for name, ctrl, tf, base_res in (("tan_K30_tf1", ConstantTangentialThrustController([sat], 0.5), 1, 30),
                                     ("const_K20_tf2", ConstantThrustController([sat], np.array([0.44, 0.7, 1.0])), 2, 10)):
        x, t, u = reference_case(sat, scale, ctrl, tf, base_res)
        d = Discretizer(const, include_drag=False, include_J2=False, use_scipy_ZOH=True)
        A, Bp, Bn, Sig, xi = d.discretize(F, x, u, tf)
        counts, nfev, nt, ny = rk_nodes(d, x, u, tf)
        d0 = Discretizer(const, include_drag=False, include_J2=False)
        A0, Bp0, Bn0, Sig0, xi0 = d0.discretize(F, x, u, tf)
        out.update({f"x_{name}": x, f"t_{name}": t, f"u_{name}": u, f"tf_{name}": np.float64(tf), f"A_{name}": A, f"Bp_{name}": Bp,
                    f"Bn_{name}": Bn, f"Sigma_{name}": Sig, f"xi_{name}": xi, f"node_counts_{name}": counts, f"node_t_{name}": nt,
                    f"foh_A_{name}": A0, f"foh_Bp_{name}": Bp0, f"foh_Bn_{name}": Bn0, f"foh_Sigma_{name}": Sig0, f"foh_xi_{name}": xi0})
    save("scipy_zoh_discretize.npz", const=const_vec(const), cases=np.array(["tan_K30_tf1", "const_K20_tf2"]), **out)


def gen_csv_case():
    """The trajectory CSV the reference writes (Simulator.save_to_csv, simulator.py:192-201, read by visualizer.m:23-28):
    file name pattern, the file's text and the run that produced it."""
    import glob
    import tempfile
    sat = Satellite(R_HUBBLE, V_HUBBLE, M_HUBBLE)
    scale = SatelliteScale(sat=sat)
    ctrl = ConstantThrustController([sat], np.array([0.1, 0.0, 0.05]))
    sim = Simulator(sats=[sat], controller=ctrl, scale=scale, base_res=12)      # truth model: drag and J2 on (defaults)
    sim.run(tf=1)
    with tempfile.TemporaryDirectory() as tmp:
        cwd = os.getcwd(); os.chdir(tmp)
        try:
            sim.save_to_csv(suffix="_ref")
            files = glob.glob("trajectory_*_ref.csv")
            assert len(files) == 1
            text = open(files[0], "rb").read()
            name = files[0].replace(str(sat.id), "<id>")
        finally:
            os.chdir(cwd)
    save("csv_reference.npz", text=np.frombuffer(text, dtype=np.uint8), file_pattern=np.array(name), x=sim.sim_data[sat.id],
         thrust=np.array([0.1, 0.0, 0.05]), base_res=np.int64(12), tf=np.float64(1.0))


if __name__ == "__main__":
    os.chdir("/tmp")  # reference code may write files into the CWD
    if len(sys.argv) > 1 and sys.argv[1] == "csv":
        gen_csv_case()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "uniform":
        gen_uniform_steps_cases()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "zoh":
        gen_scipy_zoh_case()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rk23":
        gen_rk23_cases()
        raise SystemExit(0)
    gen_constants_and_pointwise()
    gen_discretize_cases()
    gen_constellation_cases()
    gen_propagation_cases()
    gen_csv_case()
    gen_uniform_steps_cases()
    gen_rk23_cases()
    gen_scipy_zoh_case()
