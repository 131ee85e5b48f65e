"""The reference's own test scripts, run as written through the drop-in classes on the device and checked against the CPU
oracle (the reference's tests only print and plot; their configurations are what is mirrored here):
test_discretizer.py:88-118 (test_linearize_many), :120-150 (test_linearize_tangential), test_simulator.py:57-77
(test_run_segment), :149-173 (test_run_segments), :17-34 and :175-203 (the long rollouts).  test_optimizer.py and test_mpc are in test_solve_gpu.py /
test_mpc_loop_gpu.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import oracle_lib as O

pytestmark = pytest.mark.gpu
R0 = np.array([5371.4806, -4133.1393, 1399.9594]) * 1000
V0 = np.array([4.6921, 4.9848, -3.2752]) * 1000
M0 = 12200


def discrete_rollout(A, Bp, Bn, Sig, xi, x0, u, tf):
    """the forward simulation of the discretised model the reference's tests plot (test_discretizer.py:106-112)"""
    xs = [x0]
    for k in range(A.shape[0]):
        xs.append(A[k] @ xs[-1] + Bn[k] @ u[:, k] + Bp[k] @ u[:, k + 1] + Sig[:, k] * tf + xi[:, k])
    return np.column_stack(xs)


@pytest.mark.parametrize("law", ["constant", "tangential"])
def test_linearize_as_written(law):
    from mpconstellation_amd import (Satellite, SatelliteScale, Simulator, Discretizer, ConstantThrustController,
                                     ConstantTangentialThrustController)
    sat = Satellite(R0, V0, M0)
    scale = SatelliteScale(sat=sat); const = scale.get_normalized_constants(); cst = const.as_vector()
    if law == "constant":                         # test_linearize_many: T_init = [0.44, 0.7, 1.0], tf = 1, base_res = 100
        T = np.array([0.44, 0.7, 1.0]); tf, base_res = 1, 100
        c = ConstantThrustController([sat], T); oc = O.make_ctrl(O.CTRL_CONSTANT, tuple(T))
    else:                                         # test_linearize_tangential: 0.5 tangential, tf = 2, base_res = 100
        tf, base_res = 2, 100
        c = ConstantTangentialThrustController([sat], 0.5); oc = O.make_ctrl(O.CTRL_TANGENTIAL, (0.5, 0, 0))
    sim = Simulator(sats=[sat], controller=c, scale=scale, base_res=base_res, include_drag=False, include_J2=False)
    sim.run(tf=tf)
    x = sim.sim_data[sat.id]; K = x.shape[1]
    assert K == int(base_res * tf)
    xo, rc, _ = O.propagate(x[:, 0], float(tf), cst, oc, K)
    assert rc == 0 and np.abs(x - xo).max() < 1e-9
    d = Discretizer(const, use_scipy_ZOH=False, include_drag=False, include_J2=False)
    # (the reference's test_linearize_many tiles T_init as np.tile(T_init, (3, K)), a (3, 3K) table whose first-order hold
    #  is still the constant T_init; the table of the right shape gives the same matrices)
    u = np.tile(T[:, None], (1, K)) if law == "constant" else d.extract_uk(x, sim.sim_time[sat.id], c)
    A, Bp, Bn, Sig, xi = d.discretize(Simulator.satellite_dynamics, x, u, tf)
    assert A.shape == (K - 1, 7, 7) and Bp.shape == (K - 1, 7, 3) and Sig.shape == (7, K - 1)
    ref = O.discretize(xo, u, float(tf), cst)
    for name, got in (("A", A), ("Bp", Bp), ("Bn", Bn), ("Sigma", Sig), ("xi", xi)):
        assert np.abs(got - ref[name]).max() <= 1e-9 * max(1.0, np.abs(ref[name]).max()), name
    # what the reference's test looks at: the discretised model, rolled forward from x_0 under the reference inputs,
    # follows the nonlinear trajectory (linearised around that very trajectory: the defect is the quadrature error)
    xd = discrete_rollout(A, Bp, Bn, Sig, xi, x[:, 0], u, tf)
    xdo = discrete_rollout(ref["A"], ref["Bp"], ref["Bn"], ref["Sigma"], ref["xi"], xo[:, 0], u, float(tf))
    assert np.abs(xd - xdo).max() < 1e-7
    # (the reference's own trapezoid quadrature error: SURVEY 8c measured 1e-3 on xi, 2e-5 on B)
    assert np.abs(xd[:6] - x[:6]).max() < 2e-2 and np.abs(xd[:, 1] - x[:, 1]).max() < 1e-3


def test_run_segment_as_written():
    """test_simulator.py:57-77: one satellite, default controller and truth model (drag and J2 on), res = 20, segments of
    1, 1 and 2 orbits appended to the same history."""
    from mpconstellation_amd import Satellite, SatelliteScale, Simulator
    sat = Satellite(R0, V0, M0)
    scale = SatelliteScale(sat=sat); cst = scale.get_normalized_constants().as_vector()
    res = 20
    sim = Simulator(sats=[sat], scale=scale, base_res=res)
    y = scale.normalize_state(sat.get_state_vector())
    parts = []
    for tf in (1, 1, 2):
        sim.run_segment(tf=tf)
        seg, rc, _ = O.propagate(y, float(tf), cst, O.make_ctrl(O.CTRL_ZERO), int(res * tf), flags=3)
        assert rc == 0
        parts.append(seg)
        y = scale.normalize_state(scale.redim_state(seg[:, -1]))      # the state goes through the Satellite object in SI units
    assert sim.sim_time[sat.id].shape == (4 * res,) and sim.sim_data[sat.id].shape == (7, 4 * res)
    assert np.abs(sim.sim_data[sat.id] - np.concatenate(parts, axis=1)).max() < 1e-9
    assert np.all(np.diff(sim.sim_time[sat.id]) > 0)


def test_run_segments_as_written():
    """test_simulator.py:149-173: two satellites under ONE scale (the first one's), tangential thrust 0.5, res = 100, three
    orbits in four segments."""
    from mpconstellation_amd import Satellite, SatelliteScale, Simulator, ConstantTangentialThrustController
    sat, sat2 = Satellite(R0, V0, M0), Satellite(R0, V0 * 1.1, M0)
    sats = [sat, sat2]
    res, tf, nseg = 100, 3, 4
    c = ConstantTangentialThrustController(sats=sats, tangential_thrust=0.5)
    scale = SatelliteScale(sat=sat); cst = scale.get_normalized_constants().as_vector()
    y0 = [scale.normalize_state(s.get_state_vector()) for s in sats]
    sim = Simulator(sats=sats, scale=scale, base_res=res, controller=c)
    sim.run_segments(tf=tf, num_segments=nseg)
    n = int(res * tf / nseg)
    for s, y in zip(sats, y0):
        assert sim.sim_time[s.id].shape == (nseg * n,) and sim.sim_data[s.id].shape == (7, nseg * n)
        parts = []
        for _ in range(nseg):
            seg, rc, _ = O.propagate(y, tf / float(nseg), cst, O.make_ctrl(O.CTRL_TANGENTIAL, (0.5, 0, 0)), n, flags=3)
            assert rc == 0
            parts.append(seg)
            y = scale.normalize_state(scale.redim_state(seg[:, -1]))
        assert np.abs(sim.sim_data[s.id] - np.concatenate(parts, axis=1)).max() < 1e-9
    # thrusting along the velocity raises the orbit and burns mass
    assert np.linalg.norm(sim.sim_data[sat.id][:3, -1]) > 1.0 and sim.sim_data[sat.id][6, -1] < 1.0


@pytest.mark.parametrize("case", ["default_tf5", "constant_tf15", "tangential_tf5"])
def test_long_rollouts_as_written(case):
    """test_simulator.py:17-34, :175-203: the default controller for 5 orbits, ConstantThrustController(thrust=[0, 0, 0.1]) for
    15, ConstantTangentialThrustController(tangential_thrust=0.1) for 5 -- constructed with the keyword arguments the
    reference's tests use, default resolution (100 per orbit) and truth model (drag and J2 on)."""
    from mpconstellation_amd import (Satellite, SatelliteScale, Simulator, ConstantThrustController,
                                     ConstantTangentialThrustController)
    sat = Satellite(R0, V0, M0)
    scale = SatelliteScale(sat=sat); cst = scale.get_normalized_constants().as_vector()
    if case == "default_tf5":
        sim, tf, oc = Simulator(sats=[sat], scale=scale), 5, O.make_ctrl(O.CTRL_ZERO)
    elif case == "constant_tf15":
        c = ConstantThrustController(thrust=np.array([0., 0., 0.1]))
        sim, tf, oc = Simulator(sats=[sat], controller=c, scale=scale), 15, O.make_ctrl(O.CTRL_CONSTANT, (0.0, 0.0, 0.1))
    else:
        c = ConstantTangentialThrustController(tangential_thrust=0.1)
        sim, tf, oc = Simulator(sats=[sat], controller=c, scale=scale), 5, O.make_ctrl(O.CTRL_TANGENTIAL, (0.1, 0, 0))
    data, time = sim.run(tf=tf)
    x = data[sat.id]
    assert x.shape == (7, 100 * tf) and time[sat.id].shape == (100 * tf,)
    xo, rc, _ = O.propagate(scale.normalize_state(sat.get_state_vector()), float(tf), cst, oc, 100 * tf, flags=3)
    assert rc == 0 and np.abs(x - xo).max() < 1e-8 * max(1.0, np.abs(xo).max())
