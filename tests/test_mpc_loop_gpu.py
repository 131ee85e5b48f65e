"""The caller of the hot path: OptimalController / Simulator.run_segments (reference control.py:145-246,
simulator.py:50-92; the reference's test_simulator.py:79-147 test_mpc at reduced resolution)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_mpc_segments():
    from mpconstellation_amd import Satellite, SatelliteScale, Simulator, OptimalController
    sat = Satellite(np.array([5371.4806, -4133.1393, 1399.9594]) * 1000, np.array([4.6921, 4.9848, -3.2752]) * 1000, 12200)
    tf, nseg = 2, 2
    c = OptimalController(sats=[sat], base_res=15, tf_horizon=tf, tf_interval=tf / nseg, plot_inter=False,
                          opt_verbose=False, r_des=1.2)
    scale = SatelliteScale(sat=sat)
    sim = Simulator(sats=[sat], controller=c, scale=scale, base_res=40, verbose=False)
    sim.run_segments(tf=tf, num_segments=nseg)
    x_act = sim.sim_data[sat.id]
    assert x_act.shape == (7, 80) and np.isfinite(x_act).all()
    assert c.horizon == 1.0                                   # shrinking horizon (control.py:234-235)
    assert c.opt_trajectory.shape[0] == 7
    # the optimiser's plan ends on the requested circular orbit (its own linearised model)
    x_opt = c.opt_trajectory
    assert abs(np.linalg.norm(x_opt[:3, -1]) - 1.2) < 1e-4
    # mass only decreases along the flown trajectory
    assert (np.diff(x_act[6]) <= 1e-12).all()
