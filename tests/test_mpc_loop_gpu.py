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


def test_run_batched_and_csv(tmp_path, monkeypatch):
    """Simulator.run over several satellites is one batched rollout; save_to_csv keeps the reference's wire format
    (simulator.py:192-201: trajectory_{date}_{id}{suffix}.csv, T rows x 7 columns, redimensionalised)."""
    import glob
    from mpconstellation_amd import Satellite, SatelliteScale, Simulator, ConstantThrustController
    r0 = np.array([5371.4806, -4133.1393, 1399.9594]) * 1000; v0 = np.array([4.6921, 4.9848, -3.2752]) * 1000
    sats = [Satellite(r0, v0 * (1 + 0.01 * i), 12200.0) for i in range(3)]       # pattern of test_simulator.py:36-55
    scale = SatelliteScale(sat=sats[0])
    sim = Simulator(sats=sats, controller=ConstantThrustController(sats, np.array([0.1, 0.0, 0.05])), scale=scale, base_res=20)
    data, times = sim.run(tf=2)
    assert all(data[s.id].shape == (7, 40) for s in sats) and all(times[s.id].shape == (40,) for s in sats)
    assert np.abs(data[sats[0].id] - data[sats[2].id]).max() > 1e-3
    monkeypatch.chdir(tmp_path)
    sim.save_to_csv(suffix="_t")
    files = glob.glob(str(tmp_path / "trajectory_*_t.csv"))
    assert len(files) == 3
    arr = np.loadtxt(files[0], delimiter=",")
    assert arr.shape == (40, 7)
    ids = [f for f in files if str(sats[1].id) in f]
    assert np.allclose(np.loadtxt(ids[0], delimiter=","), scale.redim_state(data[sats[1].id]).T)
