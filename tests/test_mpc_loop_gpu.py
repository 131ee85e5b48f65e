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


def test_csv_matches_the_reference_written_file(golden_dir, tmp_path, monkeypatch):
    """tests/golden/csv_reference.npz holds the text of the file the reference's Simulator.save_to_csv wrote for a
    12-node constant-thrust run under the truth model (drag + J2) and the run itself: same rollout, same file name
    pattern, same text format (np.savetxt '%.18e', ',' separated, T rows x 7 columns, redimensionalised), same values."""
    import glob, re, os
    from mpconstellation_amd import Satellite, SatelliteScale, Simulator, ConstantThrustController
    g = np.load(os.path.join(golden_dir, "csv_reference.npz"))
    sat = Satellite(np.array([5371.4806, -4133.1393, 1399.9594]) * 1000, np.array([4.6921, 4.9848, -3.2752]) * 1000, 12200)
    scale = SatelliteScale(sat=sat)
    sim = Simulator(sats=[sat], controller=ConstantThrustController([sat], g["thrust"]), scale=scale, base_res=int(g["base_res"]))
    sim.run(tf=float(g["tf"]))
    assert np.abs(sim.sim_data[sat.id] - g["x"]).max() < 1e-10          # device rollout = reference rollout (truth model)
    monkeypatch.chdir(tmp_path)
    sim.save_to_csv(suffix="_ref")
    files = glob.glob("trajectory_*_ref.csv")
    assert len(files) == 1
    stamp = r"\d{4}-\d{2}-\d{2}-\d{2}-\d{2}-\d{2}"
    ref_name = str(g["file_pattern"])                                   # trajectory_<date>_<id>_ref.csv
    assert re.fullmatch("trajectory_" + stamp + "_<id>_ref\\.csv", ref_name)
    assert re.fullmatch("trajectory_" + stamp + "_" + re.escape(str(sat.id)) + "_ref\\.csv", files[0])
    ours = open(files[0], "rb").read().decode().splitlines(); ref = bytes(g["text"]).decode().splitlines()
    assert len(ours) == len(ref) == 12
    num = r"-?\d\.\d{18}e[+-]\d{2}"
    for a, b in zip(ours, ref):
        assert re.fullmatch(",".join([num] * 7), a) and re.fullmatch(",".join([num] * 7), b)
        va = np.array(a.split(","), dtype=float); vb = np.array(b.split(","), dtype=float)
        assert np.abs(va - vb).max() <= 1e-9 * np.abs(vb).max()
    assert ours[0] == ref[0]                                            # the initial state: identical text


def oracle_scp_chain(y0, cst, base_res, horizon, r_des, n_iter=2):
    """OptimalController.update (control.py:166-235) assembled from the CPU oracle's pieces"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import oracle_lib as O, nlp_ipm as N
    K = int(base_res * horizon)
    ctrl = O.make_ctrl(2, thrust=(0.5, 0.0, 0.0))
    x = O.propagate(y0, horizon, cst, ctrl, K)[0]; t = np.linspace(0, 1, K)
    tf_u = horizon; r = None
    for i in range(n_iter):
        u_bar = O.extract_uk(x, t, ctrl)
        d = O.discretize(x, u_bar, tf_u, cst)
        P = N.MpcProblem(x, u_bar, tf_u, cst[0], d, O.constraint_terms(x, u_bar, cst[0]),
                         {"r_des": r_des, "eps_r": 0.000001, "eps_vr": 0.0000000000000001, "tf_max": horizon})
        r = N.solve(P)
        assert r["status"] == 0
        tf_u = r["tf"]
        ctrl = O.make_ctrl(3, useq=np.ascontiguousarray(r["U"]), end_tau=1.0)
        Kn = int(base_res * tf_u)
        x = O.propagate(y0, tf_u, cst, ctrl, Kn)[0]; t = np.linspace(0, 1, Kn)
    return r


def test_reference_test_mpc_configuration():
    """The reference's own test_mpc (test_simulator.py:79-98) as written: base_res 30, tf_horizon 2, two segments, the
    default r_des = 1.5 (control.py:147), eps_vr = 1e-16, truth model with drag and J2 at base_res 100.  Every solve of
    both segments (K = 60, then int(30 tf_u), then K = 30 ...) ends with status 0, and the first plan equals the
    oracle chain's."""
    from mpconstellation_amd import Satellite, SatelliteScale, Simulator, OptimalController
    sat = Satellite(np.array([5371.4806, -4133.1393, 1399.9594]) * 1000, np.array([4.6921, 4.9848, -3.2752]) * 1000, 12200)
    scale = SatelliteScale(sat=sat)
    y0 = scale.normalize_state(sat.get_state_vector()); cst = scale.get_normalized_constants().as_vector()
    tf, nseg = 2, 2
    c = OptimalController(sats=[sat], base_res=30, tf_horizon=tf, tf_interval=tf / nseg, plot_inter=False, opt_verbose=False)
    assert c.r_des == 1.5
    c.update()
    assert c.last_status == [0, 0]
    r = oracle_scp_chain(y0, cst, 30, 2, 1.5)
    assert c.opt_trajectory.shape == r["X"].shape and np.abs(c.opt_trajectory - r["X"]).max() < 5e-6
    c.horizon = tf                                            # (update() shrank it; run_segments starts over like the test)
    sim = Simulator(sats=[sat], controller=c, scale=scale, base_res=100, verbose=False)
    statuses = []
    orig = c.update
    def update():
        orig(); statuses.extend(c.last_status)
    c.update = update
    sim.run_segments(tf=tf, num_segments=nseg)
    assert statuses == [0, 0, 0, 0]
    x_act = sim.sim_data[sat.id]
    assert x_act.shape == (7, 200) and np.isfinite(x_act).all() and c.horizon == 1.0
    assert np.linalg.norm(x_act[:3, -1]) > 1.05              # the orbit was raised under the truth model


def test_scp_update_vs_oracle_chain():
    """OptimalController.update (control.py:166-235: reference rollout, then 2 x (extract_uk, discretize, solve, nonlinear
    re-rollout under the optimised FOH sequence)) on the device against the same chain built from the CPU oracle's
    pieces (the second iteration's problem depends on the first one's solution and on a rollout through it)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import oracle_lib as O, nlp_ipm as N
    from mpconstellation_amd import Satellite, SatelliteScale, OptimalController
    sat = Satellite(np.array([5371.4806, -4133.1393, 1399.9594]) * 1000, np.array([4.6921, 4.9848, -3.2752]) * 1000, 12200)
    base_res, horizon, r_des = 15, 2, 1.2
    c = OptimalController(sats=[sat], base_res=base_res, tf_horizon=horizon, tf_interval=1, plot_inter=False,
                          opt_verbose=False, r_des=r_des)
    c.update()
    assert c.last_status == [0, 0]
    # the oracle chain
    scale = SatelliteScale(sat=sat)
    cst = scale.get_normalized_constants().as_vector()
    y0 = scale.normalize_state(sat.get_state_vector())
    K = int(base_res * horizon)
    ctrl = O.make_ctrl(2, thrust=(0.5, 0.0, 0.0))
    x = O.propagate(y0, horizon, cst, ctrl, K)[0]; t = np.linspace(0, 1, K)
    tf_u = horizon
    for i in range(2):
        u_bar = O.extract_uk(x, t, ctrl)
        d = O.discretize(x, u_bar, tf_u, cst)
        P = N.MpcProblem(x, u_bar, tf_u, cst[0], d, O.constraint_terms(x, u_bar, cst[0]),
                         {"r_des": r_des, "eps_r": 0.000001, "eps_vr": 0.0000000000000001, "tf_max": horizon})
        r = N.solve(P)
        assert r["status"] == 0
        tf_u = r["tf"]
        ctrl = O.make_ctrl(3, useq=np.ascontiguousarray(r["U"]), end_tau=1.0)
        Kn = int(base_res * tf_u)
        x = O.propagate(y0, tf_u, cst, ctrl, Kn)[0]; t = np.linspace(0, 1, Kn)
    assert c.opt_trajectory.shape == r["X"].shape
    # observed 2.8e-10 / 1.3e-11; asserted at the solver tolerance
    assert np.abs(c.opt_trajectory - r["X"]).max() < 5e-6
    assert abs(c.sequence_controller.end_tau - r["tf"]) < 5e-6           # end_tau = tf_u / tf_interval, tf_interval = 1
    assert c.horizon == 1.0


def test_constellation_mpc_equals_single_satellite_loops():
    """the MPC loop for several satellites at once (per-satellite scales, batched planning
    and flying) gives every satellite what the reference's structure -- one OptimalController + one Simulator per
    satellite (control.py:162, simulator.py:58-60) -- gives it.  OptimalController.update is the one-satellite case of
    ConstellationMPC.update, so this checks the batching itself (ragged launches, the segment flight riding in the update's
    call against the Simulator's own rollout of the sequence controller): a satellite's plan and flown trajectory do not
    depend on who shares its batch, bit for bit.  (Against the CPU oracle: test_constellation_mpc_plan_vs_oracle_chain,
    test_scp_update_vs_oracle_chain.)"""
    from mpconstellation_amd import Satellite, SatelliteScale, Simulator, OptimalController, ConstellationMPC
    r0 = np.array([5371.4806, -4133.1393, 1399.9594]) * 1000; v0 = np.array([4.6921, 4.9848, -3.2752]) * 1000
    make = lambda: [Satellite(r0, v0 * (1 + 0.01 * i), 12200.0) for i in range(3)]
    tf, nseg, base_res, sim_res, r_des = 2, 2, 15, 40, 1.2
    single = []
    for sat in make():
        c = OptimalController(sats=[sat], base_res=base_res, tf_horizon=tf, tf_interval=tf / nseg, plot_inter=False,
                              opt_verbose=False, r_des=r_des, time_parallel=False)      # (the default kernels: one set of bits per satellite)
        sim = Simulator(sats=[sat], controller=c, scale=SatelliteScale(sat=sat), base_res=sim_res, verbose=False)
        sim.run_segments(tf=tf, num_segments=nseg)
        single.append((sim.sim_data[sat.id], sim.sim_time[sat.id], c.opt_trajectory, sat.get_state_vector().copy(), c.last_status))
    sats = make()
    mpc = ConstellationMPC(sats, base_res=base_res, tf_horizon=tf, tf_interval=tf / nseg, r_des=r_des, sim_base_res=sim_res)
    mpc.run_segments(tf=tf, num_segments=nseg)
    assert (mpc.last_status == 0).all(), mpc.last_status
    for i, (sat, (data, time, plan, state, status)) in enumerate(zip(sats, single)):
        assert status == [0, 0]
        assert mpc.sim_data[sat.id].shape == data.shape == (7, 80)
        assert np.array_equal(mpc.sim_data[sat.id], data) and np.array_equal(mpc.sim_time[sat.id], time)
        assert np.array_equal(mpc.plan_x[i], plan)
        assert np.array_equal(sat.get_state_vector(), state)
    assert mpc.horizon == 1.0


def test_constellation_mpc_plan_vs_oracle_chain():
    """ConstellationMPC.update for satellites with different scales against the oracle chain of each (not against
    this repo's own single-satellite loop): the plan after both SCP iterations."""
    from mpconstellation_amd import Satellite, SatelliteScale, ConstellationMPC
    from mpconstellation_amd.constellation import constellation_states
    st = constellation_states(4096)[[5, 977, 3010]]
    sats = [Satellite(s[:3], s[3:6], s[6]) for s in st]
    mpc = ConstellationMPC(sats, base_res=15, tf_horizon=2, tf_interval=1, r_des=1.2, sim_base_res=40)
    mpc.update()
    assert (mpc.last_status == 0).all()
    for i, sat in enumerate(sats):
        sc = SatelliteScale(sat=sat)
        r = oracle_scp_chain(sc.normalize_state(sat.get_state_vector()), sc.get_normalized_constants().as_vector(), 15, 2, 1.2)
        assert mpc.plan_x[i].shape == r["X"].shape
        assert np.abs(mpc.plan_x[i] - r["X"]).max() < 5e-6 and np.abs(mpc.plan_u[i] - r["U"]).max() < 5e-5
        assert abs(mpc.plan_tf[i] - r["tf"]) < 5e-6


def test_fused_scp_iteration_equals_its_three_calls():
    """mpcx_scp_iteration_batch_ragged (rollout + extract_uk + discretize + solve in one call, x_bar / u_bar staying on the
    device) against the same chain through the separate entry points: bit for bit, rectangular and ragged, tangential
    reference law and sequence playback."""
    from mpconstellation_amd import _ffi, mpc_step_batch, scp_iteration_batch
    from mpconstellation_amd.simulator import propagate_batch
    from mpconstellation_amd.constellation import constellation_states, normalize_batch
    S, K = 24, 30
    y0, consts = normalize_batch(constellation_states(4096, first=500, count=S))
    tf = np.linspace(0.8, 1.2, S)
    law = (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None)
    x, st, _, u = propagate_batch(y0, tf, consts, law, K, thrust=True)
    r_des = np.linalg.norm(x[:, :3, -1], axis=1)
    ref = mpc_step_batch(x, u, tf, consts, r_des)
    one = scp_iteration_batch(y0, tf, consts, r_des, law, K, return_reference=True)
    assert (one.prop_status == 0).all() and np.array_equal(one.xbar, x) and np.array_equal(one.ubar, u)
    for f in ("X", "U", "NU", "tf", "status", "iters", "kkt"): assert np.array_equal(getattr(one, f), getattr(ref, f)), f
    # second SCP iteration: the optimised sequence played over its own horizon, int(30 tf) nodes per satellite
    Kn = (30 * ref.tf).astype(np.int32); law2 = (_ffi.CTRL_SEQUENCE, ref.U, K, 1.0)
    x2, st2, _, u2 = propagate_batch(y0, ref.tf, consts, law2, Kn, Kus=np.full(S, K), thrust=True)
    ref2 = mpc_step_batch(x2, u2, ref.tf, consts, r_des, Ks=Kn)
    two = scp_iteration_batch(y0, ref.tf, consts, r_des, law2, int(Kn.max()), Ks=Kn, Kus=np.full(S, K))
    assert two.xbar is None and (two.prop_status == 0).all() and (two.status == 0).all()
    for f in ("X", "U", "NU", "tf", "status", "iters", "kkt"): assert np.array_equal(getattr(two, f), getattr(ref2, f)), f


def test_update_in_one_call_equals_one_call_per_iteration():
    """mpcx_mpc_update_batch -- OptimalController.update (control.py:170-235) as ONE library call, node counts
    int(base_res * tf_u) computed on the device, the plan's thrust consumed in place, the segment flight riding along --
    against the same update as one call per SCP iteration with the plan crossing PCIe in between (the verbose path, which
    prints control.py:208-209's lines) and the flight as its own call: bit for bit, plans, statuses, flown trajectories."""
    import contextlib, io
    from mpconstellation_amd import Satellite, ConstellationMPC
    from mpconstellation_amd.constellation import constellation_states
    st = constellation_states(4096)[[3, 500, 1234, 2222, 4000]]
    make = lambda: [Satellite(s[:3].copy(), s[3:6].copy(), float(s[6])) for s in st]
    kw = dict(base_res=30, tf_horizon=2, tf_interval=1, r_des=1.5, sim_base_res=50)
    a = ConstellationMPC(make(), **kw)
    b = ConstellationMPC(make(), verbose=True, **kw)
    for seg in range(2):
        a.run_segment(1)
        with contextlib.redirect_stdout(io.StringIO()) as out:
            b.run_segment(1)
        assert out.getvalue().count("tf for optimizer") == 2 * 5
        assert (a.last_status == 0).all() and np.array_equal(a.last_status, b.last_status)
        assert np.array_equal(a.plan_K, b.plan_K) and np.array_equal(a.plan_tf, b.plan_tf)
        assert len(set(a.plan_K.tolist())) > 1                               # really a ragged second iteration
        for i in range(5):
            assert np.array_equal(a.plan_x[i], b.plan_x[i]) and np.array_equal(a.plan_u[i], b.plan_u[i]) and np.array_equal(a.plan_nu[i], b.plan_nu[i])
            assert a.plan_x[i].shape == (7, a.plan_K[i])
        # rows of length K, zeros behind a satellite's last node
        X = a._plan[0]
        assert X.shape[2] == int(30 * (2 - seg)) and all((X[i, :, a.plan_K[i]:] == 0).all() for i in range(5))
    for sa, sb in zip(a.sats, b.sats):
        assert np.array_equal(a.sim_data[sa.id], b.sim_data[sb.id]) and np.array_equal(sa.get_state_vector(), sb.get_state_vector())
    assert a.sim_data[a.sats[0].id].shape == (7, 100) and a.horizon == 1.0
    assert a.sim_time[a.sats[0].id] is not a.sim_time[a.sats[1].id]           # an array per id, as the reference keeps them
    # three SCP iterations: the third rollout plays a RAGGED table (the second iteration's plan) -- one call against three
    c = ConstellationMPC(make(), scp_iterations=3, **kw)
    d = ConstellationMPC(make(), scp_iterations=3, verbose=True, **kw)
    c.update()
    with contextlib.redirect_stdout(io.StringIO()):
        d.update()
    assert c.last_status.shape == (3, 5) and (c.last_status == 0).all() and np.array_equal(c.last_status, d.last_status)
    assert np.array_equal(c.plan_K, d.plan_K) and np.array_equal(c.plan_tf, d.plan_tf) and np.array_equal(c.last_iters, d.last_iters)
    for i in range(5): assert np.array_equal(c.plan_x[i], d.plan_x[i]) and np.array_equal(c.plan_u[i], d.plan_u[i])


def test_several_devices_from_the_api():
    """devices=[...] on the drop-in API (north star: satellites shard across the GPUs of a node behind the reference's API; the
    reference loops over its constellation serially, simulator.py:41,58): contiguous blocks, one host thread and one context
    per entry of the device list, results joined.  Rehearsed on this one-GPU box with devices=[0, 0] and [0, 0, 0] -- separate
    contexts and streams on the same device, calls in flight together -- against the single-device call, bit for bit
    (2 x 4096 satellites: each block on the one-wave kernel; an uneven split of 2051: blocks on the two-wave kernel against a
    whole on the one-wave kernel); and the closed loop with the constellation dealt out to three contexts."""
    from mpconstellation_amd import Satellite, ConstellationMPC, mpc_step_batch, mpc_update_batch, propagate_batch, _ffi
    from mpconstellation_amd.constellation import constellation_states, normalize_batch
    from test_full_size_gpu import workload
    for S, devs in ((8192, [0, 0]), (2051, [0, 0, 0])):
        xbar, ubar, consts, r_des = workload(8192, 30, first=0, count=S)
        tf = np.ones(S)
        one = mpc_step_batch(xbar, ubar, tf, consts, r_des)
        many = mpc_step_batch(xbar, ubar, tf, consts, r_des, devices=devs)
        assert (many.status == 0).all()
        for f in ("X", "U", "NU", "tf", "status", "iters", "kkt"): assert np.array_equal(getattr(one, f), getattr(many, f)), (S, f)
    # the closed loop: update + flight for 7 satellites on three contexts (blocks of 3, 2, 2)
    st = constellation_states(4096)[[1, 100, 900, 1500, 2500, 3100, 4090]]
    make = lambda: [Satellite(s[:3].copy(), s[3:6].copy(), float(s[6])) for s in st]
    kw = dict(base_res=30, tf_horizon=2, tf_interval=1, r_des=1.5, sim_base_res=40)
    a = ConstellationMPC(make(), **kw); b = ConstellationMPC(make(), devices=[0, 0, 0], **kw)
    a.run_segments(tf=2, num_segments=2); b.run_segments(tf=2, num_segments=2)
    assert (b.last_status == 0).all() and b.last_status.shape == (2, 7) and np.array_equal(a.last_status, b.last_status)
    for i, (sa, sb) in enumerate(zip(a.sats, b.sats)):
        assert np.array_equal(a.plan_x[i], b.plan_x[i]) and np.array_equal(a.sim_data[sa.id], b.sim_data[sb.id])
    # ragged rollouts through several contexts: rows padded to the longest satellite of the whole batch
    y0, consts = normalize_batch(st)
    n = np.array([30, 12, 25, 30, 7, 19, 28])
    law = (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None)
    y1 = propagate_batch(y0, 1.0, consts, law, n, thrust=True)
    y3 = propagate_batch(y0, 1.0, consts, law, n, thrust=True, devices=[0, 0, 0])
    for p, q in zip(y1, y3): assert np.array_equal(p, q)
    # the blocks write IN PLACE into one result set: whole C-ordered arrays come back (no join), the regularisation record and
    # page-locked results (every context's DMA lands in the one set) work with devices= as well
    S = 2051
    xbar, ubar, consts, r_des = workload(8192, 30, first=0, count=S)
    one = mpc_step_batch(xbar, ubar, np.ones(S), consts, r_des, regularised=True)
    for kw in (dict(regularised=True), dict(pinned_results=True)):
        many = mpc_step_batch(xbar, ubar, np.ones(S), consts, r_des, devices=[0, 0, 0], **kw)
        assert many.X.flags.c_contiguous and many.X.shape == (S, 7, 30)
        for f in ("X", "U", "NU", "tf", "status", "iters", "kkt"): assert np.array_equal(getattr(one, f), getattr(many, f)), (kw, f)
        if "regularised" in kw:
            assert np.array_equal(one.n_regularised, many.n_regularised) and np.array_equal(one.first_regularised, many.first_regularised)
    upd1 = mpc_update_batch(y0, 2.0, normalize_batch(st)[1], 1.5, 30, options=ConstellationMPC.OPTIONS(2.0), fly=(1.0, 1.0, 40, True, True))
    upd3 = mpc_update_batch(y0, 2.0, normalize_batch(st)[1], 1.5, 30, options=ConstellationMPC.OPTIONS(2.0), fly=(1.0, 1.0, 40, True, True), devices=[0, 0, 0])
    for f in ("X", "U", "NU", "tf", "status", "iters", "kkt", "Ks", "prop_status", "y_sim", "sim_status"):
        assert np.array_equal(getattr(upd1, f), getattr(upd3, f)), f
    assert upd3.status.shape == (2, 7) and upd3.y_sim.shape == (7, 7, 40)


def test_split_update_two_chains_equal_one():
    """mpcx_mpc_update_batch runs a large batch (>= 2048 satellites) as TWO chains -- its halves, each rollout -> discretise ->
    solve -> re-rollout -> ... -> flight on its own stream, so that one half's rollouts and the tail of its solve launch run under
    the other half's solve (control.py:178-180,221-227 are the rollouts; satellites are independent).  Same bits as ONE chain
    (MPCX_UPDATE_SPLIT=0), with and without the delayed start of the second chain (=2), three SCP iterations (a ragged table
    played back), an odd batch size, the regularisation record of both halves in one array."""
    import os
    from mpconstellation_amd import mpc_update_batch, ConstellationMPC, _ffi
    from mpconstellation_amd.constellation import constellation_states, normalize_batch
    S = 2049
    y0, consts = normalize_batch(constellation_states(4096, first=100, count=S))
    kw = dict(n_scp=3, options=ConstellationMPC.OPTIONS(2.0), fly=(1.0, 1.0, 25, True, True))
    res = {}
    old = os.environ.get("MPCX_UPDATE_SPLIT")
    try:
        for mode in ("0", "1", "2"):
            os.environ["MPCX_UPDATE_SPLIT"] = mode
            r = mpc_update_batch(y0, 2.0, consts, 1.5, 30, **kw)
            reg = np.zeros((S, 2), dtype=np.int32)
            lib = _ffi.load(); ctx = _ffi.context(0)
            _ffi.check(lib.mpcx_solve_regularised(ctx, S, _ffi.iptr(reg)), ctx, "mpcx_solve_regularised")
            res[mode] = (r, reg)
    finally:
        if old is None: os.environ.pop("MPCX_UPDATE_SPLIT", None)
        else: os.environ["MPCX_UPDATE_SPLIT"] = old
    one = res["0"][0]
    assert np.isin(one.status, (0, 7)).mean() > 0.99 and (one.prop_status == 0).all() and (one.sim_status == 0).all()
    assert len(set(one.Ks.tolist())) > 1
    for mode in ("1", "2"):
        two, reg = res[mode]
        for f in ("X", "U", "NU", "tf", "status", "iters", "kkt", "Ks", "prop_status", "y_sim", "sim_status"):
            assert np.array_equal(getattr(one, f), getattr(two, f)), (mode, f)
        assert np.array_equal(reg, res["0"][1]), mode
