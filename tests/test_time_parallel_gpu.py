"""The time-parallel solve kernel (MPCX_SOLVE_TIME_PARALLEL, csrc/solve_tp.hip: the horizon in up to four segments, a workgroup
each, joined by a coarse recursion over the cuts) against the sequential kernels through the C ABI.  It is
the same Newton direction computed another way: same statuses, same iteration counts, solutions within the solver's tolerance
(the algebra itself is pinned on the CPU oracle: tests/test_partitioned_riccati.py)."""
import numpy as np
import pytest

from test_full_size_gpu import workload

pytestmark = pytest.mark.gpu
TP = 64          # MPCX_SOLVE_TIME_PARALLEL


def _pair(S, K, first=0, **kw):
    from mpconstellation_amd import mpc_step_batch
    xbar, ubar, consts, r_des = workload(4096, K, first=first, count=S)
    tf = np.ones(S)
    a = mpc_step_batch(xbar, ubar, tf, consts, r_des, regularised=True, **kw)
    b = mpc_step_batch(xbar, ubar, tf, consts, r_des, regularised=True, flags=TP, **kw)
    return a, b


@pytest.mark.parametrize("S,K", [(64, 30), (64, 60), (37, 17), (16, 9), (5, 5), (128, 30), (3, 100)])
def test_same_statuses_iterations_and_solutions(S, K):
    """four segments from 24 nodes on, two from 8, one below (the sequential recursion on the kernel's own plumbing)"""
    a, b = _pair(S, K)
    assert (a.status == 0).all() and np.array_equal(a.status, b.status)
    # (the directions agree to ~1e-10: a convergence test within that of its tolerance may fall the other way -- one iteration
    #  more or less on a satellite in a hundred, and then a solution that differs by what the last iteration still moves)
    assert np.abs(a.iters - b.iters).max() <= 1 and (a.iters == b.iters).mean() >= 0.95, (a.iters.tolist(), b.iters.tolist())
    assert np.array_equal(a.n_regularised, b.n_regularised)
    same = a.iters == b.iters
    assert np.abs(a.X - b.X)[same].max() < 1e-7 and np.abs(a.U - b.U)[same].max() < 1e-7 and np.abs(a.tf - b.tf)[same].max() < 1e-8
    assert np.abs(a.X - b.X).max() < 1e-5 and np.abs(a.NU - b.NU).max() < 1e-5


def test_batches_above_the_limit_take_the_other_kernels():
    a, b = _pair(160, 30)
    assert np.array_equal(a.X, b.X) and np.array_equal(a.U, b.U) and np.array_equal(a.iters, b.iters)


def test_stiff_option_set_with_refinement_passes():
    """OptimalController's option set (terminal windows 1e-6 / 1e-16: barrier weights beyond the refinement threshold, regularised
    iterations): the refinement passes go through the segments' exchange as well"""
    opts = dict(eps_r=1e-6, eps_vr=1e-16, tf_max=1.0)
    a, b = _pair(32, 30, options=opts)
    ok = np.isin(a.status, (0, 7))
    assert ok.all() and np.isin(b.status, (0, 7)).all()
    assert np.abs(a.iters - b.iters).max() <= 2 and (a.iters == b.iters).mean() >= 0.75, (a.iters.tolist(), b.iters.tolist())
    assert np.abs(a.X - b.X).max() < 1e-5 and np.abs(a.tf - b.tf).max() < 1e-6


@pytest.mark.parametrize("kw", [dict(linear_vt=True), dict(fixed_tf=1.05), dict(linear_vt=True, fixed_tf=0.97), dict(options={"w_tr": 0.02})])
def test_solver_variants(kw):
    """the convex tangential pair (another border), a fixed final time (dtf out of the border), another trust-region weight"""
    kw = dict(kw)
    if "fixed_tf" in kw: kw["fixed_tf"] = np.full(32, kw["fixed_tf"])
    a, b = _pair(32, 30, **kw)
    assert (a.status == 0).all() and (b.status == 0).all()
    assert np.abs(a.iters - b.iters).max() <= 1 and (a.iters == b.iters).mean() >= 0.9
    assert np.abs(a.X - b.X).max() < 1e-6 and np.abs(a.tf - b.tf).max() < 1e-7


def test_ragged_batch_and_refused_node_counts():
    from mpconstellation_amd import mpc_step_batch
    S, K = 12, 40
    xbar, ubar, consts, r_des = workload(4096, K, first=0, count=S)
    Ks = np.array([40, 33, 25, 24, 23, 12, 8, 7, 5, 3, 2, 40], dtype=np.int32)        # 2: refused (MPCX_ST_BADK)
    tf = np.ones(S)
    a = mpc_step_batch(xbar, ubar, tf, consts, r_des, Ks=Ks)
    b = mpc_step_batch(xbar, ubar, tf, consts, r_des, Ks=Ks, flags=TP)
    assert np.array_equal(a.status, b.status) and a.status[10] == 9 and (np.delete(a.status, 10) == 0).all()
    assert np.abs(a.iters - b.iters).max() <= 1
    assert np.abs(a.X - b.X).max() < 1e-6 and np.abs(a.U - b.U).max() < 1e-6
    assert np.array_equal(a.X[10], b.X[10]) and np.array_equal(a.NU[10], b.NU[10])       # the refused satellite: reference rows back, both ways


def test_empty_constraint_set_is_reported_the_same_way():
    a, b = _pair(4, 30, options={"r_lim": [1.01, 5]})     # the fixed start node below the r_min plane: MPCX_ST_INFEASIBLE before the first iteration
    assert (a.status == 8).all() and np.array_equal(a.status, b.status) and np.array_equal(a.X, b.X) and (b.iters == 0).all()


def test_closed_loop_with_the_time_parallel_kernel():
    """ConstellationMPC(time_parallel=True): plan and flown states of the reference's test_mpc configuration for five satellites
    against the default kernels"""
    from mpconstellation_amd import Satellite, ConstellationMPC
    from mpconstellation_amd.constellation import constellation_states
    st = constellation_states(5)
    make = lambda: [Satellite(s[:3].copy(), s[3:6].copy(), float(s[6])) for s in st]
    kw = dict(base_res=30, tf_horizon=2, tf_interval=1, r_des=1.5, sim_base_res=100)
    a = ConstellationMPC(make(), **kw); b = ConstellationMPC(make(), time_parallel=True, **kw)
    a.run_segments(tf=2, num_segments=2); b.run_segments(tf=2, num_segments=2)
    assert np.isin(a.last_status, (0, 7)).all() and np.isin(b.last_status, (0, 7)).all()
    assert np.abs(a.last_iters - b.last_iters).max() <= 2
    for sa, sb in zip(a.sats, b.sats):
        assert np.allclose(sa.position, sb.position, rtol=1e-6) and np.allclose(sa.velocity, sb.velocity, rtol=1e-6)


def test_optimal_controller_on_the_time_parallel_kernel():
    """the reference's controller plans for one satellite (control.py:162): its two SCP iterations both ways"""
    from mpconstellation_amd import Satellite
    from mpconstellation_amd.control import OptimalController
    from mpconstellation_amd.constellation import constellation_states
    st = constellation_states(1)[0]
    plans = []
    for tpar in (False, True):
        sat = Satellite(st[:3].copy(), st[3:6].copy(), float(st[6]))
        c = OptimalController([sat], base_res=30, tf_horizon=2, tf_interval=1, r_des=1.5, opt_verbose=False, plot_inter=False, time_parallel=tpar)
        c.update()
        assert all(code in (0, 7) for code in c.last_status)
        plans.append((c.opt_trajectory.copy(), c.sequence_controller.u.copy(), c.sequence_controller.end_tau))
    assert np.abs(plans[0][0] - plans[1][0]).max() < 1e-6 and np.abs(plans[0][1] - plans[1][1]).max() < 1e-6 and abs(plans[0][2] - plans[1][2]) < 1e-8
    # ... and it is the controller's default (round 5): one satellite is the case the kernel is for
    sat = Satellite(st[:3].copy(), st[3:6].copy(), float(st[6]))
    c = OptimalController([sat], base_res=30, tf_horizon=2, tf_interval=1, r_des=1.5, opt_verbose=False, plot_inter=False)
    assert c.time_parallel is True
    c.update()
    assert np.array_equal(c.opt_trajectory, plans[1][0])


def test_timed_out_time_parallel_update_is_repeated_on_the_default_kernels():
    """MPCX_ST_TIMEOUT (a workgroup of a time-parallel solve did not become resident in time: forced here with the library's
    self-test flag) is not a failed plan: ConstellationMPC.update warns and repeats the update on the default kernels."""
    import warnings
    from mpconstellation_amd import Satellite, ConstellationMPC, _ffi
    from mpconstellation_amd.constellation import constellation_states
    st = constellation_states(3)
    make = lambda: [Satellite(s[:3].copy(), s[3:6].copy(), float(s[6])) for s in st]
    kw = dict(base_res=30, tf_horizon=1, tf_interval=1, r_des=1.2, sim_base_res=40)
    ref = ConstellationMPC(make(), **kw); ref.update()
    b = ConstellationMPC(make(), time_parallel=True, **kw)
    b.solver_flags |= _ffi.SOLVE_TP_SELFTEST_DEAD
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        b.update()
    assert any("timed out" in str(w.message) for w in wl)
    assert (b.last_status == 0).all() and all(np.array_equal(p, q) for p, q in zip(ref.plan_x, b.plan_x))
