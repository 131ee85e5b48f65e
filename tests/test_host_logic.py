"""Host-side mirrors of the reference API (no GPU needed): units, controllers, constraint terms, options,
constellation generator, sharding; and a world_size-2 gloo run of the multi-GPU plumbing."""
import os

import numpy as np
import pytest

import oracle_lib as O


def hubble():
    from mpconstellation_amd import Satellite
    return Satellite(np.array([5371.4806, -4133.1393, 1399.9594]) * 1000, np.array([4.6921, 4.9848, -3.2752]) * 1000, 12200)


def test_scale_and_constants(golden_dir):
    from mpconstellation_amd import SatelliteScale
    c = np.load(os.path.join(golden_dir, "constants_hubble.npz"))
    sc = SatelliteScale(sat=hubble())
    assert np.array_equal(sc.get_normalized_constants().as_vector(), c["const"])
    assert np.array_equal(sc.normalize_state(c["state"]), c["x_norm"])
    assert np.array_equal(sc.redim_state(c["x_norm"]), c["x_redim"])
    x2 = np.column_stack([c["state"], c["state"]])
    assert np.array_equal(sc.redim_state(sc.normalize_state(x2))[:, 0], c["x_redim"])
    assert np.allclose(sc.normalize_thrust(sc.redim_thrust(np.ones(3))), 1.0)
    d = SatelliteScale()                                     # default: unit scale (satellite_scale.py:25-26)
    assert d._r0 == 1 and d._m0 == 1


def test_satellite_ids_unique():
    from mpconstellation_amd import Satellite
    ids = {Satellite().id for _ in range(2000)}              # reference test_satellite.py:21-28
    assert len(ids) == 2000
    s = hubble(); s.update_state_vector(np.arange(7.0))
    assert np.array_equal(s.get_state_vector(), np.arange(7.0))


def test_controllers_match_oracle_and_golden(golden_dir):
    from mpconstellation_amd import (Discretizer, ConstantTangentialThrustController, ConstantThrustController,
                                     SequenceController, Controller)
    d = np.load(os.path.join(golden_dir, "disc_tan_K30_tf1.npz"))
    c = ConstantTangentialThrustController([], 0.5)
    u = Discretizer.extract_uk(d["x"], d["t"], c)
    assert np.abs(u - d["u"]).max() < 1e-14                  # extract_uk, linearize_discretize.py:393-411
    fo = np.load(os.path.join(golden_dir, "foh.npz"))
    sc = SequenceController(u=fo["u_30"], tf_u=0.6, tf_sim=1.0)
    f = sc.get_u_func()
    oc = O.make_ctrl(O.CTRL_SEQUENCE, useq=fo["u_30"], end_tau=0.6)
    import ctypes
    for tau in np.concatenate([fo["tau_30"], [0.6, 0.61, 1.0]]):
        out = np.zeros(3)
        O.lib().oracle_ctrl_eval.argtypes = [ctypes.POINTER(O.OracleCtrl), ctypes.POINTER(ctypes.c_double), ctypes.c_double,
                                             ctypes.POINTER(ctypes.c_double)]
        O.lib().oracle_ctrl_eval(ctypes.byref(oc), d["x"][:, 0].copy().ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                 float(tau), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        assert np.array_equal(f(None, tau), out)
    assert np.array_equal(Controller().get_u_func()(None, 0.3), np.zeros(3))
    assert np.array_equal(ConstantThrustController([], np.array([1., 2., 3.])).get_u_func()(None, 0.3), [1, 2, 3])
    assert sc.device_law()[0] == 3 and c.device_law()[0] == 2


def test_constraint_terms_mirror(golden_dir):
    from mpconstellation_amd import Optimizer, SatelliteScale, Discretizer
    d = np.load(os.path.join(golden_dir, "disc_tan_K60_tf2.npz"))
    scale = SatelliteScale(sat=hubble())
    opt = Optimizer([d["x"]], [d["u"]], [np.zeros_like(d["x"])], 2, Discretizer(scale.get_normalized_constants()),
                    None, scale, verbose=False)
    ct = opt.get_constraint_terms()
    for k, v in ct.items():
        assert np.allclose(v[0], d["ct_" + k], rtol=0, atol=1e-13, equal_nan=True), k
    assert opt.init_options({"r_des": 1.3})["r_des"] == 1.3 and opt.init_options({})["w_nu"] == 1000
    # several satellites share one tf by default, as in the reference (optimizer.py:287)
    assert Optimizer([d["x"], d["x"]], [d["u"], d["u"]], [None, None], 2, None, None, scale).shared_tf
    assert not Optimizer([d["x"], d["x"]], [d["u"], d["u"]], [None, None], 2, None, None, scale, shared_tf=False).shared_tf
    assert not opt.shared_tf


def test_host_dynamics_matches_oracle(golden_dir):
    from mpconstellation_amd import Simulator
    from mpconstellation_amd.constants import Constants
    p = np.load(os.path.join(golden_dir, "pointwise.npz"))
    const = Constants(*p["const"])
    for i in range(0, 128, 7):
        u = p["u"][i]
        f = Simulator.satellite_dynamics(0.3, p["x"][i], lambda y, t: u, p["tf"][i], const, include_drag=True, include_J2=True)
        assert np.abs(f - p["f_drag_j2"][i]).max() < 1e-12


def test_constellation_generator_and_sharding(golden_dir):
    from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
    from mpconstellation_amd.sharding import shard_block
    c64 = np.load(os.path.join(golden_dir, "constellation64.npz"))
    st = constellation_states(64)
    y0, consts = normalize_batch(st)
    for i in c64["idx"]:
        assert np.abs(st[i] - c64[f"state_{i}"]).max() < 1e-6          # metres: same generator as the golden script
        assert np.abs(consts[i] - c64[f"const_{i}"]).max() / np.abs(c64[f"const_{i}"]).max() < 1e-14
        assert np.abs(tangential_thrust(c64[f"x_{i}"][None], 0.5)[0] - c64[f"u_{i}"]).max() < 1e-14
    assert np.array_equal(constellation_states(64, first=10, count=5), st[10:15])
    for S, W in ((65536, 8), (64, 2), (10, 3), (5, 8)):
        blocks = [shard_block(S, W, r) for r in range(W)]
        assert sum(c for _, c in blocks) == S and blocks[0][0] == 0
        assert all(blocks[r][0] + blocks[r][1] == blocks[r + 1][0] for r in range(W - 1))
        assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def test_sharded_call_blocks_contexts_and_results_in_place():
    """The multi-device path of the drop-in API (ConstellationMPC / mpc_step_batch / mpc_update_batch with devices=[...]):
    contiguous blocks in device order, one context (device, slot) per entry of the device list -- a device named twice gets
    two slots --, no two calls in flight on one context (a context is not thread-safe, include/mpcx.h), the calls really
    concurrent, and every block writing IN PLACE into its slice of ONE result set for the constellation: C-contiguous views
    along the satellite axis (the library's copy-out is the only pass over the results), a small temporary only for a block's
    columns of an update's per-iteration records (n_scp, S)."""
    import threading
    import time
    from mpconstellation_amd.sharding import device_contexts, sharded_call, shard_block, OutArrays, last_call
    assert device_contexts([0, 1, 0, 0, 1]) == [(0, 0), (1, 0), (0, 1), (0, 2), (1, 1)]
    lock = threading.Lock(); busy = set(); clashes = []; peak = [0]; seen = []

    def fake_call(x, k, scale, device=0, slot=0, out=None):
        with lock:
            if (device, slot) in busy: clashes.append((device, slot))
            busy.add((device, slot)); peak[0] = max(peak[0], len(busy))
        time.sleep(0.05)
        with lock:
            busy.discard((device, slot))
        n = x.shape[0]
        oa = OutArrays(out)
        X = oa.get("X", x.shape); Ks = oa.get("Ks", (n,), np.int64); st = oa.get("status", (2, n), np.int32)
        with lock:
            seen.append((device, slot, X.base is not None and X.flags.c_contiguous, st.flags.c_contiguous and len(oa.late) == 1))
        X[...] = x * scale; Ks[...] = 7 if k is None else k; st[...] = 10 * device + slot
        oa.finish()
        return (device, slot, n)

    x = np.arange(11 * 3, dtype=np.float64).reshape(11, 3)
    outs = dict(X=np.full((11, 3), np.nan), Ks=np.zeros(11, dtype=np.int64), status=(np.full((2, 11), -1, dtype=np.int32), 1), unused=None)
    parts = sharded_call(fake_call, [0, 0, 1], [x, None], outs, 2.0)
    assert not clashes and peak[0] == 3                        # three contexts, all in flight together
    assert parts == [(0, 0, 4), (0, 1, 4), (1, 0, 3)] == [(d, s, shard_block(11, 3, r)[1]) for r, (d, s) in enumerate([(0, 0), (0, 1), (1, 0)])]
    # every block wrote through a contiguous VIEW of the whole array (no copy), the (n_scp, S) record through one temporary
    assert all(view and late for _, _, view, late in seen)
    assert np.array_equal(outs["X"], 2.0 * x) and outs["Ks"].tolist() == [7] * 11
    assert outs["status"][0][0].tolist() == [0] * 4 + [1] * 4 + [10] * 3 and np.array_equal(outs["status"][0][0], outs["status"][0][1])
    assert len(last_call["blocks"]) == 3 and last_call["wall"][1] - last_call["wall"][0] < 0.14      # (3 x 0.05 s side by side)
    # a view of the wrong shape or type is refused, a missing name means "allocate"
    with pytest.raises(ValueError):
        OutArrays({"X": np.zeros((3, 3))}).get("X", (4, 3))
    assert OutArrays(None).get("X", (4, 3)).shape == (4, 3)
    # more devices than satellites: the empty blocks are skipped; a single block runs on the caller's thread
    o2 = dict(X=np.zeros((2, 3)), Ks=np.zeros(2, dtype=np.int64), status=(np.zeros((2, 2), dtype=np.int32), 1))
    parts = sharded_call(fake_call, [0, 1, 2, 3], [x[:2], np.array([5, 6])], o2, 1.0)
    assert parts == [(0, 0, 1), (1, 0, 1)] and o2["Ks"].tolist() == [5, 6] and np.array_equal(o2["X"], x[:2])


def test_foreign_thrust_laws_and_unknown_options_are_rejected():
    """The device propagates the four thrust laws of control.py; a callable it cannot see must not be silently replaced
    by one of them (reference extension points: get_trajectory_ODE's u_func argument, simulator.py:164-189, and
    Controller.get_u_func overrides, control.py:20-29)."""
    from mpconstellation_amd import Simulator, Controller, ConstantThrustController, _ffi
    sim = Simulator(sats=[hubble()], controller=ConstantThrustController([], np.ones(3)))
    with pytest.raises(NotImplementedError):
        sim.get_trajectory_ODE(hubble(), 1.0, lambda x, tau: np.ones(3))
    with pytest.raises(NotImplementedError):                 # another controller's law is foreign too
        sim.get_trajectory_ODE(hubble(), 1.0, Controller().get_u_func())
    assert sim.controller.get_u_func()._mpcx_controller is sim.controller     # its own law passes the check

    class Mine(Controller):
        def get_u_func(self, sat_id=None):
            return lambda x, tau: np.array([1., 0., 0.])

    with pytest.raises(NotImplementedError):
        Simulator(sats=[hubble()], controller=Mine())._device_law()

    class MineWithLaw(Mine):
        def device_law(self):
            return _ffi.CTRL_CONSTANT, np.array([1., 0., 0.]), 0, None

    assert Simulator(sats=[hubble()], controller=MineWithLaw())._device_law()[0] == _ffi.CTRL_CONSTANT
    assert Simulator(sats=[hubble()], controller=Controller())._device_law()[0] == _ffi.CTRL_ZERO
    with pytest.raises(TypeError):
        _ffi.check_solver_keywords({"max_iters": 10})        # misspelt: the field is max_iter
    _ffi.check_solver_keywords({"max_iter": 10, "tol": 1e-9})


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from mpconstellation_amd.constellation import constellation_states
    from mpconstellation_amd.sharding import shard_block, gather_trajectories
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = 11
    first, count = shard_block(S, world, rank)
    local = torch.tensor(constellation_states(S, first=first, count=count))
    full = gather_trajectories(local)                       # the optional final gather (ragged blocks)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                # bench.py: max over ranks of the timed region
    dist.barrier()
    if rank == 0:
        q.put((full.numpy(), float(t.item())))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    from mpconstellation_amd.constellation import constellation_states
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    full, tmax = q.get(timeout=120)
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    assert np.array_equal(full, constellation_states(11))
    assert abs(tmax - 0.2) < 1e-12


def test_unconverged_plans_are_not_flown_silently():
    import warnings
    from mpconstellation_amd.control import _check_solver_status
    _check_solver_status(np.array([0, 7, 0]), strict=True)                 # OK / acceptable: nothing happens
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _check_solver_status(np.array([0, 5]), strict=False)
    assert len(w) == 1 and issubclass(w[0].category, RuntimeWarning) and "max_iter" in str(w[0].message)
    with pytest.raises(RuntimeError):
        _check_solver_status(np.array([6]), strict=True)


def test_shared_tf_root_reports_a_missing_bracket():
    """The scalar outer search of the shared-tf mode (Optimizer with several satellites, optimizer.py:287): a root, the
    upper bound when the range constraint is active, and -- instead of silently returning the last point tried -- a search
    marked not converged when G never changes sign."""
    from mpconstellation_amd.optimizer import shared_tf_root
    t, ev = shared_tf_root(lambda t: t - 0.7, 5.0, 1.0)
    assert ev.converged and abs(t - 0.7) < 1e-6
    t, ev = shared_tf_root(lambda t: t - 3.3, 5.0, 1.0)
    assert ev.converged and abs(t - 3.3) < 1e-6
    t, ev = shared_tf_root(lambda t: -1.0, 5.0, 1.0)                 # G < 0 up to tf_max: the bound is the solution
    assert ev.converged and t == 5.0
    t, ev = shared_tf_root(lambda t: 1.0 + t, 5.0, 1.0)              # G > 0 all the way down: no root on (0, tf_max]
    assert not ev.converged and "no root" in ev.message and len(ev) < 45 and t < 1e-5
    calls = []
    t, ev = shared_tf_root(lambda t: (calls.append(t), 1.0)[1], 5.0, 1.0, max_bracket=6)
    assert not ev.converged and len(calls) == len(ev) <= 8


def test_ragged_foh_resampling_matches_the_scalar_reference_formula():
    """ConstellationMPC's extract_uk for a whole ragged batch (foh_resample_ragged) against SequenceController.u_FOH
    (control.py:104-126, Python float floor division), satellite by satellite, bit for bit."""
    from foh_reference import foh_resample_ragged
    from mpconstellation_amd.control import SequenceController
    rng = np.random.default_rng(0)
    S, Kmax = 40, 37
    Ku = rng.integers(2, Kmax + 1, S); n = rng.integers(1, 60, S)
    u = rng.standard_normal((S, 3, Kmax))
    out = foh_resample_ragged(u, Ku, n)
    assert out.shape == (S, 3, n.max())
    for s in range(S):
        f = SequenceController(u=u[s][:, :Ku[s]], tf_u=1.3, tf_sim=1.3).get_u_func()
        ref = np.column_stack([f(None, tq) for tq in np.linspace(0, 1, n[s])])
        assert np.array_equal(ref, out[s][:, :n[s]]) and not out[s][:, n[s]:].any()


def test_result_arrays_are_recycled_only_when_dropped():
    """_ffi.result_pool: the wrappers' large result arrays are reused once the caller holds no reference to them any more (a
    fresh 7 MB numpy array is a fresh mapping whose pages fault on first write: 1.8 ms of a 8.8 ms call), and never while it
    does -- not the array, not a view of it, not a SolveResult that carries it."""
    from mpconstellation_amd import _ffi
    from mpconstellation_amd.optimizer import SolveResult
    pool = _ffi._ResultPool()
    shape = (4096, 7, 30)
    a = pool.take(shape); ida = id(a)
    b = pool.take(shape)
    assert b is not a                                    # a is still held
    res = SolveResult(b, None, None, None, None, None, None)
    del a, b
    c = pool.take(shape)
    assert id(c) == ida                                  # a was dropped: recycled
    d = pool.take(shape)
    assert d is not res.X and d is not c                 # b lives on inside the result object
    view = c[5]
    del c, d
    e = pool.take(shape)
    assert e.base is None and not np.shares_memory(e, view)      # a view keeps its base array out of circulation
    small = pool.take((64, 7, 30))
    assert pool.take((64, 7, 30)) is not small           # below 1 MB: plain allocations
    assert len(pool._pool[(shape, "<f8")]) <= pool.PER_SHAPE
    for n in range(40): pool.take((4096, 7, 31 + n))     # many shapes: the oldest leave the pool
    assert len(pool._pool) <= pool.SHAPES


def test_plot_normalized_thrust_and_verbose_devices(golden_dir):
    """the reference's Optimizer.plot_normalized_thrust (optimizer.py:47-77, called by its own test, test_optimizer.py:70): RTN
    components of a tangential thrust history are (0, |u|, 0); matplotlib is imported by the call, not by the package.
    ConstellationMPC(verbose=True, devices=[...]) is refused instead of silently solving on one device."""
    import matplotlib
    matplotlib.use("Agg")
    from mpconstellation_amd import Optimizer, ConstellationMPC
    d = np.load(os.path.join(golden_dir, "disc_tan_K20_tf2.npz"))
    x, u = d["x"], d["u"]
    rtn = Optimizer.thrust_rtn(x, u)
    assert np.abs(rtn[0]).max() < 1e-12 and np.abs(rtn[2]).max() < 1e-12 and np.allclose(rtn[1], np.linalg.norm(u, axis=0))
    fig = Optimizer.plot_normalized_thrust(x, u, show=False)
    ax = fig.axes[0]
    assert ax.get_title() == 'Normalized Thrust Commands' and [l.get_label() for l in ax.lines] == ["r", "t", "n"]
    assert np.array_equal(ax.lines[1].get_ydata(), rtn[1])
    with pytest.raises(ValueError):
        ConstellationMPC([hubble(), hubble()], verbose=True, devices=[0, 1])


def test_bench_line_carries_its_summary_ahead_of_the_lists():
    """bench.py's JSON line: the contract's keys first, then ONE short `summary` object with the secondary headline numbers
    (value_pcie, also.*, closed_loop.*, the multi-device host overhead), then roofline and cpu_baseline, then everything else --
    so that a reader of a truncated tail still finds them (round-4 verdict, bench item)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    out = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": 1.0, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "S4096_K30"},
           "host_pointer_entry": {"calls_ms": {"timed": [7.0] * 20}}, "value_pcie": 5.6e5,
           "roofline": {"kernel_ms": 5.33, "frac": 0.0053, "traffic": 1.3e10},
           "also": {"S64_K30": {"value": 46000.0, "roofline": {"kernel_ms": 1.31}},
                    "api_devices8": {"value": 6.7e5, "api_devices8_host_overhead_ms": 0.42}},
           "closed_loop": {"S4096": {"value": 96800.0}}, "cpu_baseline": {"value": 197.0, "all_cores": {"value": None}}}
    o = bench.with_summary(out)
    keys = list(o)
    assert keys[:len(bench.CONTRACT_KEYS)] == list(bench.CONTRACT_KEYS) and keys[len(bench.CONTRACT_KEYS)] == "summary"
    assert keys.index("roofline") < keys.index("host_pointer_entry") and keys.index("cpu_baseline") < keys.index("also")
    s = o["summary"]
    assert s["value_pcie"] == 560000.0 and s["also.S64_K30"] == 46000.0 and s["also.S64_K30.kernel_ms"] == 1.31
    assert s["also.api_devices8_host_overhead_ms"] == 0.42 and s["closed_loop.S4096"] == 96800.0 and s["cpu_baseline"] == 197.0
    assert "cpu_baseline_all_cores" not in s                      # (not run: nothing to show)
    line = json.dumps(o)
    assert line.index('"summary"') < 700 and set(out) <= set(o)   # within the first few hundred characters; nothing dropped
    # the CPU quota of the process's cgroup: a number of CPUs or None, never an exception
    q = bench.cpu_quota()
    assert q is None or q > 0
