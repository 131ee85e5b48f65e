"""Size-independent properties of the hot path at BASELINE.json's full single-GPU sizes (configs[2] S=4096, K=30;
configs[3] shape K=100 on a 512-satellite block), and the edge cases of the boundary (minimum horizon, a single
satellite, rejected arguments).  Everything goes through the C ABI."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import oracle_lib as O
import nlp_ipm as N

pytestmark = pytest.mark.gpu


def workload(S_total, K, first=0, count=None):
    from mpconstellation_amd import _ffi
    from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
    from mpconstellation_amd.simulator import propagate_batch
    count = S_total - first if count is None else count
    y0, consts = normalize_batch(constellation_states(S_total, first=first, count=count))
    xbar, st, _ = propagate_batch(y0, np.ones(count), consts, (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), K)
    assert (st == 0).all()
    ubar = np.ascontiguousarray(tangential_thrust(xbar, 0.5))
    return xbar, ubar, consts, np.linalg.norm(xbar[:, :3, -1], axis=1)


def dyn_defect(A, Bp, Bn, Sig, xi, X, U, NU, tf):
    """max |x_{k+1} - (A_k x_k + B_kn u_k + B_kp u_{k+1} + Sigma_k tf + xi_k + nu_k)| per satellite (optimizer.py:327-342)"""
    pred = (np.einsum("skij,sjk->sik", A, X[:, :, :-1]) + np.einsum("skij,sjk->sik", Bn, U[:, :, :-1]) +
            np.einsum("skij,sjk->sik", Bp, U[:, :, 1:]) + Sig * tf[:, None, None] + xi + NU[:, :, :-1])
    return np.abs(X[:, :, 1:] - pred).max(axis=(1, 2))


@pytest.mark.parametrize("S,K", [(4096, 30), (512, 100)])
def test_full_size_properties(S, K):
    from mpconstellation_amd import mpc_step_batch, Discretizer
    xbar, ubar, consts, r_des = workload(S, K)
    tf = np.ones(S)
    res = mpc_step_batch(xbar, ubar, tf, consts, r_des)
    # every problem converges to the solver tolerance
    assert (res.status == 0).all() and res.kkt.max() <= 1e-8
    # ... in the iteration counts of the adaptive barrier rule (clean starts, 8.9 on average and 15 at most at K = 30, 7.1 / 10
    #     at K = 100; before the clean-start rule 10.9 / 16-17)
    assert res.iters.mean() <= 10.5 and res.iters.max() <= 25
    # feasibility of the reference NLP, with the stage data recomputed by the discretize entry point
    A, Bp, Bn, Sig, xi, st = Discretizer(None).discretize_batch(xbar, ubar, tf, consts)
    assert (st == 0).all()
    assert dyn_defect(A, Bp, Bn, Sig, xi, res.X, res.U, res.NU, res.tf).max() < 1e-8
    assert np.abs(res.X[:, :, 0] - xbar[:, :, 0]).max() == 0.0                               # x_0 fixed   :344-345
    assert (res.X[:, 6, -1] >= 0.1 - 1e-6).all()                                             # final mass  :351-352
    assert np.linalg.norm(res.U, axis=1).max() <= 5 + 1e-6                                   # thrust ball :379-381
    rn = np.linalg.norm(res.X[:, :3, :], axis=1)
    assert rn.max() <= 5 + 1e-6                                                              # r_max       :393-395
    assert np.abs(rn[:, -1] - r_des).max() <= 0.01 + 1e-6                                    # eps_r       :398-403
    assert (res.tf > 0).all() and (res.tf <= 5 + 1e-6).all()                                 # tf range    :588
    assert (np.abs(res.NU) <= 1e-6).all()                                                    # no virtual control needed
    h = np.cross(res.X[:, :3, -1], res.X[:, 3:6, -1])
    vt_des = np.sqrt(consts[:, 0] / r_des)
    assert np.abs(np.linalg.norm(h, axis=1) / rn[:, -1] - vt_des).max() < 1e-7               # tangential speed :492-517
    # the objective does not exceed the (feasible up to its linearisation defect) reference's: tf decreases
    assert (res.tf < 1.0).all()
    # determinism: the same launch twice is bit-identical
    res2 = mpc_step_batch(xbar, ubar, tf, consts, r_des)
    assert np.array_equal(res.X, res2.X) and np.array_equal(res.U, res2.U) and np.array_equal(res.tf, res2.tf)
    assert np.array_equal(res.iters, res2.iters)


def test_saturated_thrust_scenarios_converge():
    """References flown at 3x the benchmark's thrust: the optimal thrust profile leans on its bounds and Q_uu gets
    barrier weights many orders above its objective part.  Regression for the 3x3 inverse: with the cofactor formula
    and a determinant test a fifth of these reported breakdowns that were not and ended in MPCX_ST_NUMERIC."""
    from mpconstellation_amd import mpc_step_batch, _ffi
    from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
    from mpconstellation_amd.simulator import propagate_batch
    S, K = 256, 30
    y0, consts = normalize_batch(constellation_states(4096, first=0, count=S))
    for thrust, tf in ((1.5, 2.0), (1.5, 0.5)):
        xbar, st, _ = propagate_batch(y0, np.full(S, tf), consts, (_ffi.CTRL_TANGENTIAL, np.array([thrust]), 0, None), K)
        assert (st == 0).all()
        ubar = np.ascontiguousarray(tangential_thrust(xbar, thrust))
        r_des = np.linalg.norm(xbar[:, :3, -1], axis=1)
        res = mpc_step_batch(xbar, ubar, np.full(S, tf), consts, r_des)
        assert (res.status == 0).sum() >= S - 1 and np.sort(res.kkt)[S - 2] <= 1e-8
        assert np.linalg.norm(res.U[res.status == 0], axis=1).max() <= 5 + 1e-6


def test_satellites_are_independent_units():
    """What the multi-GPU sharding relies on: a satellite's result does not depend on which batch, which position or
    which block it is solved in (bit for bit)."""
    from mpconstellation_amd import mpc_step_batch
    S, K = 2048, 30
    xbar, ubar, consts, r_des = workload(S, K)
    tf = np.ones(S)
    whole = mpc_step_batch(xbar, ubar, tf, consts, r_des)
    perm = np.random.default_rng(7).permutation(S)
    shuf = mpc_step_batch(xbar[perm], ubar[perm], tf, consts[perm], r_des[perm])
    assert np.array_equal(whole.X[perm], shuf.X) and np.array_equal(whole.U[perm], shuf.U)
    assert np.array_equal(whole.tf[perm], shuf.tf) and np.array_equal(whole.iters[perm], shuf.iters)
    # the launch order (longest first by the previous solve's iteration counts, the default from the second call on;
    # MPCX_SOLVE_INDEX_ORDER = 1 keeps the index order) does not change a single bit either
    again = mpc_step_batch(xbar, ubar, tf, consts, r_des)
    fifo = mpc_step_batch(xbar, ubar, tf, consts, r_des, flags=1)
    for r in (again, fifo):
        assert np.array_equal(whole.X, r.X) and np.array_equal(whole.NU, r.NU) and np.array_equal(whole.iters, r.iters)
    # contiguous blocks as the ranks of an 8-GPU job would take them, generated per block
    from mpconstellation_amd.sharding import shard_block
    for rank in (0, 3, 7):
        first, count = shard_block(S, 8, rank)
        xb, ub, cb, rb = workload(S, K, first, count)
        assert np.array_equal(xb, xbar[first:first + count])
        # (the 256-satellite block runs on the two-wave kernel, the 2048 batch on the one-wave kernel: one set of bits)
        part = mpc_step_batch(xb, ub, np.ones(count), cb, rb)
        assert np.array_equal(part.X, whole.X[first:first + count]) and np.array_equal(part.tf, whole.tf[first:first + count])


def test_two_wave_small_batch_kernel():
    """Batches of up to 1024 satellites run on the two-wave kernel (solve2w.hip: a second wave per satellite shares the
    factorisation), batches of at most one satellite per compute unit (256) with K <= 30 on its LDS-resident build (solve_lds.hip:
    the satellite's working set in the compute unit's LDS).  Both must give BIT FOR BIT what the one-wave kernel gives
    (MPCX_SOLVE_ONE_WAVE = 16, MPCX_SOLVE_NO_LDS = 32; all are compiled with -ffp-contract=on, build.py: no fusion across
    statements, so the same source expressions round alike in every compilation) -- on the benchmark constellation at K = 30
    and 100, on OptimalController's option set (stiff terminal windows: refinement passes), with a thrust limit below the
    reference thrust (regularised iterations), in a ragged batch and at the shortest horizon; and like the one-wave kernel
    they must not care who shares the batch (a batch of 64, its reversal and single-satellite calls)."""
    from mpconstellation_amd import mpc_step_batch
    F = ("X", "U", "NU", "tf", "kkt", "status", "iters", "n_regularised", "first_regularised")
    for S, K, opts in ((64, 30, {}), (48, 100, {}), (1024, 30, {}), (256, 30, {}), (128, 30, {"eps_r": 1e-6, "eps_vr": 1e-16, "tf_max": 1.0}),
                       (64, 30, {"u_lim": [0, 0.3]}), (64, 24, {}), (5, 3, {})):
        xbar, ubar, consts, r_des = workload(4096, K, first=0, count=S)
        tf = np.ones(S)
        one = mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, flags=16, regularised=True)
        two = mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, flags=32, regularised=True)
        dflt = mpc_step_batch(xbar, ubar, tf, consts, r_des, options=opts, regularised=True)
        assert np.isin(two.status, (0, 7)).all()
        for f in F:
            assert np.array_equal(getattr(one, f), getattr(two, f)), (S, K, opts, f, "two waves")
            assert np.array_equal(getattr(one, f), getattr(dflt, f)), (S, K, opts, f, "default kernel")
    # a ragged batch on the LDS-resident kernel (rows of length 30, 12..30 nodes in use)
    xbar, ubar, consts, r_des = workload(4096, 30, first=0, count=40)
    Ks = (12 + (np.arange(40) * 7) % 19).astype(np.int32)
    for s in range(40): xbar[s, :, Ks[s]:] = 0.0; ubar[s, :, Ks[s]:] = 0.0
    r_des = np.array([np.linalg.norm(xbar[s, :3, Ks[s] - 1]) for s in range(40)])
    a = mpc_step_batch(xbar, ubar, np.ones(40), consts, r_des, Ks=Ks, flags=16)
    b = mpc_step_batch(xbar, ubar, np.ones(40), consts, r_des, Ks=Ks)
    assert np.isin(a.status, (0, 7)).all()
    for f in ("X", "U", "NU", "tf", "kkt", "status", "iters"): assert np.array_equal(getattr(a, f), getattr(b, f)), f
    xbar, ubar, consts, r_des = workload(4096, 30, first=0, count=64)
    tf = np.ones(64)
    two = mpc_step_batch(xbar, ubar, tf, consts, r_des)
    rev = mpc_step_batch(xbar[::-1].copy(), ubar[::-1].copy(), tf, consts[::-1].copy(), r_des[::-1].copy())
    assert np.array_equal(two.X, rev.X[::-1]) and np.array_equal(two.tf, rev.tf[::-1]) and np.array_equal(two.iters, rev.iters[::-1])
    for s in (0, 17, 63):
        single = mpc_step_batch(xbar[s:s + 1], ubar[s:s + 1], tf[:1], consts[s:s + 1], r_des[s:s + 1])
        assert np.array_equal(single.X[0], two.X[s]) and single.tf[0] == two.tf[s]


def test_minimum_horizon_and_single_satellite():
    """K = 3 is the shortest horizon the solver accepts (one interior node); one satellite per call is the reference's
    own use of Optimizer.  Same KKT point as the oracle, compared at the stated tolerance (these tiny problems run through
    factorisation breakdowns, where device and oracle may regularise differently: tests/test_solve_gpu.py)."""
    from mpconstellation_amd import mpc_step_batch
    for K, sat in ((3, 0), (3, 5), (4, 0)):
        xbar, ubar, consts, r_des = workload(64, K, sat, 1)
        res = mpc_step_batch(xbar, ubar, [1.0], consts, r_des)
        od = O.discretize(xbar[0], ubar[0], 1.0, consts[0])
        P = N.MpcProblem(xbar[0], ubar[0], 1.0, consts[0][0], od, O.constraint_terms(xbar[0], ubar[0], consts[0][0]),
                         {"r_des": float(r_des[0])})
        ref = N.solve(P)
        assert ref["status"] == 0 and res.status[0] == 0
        assert res.X.shape == (1, 7, K) and res.U.shape == (1, 3, K) and res.NU.shape == (1, 7, K)
        assert np.abs(res.X[0] - ref["X"]).max() < 5e-6 and np.abs(res.U[0] - ref["U"]).max() < 5e-6
        assert abs(res.tf[0] - ref["tf"]) < 5e-6


def test_rejected_arguments(golden_dir):
    from mpconstellation_amd import mpc_step_batch
    from mpconstellation_amd._ffi import MpcxError
    d = np.load(os.path.join(golden_dir, "disc_K2_dimensional.npz"))
    x, u, cst = d["x"], d["u"], d["const"]
    with pytest.raises(MpcxError):                       # K = 2: no interior node, the solver refuses (MPCX_E_BADARG)
        mpc_step_batch(x[None], u[None], [1.0], cst[None], [1.0])
    with pytest.raises(ValueError):                      # wrong shapes never reach the C ABI
        mpc_step_batch(np.zeros((1, 6, 5)), np.zeros((1, 3, 5)), [1.0], np.zeros((1, 8)), [1.0])
    with pytest.raises((MpcxError, ValueError)):         # empty batch
        mpc_step_batch(np.zeros((0, 7, 5)), np.zeros((0, 3, 5)), np.zeros(0), np.zeros((0, 8)), np.zeros(0))


def test_config3_full_size_scp_loop():
    """BASELINE configs[3] at full size through the bench harness's device-resident path: 4096 satellites, K = 100,
    two SCP iterations with the nonlinear re-rollout under the optimised sequence between them, re-sampled as the
    reference does at int(base_res * tf_u) nodes per satellite (control.py:166,183-227, simulator.py:38) -- the second
    iteration is one ragged launch.  Properties over all satellites after the second iteration, and 32 satellites
    against the same chain built from the CPU oracle."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import torch
    import bench
    run = bench.Runner("S4096_K100_scp2", 0, 1, 0, n_variants=1)
    run.step()
    torch.cuda.synchronize()
    status, iters, kkt = run.solver_stats()
    S, K = run.S, run.K
    assert (status == 0).all() and kkt.max() <= 1e-8 and iters.max() <= 40
    X = run.d_X.cpu().numpy(); U = run.d_U.cpu().numpy(); NU = run.d_NU.cpu().numpy(); tfo = run.d_tfo.cpu().numpy()
    Kn = run.d_Kn[0].cpu().numpy()
    assert (run.d_pst.cpu().numpy() == 0).all() and (run.d_rst.cpu().numpy() == 0).all()    # re-rollout and resampling succeeded
    assert Kn.min() >= 80 and Kn.max() <= 99 and len(np.unique(Kn)) > 3                       # a genuinely ragged batch
    h = run.host
    assert np.abs(X[:, :, 0] - h["xbar"][:, :, 0]).max() == 0.0
    assert np.linalg.norm(U, axis=1).max() <= 5 + 1e-6 and (np.abs(NU) <= 1e-6).all()
    last = Kn - 1
    col = np.arange(K)[None, :]
    assert not X[np.broadcast_to((col >= Kn[:, None])[:, None, :], X.shape)].any()            # columns past a satellite's nodes: zero
    XK = X[np.arange(S), :, last]                                                             # (S,7): each satellite's terminal node
    rK = np.linalg.norm(XK[:, :3], axis=1)
    assert np.abs(rK - h["r_des"]).max() <= 0.01 + 1e-6
    assert (tfo > 0).all() and (tfo <= 5 + 1e-6).all()
    hK = np.cross(XK[:, :3], XK[:, 3:6])
    assert np.abs(np.linalg.norm(hK, axis=1) / rK - np.sqrt(h["consts"][:, 0] / h["r_des"])).max() < 1e-7
    # the second iteration's reference is the rollout under the first one's plan: shorter flight time than the first guess
    assert (tfo < 1.0).all()
    # 32 satellites against the oracle chain: eight fixed ones and 24 drawn with a fixed seed (round 4: eight)
    sample = sorted(set((0, 311, 1024, 1777, 2500, 3333, 4000, 4095)) | set(np.random.default_rng(3).choice(S, 24, replace=False).tolist()))
    for s in sample:
        x, u, tf = h["xbar"][s], h["ubar"][s], 1.0
        cst = h["consts"][s]
        for it in range(2):
            d = O.discretize(x, u, tf, cst)
            P = N.MpcProblem(x, u, tf, cst[0], d, O.constraint_terms(x, u, cst[0]), {"r_des": float(h["r_des"][s])})
            r = N.solve(P)
            assert r["status"] == 0
            if it == 0:
                ctrl = O.make_ctrl(3, useq=np.ascontiguousarray(r["U"]), end_tau=1.0)
                kn = int(K * r["tf"])                                  # base_res = K (tf_bar = 1): control.py:227, simulator.py:38
                x = O.propagate(h["xbar"][s][:, 0], r["tf"], cst, ctrl, kn)[0]
                u = O.extract_uk(x, np.linspace(0, 1, kn), ctrl); tf = r["tf"]
        assert Kn[s] == kn
        assert np.abs(X[s][:, :kn] - r["X"]).max() < 5e-6 and abs(tfo[s] - r["tf"]) < 5e-6


def test_config4_every_rank_block():
    """BASELINE configs[4]: 65 536 satellites over 8 GPUs = 8192 per GPU.  All eight rank blocks (different plane
    rotations and speed perturbations, mpconstellation_amd/constellation.py), one after the other on this GPU through the
    bench harness: every problem converges, the properties of test_full_size_properties hold over every satellite, 32
    satellites per block (256 in all) agree with the CPU oracle (which discretises on its own), and a satellite's result is bit for bit
    what it is when its generator index is solved in another batch (no cross-satellite state: shards need no exchange)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import torch
    import bench
    from mpconstellation_amd import mpc_step_batch
    from mpconstellation_amd.sharding import shard_block
    world, K = 8, 30
    for rank in range(world):
        run = bench.Runner("S8192_K30", rank, world, 0, n_variants=1)
        run.step()
        torch.cuda.synchronize()
        status, iters, kkt = run.solver_stats()
        assert run.S == 8192 and (status == 0).all() and kkt.max() <= 1e-8 and iters.mean() <= 13 and iters.max() <= 30, rank
        first, count = shard_block(65536, world, rank)
        assert (first, count) == (rank * 8192, 8192)
        X = run.d_X.cpu().numpy(); U = run.d_U.cpu().numpy(); NU = run.d_NU.cpu().numpy(); tfo = run.d_tfo.cpu().numpy()
        h = run.host
        assert np.abs(X[:, :, 0] - h["xbar"][:, :, 0]).max() == 0.0                               # x_0 fixed   :344-345
        assert (X[:, 6, -1] >= 0.1 - 1e-6).all()                                                  # final mass  :351-352
        assert np.linalg.norm(U, axis=1).max() <= 5 + 1e-6 and (np.abs(NU) <= 1e-6).all()         # thrust ball :379-381
        rn = np.linalg.norm(X[:, :3, :], axis=1)
        assert rn.max() <= 5 + 1e-6 and np.abs(rn[:, -1] - h["r_des"]).max() <= 0.01 + 1e-6       # r_max, eps_r window
        assert (tfo > 0).all() and (tfo < 1.0).all()
        hK = np.cross(X[:, :3, -1], X[:, 3:6, -1])
        assert np.abs(np.linalg.norm(hK, axis=1) / rn[:, -1] - np.sqrt(h["consts"][:, 0] / h["r_des"])).max() < 1e-7
        pick = np.array([(977 * (rank + 1)) % 8192, 8191 - 311 * rank, (3001 * (rank + 2)) % 8192, 17 + 1000 * rank, 4096 + 53 * rank, 7000 - 777 * rank])
        more = np.random.default_rng(40 + rank).choice(8192, 26, replace=False)      # (round 5: 32 per block, 256 in all; round 4: 48)
        for s in np.concatenate([pick, more]):             # against the CPU oracle (its own discretisation)
            x, u, cst, rd = h["xbar"][s], h["ubar"][s], h["consts"][s], float(h["r_des"][s])
            P = N.MpcProblem(x, u, 1.0, cst[0], O.discretize(x, u, 1.0, cst), O.constraint_terms(x, u, cst[0]), {"r_des": rd})
            ref = N.solve(P)
            assert ref["status"] == 0
            assert np.abs(X[s] - ref["X"]).max() < 5e-6 and np.abs(U[s] - ref["U"]).max() < 5e-6 and abs(tfo[s] - ref["tf"]) < 5e-6
        res = mpc_step_batch(h["xbar"][pick], h["ubar"][pick], np.ones(len(pick)), h["consts"][pick], h["r_des"][pick])
        assert np.array_equal(res.X, X[pick]) and np.array_equal(res.tf, tfo[pick])
        if rank == 3:                                      # the block's inputs are what the generator gives for these indices
            xb, ub, cs, rd = workload(65536, K, first=first, count=count)
            assert np.array_equal(xb, h["xbar"]) and np.array_equal(rd, h["r_des"])
        del run
        torch.cuda.empty_cache()


def test_bench_two_ranks_launched_as_child_processes():
    """`python bench.py --gpus 2` from a process that is not itself a rank: bench.py starts its two ranks through
    torch.distributed.run from a parent that never touches the GPU (bench.py: spawn_ranks); on this one-GPU box both ranks
    share device 0 (MPCX_BENCH_SINGLE_DEVICE=1, gloo for the timing reduction).  The JSON line must report the whole job:
    two ranks, 2 x 4096 satellites (the N = 1 line's work per GPU) and, beside it, 2 x 8192 (configs[4]'s), every problem converged."""
    import json
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = dict(os.environ, MPCX_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    # the per-GPU work of the N = 1 line at every N (one weak-scaling curve); configs[4]'s 8192 per GPU in the same run beside it
    assert out["n_gpus"] == 2 and out["config"]["satellites_total"] == 8192 and out["config"]["workload"] == "S4096_K30"
    a4 = out["also"]["S8192_K30"]
    assert a4["satellites_total"] == 16384 and a4["converged"] == a4["of"] and a4["value"] > 0
    assert out["scaling"] == "weak" and out["solver"]["converged"] == out["solver"]["of"] == 8192
    assert out["value"] > 0 and abs(out["value"] - 8192 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]


@pytest.mark.parametrize("K,tf,r_des", [(30, 1.0, 1.5), (30, 2.0, 1.2), (60, 2.0, 1.5), (30, 1.0, 1.05)])
def test_optimal_controller_option_set_converges(K, tf, r_des):
    """The options OptimalController passes (control.py:192-197: eps_r 1e-6, eps_vr 1e-16, tf_max = horizon, fixed r_des) on
    512 satellites of the constellation: every problem ends with status 0 (round 1: up to 40 % at MAXITER, the rest only
    'acceptable' -- the structured linear solve lost the digits these stiff terminal windows need)."""
    from mpconstellation_amd import mpc_step_batch, _ffi
    from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
    from mpconstellation_amd.simulator import propagate_batch
    S = 512
    y0, consts = normalize_batch(constellation_states(4096, first=0, count=S))
    xbar, st, _ = propagate_batch(y0, np.full(S, tf), consts, (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), K)
    assert (st == 0).all()
    ubar = np.ascontiguousarray(tangential_thrust(xbar, 0.5))
    opts = {"eps_r": 0.000001, "eps_vr": 0.0000000000000001, "tf_max": tf}
    res = mpc_step_batch(xbar, ubar, np.full(S, tf), consts, np.full(S, r_des), options=opts)
    assert (res.status == 0).all() and res.kkt.max() <= 1e-8
    assert res.iters.mean() <= 22 and res.iters.max() <= 60
    rK = np.linalg.norm(res.X[:, :3, -1], axis=1)
    assert np.abs(rK - r_des).max() <= 1e-6 + 2e-8                     # the eps_r window (with ipopt's 1e-8 relaxation)
    assert (res.tf <= tf * (1 + 1e-8) + 1e-8).all()


def test_off_nominal_option_sets():
    """Option sets far from the reference's (profiles/tools/edge_cases.py): a thrust limit below the reference thrust
    (the start violates it everywhere), a target radius out of reach, a final-mass floor just under the start mass, tf_max
    below the reference flight time.  The adaptive barrier rule alone jams on some of these (round-2 finding: status
    NUMERIC); with its monotone fallback every problem converges, bar the rare one that needs more than max_iter."""
    from mpconstellation_amd import mpc_step_batch
    S, K = 256, 30
    xbar, ubar, consts, r_des = workload(4096, K, first=0, count=S)
    tf = np.ones(S)
    # u_max 0.05 and r_des 3 leave 1-4 of 256 at MAXITER: the curvature of their reduced problem along tf is negative
    # next to the optimum (the border's last pivot), every iteration of the endgame is regularised (delta_w ~ 1) and the
    # regularised Newton step makes only linear progress (known limits).  The thrust-limited sets are also
    # sensitive to rounding: scaling u_bar by 1 + j 2^-50 moves the iteration counts of ~12 of the 256 problems by up to
    # 60 and decides whether one of them jams (profiles/r03/umax_chaos.txt: about one problem in 2000, the same rate for
    # this build and the round-2 build) -- hence one failure is allowed where none is expected
    for opts, rd, min_ok in (({"u_lim": [0, 0.3]}, r_des, S - 1), ({"u_lim": [0, 0.05]}, r_des, S - 6), ({"min_mass": 0.999}, r_des, S),
                             ({}, np.full(S, 3.0), S - 3), ({"tf_max": 0.5}, r_des, S), ({"r_lim": [1.0, 5]}, r_des, S)):
        res = mpc_step_batch(xbar, ubar, tf, consts, rd, options=opts)
        assert (res.status == 6).sum() <= (1 if "u_lim" in opts else 0), opts      # no numeric breakdown
        assert np.isin(res.status, (0, 7)).sum() >= min_ok, (opts, np.unique(res.status, return_counts=True))
        ok = res.status == 0
        assert res.kkt[ok].max() <= 1e-8
        if "u_lim" in opts:
            # the thrust ball is |u|^2 <= u_max^2, relaxed by 1e-8 like every bound (ipopt's bound_relax_factor) and met to
            # the solver tolerance 1e-8
            assert (np.linalg.norm(res.U[ok], axis=1) ** 2).max() <= opts["u_lim"][1] ** 2 + 2.5e-8
    # an empty constraint set is reported before the first iteration (MPCX_ST_INFEASIBLE), not iterated on for max_iter:
    # the start node below the r_min plane (x_0 is fixed), a terminal window outside r_max, r_min > r_max, an empty
    # velocity window, an empty tf range -- in one launch with feasible neighbours, which are not disturbed
    ref = mpc_step_batch(xbar, ubar, tf, consts, r_des)
    for opts, rd in (({"r_lim": [1.01, 5]}, r_des), ({}, np.full(S, 5.5)), ({"r_lim": [0.99, 1.2]}, np.full(S, 1.5)),
                     ({"r_lim": [2.0, 1.5]}, r_des), ({"eps_vr": -1e-3}, r_des), ({"tf_max": -1.0}, r_des)):
        res = mpc_step_batch(xbar, ubar, tf, consts, rd, options=opts)
        assert (res.status == 8).all() and (res.iters == 0).all() and (res.kkt > 0).all(), opts
        assert np.array_equal(res.X, xbar) and np.array_equal(res.U, ubar) and not res.NU.any() and np.array_equal(res.tf, tf)
    rd = r_des.copy(); rd[5] = 7.0                         # one unreachable target among feasible satellites
    res = mpc_step_batch(xbar, ubar, tf, consts, rd)
    assert res.status[5] == 8 and (np.delete(res.status, 5) == 0).all()
    assert np.array_equal(np.delete(res.X, 5, axis=0), np.delete(ref.X, 5, axis=0))
    # a NaN in one satellite's reference is that satellite's problem only
    xn = xbar.copy(); xn[3, 2, 7] = np.nan
    res = mpc_step_batch(xn, ubar, tf, consts, r_des)
    assert res.status[3] != 0 and (np.delete(res.status, 3) == 0).all()
    assert np.array_equal(np.delete(res.X, 3, axis=0), np.delete(ref.X, 3, axis=0))


def test_host_pointer_calls_have_no_stragglers():
    """BENCH_r03 held one 53 ms call among 1.5 ms ones; round 4 saw 67 ms among 1.4 ms at 64 satellites and 11-13 ms among 7.4 ms
    at 4096 and explained them, in a docstring behind a retry, as time "before the stream executes the call's first packet".
    Round 5 made the call trace data (include/mpcx.h: mpcx_trace_enable / mpcx_last_call_trace) and measured 10 000 + 2 500 traced
    calls on two boxes (profiles/r05/host_trace_stats.txt): the first-packet wait never exceeded 0.06 ms -- that explanation does
    not describe what is seen now.  What is seen: 3 of 500 and 0 of 2000 calls at 4096 satellites beyond 1.5 x the median, 1 of
    10 000 at 64; every extra millisecond of them inside ONE host step of the call -- a hipMemcpyAsync enqueue that blocks for
    3-12 ms inside the runtime (up:enq / down:enq), or a staging memcpy that takes 3-6 ms instead of 0.3 (fin:copy) -- on boxes
    with a load average of 14-24; the device's kernels of those calls took their usual time.
    ONE round of 50 consecutive host-pointer calls (numpy in, numpy out) at 64 and at 4096 satellites, trace armed, no retry:
      (i)   the call's kernels as the device ran them (events on the call's stream: last upload done -> kernels done) are within
            1.25 x their median for EVERY call: the device work of a call does not straggle;
      (ii)  the fields of the record account for the call's wall time (nothing hides between them), the first-packet wait
            stays below 1 ms, and the host work behind it (staging, enqueueing, waits, copy-out) is uniform: 90 % of the calls
            within 1.25 x its median;
      (iii) at most 2 of the 50 calls may exceed 1.5 x that median (measured rate per call: 0.1-0.6 %, i.e. P(3 or more) < 0.4 %),
            and each one is printed with its breakdown.  A stall of the library's own making -- in every call, or in the
            kernels -- fails (i) or (ii)."""
    from mpconstellation_amd import mpc_step_batch, _ffi
    lines = []
    for S in (64, 4096):
        xbar, ubar, consts, r_des = workload(4096, 30, first=0, count=S)
        tf = np.ones(S)
        for _ in range(3): r = mpc_step_batch(xbar, ubar, tf, consts, r_des)
        _ffi.trace_enable(True)
        try:
            recs = []
            for _ in range(50):
                r = mpc_step_batch(xbar, ubar, tf, consts, r_des)
                recs.append(_ffi.last_call_trace())
        finally:
            _ffi.trace_enable(False)
        assert (r.status == 0).all() and all(t is not None for t in recs)
        f = {k: np.array([t[k] for t in recs]) for k in _ffi.TRACE_FIELDS}
        wall, first, kern = f["wall_ms"], f["first_marker_ms"], f["dev_kernels_ms"]
        after = f["host_stage_ms"] + f["host_wait_ms"] + f["host_copyout_ms"]
        lines.append(f"S {S}: wall median {np.median(wall):.3f} max {wall.max():.3f} | kernels on the device median {np.median(kern):.3f} max {kern.max():.3f} | "
                     f"host after first packet median {np.median(after):.3f} p90 {np.percentile(after, 90):.3f} max {after.max():.3f} | first marker max {first.max():.3f}")
        slow = np.nonzero(after > 1.5 * np.median(after))[0]
        for i in slow:
            lines.append(f"   slow call {i}: " + ", ".join(f"{k} {recs[i][k]:.3f}" for k in _ffi.TRACE_FIELDS[:-1]))
        print("\n".join(lines))
        assert (kern <= 1.25 * np.median(kern)).all(), lines                                 # (i)
        assert np.abs(first + after - wall).max() < 0.02 * np.median(wall) + 0.01, lines     # (ii)
        assert first.max() < 1.0 and np.percentile(after, 90) <= 1.25 * np.median(after), lines
        assert len(slow) <= 2, lines                                                         # (iii)
    # an untraced call leaves no record behind a disabled trace, and tracing does not change results
    r0 = mpc_step_batch(xbar, ubar, tf, consts, r_des)
    assert np.array_equal(r0.X, r.X) and _ffi.last_call_trace() is None
