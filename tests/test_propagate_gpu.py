"""Device rollout (mpcx_propagate_batch) against the reference's golden rollouts and the oracle."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_propagate_vs_reference_golden(golden_dir):
    from mpconstellation_amd import _ffi
    from mpconstellation_amd.simulator import propagate_batch
    p = np.load(os.path.join(golden_dir, "propagate.npz"))
    cst, y0 = p["const"], p["y0"]
    tan = (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None)
    y, st, ns = propagate_batch(y0[None], [1.0], cst[None], tan, 30)
    assert st[0] == 0 and ns[0] == 1000 and np.abs(y[0] - p["x_tan_plain"]).max() < 1e-11
    y, st, _ = propagate_batch(y0[None], [1.0], cst[None], tan, 30, include_drag=True, include_J2=True)
    assert st[0] == 0 and np.abs(y[0] - p["x_tan_dragj2"]).max() < 1e-11
    y, st, _ = propagate_batch(y0[None], [0.8], cst[None], (_ffi.CTRL_SEQUENCE, p["useq"], 12, 1.0), 32)
    assert st[0] == 0 and np.abs(y[0] - p["x_seq_full"]).max() < 1e-11
    y, st, _ = propagate_batch(y0[None], [1.0], cst[None], (_ffi.CTRL_SEQUENCE, p["useq"], 12, 0.6), 40)
    assert st[0] == 0 and np.abs(y[0] - p["x_seq_tail"]).max() < 1e-11


def test_propagate_batch_vs_oracle_and_status(golden_dir):
    from mpconstellation_amd import _ffi
    from mpconstellation_amd.constellation import constellation_states, normalize_batch
    from mpconstellation_amd.simulator import propagate_batch
    S = 70            # more than one wave, ragged tail
    y0, consts = normalize_batch(constellation_states(S))
    tf = np.linspace(0.5, 1.5, S)
    thr = np.random.default_rng(5).normal(size=(S, 3)) * 0.3
    y, st, ns = propagate_batch(y0, tf, consts, (_ffi.CTRL_CONSTANT, thr, 0, None), 17, include_J2=True)
    assert (st == 0).all()
    for s in (0, 33, 69):
        c = O.make_ctrl(O.CTRL_CONSTANT, thr[s])
        yo, rc, n = O.propagate(y0[s], tf[s], consts[s], c, 17, O.FLAG_J2)
        assert rc == 0 and n == ns[s] and np.abs(y[s] - yo).max() < 1e-11
    # zero thrust law and the mass guard (simulator.py:135-136)
    ybad = y0.copy(); ybad[3, 6] = -0.5
    y, st, _ = propagate_batch(ybad, 1.0, consts, (_ffi.CTRL_ZERO, None, 0, None), 5)
    assert st[3] == 1 and (np.delete(st, 3) == 0).all()
    c64 = np.load(os.path.join(golden_dir, "constellation64.npz"))
    yt, st, _ = propagate_batch(y0[:64], 1.0, consts[:64], (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), 30)
    y064, _ = normalize_batch(constellation_states(64))
    yt, st, _ = propagate_batch(y064, 1.0, _ := normalize_batch(constellation_states(64))[1],
                                (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), 30)
    for i in c64["idx"]:
        assert np.abs(yt[i] - c64[f"x_{i}"]).max() < 1e-11          # the reference's own rollouts


def test_rollout_returns_the_thrust_at_its_output_points():
    """mpcx_propagate_thrust_batch_ragged: u_out = Discretizer.extract_uk of the rollout's own controller
    (linearize_discretize.py:393-411) from the same launch -- against the host forms: tangential_thrust on the returned
    trajectory (control.py:66-84), the first-order hold of SequenceController at linspace(0, 1, n) (control.py:104-142:
    the same node index and weights, to one rounding), the constant vector, zeros; ragged batches keep zeros past a satellite's count; the
    trajectories themselves are those of the plain entry point, bit for bit."""
    from mpconstellation_amd import _ffi
    from mpconstellation_amd.simulator import propagate_batch
    from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
    from foh_reference import foh_resample_ragged
    S = 37
    y0, consts = normalize_batch(constellation_states(4096, first=100, count=S))
    tf = np.linspace(0.6, 1.4, S)
    y, st, ns = propagate_batch(y0, tf, consts, (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), 40)
    y2, st2, ns2, u = propagate_batch(y0, tf, consts, (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), 40, thrust=True)
    assert (st == 0).all() and np.array_equal(y, y2) and np.array_equal(ns, ns2)
    assert np.abs(u - tangential_thrust(y, 0.5)).max() < 1e-14
    # sequence playback over its own horizon, ragged: per-satellite table lengths and sample counts
    rng = np.random.default_rng(5)
    Ku = rng.integers(5, 31, S); n = rng.integers(4, 41, S); Kmax = int(Ku.max())
    table = rng.normal(size=(S, 3, Kmax)) * 0.3
    for s in range(S): table[s, :, Ku[s]:] = 0.0
    y3, st3, _, u3 = propagate_batch(y0, tf, consts, (_ffi.CTRL_SEQUENCE, table, Kmax, 1.0), n, Kus=Ku, thrust=True)
    assert (st3 == 0).all()
    ref3 = foh_resample_ragged(table, Ku, n)
    assert np.abs(u3 - ref3).max() <= 1e-15 * max(1.0, np.abs(ref3).max())      # (the device contracts the blend into an fma: one rounding)
    for s in (0, 11, S - 1): assert (u3[s, :, n[s]:] == 0.0).all() and (y3[s, :, n[s]:] == 0.0).all()
    # a table that ends before the rollout does: zero thrust afterwards (control.py:139-141)
    _, st4, _, u4 = propagate_batch(y0[:3], 1.0, consts[:3], (_ffi.CTRL_SEQUENCE, table[:3, :, :5], 5, 0.5), 21, thrust=True)
    assert (st4 == 0).all() and (u4[:, :, 11:] == 0.0).all() and np.abs(u4[:, :, :10]).max() > 0
    vec = np.array([0.1, -0.2, 0.05])
    _, _, _, u5 = propagate_batch(y0[:3], 1.0, consts[:3], (_ffi.CTRL_CONSTANT, vec, 0, None), 9, thrust=True)
    assert np.array_equal(u5, np.broadcast_to(vec[None, :, None], (3, 3, 9)))
    _, _, _, u6 = propagate_batch(y0[:3], 1.0, consts[:3], (_ffi.CTRL_ZERO, None, 0, None), 9, thrust=True)
    assert (u6 == 0.0).all()
