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
