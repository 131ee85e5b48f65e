"""Synthetic constellation used by bench.py and the tests:
satellite i = the Hubble state of the reference's tests (test_optimizer.py:18-22) with
v <- v (1 + 0.1 U_i), U_i ~ U[0,1) from numpy default_rng(seed) (pattern of test_simulator.py:47),
rotated about z by 2 pi i / S and about x by pi ((0.61803 i) mod 1) / 3; one SatelliteScale per satellite."""
import numpy as np

from . import constants as _k

R_HUBBLE = np.array([5371.4806, -4133.1393, 1399.9594]) * 1000.0
V_HUBBLE = np.array([4.6921, 4.9848, -3.2752]) * 1000.0
M_HUBBLE = 12200.0
SEED = 20260101


def _rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def _rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def constellation_states(S, seed=SEED, first=0, count=None):
    """Dimensional states (count,7) of satellites first..first+count-1 of an S-satellite constellation."""
    count = S - first if count is None else count
    U = np.random.default_rng(seed).random(S)
    out = np.zeros((count, 7))
    for n in range(count):
        i = first + n
        R = _rot_x(np.pi * ((i * 0.61803) % 1.0) / 3.0) @ _rot_z(2.0 * np.pi * i / S)
        out[n, 0:3] = R @ R_HUBBLE
        out[n, 3:6] = R @ (V_HUBBLE * (1.0 + 0.1 * U[i]))
        out[n, 6] = M_HUBBLE
    return out


def normalize_batch(states):
    """Per-satellite designer units (satellite_scale.py:28-44): returns y0 (S,7) and consts (S,8)."""
    r0 = np.linalg.norm(states[:, 0:3], axis=1)
    s0 = 2 * np.pi * np.sqrt(r0 ** 3 / _k.MU_EARTH)
    v0 = r0 / s0; a0 = r0 / s0 ** 2; m0 = states[:, 6]; mu0 = r0 ** 3 / s0 ** 2
    y0 = np.column_stack([states[:, 0:3] / r0[:, None], states[:, 3:6] / v0[:, None], states[:, 6] / m0])
    consts = np.column_stack([_k.MU_EARTH / mu0, _k.R_EARTH / r0, np.full_like(r0, _k.J2), _k.G0 / a0, _k.ISP / s0,
                              _k.S / r0 ** 2, r0, m0 / r0 ** 3])
    return y0, consts


def tangential_thrust(x, mag):
    """u_k = mag * t_hat(x_k) for x (S,7,K): extract_uk of ConstantTangentialThrustController (control.py:66-84),
    t_hat = h_hat x r_hat with h = r x v, component by component on (S,K) arrays (np.cross / np.linalg.norm over the
    middle axis of a (S,3,K) array are several times slower)."""
    r0, r1, r2 = x[:, 0, :], x[:, 1, :], x[:, 2, :]
    v0, v1, v2 = x[:, 3, :], x[:, 4, :], x[:, 5, :]
    rn = np.sqrt(r0 * r0 + r1 * r1 + r2 * r2)
    a0, a1, a2 = r0 / rn, r1 / rn, r2 / rn                                   # r_hat
    h0 = r1 * v2 - r2 * v1; h1 = r2 * v0 - r0 * v2; h2 = r0 * v1 - r1 * v0   # h = r x v
    hn = np.sqrt(h0 * h0 + h1 * h1 + h2 * h2)
    b0, b1, b2 = h0 / hn, h1 / hn, h2 / hn                                   # h_hat
    out = np.empty((x.shape[0], 3, x.shape[2]))
    out[:, 0, :] = mag * (b1 * a2 - b2 * a1)                                 # h_hat x r_hat
    out[:, 1, :] = mag * (b2 * a0 - b0 * a2)
    out[:, 2, :] = mag * (b0 * a1 - b1 * a0)
    return out
