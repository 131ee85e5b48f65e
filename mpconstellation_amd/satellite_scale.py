"""Designer-unit scaling (mirrors reference satellite_scale.py:4-100)."""
import numpy as np

from . import constants as _k
from .constants import Constants


class SatelliteScale:
    def __init__(self, x=None, sat=None):
        if sat is not None:
            x = sat.get_state_vector()
        elif x is None:
            x = np.array([1, 0, 0, 0, 0, 0, 1])
        self._r0 = np.linalg.norm(x[0:3])
        self._s0 = 2 * np.pi * np.sqrt(self._r0 ** 3 / _k.MU_EARTH)
        self._v0 = self._r0 / self._s0
        self._a0 = self._r0 / self._s0 ** 2
        self._m0 = x[6]
        self._T0 = self._m0 * self._r0 / self._s0 ** 2
        self._mu0 = self._r0 ** 3 / self._s0 ** 2

    def get_normalized_constants(self):
        return Constants(MU=_k.MU_EARTH / self._mu0, R_E=_k.R_EARTH / self._r0, J2=_k.J2,
                         G0=_k.G0 / self._a0, ISP=_k.ISP / self._s0, S=_k.S / self._r0 ** 2,
                         R0=self._r0, RHO=self._m0 / self._r0 ** 3)

    def _apply(self, x, fr, fv, fm):
        if x.ndim == 1:
            return np.concatenate([x[0:3] * fr, x[3:6] * fv, [x[6] * fm]])
        assert x.shape[0] == 7, "If x is 2D, must be shaped as 7 x N"
        return np.vstack([x[0:3, :] * fr, x[3:6, :] * fv, x[6, :] * fm])

    def redim_state(self, x):
        return self._apply(x, self._r0, self._v0, self._m0)

    def normalize_state(self, x):
        if x.ndim == 1:
            return np.concatenate([x[0:3] / self._r0, x[3:6] / self._v0, np.array([x[6] / self._m0])])
        assert x.shape[0] == 7, "If x is 2D, must be shaped as 7 x N"
        return np.vstack([x[0:3, :] / self._r0, x[3:6, :] / self._v0, x[6, :] / self._m0])

    def redim_thrust(self, u):
        return u * self._T0

    def normalize_thrust(self, u):
        return u / self._T0
