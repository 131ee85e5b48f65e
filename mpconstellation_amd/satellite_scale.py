"""Designer units of one satellite: the drop-in for the reference's SatelliteScale (satellite_scale.py:4-100).

The reference derives seven units from the start state -- length = |r|, time = the circular-orbit period at that radius,
mass = the start mass, and speed, acceleration, force and gravitational parameter from those -- and keeps them in the
private attributes its own tests read (`scale._r0`, `scale._v0`, test_simulator.py:138).  Here the units live in one table
computed by `derived_units` (same expressions, so the values are bit for bit the reference's: tests/golden/
constants_hubble.npz), a state is scaled by ONE per-component unit vector, and the reference's attribute names are
read-only views of the table."""
import numpy as np

from . import constants as _k
from .constants import Constants

_STATE_SHAPE_MSG = "If x is 2D, must be shaped as 7 x N"


def derived_units(length, mass):
    """The unit table of a satellite that starts at radius `length` with mass `mass` (satellite_scale.py:22-35)."""
    time = 2 * np.pi * np.sqrt(length ** 3 / _k.MU_EARTH)
    return {"length": length, "time": time, "speed": length / time, "accel": length / time ** 2, "mass": mass,
            "force": mass * length / time ** 2, "mu": length ** 3 / time ** 2}


class SatelliteScale:
    def __init__(self, x=None, sat=None):
        state = sat.get_state_vector() if sat is not None else (np.array([1, 0, 0, 0, 0, 0, 1]) if x is None else x)
        self.units = derived_units(np.linalg.norm(state[0:3]), state[6])
        q = self.units
        self._per_component = np.array([q["length"]] * 3 + [q["speed"]] * 3 + [q["mass"]], dtype=np.float64)

    # the reference's attribute names
    _r0 = property(lambda self: self.units["length"])
    _s0 = property(lambda self: self.units["time"])
    _v0 = property(lambda self: self.units["speed"])
    _a0 = property(lambda self: self.units["accel"])
    _m0 = property(lambda self: self.units["mass"])
    _T0 = property(lambda self: self.units["force"])
    _mu0 = property(lambda self: self.units["mu"])

    def get_normalized_constants(self):
        """Constants in designer units (satellite_scale.py:37-52); R0 and RHO are the two the drag term needs."""
        q = self.units
        return Constants(MU=_k.MU_EARTH / q["mu"], R_E=_k.R_EARTH / q["length"], J2=_k.J2, G0=_k.G0 / q["accel"],
                         ISP=_k.ISP / q["time"], S=_k.S / q["length"] ** 2, R0=q["length"], RHO=q["mass"] / q["length"] ** 3)

    def _units_like(self, x):
        """the per-component unit vector shaped for a state (7,) or a trajectory (7, N)"""
        if x.ndim == 1:
            return self._per_component
        assert x.shape[0] == 7, _STATE_SHAPE_MSG
        return self._per_component[:, None]

    def redim_state(self, x):
        return x * self._units_like(x)

    def normalize_state(self, x):
        return x / self._units_like(x)

    def redim_thrust(self, u):
        return u * self.units["force"]

    def normalize_thrust(self, u):
        return u / self.units["force"]
