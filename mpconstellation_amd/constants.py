"""Physical constants and the normalised-constants record (mirrors reference constants.py:1-20)."""
MU_EARTH = 3.986004418E14   # m^3 / s^2
R_EARTH = 6.371E6           # m, mean radius
J2 = 1.08262668E-3
G0 = 9.80665                # m / s^2
ISP = 500                   # s
C_D = 2.5
S = 55.44                   # m^2

CONST_FIELDS = ("MU", "R_E", "J2", "G0", "ISP", "S", "R0", "RHO")


class Constants:
    """Same attribute names as the reference's Constants (constants.py:11-20)."""

    def __init__(self, MU, R_E, J2, G0, ISP, S, R0, RHO):
        self.MU, self.R_E, self.J2, self.G0 = MU, R_E, J2, G0
        self.ISP, self.S, self.R0, self.RHO = ISP, S, R0, RHO

    def as_vector(self):
        """Packed in the order libmpcx expects (include/mpcx.h MPCX_C_*)."""
        import numpy as np
        return np.array([getattr(self, f) for f in CONST_FIELDS], dtype=np.float64)
