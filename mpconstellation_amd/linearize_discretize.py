"""Discretizer: drop-in for the reference class of the same name
(reference linearize_discretize.py:85-411) whose discretize() runs on the MI355X through
libmpcx.so.  There is no host fallback: without the HIP library / a gfx950 device it raises."""
import ctypes as C

import numpy as np

from . import _ffi


def _is_native_dynamics(f):
    return getattr(f, "__name__", "") == "satellite_dynamics"


class Discretizer:
    def __init__(self, const, rho_func=None, drho_func=None, include_drag=False, include_J2=False,
                 use_scipy_ZOH=False, device=0):
        self.const = const
        self.include_drag = include_drag
        self.include_J2 = include_J2
        # use_scipy_ZOH (linearize_discretize.py:327-329): the reference then evaluates the hold with scipy's interp1d(kind='linear')
        # -- the same piecewise-linear function as u_FOH, its own two evaluations <= 1e-16 relative apart on the goldens.  The device
        # evaluates the FOH formula in both modes and is pinned against arrays the reference produced WITH the flag
        # (tests/golden/scipy_zoh_discretize.npz, tests/test_discretize_gpu.py::test_scipy_zoh_mode_vs_reference: 1e-10 relative)
        self.use_scipy_ZOH = use_scipy_ZOH
        self.rho_func = rho_func
        self.drho_func = drho_func
        # ODE / quadrature settings, same names and defaults as the reference (:104-109)
        self.ivp_max_step = 1e-2
        self.ivp_solver = 'RK45'
        self.integrator_steps = 101
        self.use_uniform_steps = False
        self.device = device

    # ---- batched entry point (S satellites at once) -------------------------------------
    def discretize_batch(self, x, u, tf, consts):
        """x (S,7,K), u (S,3,Ku), tf (S,), consts (S,8) -> A (S,K-1,7,7), B_kp, B_kn (S,K-1,7,3),
        Sigma, xi (S,7,K-1), status (S,)"""
        self._check_modes()
        x = _ffi.as_f64(x); u = _ffi.as_f64(u)
        S, _, K = x.shape
        Ku = u.shape[2]
        tf = _ffi.as_f64(np.broadcast_to(np.asarray(tf, dtype=np.float64), (S,)))
        consts = _ffi.as_f64(consts)
        if x.shape[1] != 7 or u.shape[:2] != (S, 3) or consts.shape != (S, _ffi.NCONST):
            raise ValueError("expected x (S,7,K), u (S,3,Ku), consts (S,8)")
        A = np.empty((S, K - 1, 7, 7)); Bp = np.empty((S, K - 1, 7, 3)); Bn = np.empty((S, K - 1, 7, 3))
        Sig = np.empty((S, 7, K - 1)); xi = np.empty((S, 7, K - 1))
        status = np.zeros(S, dtype=np.int32)
        lib = _ffi.load()
        ctx = _ffi.context(self.device)
        flags = _ffi.FLAG_J2 if self.include_J2 else 0
        flags |= self.device_flags()
        rc = lib.mpcx_discretize_batch(ctx, S, K, Ku, _ffi.dptr(x), _ffi.dptr(u), _ffi.dptr(tf),
                                       _ffi.dptr(consts), flags, float(self.ivp_max_step),
                                       _ffi.dptr(A), _ffi.dptr(Bp), _ffi.dptr(Bn), _ffi.dptr(Sig),
                                       _ffi.dptr(xi), _ffi.iptr(status))
        _ffi.check(rc, ctx, "mpcx_discretize_batch")
        return A, Bp, Bn, Sig, xi, status

    # ---- reference signature ---------------------------------------------------------------
    def discretize(self, f, x, u, tf):
        """Same contract as the reference (:334-390): returns A_k (K-1,7,7), B_kp (K-1,7,3),
        B_kn (K-1,7,3), Sigma_k (7,K-1), xi_k (7,K-1) for one satellite."""
        if not _is_native_dynamics(f):
            raise NotImplementedError("only Simulator.satellite_dynamics is implemented on the device")
        x = np.asarray(x, dtype=np.float64); u = np.asarray(u, dtype=np.float64)
        A, Bp, Bn, Sig, xi, status = self.discretize_batch(x[None], u[None], [tf],
                                                           self.const.as_vector()[None])
        if status[0] == 1:
            raise Exception("ERROR: INVALID SATELLITE MASS")          # simulator.py:135-136
        if status[0] == 3:
            raise IndexError("FOH index outside the input table")      # u[:, k+1] in the reference
        if status[0] == 4:
            raise np.linalg.LinAlgError("Singular matrix")             # np.linalg.inv in the reference
        if status[0] != 0:
            raise RuntimeError(_ffi.STATUS_TEXT.get(int(status[0]), "discretize failed"))
        return A[0], Bp[0], Bn[0], Sig[0], xi[0]

    def device_flags(self):
        """the integration settings as flags of the discretize / fused-step entry points (include/mpcx.h): use_uniform_steps
        with integrator_steps (linearize_discretize.py:27-30: t_eval = linspace(.., integrator_steps)), ivp_solver (:40)"""
        flags = 0
        if self.use_uniform_steps:
            flags |= _ffi.FLAG_UNIFORM_STEPS | (int(self.integrator_steps) << 8)
        if self.ivp_solver == 'RK23':
            flags |= _ffi.FLAG_RK23
        return flags

    def _check_modes(self):
        if self.include_drag:
            # the reference's drag branch cannot run either (Constants has no CD, rho_func is None)
            raise NotImplementedError("drag in the linearisation is not supported")
        if self.ivp_solver not in ('RK45', 'RK23'):
            raise NotImplementedError("ivp_solver: 'RK45' (the reference's default) and 'RK23' are implemented on the device; "
                                      "scipy's DOP853 and its implicit methods (Radau, BDF, LSODA) are not")
        if self.use_uniform_steps and int(self.integrator_steps) < 2:
            raise ValueError("use_uniform_steps needs integrator_steps >= 2")

    @staticmethod
    def extract_uk(x_k, tau_k, controller):
        """Reference linearize_discretize.py:393-411."""
        u_func = controller.get_u_func()
        return np.column_stack([u_func(x_k[:, i], tau_k[i]) for i in range(x_k.shape[1])])
