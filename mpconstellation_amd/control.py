"""Controllers: drop-in for reference control.py.  Every thrust law the reference ships can be
evaluated on the host (get_u_func, same signature u(x, tau)) and has a device form (device_law)
that the batched propagator consumes."""
import numpy as np

from . import _ffi


def _check_solver_status(status, strict):
    """the plan of a solve that ended neither OK (0) nor at the acceptable level (7) is not silently flown"""
    import warnings
    bad = [int(c) for c in np.atleast_1d(status) if c not in (0, 7)]
    if bad:
        msg = f"MPC solve did not converge: status {[_ffi.STATUS_TEXT.get(c, c) for c in bad]}"
        if strict:
            raise RuntimeError(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


class Controller:
    """Zero thrust (reference control.py:8-35)."""

    def __init__(self, sats=[]):
        self.sats = sats
        self.sat_ids = set(s.id for s in sats)

    def _own(self, f):
        """marks a thrust function as this controller's, so that Simulator.get_trajectory_ODE can tell it from a foreign
        callable (which the device cannot integrate)"""
        f._mpcx_controller = self
        return f

    def get_u_func(self, sat_id=None):
        zero = np.array([0., 0., 0.])
        return self._own(lambda x, tau: zero)

    def update(self):
        pass

    def device_law(self):
        return _ffi.CTRL_ZERO, None, 0, None


class ConstantThrustController(Controller):
    """reference control.py:37-53"""

    def __init__(self, sats=[], thrust=np.array([1., 1., 1.])):
        super().__init__(sats)
        self.thrust = thrust

    def get_u_func(self, sat_id=None):
        return self._own(lambda x, tau: self.thrust)

    def device_law(self):
        return _ffi.CTRL_CONSTANT, np.asarray(self.thrust, dtype=np.float64), 0, None


class ConstantTangentialThrustController(Controller):
    """reference control.py:55-84"""

    def __init__(self, sats=[], tangential_thrust=1):
        super().__init__(sats)
        self.tangential_thrust = tangential_thrust

    def compute_rotation(self, x):
        r = x[0:3]; v = x[3:6]
        r_hat = r / np.linalg.norm(r)
        h = np.cross(r, v); h_hat = h / np.linalg.norm(h)
        return np.column_stack([r_hat, np.cross(h_hat, r_hat), h_hat])

    def get_u_func(self, sat_id=None):
        return self._own(lambda x, tau: self.compute_rotation(x) @ np.array([0, self.tangential_thrust, 0]))

    def device_law(self):
        return _ffi.CTRL_TANGENTIAL, np.array([float(self.tangential_thrust)]), 0, None


class SequenceController(Controller):
    """First-order-hold playback of a (3,K) table over tau in [0, tf_u/tf_sim], zero afterwards
    (reference control.py:86-143)."""

    def __init__(self, sats=[], u=np.array([]), tf_u=1, tf_sim=1):
        super().__init__(sats)
        self.end_tau = tf_u / tf_sim
        self.u = u

    def u_FOH(self, tau):
        if tau == 1:
            return self.u[:, -1]
        K = self.u.shape[1]
        dtau = 1 / (K - 1)
        k = int(tau // dtau)
        tau_k = k / (K - 1); tau_kp1 = (k + 1) / (K - 1)
        lam_n = (tau_kp1 - tau) / (tau_kp1 - tau_k); lam_p = (tau - tau_k) / (tau_kp1 - tau_k)
        return lam_n * self.u[:, k] + lam_p * self.u[:, k + 1]

    def get_u_func(self, sat_id=None):
        def u(x, tau):
            if tau <= self.end_tau:
                return self.u_FOH(tau / self.end_tau)
            return np.array([0., 0., 0.])
        return self._own(u)

    def device_law(self):
        return _ffi.CTRL_SEQUENCE, np.ascontiguousarray(self.u, dtype=np.float64), self.u.shape[1], float(self.end_tau)


class OptimalController(Controller):
    """MPC / SCP controller (reference control.py:145-246): reference rollout -> SCPn x (discretize + solve
    + nonlinear re-rollout) with the discretize+solve step on the device."""

    def __init__(self, sats=[], objective=None, base_res=100, tf_horizon=1, tf_interval=1, plot_inter=True,
                 opt_verbose=True, r_des=1.5, strict=False):
        super().__init__(sats)
        from .satellite_scale import SatelliteScale
        self.u = np.zeros((3, 1))
        self.horizon = tf_horizon
        self.interval = tf_interval
        self.base_res = base_res
        self.sat = self.sats[0]
        self.scale = SatelliteScale(sat=self.sat)
        self.r_des = r_des
        self.SCPn_iterations = 2
        self.plot_intermediate = plot_inter
        self.opt_verbose = opt_verbose
        self.last_status = []
        # The reference never looks at ipopt's return status (optimizer.py:603).  Here a solve that ends neither OK nor at
        # the acceptable level always warns (RuntimeWarning, whatever opt_verbose says) and, with strict=True, raises
        # instead of flying an unconverged plan.
        self.strict = strict

    def update(self):
        from . import simulator
        from .linearize_discretize import Discretizer
        from .optimizer import Optimizer
        const = self.scale.get_normalized_constants()
        c = ConstantTangentialThrustController([self.sat], 0.5)       # control.py:178-180
        x, t = self.run_nonlinear(c, self.horizon)
        tf_u = self.horizon
        self.last_status = []
        for i in range(self.SCPn_iterations):
            K = x.shape[1]
            d = Discretizer(const, use_scipy_ZOH=False, include_drag=False, include_J2=False)
            u_bar = Discretizer.extract_uk(x, t, c)
            nu_bar = np.zeros((7, K))
            f = simulator.Simulator.satellite_dynamics
            opt_options = {'r_des': self.r_des, 'eps_r': 0.000001, 'eps_vr': 0.0000000000000001, 'eps_vt': 0.01,
                           'tf_max': self.horizon}                    # control.py:192-197
            opt = Optimizer([x], [u_bar], [nu_bar], tf_u, d, f, self.scale, verbose=self.opt_verbose)
            opt.solve_OPT(input_options=opt_options)
            self.last_status.append(int(opt.status[0]))
            _check_solver_status(opt.status, self.strict)
            tf_u = opt.get_solved_tf(0)
            u_opt = opt.get_solved_u(0)
            nu_opt = opt.get_solved_nu(0)
            if self.opt_verbose:
                print(f"tf for optimizer: {tf_u}")
                print(f"Total virtual control effort: {np.abs(nu_opt).sum()}")
            self.opt_trajectory = opt.get_solved_trajectory(0)
            self.sequence_controller = SequenceController(u=u_opt, tf_u=tf_u, tf_sim=self.interval)
            c = SequenceController(u=u_opt, tf_u=tf_u, tf_sim=tf_u)
            x, t = self.run_nonlinear(c=c, tf=tf_u)
        if self.horizon - self.interval > 0.1:                        # control.py:234-235
            self.horizon -= self.interval

    def run_nonlinear(self, c, tf):
        from . import simulator
        s = simulator.Simulator(sats=[self.sat], controller=c, scale=self.scale, base_res=self.base_res,
                                include_drag=False, include_J2=False)
        s.run(tf=tf)
        return s.sim_data[self.sat.id], s.sim_time[self.sat.id]

    def get_u_func(self):
        return self.sequence_controller.get_u_func()

    def device_law(self):
        return self.sequence_controller.device_law()
