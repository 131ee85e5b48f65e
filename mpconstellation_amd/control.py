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
    """MPC / SCP controller (reference control.py:145-246) for `sats[0]` (:162).  Its plan is the one-satellite case of
    ConstellationMPC.update -- reference rollout, then SCPn x (extract u_bar, discretize + solve on the device, nonlinear
    re-rollout under the optimised sequence), horizon shrink -- and this class keeps the reference's attributes on top of
    it: horizon, interval, base_res, r_des, SCPn_iterations (read before every update, as the reference reads them),
    opt_trajectory, sequence_controller."""

    def __init__(self, sats=[], objective=None, base_res=100, tf_horizon=1, tf_interval=1, plot_inter=True,
                 opt_verbose=True, r_des=1.5, strict=False, device=0, time_parallel=None):
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
        self.device = device
        # The plan's solves run on the time-parallel kernel (include/mpcx.h, MPCX_SOLVE_TIME_PARALLEL: the horizon in four segments
        # side by side) BY DEFAULT since round 5: this controller plans for ONE satellite (control.py:162), the case that kernel
        # is for -- 1.19 against 1.39 ms per solve at 30 nodes, 1.33 / 1.86 at 60 -- and the kernel is pinned directly against the
        # oracle and the independent scipy solutions (tests/test_time_parallel_oracle_gpu.py: 4e-15 ... 1e-12 on the 30-node
        # fixtures and OptimalController's stiff option set).  It needs the device to itself while it runs: a solve whose
        # workgroups could not all become resident reports MPCX_ST_TIMEOUT and the update is repeated on the default kernels
        # (ConstellationMPC.update).  time_parallel=False: the default kernels, whose bits do not depend on the batch size.
        self.time_parallel = True if time_parallel is None else bool(time_parallel)

    def update(self):
        from .constellation_mpc import ConstellationMPC
        mpc = ConstellationMPC([self.sat], base_res=self.base_res, tf_horizon=self.horizon, tf_interval=self.interval,
                               r_des=self.r_des, scp_iterations=self.SCPn_iterations, device=self.device, strict=self.strict,
                               scales=[self.scale], verbose=self.opt_verbose, time_parallel=self.time_parallel)
        mpc.update()
        self.last_status = [int(c) for c in mpc.last_status[:, 0]]
        self.opt_trajectory = mpc.plan_x[0]
        self.sequence_controller = SequenceController(u=mpc.plan_u[0], tf_u=float(mpc.plan_tf[0]), tf_sim=self.interval)
        self.horizon = mpc.horizon                                   # control.py:234-235

    def run_nonlinear(self, c, tf):
        from . import simulator
        s = simulator.Simulator(sats=[self.sat], controller=c, scale=self.scale, base_res=self.base_res,
                                include_drag=False, include_J2=False, device=self.device)
        s.run(tf=tf)
        return s.sim_data[self.sat.id], s.sim_time[self.sat.id]

    def get_u_func(self):
        return self.sequence_controller.get_u_func()

    def device_law(self):
        return self.sequence_controller.device_law()
