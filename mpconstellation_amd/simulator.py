"""Simulator: drop-in for reference simulator.py.  The nonlinear rollouts (get_trajectory_ODE, one
scipy solve_ivp per satellite in the reference) run as one batched kernel on the device
(mpcx_propagate_batch); there is no host integrator fallback."""
from datetime import datetime

import numpy as np

from . import _ffi
from .constants import C_D, R_EARTH
from .control import Controller
from .satellite_scale import SatelliteScale


class OdeResult:
    """The fields of scipy's OdeResult that the reference reads (sol.y, sol.t)."""

    def __init__(self, t, y, status, nsteps):
        self.t, self.y, self.status, self.nsteps = t, y, status, nsteps
        self.success = (status == 0)


def propagate_batch(y0, tf, consts, law, n_eval, include_drag=False, include_J2=False, max_step=1e-3, device=0, slot=0,
                    Kus=None, thrust=False, devices=None, out=None, row_len=None):
    """y0 (S,7) normalised, tf (S,), consts (S,8); law = (kind, vec, Ku, end_tau) with per-satellite or
    broadcastable parameters.  Returns y (S,7,n_eval), status (S,), nsteps (S,) -- and, with thrust=True, u (S,3,n_eval) as a
    fourth value: the law evaluated at the output points, Discretizer.extract_uk of the rollout's own controller
    (linearize_discretize.py:393-411), from the same launch.
    Ragged batches: n_eval may be an (S,) integer array -- satellite s is sampled at linspace(0, 1, n_eval[s]), y has
    max(n_eval) columns, zero past a satellite's count -- and Kus (S,) gives the columns in use of each satellite's
    thrust table (law SEQUENCE, table rows of length Ku).
    devices=[d0, d1, ...]: contiguous blocks of satellites on several devices at once (sharding.sharded_call), every block
    writing in place into its slice of one result set (out / row_len: internal -- a block's views of that set, and the set's
    row length, which a ragged block uses instead of its own longest satellite's)."""
    y0 = _ffi.as_f64(y0); S = y0.shape[0]
    if devices is not None and len(devices) > 1:
        from .sharding import sharded_call
        kind, vec, Ku, end_tau = law
        bc = lambda a, shape: None if a is None else _ffi.as_f64(np.broadcast_to(np.asarray(a, dtype=np.float64), shape))
        if kind == _ffi.CTRL_CONSTANT: vec = bc(np.asarray(vec, dtype=np.float64).reshape(-1, 3), (S, 3))
        elif kind == _ffi.CTRL_TANGENTIAL: vec = bc(np.asarray(vec, dtype=np.float64).reshape(-1), (S,))
        elif kind == _ffi.CTRL_SEQUENCE:
            vec = np.asarray(vec, dtype=np.float64); vec = bc(vec if vec.ndim == 3 else vec[None], (S, 3, Ku)); end_tau = bc(end_tau, (S,))
        else: vec = None
        if kind != _ffi.CTRL_SEQUENCE: end_tau = None
        ne = np.ascontiguousarray(np.broadcast_to(np.asarray(n_eval), (S,)), dtype=np.int32) if np.ndim(n_eval) > 0 else None
        ku = None if Kus is None else np.ascontiguousarray(np.broadcast_to(np.asarray(Kus), (S,)), dtype=np.int32)
        nmax = int(ne.max()) if ne is not None else int(n_eval)
        y = _ffi.result_pool.take((S, 7, nmax)); status = np.zeros(S, dtype=np.int32); nsteps = np.zeros(S, dtype=np.int32)
        u = _ffi.result_pool.take((S, 3, nmax)) if thrust else None

        def fn(y0b, t, c, v, e, n, k, device, slot, out):
            propagate_batch(y0b, t, c, (kind, v, Ku, e), n_eval if n is None else n, include_drag, include_J2, max_step, device, slot, k, thrust,
                            None, out, nmax)
        sharded_call(fn, devices, [y0, bc(tf, (S,)), _ffi.as_f64(consts), vec, end_tau, ne, ku], dict(y=y, status=status, nsteps=nsteps, u=u))
        return (y, status, nsteps, u) if thrust else (y, status, nsteps)
    if devices is not None and len(devices) == 1:
        device = int(devices[0])
    n_evals = None
    if np.ndim(n_eval) > 0:
        n_evals = np.ascontiguousarray(np.broadcast_to(np.asarray(n_eval), (S,)), dtype=np.int32)
        n_eval = int(n_evals.max()) if row_len is None else int(row_len)
    if Kus is not None:
        Kus = np.ascontiguousarray(np.broadcast_to(np.asarray(Kus), (S,)), dtype=np.int32)
    tf = _ffi.as_f64(np.broadcast_to(np.asarray(tf, dtype=np.float64), (S,)))
    consts = _ffi.as_f64(consts)
    kind, vec, Ku, end_tau = law
    vec_p = None; et_p = None
    if kind == _ffi.CTRL_CONSTANT:
        vec = _ffi.as_f64(np.broadcast_to(np.asarray(vec, dtype=np.float64).reshape(-1, 3), (S, 3))); vec_p = _ffi.dptr(vec)
    elif kind == _ffi.CTRL_TANGENTIAL:
        vec = _ffi.as_f64(np.broadcast_to(np.asarray(vec, dtype=np.float64).reshape(-1), (S,))); vec_p = _ffi.dptr(vec)
    elif kind == _ffi.CTRL_SEQUENCE:
        vec = np.asarray(vec, dtype=np.float64)
        vec = _ffi.as_f64(np.broadcast_to(vec if vec.ndim == 3 else vec[None], (S, 3, Ku))); vec_p = _ffi.dptr(vec)
        end_tau = _ffi.as_f64(np.broadcast_to(np.asarray(end_tau, dtype=np.float64), (S,))); et_p = _ffi.dptr(end_tau)
    from .sharding import OutArrays
    oa = OutArrays(out)
    y = oa.get("y", (S, 7, n_eval)); status = oa.get("status", (S,), np.int32, lambda: np.zeros(S, dtype=np.int32))
    nsteps = oa.get("nsteps", (S,), np.int32, lambda: np.zeros(S, dtype=np.int32))
    flags = (_ffi.FLAG_DRAG if include_drag else 0) | (_ffi.FLAG_J2 if include_J2 else 0)
    lib = _ffi.load(); ctx = _ffi.context(device, slot)
    if thrust:
        u = oa.get("u", (S, 3, n_eval))
        rc = lib.mpcx_propagate_thrust_batch_ragged(ctx, S, int(n_eval), None if n_evals is None else _ffi.iptr(n_evals), _ffi.dptr(y0),
                                                    _ffi.dptr(tf), _ffi.dptr(consts), flags, kind, vec_p, int(Ku),
                                                    None if Kus is None else _ffi.iptr(Kus), et_p, float(max_step), _ffi.dptr(y),
                                                    _ffi.dptr(u), _ffi.iptr(status), _ffi.iptr(nsteps))
        _ffi.check(rc, ctx, "mpcx_propagate_thrust_batch_ragged")
        return y, status, nsteps, u
    if n_evals is None and Kus is None:
        rc = lib.mpcx_propagate_batch(ctx, S, int(n_eval), _ffi.dptr(y0), _ffi.dptr(tf), _ffi.dptr(consts), flags, kind,
                                      vec_p, int(Ku), et_p, float(max_step), _ffi.dptr(y), _ffi.iptr(status),
                                      _ffi.iptr(nsteps))
    else:
        rc = lib.mpcx_propagate_batch_ragged(ctx, S, int(n_eval), None if n_evals is None else _ffi.iptr(n_evals), _ffi.dptr(y0),
                                             _ffi.dptr(tf), _ffi.dptr(consts), flags, kind, vec_p, int(Ku),
                                             None if Kus is None else _ffi.iptr(Kus), et_p, float(max_step), _ffi.dptr(y),
                                             _ffi.iptr(status), _ffi.iptr(nsteps))
    _ffi.check(rc, ctx, "mpcx_propagate_batch")
    return y, status, nsteps


class Simulator:
    def __init__(self, sats=[], controller=Controller(), scale=SatelliteScale(), base_res=100, include_drag=True,
                 include_J2=True, verbose=False, device=0, devices=None):
        self.sim_data = {}
        self.sim_time = {}
        self.sats = sats
        self.base_res = base_res
        self.eval_points = self.base_res
        self.controller = controller
        self.include_drag = include_drag
        self.include_J2 = include_J2
        self.scale = scale
        self.verbose = verbose
        self.device = device
        self.devices = devices        # several devices: the satellites are dealt out in contiguous blocks (sharding.sharded_call)

    # ---- batched rollout of all satellites (one kernel) ----
    def _rollout(self, sats, tf):
        const = self.scale.get_normalized_constants().as_vector()
        y0 = np.stack([self.scale.normalize_state(s.get_state_vector()) for s in sats])
        law = self._device_law()
        y, status, nsteps = propagate_batch(y0, tf, np.tile(const, (len(sats), 1)), law, self.eval_points,
                                            self.include_drag, self.include_J2, 0.001, self.device, devices=self.devices)
        if (status == 1).any():
            raise Exception("ERROR: INVALID SATELLITE MASS")           # simulator.py:135-136
        if (status != 0).any():
            raise RuntimeError(f"propagation failed: {[_ffi.STATUS_TEXT.get(int(c), c) for c in status if c]}")
        t = np.linspace(0, 1, self.eval_points)
        return [OdeResult(t.copy(), y[i], int(status[i]), int(nsteps[i])) for i in range(len(sats))]

    def _device_law(self):
        """The controller's thrust law in the form the device propagator takes.  A Controller subclass that overrides
        get_u_func (the reference's extension point, control.py:20-29) without providing the matching device_law
        would otherwise be flown with its base class's law."""
        cls = type(self.controller)
        owner = lambda name: next(k for k in cls.__mro__ if name in k.__dict__)
        if not hasattr(cls, "device_law") or not issubclass(owner("device_law"), owner("get_u_func")):
            raise NotImplementedError(f"{cls.__name__} overrides get_u_func without a device_law(): only the thrust laws "
                                      "of control.py run on the device (there is no host integrator)")
        return self.controller.device_law()

    def run(self, tf=10):
        """reference simulator.py:29-48"""
        self.eval_points = int(self.base_res * tf)
        sols = self._rollout(self.sats, tf) if self.sats else []
        self.sim_data = {s.id: sol.y for s, sol in zip(self.sats, sols)}
        self.sim_time = {s.id: sol.t for s, sol in zip(self.sats, sols)}
        return self.sim_data, self.sim_time

    def run_segment(self, tf=1):
        """reference simulator.py:50-77 (the controller re-plans once per satellite, as there)"""
        self.eval_points = int(self.base_res * tf)
        for sat in self.sats:
            self.controller.update()
            sol = self._rollout([sat], tf)[0]
            sat.update_state_vector(self.scale.redim_state(sol.y[:, -1]))
            if sat.id in self.sim_data and sat.id in self.sim_time:
                time = sol.t + self.sim_time.get(sat.id, [0])[-1] * tf + 0.0000001
                self.sim_data[sat.id] = np.concatenate([self.sim_data[sat.id], sol.y], axis=1)
                self.sim_time[sat.id] = np.concatenate([self.sim_time[sat.id], time])
            else:
                self.sim_data[sat.id] = sol.y
                self.sim_time[sat.id] = sol.t

    def run_segments(self, tf=1, num_segments=1):
        """reference simulator.py:79-92"""
        tf_step = tf / float(num_segments)
        for n in range(num_segments):
            if self.verbose:
                print(f"\nRunning segment {n+1} of {num_segments}; tf {tf_step*(n+1)} of {tf}")
            self.run_segment(tf=tf_step)

    def get_trajectory_ODE(self, sat, tf, u_func=None):
        """reference simulator.py:164-189; the thrust law is the controller's (device form).  The reference integrates
        whatever callable it is handed; here only the controller's own law (what run/run_segment pass, :42,61) is
        accepted, anything else raises instead of being ignored."""
        mine = (self.controller, getattr(self.controller, "sequence_controller", None))
        owner = getattr(u_func, "_mpcx_controller", None)
        if u_func is not None and (owner is None or not any(owner is m for m in mine)):
            raise NotImplementedError("get_trajectory_ODE integrates the controller's device thrust law; an arbitrary "
                                      "u_func callable cannot run on the device (there is no host integrator)")
        return self._rollout([sat], tf)[0]

    @staticmethod
    def get_atmo_density(r, r0):
        return 9.983E-13                                               # simulator.py:97-112

    @staticmethod
    def satellite_dynamics(tau, y, u_func, tf, const, include_drag=True, include_J2=True):
        """Host evaluation of the dynamics for callers that want f(tau, y) itself (reference
        simulator.py:116-161).  The device kernels carry their own copy; this is never used by them."""
        r = y[0:3]; v = y[3:6]; m = y[6]
        if m <= 0.1:
            print(f"WARNING: low mass {m}")
        if m <= 0:
            raise Exception(f"ERROR: INVALID SATELLITE MASS: {m}")
        r_norm = np.linalg.norm(r)
        y_dot = np.zeros((7,))
        y_dot[0:3] = v
        u = u_func(y, tau)
        y_dot[3:6] = -const.MU / r_norm ** 3 * r + u / m
        if include_drag:
            y_dot[3:6] += -1 / 2 * C_D * const.S * (1 / m) * (Simulator.get_atmo_density(r, const.R0) / const.RHO) \
                * np.linalg.norm(v) * v
        if include_J2:
            q = (r[2] / r_norm) ** 2
            y_dot[3:6] += 1.5 * const.J2 * const.MU * const.R_E ** 2 / r_norm ** 5 \
                * np.array([5 * q - 1, 5 * q - 1, 5 * q - 3]) * r
        y_dot[6] = -np.linalg.norm(u) / (const.G0 * const.ISP)
        return tf * y_dot

    def save_to_csv(self, suffix="", redimensionalize=True):
        """reference simulator.py:192-201: trajectory_{date}_{id}{suffix}.csv, T rows x 7 columns"""
        date = datetime.today().strftime('%Y-%m-%d-%H-%M-%S')
        for sat in self.sats:
            data = self.scale.redim_state(self.sim_data[sat.id]) if redimensionalize else self.sim_data[sat.id]
            np.savetxt(f"trajectory_{date}_{sat.id}{suffix}.csv", data.T, delimiter=",")
