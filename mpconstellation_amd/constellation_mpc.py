"""ConstellationMPC -- the reference's MPC loop for a whole constellation at once.

The reference runs OptimalController.update (control.py:166-235) for ONE satellite (`self.sats[0]`, :162) and
Simulator.run_segment calls it once per satellite in a Python loop (simulator.py:58-60).  Here every step of that loop
is one batched device call over all satellites, each with its own SatelliteScale (so each sees MU = 4 pi^2):

    reference rollout (tangential 0.5)  ->  SCPn x [ extract u_bar, discretise + solve, re-rollout under the
    optimised first-order-hold sequence over tf_u ]  ->  fly the segment under the truth model, update the states.

The number of nodes of the second SCP iteration is int(base_res * tf_u) and differs between satellites, as does the
length of the thrust table played back during the segment: those steps are ragged launches (include/mpcx.h,
mpcx_*_ragged: per-satellite node counts inside one rectangular batch), so each satellite gets exactly the result of the
single-satellite path (tests/test_mpc_loop_gpu.py) and the host never groups or loops over satellites."""
import time

import numpy as np

from . import _ffi
from .control import _check_solver_status
from .optimizer import mpc_step_batch, mpc_update_batch, scp_iteration_batch
from .satellite_scale import SatelliteScale
from .simulator import propagate_batch


class ConstellationMPC:
    def __init__(self, sats, base_res=100, tf_horizon=1, tf_interval=1, r_des=1.5, scp_iterations=2, sim_base_res=100,
                 include_drag=True, include_J2=True, device=0, strict=False, scales=None, verbose=False, devices=None,
                 time_parallel=False):
        self.sats = list(sats)
        # every satellite in its own "designer units" (so that each sees MU = 4 pi^2) unless the caller brings the scales
        self.scales = list(scales) if scales is not None else [SatelliteScale(sat=s) for s in self.sats]
        self.verbose = verbose                # control.py:208-209's prints, per satellite
        self.consts = np.stack([sc.get_normalized_constants().as_vector() for sc in self.scales])
        self.base_res, self.sim_base_res = base_res, sim_base_res
        self.horizon, self.interval = tf_horizon, tf_interval
        self.r_des = np.broadcast_to(np.asarray(r_des, dtype=np.float64), (len(self.sats),)).copy()
        self.scp_iterations = scp_iterations
        self.include_drag, self.include_J2 = include_drag, include_J2
        self.device = device
        # devices=[0, 1, ..., 7]: the constellation is dealt out in contiguous blocks to these devices, one host thread and one
        # context per device, no exchange between them (sharding.sharded_call) -- the reference loops over
        # its satellites serially (simulator.py:41,58)
        self.devices = list(devices) if devices is not None else None
        if verbose and self.devices is not None and len(self.devices) > 1:
            # (verbose prints control.py:208-209's lines between the SCP iterations: one library call per iteration on ONE
            #  context; silently solving on self.device while the flight is sharded would be neither of the two things asked for)
            raise ValueError("ConstellationMPC: verbose=True runs the SCP iterations as separate calls on one device; "
                             "use devices=[...] without verbose, or verbose with a single device")
        # time_parallel: the solves of small constellations (up to 128 satellites) on the time-parallel kernel (include/mpcx.h,
        # MPCX_SOLVE_TIME_PARALLEL: the horizon in four segments side by side; same iterations, not the other kernels' bits)
        self.solver_flags = _ffi.SOLVE_TIME_PARALLEL if time_parallel else 0
        self.strict = strict                  # raise instead of warning when a solve does not converge (control.py mirror)
        # per-satellite unit factors as vectors: the whole constellation is (re)dimensionalised in one array expression
        # (satellite_scale.py:46-100: r / r0, v / v0, m / m0 and back)
        self._f = np.array([[sc._r0, sc._v0, sc._m0] for sc in self.scales]).reshape(len(self.sats), 3)
        self._seg_y, self._seg_t, self._sim_cache = [], [], None       # flown segments (S,7,n) / (n,), and the dict view of them
        self.last_status = None           # (scp_iterations, S): every solve's MPCX_ST_* code of the last update
        self.last_iters = None            # ... and its interior-point iteration count
        self.plan_tf, self.plan_K = None, None
        self._plan = None                                               # (X, U, NU) of the last plan, rows of length Kmax
        self._plan_lists = None
        # wall-clock seconds spent inside the batched device calls (host staging included), accumulated over the updates
        self.timing = {"update": 0.0, "truth_propagation": 0.0}     # seconds inside the library calls (update [+ segment flight]; separate flight)

    def _timed(self, key, fn, *a, **kw):
        t0 = time.perf_counter()
        out = fn(*a, **kw)
        self.timing[key] += time.perf_counter() - t0
        return out

    def _y0(self):
        """normalised states of all satellites (SatelliteScale.normalize_state of each, in one expression)"""
        pos = np.array([s.position for s in self.sats], dtype=np.float64).reshape(-1, 3)
        vel = np.array([s.velocity for s in self.sats], dtype=np.float64).reshape(-1, 3)
        m = np.array([s.mass for s in self.sats], dtype=np.float64)
        f = self._f
        return np.column_stack([pos / f[:, 0:1], vel / f[:, 1:2], m / f[:, 2]])

    # the reference keeps sim_data / sim_time as dicts id -> array (simulator.py:18-19); built on demand from the batched
    # segments (one concatenation for the constellation instead of one per satellite and segment)
    def _sim_dicts(self):
        if self._sim_cache is None:
            if not self._seg_y:
                self._sim_cache = ({}, {})
            else:
                Y = np.concatenate(self._seg_y, axis=2); T = np.concatenate(self._seg_t)
                self._sim_cache = ({sat.id: Y[i] for i, sat in enumerate(self.sats)}, {sat.id: T.copy() for sat in self.sats})      # (an array per id, simulator.py:69-76)
        return self._sim_cache

    @property
    def sim_data(self):
        return self._sim_dicts()[0]

    @property
    def sim_time(self):
        return self._sim_dicts()[1]

    # the plan per satellite, trimmed to its own node count: lists of views, built on demand
    def _plan_views(self):
        if self._plan_lists is None and self._plan is not None:
            Kp = self.plan_K
            self._plan_lists = tuple([a[s][:, :Kp[s]] for s in range(len(self.sats))] for a in self._plan)
        return self._plan_lists or (None, None, None)

    @property
    def plan_x(self):
        return self._plan_views()[0]

    @property
    def plan_u(self):
        return self._plan_views()[1]

    @property
    def plan_nu(self):
        return self._plan_views()[2]

    # ---- OptimalController.update for every satellite ----
    OPTIONS = staticmethod(lambda horizon: {"eps_r": 0.000001, "eps_vr": 0.0000000000000001, "tf_max": horizon})      # control.py:192-197

    def update(self, y0=None, fly=None):
        """The whole update is ONE library call (mpc_update_batch -> mpcx_mpc_update_batch): reference rollout under the tangential
        controller (control.py:178-180), then per SCP iteration extract_uk, discretisation and solve, the re-rollout under the
        sequence just optimised sampled at int(base_res * tf_u) nodes per satellite (control.py:217-227, simulator.py:38: node
        counts computed on the device, ragged launches), the plan's thrust consumed in place -- only the final plan comes back.
        fly: also fly the segment (run_segment).  verbose=True prints control.py:208-209's lines per iteration and therefore runs
        the iterations as separate calls (scp_iteration_batch), with the same results bit for bit."""
        S = len(self.sats)
        y0 = self._y0() if y0 is None else y0
        K = int(self.base_res * self.horizon)
        opts = self.OPTIONS(self.horizon)
        for flags in ((self.solver_flags, 0) if self.solver_flags & _ffi.SOLVE_TIME_PARALLEL else (self.solver_flags,)):
            if self.verbose:
                res = self._update_by_iterations(y0, K, opts, flags)
                flown = None
            else:
                res = self._timed("update", mpc_update_batch, y0, float(self.horizon), self.consts, self.r_des, self.base_res,
                                  n_scp=self.scp_iterations, options=opts, device=self.device, fly=fly, devices=self.devices,
                                  flags=flags)
                self._check(res.prop_status)
                self.last_status = res.status; self.last_iters = res.iters
                flown = (res.y_sim, res.sim_status) if fly is not None else None
            # A time-parallel solve whose workgroups could not all run at once (the device shared with another long kernel:
            # MPCX_ST_TIMEOUT, include/mpcx.h) is not a failed plan: the update is done again on the default kernels.
            if flags & _ffi.SOLVE_TIME_PARALLEL and (np.asarray(self.last_status) == 10).any():
                import warnings
                warnings.warn("time-parallel solve timed out waiting for its workgroups (device busy?): update repeated on the default kernels",
                              RuntimeWarning, stacklevel=2)
                continue
            break
        for it in range(self.scp_iterations):
            _check_solver_status(np.asarray(self.last_status)[it], self.strict)
        self.plan_K = res.Ks.astype(np.int32)
        self._plan = (res.X, res.U, res.NU)                                # rows of length K; U is the table the segment is flown with
        self._plan_lists = None
        self.plan_tf = res.tf.copy()
        if self.horizon - self.interval > 0.1:                               # control.py:234-235
            self.horizon -= self.interval
        return flown

    def _update_by_iterations(self, y0, K, opts, flags=0):
        """the same update as one library call per SCP iteration (the plan crosses PCIe between them): what verbose mode needs"""
        S = len(self.sats)
        tf_u = np.full(S, float(self.horizon))
        law = (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None)
        Ks = None; Kus = None                                              # first iteration: K nodes for everybody
        self.last_status = np.zeros((self.scp_iterations, S), dtype=np.int32); self.last_iters = np.zeros_like(self.last_status)
        res = None
        for it in range(self.scp_iterations):
            res = self._timed("update", scp_iteration_batch, y0, tf_u, self.consts, self.r_des, law, K, options=opts,
                              Ks=Ks, Kus=Kus, device=self.device, flags=flags)
            self._check(res.prop_status)
            self.last_status[it] = res.status; self.last_iters[it] = res.iters
            for j in range(S):
                print(f"tf for optimizer: {res.tf[j]}")
                print(f"Total virtual control effort: {np.abs(res.NU[j]).sum()}")
            tf_u = res.tf.copy()
            if it == self.scp_iterations - 1:
                break                 # (the reference re-rolls once more, control.py:227, and drops the result)
            Kus = Ks                                                       # columns in use of the table the next rollout plays
            Ks = (self.base_res * res.tf).astype(np.int32)                 # ... sampled at int(base_res * tf_u) nodes
            law = (_ffi.CTRL_SEQUENCE, res.U, K, 1.0)                      # SequenceController(u_opt, tf_u, tf_sim = tf_u)
        res.Ks = np.full(S, K, dtype=np.int32) if Ks is None else Ks
        return res

    # ---- Simulator.run_segment for every satellite: plan, fly tf under the truth model, update the states ----
    def run_segment(self, tf=1):
        y0 = self._y0()
        n_eval = int(self.sim_base_res * tf)
        # SequenceController(tf_u, tf_sim = interval): end_tau = tf_u / interval; the flight rides in the update's call
        flown = self.update(y0, fly=(tf, self.interval, n_eval, self.include_drag, self.include_J2, 0.001))
        if flown is None:                                                    # (verbose path: the flight as its own call)
            U = self._plan[1]
            y, st, _ = self._timed("truth_propagation", propagate_batch, y0, tf, self.consts,
                                   (_ffi.CTRL_SEQUENCE, U, U.shape[2], self.plan_tf / self.interval), n_eval,
                                   self.include_drag, self.include_J2, 0.001, self.device, Kus=self.plan_K, devices=self.devices)
        else:
            y, st = flown
        self._check(st)
        t = np.linspace(0, 1, n_eval)
        f = self._f
        end = np.column_stack([y[:, 0:3, -1] * f[:, 0:1], y[:, 3:6, -1] * f[:, 1:2], y[:, 6, -1] * f[:, 2]])     # redim_state
        for i, sat in enumerate(self.sats):
            sat.update_state_vector(end[i])
        if self._seg_t:
            t = t + self._seg_t[-1][-1] * tf + 0.0000001                  # simulator.py:69-76
        self._seg_y.append(y); self._seg_t.append(t); self._sim_cache = None

    def run_segments(self, tf=1, num_segments=1):
        for _ in range(num_segments):
            self.run_segment(tf=tf / float(num_segments))

    @staticmethod
    def _check(status):
        if (status == 1).any():
            raise Exception("ERROR: INVALID SATELLITE MASS")               # simulator.py:135-136
        if (status != 0).any():
            raise RuntimeError(f"propagation failed: {[_ffi.STATUS_TEXT.get(int(c), c) for c in status if c]}")
