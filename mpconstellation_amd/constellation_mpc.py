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
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _ffi
from .control import _check_solver_status
from .optimizer import mpc_step_batch, scp_iteration_batch
from .satellite_scale import SatelliteScale
from .simulator import propagate_batch


def foh_resample(u, n):
    """SequenceController(u, tf_u, tf_sim=tf_u).get_u_func() evaluated at linspace(0, 1, n) (control.py:103-131:
    first-order hold with Python's float floor division, the last column at tau == 1) for a batch u (S,3,K)."""
    S, _, K = u.shape
    return foh_resample_ragged(u, np.full(S, K), np.full(S, n))


def foh_resample_ragged(u, Ku, n):
    """The same for a ragged batch, every satellite at once: table u (S,3,Kmax) with Ku[s] columns in use, evaluated at
    linspace(0, 1, n[s]) -> (S,3,max(n)), zero past a satellite's last node.  Extract_uk of the reference
    (linearize_discretize.py:393-411) for SequenceController(u_s, tf_u, tf_sim = tf_u): np.linspace's nodes (i * step, the
    last one exactly 1), k = int(tau // dtau) (numpy's float floor_divide is CPython's algorithm), tau_k = k / (K-1),
    the blend as written in control.py:122-126."""
    u = np.ascontiguousarray(u, dtype=np.float64)
    S, _, Kmax = u.shape
    Ku = np.asarray(Ku).reshape(S, 1); nn = np.asarray(n).reshape(S, 1)
    nmax = int(nn.max())
    i = np.arange(nmax, dtype=np.float64)[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        step = 1.0 / (nn - 1.0)
        tau = i * step
        tau[np.broadcast_to(nn <= 1, tau.shape)] = 0.0
        at1 = (i == nn - 1) & (nn > 1)              # np.linspace's last node is exactly 1
        tau[at1] = 1.0
        km1 = (Ku - 1).astype(np.float64)
        dtau = 1 / km1
        k = np.floor_divide(tau, dtau)
    k = np.clip(k, 0, Ku - 2).astype(np.int64)
    k[at1] = 0
    tau_k = k / km1; tau_kp1 = (k + 1) / km1
    den = tau_kp1 - tau_k
    lam_n = (tau_kp1 - tau) / den; lam_p = (tau - tau_k) / den
    keep = i < nn
    out = np.zeros((S, 3, nmax))
    base = np.arange(S, dtype=np.int64)[:, None] * (3 * Kmax)
    flat = u.reshape(-1)
    last = (Ku - 1).astype(np.int64)
    for c in range(3):                          # (flat gathers: much cheaper than take_along_axis on a broadcast index)
        off = base + c * Kmax
        val = lam_n * flat[off + k] + lam_p * flat[off + k + 1]
        val[at1] = np.broadcast_to(flat[off + last], val.shape)[at1]
        oc = out[:, c, :]
        oc[keep] = val[keep]
    return out


MAX_SLOTS = 8      # contexts (streams) used side by side (run_concurrently)


def _concurrently(fn, jobs, device, **kw):
    """Several batched calls at once, each on its own context / stream (a context is not thread-safe, include/mpcx.h:
    every worker thread of the pool owns one context slot for its whole life).  ConstellationMPC itself no longer needs
    it -- a planning step is one ragged launch -- it stays for callers that drive several constellations from one
    process."""
    if len(jobs) == 1:
        return [fn(*jobs[0], device=device, **kw)]
    n = min(MAX_SLOTS, len(jobs))
    free = queue.SimpleQueue()
    for slot in range(1, n + 1):
        free.put(slot)
    own = threading.local()

    def take_slot():
        own.slot = free.get_nowait()           # n workers, n slots: never empty

    def run(job):
        return fn(*job, device=device, slot=own.slot, **kw)

    with ThreadPoolExecutor(max_workers=n, initializer=take_slot) as pool:
        return list(pool.map(run, jobs))


class ConstellationMPC:
    def __init__(self, sats, base_res=100, tf_horizon=1, tf_interval=1, r_des=1.5, scp_iterations=2, sim_base_res=100,
                 include_drag=True, include_J2=True, device=0, strict=False, scales=None, verbose=False):
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
        self.strict = strict                  # raise instead of warning when a solve does not converge (control.py mirror)
        # per-satellite unit factors as vectors: the whole constellation is (re)dimensionalised in one array expression
        # (satellite_scale.py:46-100: r / r0, v / v0, m / m0 and back)
        self._f = np.array([[sc._r0, sc._v0, sc._m0] for sc in self.scales]).reshape(len(self.sats), 3)
        self._seg_y, self._seg_t, self._sim_cache = [], [], None       # flown segments (S,7,n) / (n,), and the dict view of them
        self.last_status = None
        self.plan_tf, self.plan_K = None, None
        self._plan = None                                               # (X, U, NU) of the last plan, rows of length Kmax
        self._plan_lists = None
        # wall-clock seconds spent inside the batched device calls (host staging included), accumulated over the updates
        self.timing = {"scp_iteration": 0.0, "truth_propagation": 0.0}     # seconds inside the library calls (rollout + discretize + solve; truth flight)

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
                self._sim_cache = ({sat.id: Y[i] for i, sat in enumerate(self.sats)}, {sat.id: T for sat in self.sats})
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
    def update(self, y0=None):
        S = len(self.sats)
        y0 = self._y0() if y0 is None else y0
        K = int(self.base_res * self.horizon)
        # Every SCP iteration is ONE library call (scp_iteration_batch): the nonlinear rollout under the iteration's thrust law
        # -- the tangential reference controller first (control.py:183-187), then the sequence just optimised, played over its
        # own horizon and sampled at int(base_res * tf_u) nodes per satellite (control.py:217-227, simulator.py:38: a ragged
        # batch) --, extract_uk at its nodes, discretisation and solve.  x_bar and u_bar never come to the host.
        tf_u = np.full(S, float(self.horizon))
        law = (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None)
        Ks = None; Kus = None                                              # first iteration: K nodes for everybody
        self.last_status = np.zeros((self.scp_iterations, S), dtype=np.int32)
        opts = {"eps_r": 0.000001, "eps_vr": 0.0000000000000001, "tf_max": self.horizon}     # control.py:192-197
        res = None
        for it in range(self.scp_iterations):
            Krow = K if Ks is None else int(Ks.max())
            res = self._timed("scp_iteration", scp_iteration_batch, y0, tf_u, self.consts, self.r_des, law, Krow, options=opts,
                              Ks=Ks, Kus=Kus, device=self.device)
            self._check(res.prop_status)
            self.last_status[it] = res.status
            _check_solver_status(res.status, self.strict)
            if self.verbose:
                for j in range(S):
                    print(f"tf for optimizer: {res.tf[j]}")
                    print(f"Total virtual control effort: {np.abs(res.NU[j]).sum()}")
            tf_u = res.tf.copy()
            if it == self.scp_iterations - 1:
                break                 # (the reference re-rolls once more, control.py:227, and drops the result)
            Kus = np.full(S, Krow) if Ks is None else Ks                   # columns in use of the table the next rollout plays
            Ks = (self.base_res * res.tf).astype(np.int32)                 # ... sampled at int(base_res * tf_u) nodes
            law = (_ffi.CTRL_SEQUENCE, res.U, res.U.shape[2], 1.0)         # SequenceController(u_opt, tf_u, tf_sim = tf_u)
        Kp = np.full(S, res.X.shape[2]) if Ks is None else Ks
        self.plan_K = Kp.astype(np.int32)
        self._plan = (res.X, res.U, res.NU)                                # rows of length Kmax; U is the table the segment is flown with
        self._plan_lists = None
        self.plan_tf = tf_u.copy()
        if self.horizon - self.interval > 0.1:                               # control.py:234-235
            self.horizon -= self.interval

    # ---- Simulator.run_segment for every satellite: plan, fly tf under the truth model, update the states ----
    def run_segment(self, tf=1):
        y0 = self._y0()
        self.update(y0)
        n_eval = int(self.sim_base_res * tf)
        U = self._plan[1]
        # SequenceController(tf_u, tf_sim = interval): end_tau = tf_u / interval; one launch, tables of plan_K[s] columns
        y, st, _ = self._timed("truth_propagation", propagate_batch, y0, tf, self.consts,
                               (_ffi.CTRL_SEQUENCE, U, U.shape[2], self.plan_tf / self.interval), n_eval,
                               self.include_drag, self.include_J2, 0.001, self.device, Kus=self.plan_K)
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
