"""ConstellationMPC -- the reference's MPC loop for a whole constellation at once.

The reference runs OptimalController.update (control.py:166-235) for ONE satellite (`self.sats[0]`, :162) and
Simulator.run_segment calls it once per satellite in a Python loop (simulator.py:58-60).  Here every step of that loop
is one batched device call over all satellites, each with its own SatelliteScale (so each sees MU = 4 pi^2):

    reference rollout (tangential 0.5)  ->  SCPn x [ extract u_bar, discretise + solve, re-rollout under the
    optimised first-order-hold sequence over tf_u ]  ->  fly the segment under the truth model, update the states.

The number of nodes of the second SCP iteration is int(base_res * tf_u) and differs between satellites, as does the
length of the thrust table played back during the segment; satellites are grouped by node count and every group is
one batch, so each satellite gets exactly the result of the single-satellite path (tests/test_mpc_loop_gpu.py)."""
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _ffi
from .constellation import tangential_thrust
from .control import _check_solver_status
from .optimizer import mpc_step_batch
from .satellite_scale import SatelliteScale
from .simulator import propagate_batch


def foh_resample(u, n):
    """SequenceController(u, tf_u, tf_sim=tf_u).get_u_func() evaluated at linspace(0, 1, n) (control.py:103-131:
    first-order hold with Python's float floor division, the last column at tau == 1) for a batch u (S,3,K)."""
    S, _, K = u.shape
    tau = np.linspace(0, 1, n)
    dtau = 1 / (K - 1)
    out = np.empty((S, 3, n))
    for i, tq in enumerate(tau):
        if tq == 1:
            out[:, :, i] = u[:, :, -1]
            continue
        k = int(tq // dtau)
        tau_k = k / (K - 1); tau_kp1 = (k + 1) / (K - 1)
        lam_n = (tau_kp1 - tq) / (tau_kp1 - tau_k); lam_p = (tq - tau_k) / (tau_kp1 - tau_k)
        out[:, :, i] = lam_n * u[:, :, k] + lam_p * u[:, :, k + 1]
    return out


MAX_SLOTS = 8      # contexts (streams) used side by side by one ConstellationMPC step


def _concurrently(fn, jobs, device, **kw):
    """Several batched calls at once.  A rollout is ~1000 sequential RK45 steps and a solve ~25-50 sequential
    interior-point iterations however few satellites the call carries, so the groups run concurrently.  A context is
    not thread-safe (include/mpcx.h: one context per host thread): every worker thread of the pool owns one context
    slot for its whole life (taken from a queue by the pool initialiser), so however many jobs there are no two calls
    in flight ever share a context."""
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


def _rollouts(jobs, device):
    return _concurrently(propagate_batch, jobs, device)


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
        self.sim_data, self.sim_time = {}, {}
        self.last_status = None
        self.plan_u, self.plan_tf, self.plan_x, self.plan_nu = None, None, None, None
        # wall-clock seconds spent inside the batched device calls (host staging included), accumulated over the updates
        self.timing = {"rollouts": 0.0, "discretize_solve": 0.0, "truth_propagation": 0.0}

    def _timed(self, key, fn, *a, **kw):
        t0 = time.perf_counter()
        out = fn(*a, **kw)
        self.timing[key] += time.perf_counter() - t0
        return out

    def _y0(self):
        return np.stack([sc.normalize_state(s.get_state_vector()) for sc, s in zip(self.scales, self.sats)])

    # ---- OptimalController.update for every satellite ----
    def update(self):
        S = len(self.sats)
        y0 = self._y0()
        K = int(self.base_res * self.horizon)
        x, st, _ = self._timed("rollouts", propagate_batch, y0, self.horizon, self.consts,
                               (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), K, False, False, 0.001, self.device)
        self._check(st)
        u_bar = tangential_thrust(x, 0.5)                                  # extract_uk of the tangential controller
        tf_u = np.full(S, float(self.horizon))
        groups = {K: np.arange(S)}
        xs = {K: x}; us = {K: u_bar}
        self.last_status = np.zeros((self.scp_iterations, S), dtype=np.int32)
        plan_u = [None] * S; plan_x = [None] * S; plan_nu = [None] * S
        opts = {"eps_r": 0.000001, "eps_vr": 0.0000000000000001, "tf_max": self.horizon}     # control.py:192-197
        for it in range(self.scp_iterations):
            nxt_groups, nxt_x, nxt_u = {}, {}, {}
            keys = list(groups)
            jobs = [(xs[Kg], us[Kg], tf_u[groups[Kg]], self.consts[groups[Kg]], self.r_des[groups[Kg]]) for Kg in keys]
            if S <= 256:      # small groups: the solves are latency bound and overlap (measured: -12 % at 64 satellites)
                solved = self._timed("discretize_solve", _concurrently, mpc_step_batch, jobs, self.device, options=opts)
            else:             # large ones fill the device on their own; separate contexts would only regrow workspaces
                solved = self._timed("discretize_solve", lambda: [mpc_step_batch(*job, options=opts, device=self.device) for job in jobs])
            for Kg, res in zip(keys, solved):
                idx = groups[Kg]
                self.last_status[it, idx] = res.status
                _check_solver_status(res.status, self.strict)
                tf_u[idx] = res.tf
                for j, s in enumerate(idx):
                    plan_u[s] = res.U[j]; plan_x[s] = res.X[j]; plan_nu[s] = res.NU[j]
                    if self.verbose:
                        print(f"tf for optimizer: {res.tf[j]}")
                        print(f"Total virtual control effort: {np.abs(res.NU[j]).sum()}")
                if it == self.scp_iterations - 1:
                    continue          # (the reference re-rolls once more, control.py:227, and drops the result)
                # nonlinear re-rollout under the optimised sequence over tf_u, sampled at int(base_res * tf_u) nodes
                Kn = (self.base_res * res.tf).astype(int)
                kns = np.unique(Kn)
                sels = [np.nonzero(Kn == kn)[0] for kn in kns]
                outs = self._timed("rollouts", _rollouts,
                                   [(y0[idx[sel]], tf_u[idx[sel]], self.consts[idx[sel]], (_ffi.CTRL_SEQUENCE, res.U[sel], Kg, 1.0),
                                     int(kn), False, False, 0.001) for kn, sel in zip(kns, sels)], self.device)
                for kn, sel, (xr, st, _) in zip(kns, sels, outs):
                    gi = idx[sel]
                    self._check(st)
                    ur = foh_resample(res.U[sel], int(kn))              # extract_uk of SequenceController(tf_sim = tf_u)
                    if int(kn) in nxt_groups:
                        nxt_groups[int(kn)] = np.concatenate([nxt_groups[int(kn)], gi])
                        nxt_x[int(kn)] = np.concatenate([nxt_x[int(kn)], xr]); nxt_u[int(kn)] = np.concatenate([nxt_u[int(kn)], ur])
                    else:
                        nxt_groups[int(kn)] = gi; nxt_x[int(kn)] = xr; nxt_u[int(kn)] = ur
            groups, xs, us = nxt_groups, nxt_x, nxt_u
        self.plan_u, self.plan_x, self.plan_nu, self.plan_tf = plan_u, plan_x, plan_nu, tf_u.copy()
        if self.horizon - self.interval > 0.1:                               # control.py:234-235
            self.horizon -= self.interval

    # ---- Simulator.run_segment for every satellite: plan, fly tf under the truth model, update the states ----
    def run_segment(self, tf=1):
        self.update()
        n_eval = int(self.sim_base_res * tf)
        S = len(self.sats)
        y0 = self._y0()
        Ku = np.array([u.shape[1] for u in self.plan_u])
        y = np.empty((S, 7, n_eval))
        kus = np.unique(Ku)
        gis = [np.nonzero(Ku == ku)[0] for ku in kus]
        # SequenceController(tf_u, tf_sim = interval): end_tau = tf_u / interval
        outs = self._timed("truth_propagation", _rollouts,
                           [(y0[gi], tf, self.consts[gi], (_ffi.CTRL_SEQUENCE, np.stack([self.plan_u[s] for s in gi]), int(ku),
                                                           self.plan_tf[gi] / self.interval), n_eval, self.include_drag, self.include_J2,
                             0.001) for ku, gi in zip(kus, gis)], self.device)
        for gi, (yy, st, _) in zip(gis, outs):
            self._check(st)
            y[gi] = yy
        t = np.linspace(0, 1, n_eval)
        for i, (sat, sc) in enumerate(zip(self.sats, self.scales)):
            sat.update_state_vector(sc.redim_state(y[i][:, -1]))
            if sat.id in self.sim_data:
                time = t + self.sim_time[sat.id][-1] * tf + 0.0000001       # simulator.py:69-76
                self.sim_data[sat.id] = np.concatenate([self.sim_data[sat.id], y[i]], axis=1)
                self.sim_time[sat.id] = np.concatenate([self.sim_time[sat.id], time])
            else:
                self.sim_data[sat.id] = y[i]; self.sim_time[sat.id] = t.copy()

    def run_segments(self, tf=1, num_segments=1):
        for _ in range(num_segments):
            self.run_segment(tf=tf / float(num_segments))

    @staticmethod
    def _check(status):
        if (status == 1).any():
            raise Exception("ERROR: INVALID SATELLITE MASS")               # simulator.py:135-136
        if (status != 0).any():
            raise RuntimeError(f"propagation failed: {[_ffi.STATUS_TEXT.get(int(c), c) for c in status if c]}")
