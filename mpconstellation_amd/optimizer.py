"""Optimizer: drop-in for the reference class of the same name (reference optimizer.py:12-613).
solve_OPT runs discretize -> constraint terms -> interior-point solve on the MI355X through libmpcx.so
(mpcx_mpc_step_batch); there is no pyomo model and no ipopt subprocess, and no host fallback."""
import numpy as np

from . import _ffi

DEFAULT_OPTIONS = {'min_mass': 0.1, 'u_lim': [0, 5], 'r_lim': [0.99, 5], 'r_des': 1, 'eps_r': 0.01,
                   'eps_vr': 0.00001, 'eps_vn': 0.00001, 'eps_vt': 0.00001, 'tf_max': 5, 'w_nu': 1000, 'w_tr': 0.002}


class SolveResult:
    """What the reference keeps in self.model (a pyomo object) reduced to what its callers read."""

    def __init__(self, X, U, NU, tf, status, iters, kkt, g_tf=None, regularised=None):
        self.X, self.U, self.NU, self.tf, self.status, self.iters, self.kkt = X, U, NU, tf, status, iters, kkt
        self.g_tf = g_tf          # fixed-tf solves only: each satellite's term of the tf stationarity row (include/mpcx.h)
        # regularised=True calls only: per satellite, the iterations that needed delta_w > 0 and the first of them (-1: none)
        self.n_regularised = None if regularised is None else regularised[:, 0]
        self.first_regularised = None if regularised is None else regularised[:, 1]


def _regularised(lib, ctx, S, out=None):
    """mpcx_solve_regularised of the solve that just returned on this context"""
    out = np.zeros((S, 2), dtype=np.int32) if out is None else out
    _ffi.check(lib.mpcx_solve_regularised(ctx, S, _ffi.iptr(out)), ctx, "mpcx_solve_regularised")
    return out


def constraint_terms_batch(xbar, consts, r_des, options=None, device=0, linear_vt=False):
    """What the device builds from Optimizer.get_constraint_terms (optimizer.py:80-170) before its first iteration
    (include/mpcx.h, mpcx_constraint_terms): aT (S,8,7), bT (S,8), scalars (S,8)."""
    xbar = _ffi.as_f64(xbar)
    S, _, K = xbar.shape
    consts = _ffi.as_f64(consts)
    r_des = _ffi.as_f64(np.broadcast_to(np.asarray(r_des, dtype=np.float64), (S,)))
    opts = _ffi.make_solve_opts(options, **_solver_flags({}, linear_vt))
    aT = np.empty((S, 8, 7)); bT = np.empty((S, 8)); sc = np.empty((S, _ffi.NTERM_SCALARS))
    lib = _ffi.load(); ctx = _ffi.context(device)
    import ctypes as C
    rc = lib.mpcx_constraint_terms(ctx, S, K, _ffi.dptr(xbar), _ffi.dptr(consts), _ffi.dptr(r_des), C.byref(opts),
                                   _ffi.dptr(aT), _ffi.dptr(bT), _ffi.dptr(sc))
    _ffi.check(rc, ctx, "mpcx_constraint_terms")
    return aT, bT, sc


def _solver_flags(solver, linear_vt, fixed_tf=None, shared_tf=False):
    """linear_vt: the linearised tangential pair the reference keeps commented out (optimizer.py:471-489, 575-576;
    tolerance options['eps_vt']) instead of the exact equality it enables (:577) -- MPCX_SOLVE_LINEAR_VT.
    fixed_tf: hold every satellite's final time at the given value(s) -- MPCX_SOLVE_FIXED_TF.
    shared_tf: ONE final time for the whole batch, solved as one problem on the device -- MPCX_SOLVE_SHARED_TF."""
    solver = dict(solver)
    if shared_tf:
        solver["flags"] = int(solver.get("flags", 0)) | _ffi.SOLVE_SHARED_TF
    if linear_vt:
        solver["flags"] = int(solver.get("flags", 0)) | _ffi.SOLVE_LINEAR_VT
    if fixed_tf is not None:
        solver["flags"] = int(solver.get("flags", 0)) | _ffi.SOLVE_FIXED_TF
    return solver


_pinned_results = {}


def _result_arrays(S, K, device, pinned):
    """X, U, NU, kkt, status, iters for one call.  pinned: page-locked buffers owned by this module and REUSED by the next
    call of the same shape on the device (results are DMA targets, no staging copy; copy what must outlive the next call)."""
    if not pinned:
        take = _ffi.result_pool.take         # (large arrays are recycled once the caller has dropped the previous results)
        return (take((S, 7, K)), take((S, 3, K)), take((S, 7, K)), np.empty(S), np.zeros(S, dtype=np.int32),
                np.zeros(S, dtype=np.int32))
    key = (S, K, device)          # (device: an index, or the tuple of a multi-device call -- one page-locked set for the constellation)
    device = device[0] if isinstance(device, tuple) else device
    if key not in _pinned_results:
        _pinned_results[key] = (_ffi.pinned_empty((S, 7, K), device=device), _ffi.pinned_empty((S, 3, K), device=device),
                                _ffi.pinned_empty((S, 7, K), device=device), _ffi.pinned_empty((S,), device=device),
                                _ffi.pinned_empty((S,), np.int32, device), _ffi.pinned_empty((S,), np.int32, device))
    return _pinned_results[key]


def _tf_io(S, fixed_tf):
    """tf_out buffer: plain output, or (fixed-tf mode) the values to hold on entry and g_s on exit"""
    if fixed_tf is None:
        return np.empty(S), None
    held = _ffi.as_f64(np.broadcast_to(np.asarray(fixed_tf, dtype=np.float64), (S,))).copy()
    return held.copy(), held


def mpc_step_batch(xbar, ubar, tf, consts, r_des, options=None, include_J2=False, max_step=1e-2, device=0, slot=0,
                   linear_vt=False, fixed_tf=None, pinned_results=False, uniform_steps=0, regularised=False, Ks=None, shared_tf=False,
                   devices=None, rk23=False, out=None, **solver):
    """S independent satellite-MPC-steps (discretize + solve) on the device.
    xbar (S,7,K), ubar (S,3,K), tf (S,), consts (S,8), r_des (S,) -> SolveResult with batched arrays.
    Ks (S,) int: a ragged batch -- satellite s has Ks[s] <= K nodes in the first columns of its rows (what the reference's
    second SCP iteration poses: int(base_res * tf_u) nodes per satellite, control.py:227); result columns past a
    satellite's count are zero.
    Inputs that live in page-locked memory (_ffi.pinned_copy) are transferred without a staging copy; pinned_results=True
    returns the results in page-locked buffers that the next call of the same shape overwrites.
    devices=[d0, d1, ...]: the satellites are dealt out in contiguous blocks to these devices (sharding.sharded_call: one
    host thread and context per device, no exchange between them), every block writing its results in place into its slice of
    ONE result set for the constellation (page-locked with pinned_results=True: then every device's DMA lands in the caller's
    arrays directly); satellites are independent units, so every satellite gets bit for bit what a single-device call gives it.
    out: (internal) views of such a result set for this call's satellites -- X, U, NU, tf, status, iters, kkt[, regularised]."""
    if devices is not None and len(devices) > 1:
        if shared_tf or fixed_tf is not None:
            raise ValueError("devices=[...]: independent per-satellite problems only (no shared / fixed tf)")
        from .sharding import sharded_call
        xbar = _ffi.as_f64(xbar); S, _, K = xbar.shape
        bc = lambda a: _ffi.as_f64(np.broadcast_to(np.asarray(a, dtype=np.float64), (S,)))
        Ksb = None if Ks is None else np.ascontiguousarray(np.broadcast_to(np.asarray(Ks), (S,)), dtype=np.int32)
        X, U, NU, kkt, status, iters = _result_arrays(S, K, tuple(int(d) for d in devices) if pinned_results else int(devices[0]), pinned_results)
        tfo = np.empty(S); reg = np.zeros((S, 2), dtype=np.int32) if regularised else None
        fn = lambda x, u, t, c, r, k, device, slot, out: mpc_step_batch(x, u, t, c, r, options, include_J2, max_step, device, slot, linear_vt,
                                                                        None, False, uniform_steps, regularised, k, False, None, rk23, out, **solver)
        sharded_call(fn, devices, [xbar, _ffi.as_f64(ubar), bc(tf), _ffi.as_f64(consts), bc(r_des), Ksb],
                     dict(X=X, U=U, NU=NU, kkt=kkt, status=status, iters=iters, tf=tfo, regularised=reg))
        return SolveResult(X, U, NU, tfo, status, iters, kkt, regularised=reg)
    if devices is not None and len(devices) == 1:
        device = int(devices[0])
    solver = _solver_flags(solver, linear_vt, fixed_tf, shared_tf)
    xbar = _ffi.as_f64(xbar); ubar = _ffi.as_f64(ubar)
    S, _, K = xbar.shape
    if xbar.shape[1] != 7 or ubar.shape != (S, 3, K):
        raise ValueError("expected xbar (S,7,K) and ubar (S,3,K)")
    tf = _ffi.as_f64(np.broadcast_to(np.asarray(tf, dtype=np.float64), (S,)))
    r_des = _ffi.as_f64(np.broadcast_to(np.asarray(r_des, dtype=np.float64), (S,)))
    consts = _ffi.as_f64(consts)
    opts = _ffi.make_solve_opts(options, **solver)
    if out is None:
        X, U, NU, kkt, status, iters = _result_arrays(S, K, device, pinned_results and slot == 0)
        tfo, held = _tf_io(S, fixed_tf)
    else:                                                  # (a block of a multi-device call: its slice of the constellation's arrays)
        from .sharding import OutArrays
        oa = OutArrays(out)
        X, U, NU = oa.get("X", (S, 7, K)), oa.get("U", (S, 3, K)), oa.get("NU", (S, 7, K))
        kkt, status, iters = oa.get("kkt", (S,)), oa.get("status", (S,), np.int32), oa.get("iters", (S,), np.int32)
        tfo, held = oa.get("tf", (S,)), None
    lib = _ffi.load(); ctx = _ffi.context(device, slot)
    import ctypes as C
    dflags = _ffi.FLAG_J2 if include_J2 else 0
    if uniform_steps:                         # Discretizer.use_uniform_steps with integrator_steps = uniform_steps
        dflags |= _ffi.FLAG_UNIFORM_STEPS | (int(uniform_steps) << 8)
    if rk23:                                  # Discretizer.ivp_solver = 'RK23'
        dflags |= _ffi.FLAG_RK23
    if Ks is None:
        rc = lib.mpcx_mpc_step_batch(ctx, S, K, _ffi.dptr(xbar), _ffi.dptr(ubar), _ffi.dptr(tf), _ffi.dptr(consts),
                                     _ffi.dptr(r_des), dflags, float(max_step), C.byref(opts),
                                     _ffi.dptr(X), _ffi.dptr(U), _ffi.dptr(NU), _ffi.dptr(tfo), _ffi.iptr(status),
                                     _ffi.iptr(iters), _ffi.dptr(kkt))
    else:
        Ks = np.ascontiguousarray(np.broadcast_to(np.asarray(Ks), (S,)), dtype=np.int32)
        rc = lib.mpcx_mpc_step_batch_ragged(ctx, S, K, _ffi.iptr(Ks), _ffi.dptr(xbar), _ffi.dptr(ubar), _ffi.dptr(tf),
                                            _ffi.dptr(consts), _ffi.dptr(r_des), dflags, float(max_step), C.byref(opts),
                                            _ffi.dptr(X), _ffi.dptr(U), _ffi.dptr(NU), _ffi.dptr(tfo), _ffi.iptr(status),
                                            _ffi.iptr(iters), _ffi.dptr(kkt))
    _ffi.check(rc, ctx, "mpcx_mpc_step_batch")
    reg = _regularised(lib, ctx, S, None if out is None else out.get("regularised")) if regularised else None
    if out is not None:
        oa.finish()
    return SolveResult(X, U, NU, tfo, status, iters, kkt, regularised=reg) if held is None else \
        SolveResult(X, U, NU, held, status, iters, kkt, tfo, reg)


def scp_iteration_batch(y0, tf, consts, r_des, law, K, options=None, Ks=None, Kus=None, include_J2=False, max_step=1e-2,
                        prop_max_step=1e-3, device=0, slot=0, linear_vt=False, return_reference=False, **solver):
    """One SCP iteration of OptimalController.update (control.py:183-227) for S satellites in ONE library call
    (mpcx_scp_iteration_batch_ragged): the nonlinear rollout from y0 (S,7) over tf (S,) under `law` = (kind, vec, Ku, end_tau)
    (simulator.propagate_batch's law tuple) sampled at K nodes -- Ks[s] of them in a ragged batch --, its controller's thrust
    at those nodes (extract_uk), the discretisation about them and the solve.  The reference trajectory stays on the device
    unless return_reference=True (then the result carries .xbar (S,7,K) and .ubar (S,3,K)).  Returns a SolveResult with the
    extra attribute prop_status (S,)."""
    solver = _solver_flags(solver, linear_vt, None, False)
    y0 = _ffi.as_f64(y0); S = y0.shape[0]; K = int(K)
    tf = _ffi.as_f64(np.broadcast_to(np.asarray(tf, dtype=np.float64), (S,)))
    r_des = _ffi.as_f64(np.broadcast_to(np.asarray(r_des, dtype=np.float64), (S,)))
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
    if Ks is not None: Ks = np.ascontiguousarray(np.broadcast_to(np.asarray(Ks), (S,)), dtype=np.int32)
    if Kus is not None: Kus = np.ascontiguousarray(np.broadcast_to(np.asarray(Kus), (S,)), dtype=np.int32)
    opts = _ffi.make_solve_opts(options, **solver)
    X, U, NU, kkt, status, iters = _result_arrays(S, K, device, False)
    tfo = np.empty(S); pst = np.zeros(S, dtype=np.int32)
    xb = np.empty((S, 7, K)) if return_reference else None
    ub = np.empty((S, 3, K)) if return_reference else None
    lib = _ffi.load(); ctx = _ffi.context(device, slot)
    import ctypes as C
    rc = lib.mpcx_scp_iteration_batch_ragged(ctx, S, K, None if Ks is None else _ffi.iptr(Ks), _ffi.dptr(y0), _ffi.dptr(tf),
                                             _ffi.dptr(consts), _ffi.dptr(r_des), 0, kind, vec_p, int(Ku),
                                             None if Kus is None else _ffi.iptr(Kus), et_p, float(prop_max_step),
                                             _ffi.FLAG_J2 if include_J2 else 0, float(max_step), C.byref(opts),
                                             None if xb is None else _ffi.dptr(xb), None if ub is None else _ffi.dptr(ub),
                                             _ffi.dptr(X), _ffi.dptr(U), _ffi.dptr(NU), _ffi.dptr(tfo), _ffi.iptr(status),
                                             _ffi.iptr(iters), _ffi.dptr(kkt), _ffi.iptr(pst))
    _ffi.check(rc, ctx, "mpcx_scp_iteration_batch_ragged")
    res = SolveResult(X, U, NU, tfo, status, iters, kkt)
    res.prop_status = pst; res.xbar = xb; res.ubar = ub
    return res


class UpdateResult(SolveResult):
    """mpc_update_batch's result: the last SCP iteration's plan (rows of length K, Ks[s] columns in use), every iteration's
    status / iteration counts (n_scp, S), the rollouts' status and -- when the segment was flown -- its trajectory."""


def mpc_update_batch(y0, horizon, consts, r_des, base_res, n_scp=2, options=None, ref_thrust=0.5, include_J2=False, max_step=1e-2,
                     prop_max_step=1e-3, device=0, slot=0, linear_vt=False, fly=None, devices=None, out=None, **solver):
    """OptimalController.update (control.py:170-235) for S satellites in ONE library call (mpcx_mpc_update_batch): the tangential
    reference rollout over `horizon` sampled at K = int(base_res * horizon) nodes, n_scp x (extract_uk, discretise, solve) with
    the nonlinear re-rollout under the optimised sequence -- sampled at int(base_res * tf_u) nodes per satellite -- between
    two iterations; nothing but the final plan crosses PCIe.
    fly = (tf, interval, n_eval, include_drag, include_J2[, max_step]): also Simulator.run_segment's flight (simulator.py:58-65)
    of the plan over tf under the truth model, SequenceController(u_opt, tf_u, tf_sim = interval), from the same start states;
    the result then carries y_sim (S,7,n_eval) and sim_status.
    devices=[d0, d1, ...]: contiguous blocks of satellites on several devices at once (see mpc_step_batch)."""
    if devices is not None and len(devices) > 1:
        from .sharding import sharded_call
        y0 = _ffi.as_f64(y0); S = y0.shape[0]
        bc = lambda a: _ffi.as_f64(np.broadcast_to(np.asarray(a, dtype=np.float64), (S,)))
        hz = bc(horizon); K = int(base_res * float(hz[0]))
        res = _update_result(S, K, n_scp, int(devices[0]), fly)          # ONE result set; every block fills its satellites' part
        fn = lambda y, h, c, r, device, slot, out: mpc_update_batch(y, h, c, r, base_res, n_scp, options, ref_thrust, include_J2, max_step,
                                                                    prop_max_step, device, slot, linear_vt, fly, None, out, **solver)
        sharded_call(fn, devices, [y0, hz, _ffi.as_f64(consts), bc(r_des)],
                     dict(X=res.X, U=res.U, NU=res.NU, kkt=res.kkt, tf=res.tf, Ks=res.Ks, prop_status=res.prop_status,
                          status=(res.status, 1), iters=(res.iters, 1), y_sim=res.y_sim, sim_status=res.sim_status))
        return res
    if devices is not None and len(devices) == 1:
        device = int(devices[0])
    solver = _solver_flags(solver, linear_vt, None, False)
    y0 = _ffi.as_f64(y0); S = y0.shape[0]
    horizon = _ffi.as_f64(np.broadcast_to(np.asarray(horizon, dtype=np.float64), (S,)))
    K = int(base_res * float(horizon[0]))
    if not (horizon == horizon[0]).all():
        raise ValueError("mpc_update_batch: one horizon for the whole batch (the row length K = int(base_res * horizon))")
    r_des = _ffi.as_f64(np.broadcast_to(np.asarray(r_des, dtype=np.float64), (S,)))
    consts = _ffi.as_f64(consts)
    opts = _ffi.make_solve_opts(options, **solver)
    sim = (0.0, 0.0, 0, 0, 1e-3)
    if fly is not None:
        tf_sim, interval, n_eval, drag, j2 = fly[:5]
        sim = (float(tf_sim), float(interval), int(n_eval), (_ffi.FLAG_DRAG if drag else 0) | (_ffi.FLAG_J2 if j2 else 0),
               float(fly[5]) if len(fly) > 5 else 1e-3)
    if out is None:
        res = _update_result(S, K, n_scp, device, fly)
        X, U, NU, kkt, status, iters, tfo, Ks, pst, y_sim, sst = (res.X, res.U, res.NU, res.kkt, res.status, res.iters, res.tf, res.Ks,
                                                                  res.prop_status, res.y_sim, res.sim_status)
    else:                                                  # (a block of a multi-device call: its slice of the constellation's arrays)
        from .sharding import OutArrays
        oa = OutArrays(out)
        X, U, NU, kkt = oa.get("X", (S, 7, K)), oa.get("U", (S, 3, K)), oa.get("NU", (S, 7, K)), oa.get("kkt", (S,))
        status, iters = oa.get("status", (n_scp, S), np.int32), oa.get("iters", (n_scp, S), np.int32)
        tfo, Ks, pst = oa.get("tf", (S,)), oa.get("Ks", (S,), np.int32), oa.get("prop_status", (S,), np.int32)
        y_sim = oa.get("y_sim", (S, 7, sim[2])) if fly is not None else None
        sst = oa.get("sim_status", (S,), np.int32) if fly is not None else None
        res = UpdateResult(X, U, NU, tfo, status, iters, kkt)
        res.Ks = Ks; res.prop_status = pst; res.y_sim = y_sim; res.sim_status = sst
    lib = _ffi.load(); ctx = _ffi.context(device, slot)
    import ctypes as C
    rc = lib.mpcx_mpc_update_batch(ctx, S, K, int(n_scp), float(base_res), _ffi.dptr(y0), _ffi.dptr(horizon), _ffi.dptr(consts),
                                   _ffi.dptr(r_des), float(ref_thrust), float(prop_max_step), _ffi.FLAG_J2 if include_J2 else 0,
                                   float(max_step), C.byref(opts), _ffi.dptr(X), _ffi.dptr(U), _ffi.dptr(NU), _ffi.dptr(tfo),
                                   _ffi.iptr(Ks), _ffi.iptr(status), _ffi.iptr(iters), _ffi.dptr(kkt), _ffi.iptr(pst), sim[0], sim[1],
                                   sim[2], sim[3], sim[4], None if y_sim is None else _ffi.dptr(y_sim), None if sst is None else _ffi.iptr(sst))
    _ffi.check(rc, ctx, "mpcx_mpc_update_batch")
    if out is not None:
        oa.finish()
    return res


def _update_result(S, K, n_scp, device, fly):
    """the result set of an update of S satellites (large arrays from the recycling pool: _ffi.result_pool)"""
    X, U, NU, kkt, _, _ = _result_arrays(S, K, device, False)
    res = UpdateResult(X, U, NU, np.empty(S), np.zeros((n_scp, S), dtype=np.int32), np.zeros((n_scp, S), dtype=np.int32), kkt)
    res.Ks = np.zeros(S, dtype=np.int32); res.prop_status = np.zeros(S, dtype=np.int32)
    res.y_sim = _ffi.result_pool.take((S, 7, int(fly[2]))) if fly is not None else None
    res.sim_status = np.zeros(S, dtype=np.int32) if fly is not None else None
    return res


def solve_batch(A, Bp, Bn, Sigma, xi, xbar, ubar, tf, consts, r_des, options=None, device=0, linear_vt=False, fixed_tf=None,
                regularised=False, shared_tf=False, **solver):
    """Solve only (dynamics already discretised, reference-shaped arrays with a leading satellite axis)."""
    solver = _solver_flags(solver, linear_vt, fixed_tf, shared_tf)
    xbar = _ffi.as_f64(xbar); ubar = _ffi.as_f64(ubar)
    S, _, K = xbar.shape
    arrs = [_ffi.as_f64(a) for a in (A, Bp, Bn, Sigma, xi)]
    tf = _ffi.as_f64(np.broadcast_to(np.asarray(tf, dtype=np.float64), (S,)))
    r_des = _ffi.as_f64(np.broadcast_to(np.asarray(r_des, dtype=np.float64), (S,)))
    consts = _ffi.as_f64(consts)
    opts = _ffi.make_solve_opts(options, **solver)
    X = np.empty((S, 7, K)); U = np.empty((S, 3, K)); NU = np.empty((S, 7, K)); kkt = np.empty(S)
    tfo, held = _tf_io(S, fixed_tf)
    status = np.zeros(S, dtype=np.int32); iters = np.zeros(S, dtype=np.int32)
    lib = _ffi.load(); ctx = _ffi.context(device)
    import ctypes as C
    rc = lib.mpcx_solve_batch(ctx, S, K, *[_ffi.dptr(a) for a in arrs], _ffi.dptr(xbar), _ffi.dptr(ubar),
                              _ffi.dptr(tf), _ffi.dptr(consts), _ffi.dptr(r_des), C.byref(opts), _ffi.dptr(X),
                              _ffi.dptr(U), _ffi.dptr(NU), _ffi.dptr(tfo), _ffi.iptr(status), _ffi.iptr(iters),
                              _ffi.dptr(kkt))
    _ffi.check(rc, ctx, "mpcx_solve_batch")
    reg = _regularised(lib, ctx, S) if regularised else None
    return SolveResult(X, U, NU, tfo, status, iters, kkt, regularised=reg) if held is None else \
        SolveResult(X, U, NU, held, status, iters, kkt, tfo, reg)


class SharedTfSearch(list):
    """The (tf, G(tf)) evaluations of a shared-tf root search, with its outcome: `converged` False when no sign change
    of G was found within the evaluation budget (the returned tf is then the last point tried, not a root), `message`
    says why."""
    converged = True
    message = "root found"


def shared_tf_root(G, tf_max, tf0, gtol=1e-7, xtol=2e-8, max_bracket=40):
    """Root of the tf stationarity row G(tf) = 1 + sum_s g_s(tf) on (0, tf_max] (G increasing), or tf_max when
    G(tf_max) <= 0 (range constraint optimizer.py:588 active).  From the reference final time towards the root with doubling
    steps until the sign changes, then a bracketing secant (Illinois); every G is one batched device solve at fixed tf.
    Returns (tf, SharedTfSearch); the search's `converged` is False when G keeps one sign over the whole search (the
    downward search stops at 1e-6 tf_max: G > 0 all the way down means the problem wants tf -> 0, which 0 <= tf allows only
    in the limit)."""
    ev = SharedTfSearch()

    def g(t):
        v = G(t); ev.append((t, v)); return v

    def give_up(t, why):
        ev.converged = False; ev.message = why
        return t, ev
    a = min(tf0, tf_max); ga = g(a)
    if abs(ga) <= gtol or (a == tf_max and ga <= 0.0): return a, ev
    h = 0.05 * a
    t_floor = 1e-6 * tf_max
    while True:
        b = a - h if ga > 0.0 else a + h
        b = min(max(b, 0.05 * a), tf_max)                # (tf stays positive)
        gb = g(b)
        if abs(gb) <= gtol: return b, ev
        if (ga > 0.0) != (gb > 0.0): break
        if b == tf_max and gb <= 0.0: return tf_max, ev
        a, ga = b, gb; h *= 2.0
        if b <= t_floor:
            return give_up(b, f"G > 0 down to tf = {b:.3g}: no root on (0, tf_max]")
        if len(ev) > max_bracket:
            return give_up(b, f"no sign change of G in {len(ev)} evaluations (last tf {b:.6g}, G {gb:.3g})")
    lo, glo, hi, ghi = (a, ga, b, gb) if ga < 0.0 else (b, gb, a, ga)
    side = 0; t = 0.5 * (lo + hi)
    for _ in range(40):
        if hi - lo <= xtol: break
        t_prev = t
        t = (lo * ghi - hi * glo) / (ghi - glo)
        if abs(t - t_prev) <= 1e-9 * max(1.0, abs(t)): break       # G carries the noise of the inner multipliers: no finer root
        gt = g(t)
        if abs(gt) <= gtol: return t, ev
        if gt > 0.0:
            hi, ghi = t, gt
            if side == 1: glo *= 0.5
            side = 1
        else:
            lo, glo = t, gt
            if side == -1: ghi *= 0.5
            side = -1
    return t, ev


def solve_shared_tf(A, Bp, Bn, Sigma, xi, xbar, ubar, tf, consts, r_des, options=None, device=0, linear_vt=False, monolithic=True,
                    **solver):
    """S satellites that share ONE final time, as in a reference Optimizer holding several satellites
    (optimizer.py:287,311,322,336).
    monolithic (default): one device solve of the whole NLP (MPCX_SOLVE_SHARED_TF: one interior-point iteration for all
    satellites, the tf row assembled across them in a cooperative launch).
    monolithic=False: the decomposition of rounds 1-2 -- given tf the NLP separates into the S per-satellite problems the
    device solves in one batch (MPCX_SOLVE_FIXED_TF); what remains is the scalar row 1 + sum_s g_s(tf) = 0 (or tf on its
    bound), solved here by a bracketing secant (8-35 batched inner solves).  Kept as an independent cross-check.
    Returns (SolveResult, SharedTfSearch: the (tf, G(tf)) evaluations of the decomposition -- empty for the monolithic
    solve -- and whether a solution was found).  In the decomposition an inner solve that ends with a numeric breakdown
    raises; one whose constraint set is empty (MPCX_ST_INFEASIBLE: it is empty at every tf) ends the search at once, its
    result is returned with the status set and the search marked not converged; one that stops at max_iter is used as it
    is and recorded in the search's message -- the reference never raises from solve_OPT (optimizer.py:603 ignores
    ipopt's status)."""
    opts = {**DEFAULT_OPTIONS, **(options or {})}
    S = np.asarray(xbar).shape[0]
    if monolithic:
        r = solve_batch(A, Bp, Bn, Sigma, xi, xbar, ubar, tf, consts, r_des, options, device, linear_vt, shared_tf=True, **solver)
        ev = SharedTfSearch()
        ev.converged = bool(np.isin(r.status, (0, 7)).all())
        ev.message = "monolithic device solve" + ("" if ev.converged else f": status {sorted(set(int(c) for c in r.status))}")
        return r, ev

    class _Infeasible(Exception):
        pass
    notes = []

    def inner(t):
        return solve_batch(A, Bp, Bn, Sigma, xi, xbar, ubar, tf, consts, r_des, options, device, linear_vt,
                           fixed_tf=np.full(S, float(t)), **solver)

    def G(t):
        r = inner(t)
        if (r.status == 6).any():
            raise _ffi.MpcxError(f"shared-tf inner solve at tf = {t}: status {r.status.tolist()}")
        if (r.status == 8).any():
            raise _Infeasible(r)
        if not np.isin(r.status, (0, 7)).all():
            notes.append(f"inner solve at tf = {t:.6g} stopped at max_iter for {int((r.status == 5).sum())} satellite(s)")
        return 1.0 + float(np.sum(r.g_tf))
    try:
        tfs, ev = shared_tf_root(G, float(opts["tf_max"]), float(np.asarray(tf).reshape(-1)[0]))
    except _Infeasible as e:
        ev = SharedTfSearch(); ev.converged = False
        ev.message = "constraint set empty for at least one satellite (MPCX_ST_INFEASIBLE), at every tf"
        return e.args[0], ev
    if notes:
        ev.message += "; " + "; ".join(notes)
    return inner(tfs), ev


class Optimizer:
    def __init__(self, x_bar, u_bar, nu_bar, tf, d, f, scale, verbose=True, shared_tf=None):
        """Same arguments as the reference (optimizer.py:13-39).  With more than one satellite the
        reference couples all of them through a single tf variable (:287): that is the default here too
        (solve_shared_tf: the whole NLP as ONE cooperative device launch, MPCX_SOLVE_SHARED_TF);
        pass shared_tf=False to solve the satellites as independent problems (own tf each, one device call)."""
        self.x_bar, self.u_bar, self.nu_bar = x_bar, u_bar, nu_bar
        self.tf, self.d, self.f, self.scale = tf, d, f, scale
        self.const = scale.get_normalized_constants()
        self._N = len(x_bar)
        self._K = x_bar[0].shape[1]
        self.verbose = verbose
        self.shared_tf = (self._N > 1) if shared_tf is None else bool(shared_tf)
        self.result = None

    @staticmethod
    def skew(x):
        return np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]])

    @staticmethod
    def thrust_rtn(x, u):
        """u (3,K) in the radial / tangential / normal frame of the states x (7,K) -- what plot_normalized_thrust draws"""
        x = np.asarray(x, dtype=np.float64); u = np.asarray(u, dtype=np.float64)
        r, v = x[0:3], x[3:6]
        unit = lambda a: a / np.linalg.norm(a, axis=0)
        r_hat = unit(r); h_hat = unit(np.cross(r, v, axis=0)); t_hat = np.cross(h_hat, r_hat, axis=0)
        return np.stack([(b * u).sum(axis=0) for b in (r_hat, t_hat, h_hat)])

    @staticmethod
    def plot_normalized_thrust(x, u, show=True):
        """The reference's diagnostic plot (optimizer.py:47-77; its test calls it, test_optimizer.py:70): the thrust history in the
        RTN frame over normalised time.  matplotlib is imported here, not with the module (plotting is outside the hot path);
        returns the figure, show=False leaves it unshown."""
        import matplotlib.pyplot as plt
        u_rtn = Optimizer.thrust_rtn(x, u)
        print(f"u shape\n:{np.shape(u)}")
        fig, ax = plt.subplots()
        tau = np.linspace(0, 1, u_rtn.shape[1])
        for row, name in zip(u_rtn, "rtn"):
            ax.plot(tau, row, label=name)
        ax.set_title('Normalized Thrust Commands')
        ax.legend()
        if show:
            plt.show()
        return fig

    def init_options(self, options):
        return {**DEFAULT_OPTIONS, **options}

    def get_constraint_terms(self):
        """The dictionary of optimizer.py:80-170 (same keys, one list entry per satellite), computed for all satellites at once
        as array expressions on the stacked terminal states -- the Jacobians of the unit vectors r_hat, h_hat, t_hat as (N,3,3)
        stacks, incl. the expression form of Dv_h_hat as the reference evaluates it (:122: only h h^T / |h|^3 is multiplied by
        skew(r)).  Checked against dictionaries the reference itself produced (tests/golden ct_*, 1e-13).  The solver does not
        read this dictionary: the device builds the same terms (mpcx_constraint_terms, constraint_terms_batch above)."""
        N = self._N
        xK = np.stack([np.asarray(x, dtype=np.float64)[0:6, -1] for x in self.x_bar])                # (N,6) terminal r, v
        r, v = xK[:, 0:3], xK[:, 3:6]
        eye = np.eye(3)[None]

        def cross_matrix(a):                      # (N,3) -> (N,3,3), cross_matrix(a) @ b = a x b
            z = np.zeros(len(a))
            return np.stack([np.stack([z, -a[:, 2], a[:, 1]], -1), np.stack([a[:, 2], z, -a[:, 0]], -1),
                             np.stack([-a[:, 1], a[:, 0], z], -1)], -2)
        unit_jac = lambda a, n: eye / n[:, None, None] - a[:, :, None] * a[:, None, :] / n[:, None, None] ** 3      # d(a/|a|)/da
        rn = np.linalg.norm(r, axis=1); h = np.cross(r, v); hn = np.linalg.norm(h, axis=1)
        r_hat = r / rn[:, None]; h_hat = h / hn[:, None]; t_hat = np.cross(h_hat, r_hat)
        hh = h[:, :, None] * h[:, None, :] / hn[:, None, None] ** 3
        Dr_h = unit_jac(h, hn) @ (-cross_matrix(v))
        Dv_h = eye / hn[:, None, None] - hh @ cross_matrix(r)                                        # (as written, :122)
        Dr_r = unit_jac(r, rn)
        Dr_t = -cross_matrix(r_hat) @ Dr_h + cross_matrix(h_hat) @ Dr_r
        Dv_t = -cross_matrix(r_hat) @ Dv_h
        left = lambda M: np.einsum("ni,nij->nj", v, M)                                               # v^T M per satellite
        grads = {"t": np.concatenate([left(Dr_t), t_hat + left(Dv_t)], axis=1),
                 "r": np.concatenate([left(Dr_r), r_hat], axis=1),
                 "n": np.concatenate([left(Dr_h), h_hat + left(Dv_h)], axis=1)}
        mu = self.const.MU
        DrVc = (-0.5 * mu ** 0.5) * rn[:, None] ** (-5 / 2) * r
        per_sat = {'rf_hat': r_hat, 'Vc': np.sqrt(mu / rn), 'DrVc': DrVc, 'DrVc_rbar': np.einsum("ni,ni->n", DrVc, r),
                   'Vt': np.einsum("ni,ni->n", v, t_hat), 'DrVt_DvVt': grads["t"], 'DrVt_DvVt_bar': np.einsum("ni,ni->n", grads["t"], xK),
                   'Vr': np.einsum("ni,ni->n", v, r_hat), 'DrVr_DvVr': grads["r"], 'DrVr_DvVr_bar': np.einsum("ni,ni->n", grads["r"], xK),
                   'Vn': np.einsum("ni,ni->n", v, h_hat), 'DrVn_DvVn': grads["n"], 'DrVn_DvVn_bar': np.einsum("ni,ni->n", grads["n"], xK)}
        out = {k: [a[i] for i in range(N)] for k, a in per_sat.items()}
        # per-node unit vectors of the reference trajectory / thrust (:129-138; the mask of ubar_hat as the reference has it:
        # columns with |u| <= eps are the ones divided, the others stay zero)
        out['rbar_hat'] = []; out['ubar_hat'] = []
        for x, u in zip(self.x_bar, self.u_bar):
            rb = np.asarray(x, dtype=np.float64)[0:3, :-1]
            out['rbar_hat'].append(rb / np.linalg.norm(rb, axis=0))
            ub = np.asarray(u, dtype=np.float64); un = np.linalg.norm(ub, axis=0)
            uh = np.zeros(ub.shape); tiny = un <= np.finfo(float).eps
            with np.errstate(all='ignore'):
                uh[:, tiny] = ub[:, tiny] / un[tiny]
            out['ubar_hat'].append(uh)
        keys = ['rbar_hat', 'ubar_hat', 'rf_hat', 'Vc', 'DrVc', 'DrVc_rbar', 'Vt', 'DrVt_DvVt', 'DrVt_DvVt_bar',
                'Vr', 'DrVr_DvVr', 'DrVr_DvVr_bar', 'Vn', 'DrVn_DvVn', 'DrVn_DvVn_bar']                # (the reference's key order)
        return {k: out[k] for k in keys}

    def solve_OPT(self, input_options={}, **solver):
        """Transcribe-and-solve replacement (optimizer.py:219-613).  Extra keyword arguments are solver
        controls (tol, acceptable_tol, max_iter, acceptable_iter, n_refine)."""
        options = self.init_options(input_options)
        if getattr(self.f, "__name__", "") != "satellite_dynamics":
            raise NotImplementedError("only Simulator.satellite_dynamics is implemented on the device")
        self.d._check_modes()
        xbar = np.stack([np.asarray(x, dtype=np.float64) for x in self.x_bar])
        ubar = np.stack([np.asarray(u, dtype=np.float64) for u in self.u_bar])
        consts = np.tile(self.const.as_vector(), (self._N, 1))
        if self.shared_tf and self._N > 1:
            A, Bp, Bn, Sig, xi, dst = self.d.discretize_batch(xbar, ubar, self.tf, consts)       # optimizer.py:243-249
            if (dst != 0).any():
                raise RuntimeError(f"discretize failed: {[_ffi.STATUS_TEXT.get(int(c), c) for c in dst if c]}")
            self.result, self.tf_search = solve_shared_tf(A, Bp, Bn, Sig, xi, xbar, ubar, self.tf, consts, options['r_des'],
                                                          options, device=getattr(self.d, "device", 0), **solver)
        else:
            self.result = mpc_step_batch(xbar, ubar, self.tf, consts, options['r_des'], options,
                                         include_J2=self.d.include_J2, max_step=self.d.ivp_max_step,
                                         device=getattr(self.d, "device", 0),
                                         uniform_steps=int(self.d.integrator_steps) if self.d.use_uniform_steps else 0,
                                         rk23=(self.d.ivp_solver == 'RK23'), **solver)
        self.status = self.result.status
        if self.shared_tf and self._N > 1 and not self.tf_search.converged:
            import warnings
            warnings.warn(f"shared-tf search did not find a root: {self.tf_search.message}", RuntimeWarning, stacklevel=2)
        bad = [int(c) for c in self.result.status if c not in (0, 7)]
        if bad and self.verbose:
            print(f"WARNING: solve_OPT status {[_ffi.STATUS_TEXT.get(c, c) for c in bad]}")
        return None

    def get_solved_trajectory(self, s):
        return self.result.X[s].copy()

    def get_solved_tf(self, s):
        """(the reference ignores s, :199-203: one tf for all; so does the shared mode, where every entry is the same)"""
        return float(self.result.tf[s])

    def get_solved_u(self, s):
        return self.result.U[s].copy()

    def get_solved_nu(self, s):
        return self.result.NU[s].copy()
