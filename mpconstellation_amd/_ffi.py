"""ctypes binding of libmpcx.so (include/mpcx.h).  No CPU fallback: if the HIP library or a
gfx950 device is missing, every compute entry point raises."""
import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPCX_LIB") or os.path.join(HERE, "libmpcx.so")      # (MPCX_LIB: another build of the same library, for A/B measurements)

STATUS_TEXT = {
    0: "ok", 1: "satellite mass <= 0", 2: "RK45 step size underflow",
    3: "FOH index outside the input table", 4: "state-transition matrix singular",
    5: "solver hit max_iter", 6: "solver numeric breakdown", 7: "solver stopped at acceptable level",
    9: "ragged batch: node / table-column / output-point count outside the accepted range",
    8: "constraint set empty (start node outside its radius bounds, terminal window outside r_max, empty window or tf range)",
    10: "time-parallel solve: a workgroup of the satellite did not answer within the wait limit (device shared with another long kernel?)",
}
FLAG_DRAG, FLAG_J2, FLAG_UNIFORM_STEPS, FLAG_RK23 = 1, 2, 4, 8
CTRL_ZERO, CTRL_CONSTANT, CTRL_TANGENTIAL, CTRL_SEQUENCE = 0, 1, 2, 3
NCONST = 8
STAGE_DOUBLES = 105
NTERM_SCALARS = 8

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p


class MpcxError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()
_ctxs = {}

class SolveOpts(C.Structure):
    """mpcx_solve_opts (include/mpcx.h)"""
    _fields_ = [(n, C.c_double) for n in ("min_mass", "u_max", "r_min", "r_max", "eps_r", "eps_vr", "eps_vn", "eps_vt",
                                          "tf_max", "w_nu", "w_tr", "tol", "acceptable_tol")] + \
               [(n, C.c_int32) for n in ("max_iter", "acceptable_iter", "n_refine", "flags")]


_po = C.POINTER(SolveOpts)

_SIGS = {
    "mpcx_version": (C.c_int, []),
    "mpcx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "mpcx_destroy": (None, [_vp]),
    "mpcx_last_error": (C.c_char_p, [_vp]),
    "mpcx_synchronize": (C.c_int, [_vp, _vp]),
    "mpcx_set_stream": (C.c_int, [_vp, _vp]),
    "mpcx_trace_enable": (C.c_int, [_vp, C.c_int]),
    "mpcx_last_call_trace": (C.c_int, [_vp, _dp, C.c_int]),
    "mpcx_host_alloc": (_vp, [_vp, C.c_size_t]),
    "mpcx_host_free": (None, [_vp, _vp]),
    "mpcx_discretize_batch": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int,
                                        C.c_double, _dp, _dp, _dp, _dp, _dp, _ip]),
    "mpcx_discretize_batch_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp,
                                            C.c_int, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mpcx_discretize_stages_dev": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp,
                                             C.c_int, C.c_double, _vp, _vp, _vp]),
    "mpcx_propagate_batch": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, _dp, C.c_int, _dp,
                                       C.c_double, _dp, _ip, _ip]),
    "mpcx_propagate_batch_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp,
                                           C.c_double, _vp, _vp, _vp, _vp]),
    "mpcx_default_solve_opts": (None, [_po]),
    "mpcx_solve_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mpcx_mpc_step_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mpcx_solve_workspace_bytes_ctx": (C.c_size_t, [_vp, C.c_int, C.c_int]),
    "mpcx_mpc_step_workspace_bytes_ctx": (C.c_size_t, [_vp, C.c_int, C.c_int]),
    "mpcx_solve_batch": (C.c_int, [_vp, C.c_int, C.c_int] + [_dp] * 10 + [_po, _dp, _dp, _dp, _dp, _ip, _ip, _dp]),
    "mpcx_solve_regularised": (C.c_int, [_vp, C.c_int, _ip]),
    "mpcx_solve_regularised_dev": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "mpcx_constraint_terms": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _po, _dp, _dp, _dp]),
    "mpcx_constraint_terms_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _po, _vp, _vp, _vp, _vp]),
    "mpcx_solve_batch_dev": (C.c_int, [_vp, C.c_int, C.c_int] + [_vp] * 6 + [_po] + [_vp] * 7 + [_vp, _vp]),
    "mpcx_mpc_step_batch": (C.c_int, [_vp, C.c_int, C.c_int] + [_dp] * 5 + [C.c_int, C.c_double, _po, _dp, _dp, _dp, _dp,
                                                                         _ip, _ip, _dp]),
    "mpcx_mpc_step_batch_dev": (C.c_int, [_vp, C.c_int, C.c_int] + [_vp] * 5 + [C.c_int, C.c_double, _po] + [_vp] * 7
                                + [_vp, _vp]),
    # ragged batches (per-satellite node counts)
    "mpcx_discretize_stages_ragged_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp,
                                                    C.c_int, C.c_double, _vp, _vp, _vp]),
    "mpcx_solve_batch_ragged_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp] + [_vp] * 6 + [_po] + [_vp] * 7 + [_vp, _vp]),
    "mpcx_mpc_step_batch_ragged": (C.c_int, [_vp, C.c_int, C.c_int, _ip] + [_dp] * 5 + [C.c_int, C.c_double, _po, _dp, _dp, _dp,
                                                                                     _dp, _ip, _ip, _dp]),
    "mpcx_mpc_step_batch_ragged_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp] + [_vp] * 5 + [C.c_int, C.c_double, _po] + [_vp] * 7
                                       + [_vp, _vp]),
    "mpcx_propagate_batch_ragged": (C.c_int, [_vp, C.c_int, C.c_int, _ip, _dp, _dp, _dp, C.c_int, C.c_int, _dp, C.c_int, _ip,
                                              _dp, C.c_double, _dp, _ip, _ip]),
    "mpcx_propagate_batch_ragged_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int,
                                                  _vp, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "mpcx_propagate_thrust_batch_ragged": (C.c_int, [_vp, C.c_int, C.c_int, _ip, _dp, _dp, _dp, C.c_int, C.c_int, _dp, C.c_int, _ip,
                                                     _dp, C.c_double, _dp, _dp, _ip, _ip]),
    "mpcx_propagate_thrust_batch_ragged_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int,
                                                         _vp, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "mpcx_scp_iteration_batch_ragged": (C.c_int, [_vp, C.c_int, C.c_int, _ip, _dp, _dp, _dp, _dp, C.c_int, C.c_int, _dp, C.c_int, _ip, _dp,
                                                  C.c_double, C.c_int, C.c_double, _po, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _ip]),
    "mpcx_mpc_update_batch": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_int,
                                        C.c_double, _po, _dp, _dp, _dp, _dp, _ip, _ip, _ip, _dp, _ip, C.c_double, C.c_double, C.c_int, C.c_int,
                                        C.c_double, _dp, _ip]),
    "mpcx_resample_sequence_dev": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
}


def exported_symbols():
    return list(_SIGS)


def load():
    """dlopen libmpcx.so and declare signatures.  Raises MpcxError if it is not built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise MpcxError(f"{LIB_PATH} is missing: run `python -m mpconstellation_amd.build` "
                                "(hipcc, gfx950). There is no CPU fallback.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in _SIGS.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def context(device=0, slot=0):
    """One mpcx context per (process, device, slot).  Calls on different contexts are thread-safe and run on their own
    streams (include/mpcx.h): host threads that want to overlap small launches take different slots."""
    lib = load()
    device = (device, slot) if slot else device
    with _lock:
        if device not in _ctxs:
            h = _vp()
            rc = lib.mpcx_create(device[0] if isinstance(device, tuple) else device, C.byref(h))
            if rc != 0:
                raise MpcxError(f"mpcx_create(device={device}) failed ({rc}): "
                                f"{lib.mpcx_last_error(None).decode()}")
            _ctxs[device] = h
        return _ctxs[device]


STREAM_PRIVATE = C.c_void_p(-1)


def set_stream(stream, device=0, slot=0):
    """Run the host-pointer entry points of this context on `stream` (a hipStream_t as an integer, e.g.
    torch.cuda.current_stream().cuda_stream; None / 0: the device's default stream; STREAM_PRIVATE: a private stream again).
    For processes that also drive the device through another stream: include/mpcx.h, mpcx_set_stream."""
    lib = load(); ctx = context(device, slot)
    check(lib.mpcx_set_stream(ctx, stream if isinstance(stream, C.c_void_p) else C.c_void_p(stream or 0)), ctx, "mpcx_set_stream")


TRACE_FIELDS = ("wall_ms", "first_marker_ms", "host_stage_ms", "host_wait_ms", "host_copyout_ms", "dev_span_ms", "dev_kernels_ms", "valid")


def trace_enable(on=True, device=0, slot=0):
    """every following host-pointer call of this context records where its time went (include/mpcx.h, mpcx_trace_enable)"""
    lib = load(); ctx = context(device, slot)
    check(lib.mpcx_trace_enable(ctx, 1 if on else 0), ctx, "mpcx_trace_enable")


def last_call_trace(device=0, slot=0):
    """the record of the context's last traced host-pointer call as a dict (TRACE_FIELDS), or None if there is none"""
    lib = load(); ctx = context(device, slot)
    out = np.zeros(len(TRACE_FIELDS))
    check(lib.mpcx_last_call_trace(ctx, dptr(out), len(out)), ctx, "mpcx_last_call_trace")
    return dict(zip(TRACE_FIELDS, out.tolist())) if out[-1] else None


class _PinnedOwner:
    """keeps a page-locked allocation alive as long as a numpy array views it"""

    def __init__(self, ctx, ptr, nbytes):
        self.ctx, self.ptr = ctx, ptr
        self.buf = (C.c_char * nbytes).from_address(ptr)

    def __del__(self):
        try:
            load().mpcx_host_free(self.ctx, self.ptr)
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64, device=0):
    """numpy array in page-locked host memory (mpcx_host_alloc): the host-pointer entry points transfer such arrays by DMA
    without a staging copy.  Use it for arrays handed to mpc_step_batch / solve_batch repeatedly."""
    lib = load(); ctx = context(device)
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    ptr = lib.mpcx_host_alloc(ctx, max(n, 1))
    if not ptr:
        raise MpcxError(f"mpcx_host_alloc({n}) failed: {lib.mpcx_last_error(ctx).decode()}")
    owner = _PinnedOwner(ctx, ptr, max(n, 1))
    owner.buf._mpcx_owner = owner            # the array keeps buf alive, buf keeps the allocation's owner alive
    arr = np.frombuffer(owner.buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    return arr


def pinned_copy(a, device=0):
    a = np.asarray(a)
    out = pinned_empty(a.shape, a.dtype, device)
    out[...] = a
    return out


class _ResultPool:
    """Result arrays of the batched wrappers, recycled.  A large numpy array is a fresh anonymous mapping: the first write to
    each of its pages is a page fault, and for the 17 MB a 4096-satellite step returns those faults are 1.8 ms of a 8.8 ms call on
    average and 3-5 ms now and then -- every call, because numpy unmaps the arrays of the previous results
    when the caller drops them.  take() hands out an array of the pool when NOBODY else holds a reference to it any more (the
    previous results have been dropped: reference count of the pooled object = the pool's own) and a new one otherwise, so a
    caller that keeps its results keeps them untouched.  A few arrays per shape, a few shapes."""
    PER_SHAPE, SHAPES = 3, 24

    def __init__(self):
        self._lock = threading.Lock()
        self._pool = {}          # (shape, dtype) -> [arrays]

    def take(self, shape, dtype=np.float64):
        import sys
        key = (tuple(int(n) for n in shape), np.dtype(dtype).str)
        nbytes = int(np.prod(key[0])) * np.dtype(dtype).itemsize
        if nbytes < (1 << 20):                                   # small arrays come from malloc's own free lists: nothing to gain
            return np.empty(shape, dtype=dtype)
        with self._lock:
            lst = self._pool.get(key)
            if lst is None:
                if len(self._pool) >= self.SHAPES:
                    self._pool.pop(next(iter(self._pool)))
                lst = self._pool[key] = []
            else:
                self._pool[key] = self._pool.pop(key)            # (most recently used last)
            for i in range(len(lst)):
                if sys.getrefcount(lst[i]) == 2:                 # the list's reference and getrefcount's argument: nobody else
                    return lst[i]
            a = np.empty(shape, dtype=dtype)
            if len(lst) < self.PER_SHAPE:
                lst.append(a)
            return a


result_pool = _ResultPool()


def check(rc, ctx, what):
    if rc != 0:
        raise MpcxError(f"{what} failed ({rc}): {load().mpcx_last_error(ctx).decode()}")


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


SOLVER_KEYWORDS = ("tol", "acceptable_tol", "max_iter", "acceptable_iter", "n_refine", "flags")
SOLVE_INDEX_ORDER, SOLVE_LINEAR_VT, SOLVE_FIXED_TF, SOLVE_SHARED_TF, SOLVE_ONE_WAVE, SOLVE_NO_LDS, SOLVE_TIME_PARALLEL = 1, 2, 4, 8, 16, 32, 64      # mpcx_solve_opts.flags (include/mpcx.h)
SOLVE_TP_SELFTEST_DEAD = 1 << 30


def check_solver_keywords(solver):
    bad = sorted(set(solver) - set(SOLVER_KEYWORDS))
    if bad:
        raise TypeError(f"unknown solver option(s) {bad}; known: {list(SOLVER_KEYWORDS)}")


# reference option keys (optimizer.py:178-188) -> mpcx_solve_opts
def make_solve_opts(options=None, **solver):
    """options: dict with the reference's keys (min_mass, u_lim, r_lim, eps_r, eps_vr, eps_vn, tf_max, w_nu, w_tr;
    r_des is passed per satellite); solver: tol, acceptable_tol, max_iter, acceptable_iter, n_refine."""
    o = SolveOpts()
    load().mpcx_default_solve_opts(C.byref(o))
    options = options or {}
    if "min_mass" in options: o.min_mass = options["min_mass"]
    if "u_lim" in options: o.u_max = options["u_lim"][1]
    if "r_lim" in options: o.r_min, o.r_max = options["r_lim"][0], options["r_lim"][1]
    for k in ("eps_r", "eps_vr", "eps_vn", "eps_vt", "tf_max", "w_nu", "w_tr"):
        if k in options: setattr(o, k, options[k])
    check_solver_keywords(solver)
    for k, v in solver.items():
        setattr(o, k, v)
    return o
