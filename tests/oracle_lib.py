"""ctypes binding of oracle/liboracle.so (the CPU restatement; test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

FLAG_DRAG, FLAG_J2, FLAG_RK23 = 1, 2, 8      # (FLAG_RK23: Discretizer.ivp_solver = 'RK23' in oracle_discretize*)
CTRL_ZERO, CTRL_CONSTANT, CTRL_TANGENTIAL, CTRL_SEQUENCE = 0, 1, 2, 3
CT_NTERMS = 32
CT_SLICES = {
    "rf_hat": slice(0, 3), "Vc": 3, "DrVc": slice(4, 7), "DrVc_rbar": 7, "Vt": 8,
    "DrVt_DvVt": slice(9, 15), "DrVt_DvVt_bar": 15, "Vr": 16, "DrVr_DvVr": slice(17, 23),
    "DrVr_DvVr_bar": 23, "Vn": 24, "DrVn_DvVn": slice(25, 31), "DrVn_DvVn_bar": 31,
}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class OracleCtrl(C.Structure):
    _fields_ = [("kind", C.c_int), ("thrust", C.c_double * 3), ("useq", _dp), ("Ku", C.c_int),
                ("end_tau", C.c_double)]


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
            for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        # (ORACLE_LIB: another build of the same sources, e.g. the -fsanitize=address,undefined one of `make -C oracle asan`)
        _lib = C.CDLL(os.environ.get("ORACLE_LIB") or build())
        _lib.oracle_dynamics.restype = C.c_int
        _lib.oracle_u_foh.restype = C.c_int
        _lib.oracle_discretize.restype = C.c_int
        _lib.oracle_propagate.restype = C.c_int
        _lib.oracle_u_foh.argtypes = [C.c_double, _dp, C.c_int, _dp]
        _lib.oracle_dynamics.argtypes = [_dp, _dp, C.c_double, _dp, C.c_int, _dp]
        _lib.oracle_A_func.argtypes = [_dp, _dp, C.c_double, _dp, C.c_int, _dp]
        _lib.oracle_B_func.argtypes = [_dp, _dp, C.c_double, _dp, _dp]
        _lib.oracle_xi_func.argtypes = [_dp, _dp, C.c_double, _dp, C.c_int, _dp]
        _lib.oracle_discretize.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, _dp, C.c_int,
                                           C.c_double, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp,
                                           C.c_int]
        _lib.oracle_discretize_mode.restype = C.c_int
        _lib.oracle_discretize_mode.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, _dp, C.c_int,
                                                C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp,
                                                C.c_int]
        _lib.oracle_propagate.argtypes = [_dp, C.c_double, _dp, C.c_int, C.POINTER(OracleCtrl),
                                          C.c_int, C.c_double, _dp, _ip]
        _lib.oracle_extract_uk.argtypes = [C.c_int, _dp, _dp, C.POINTER(OracleCtrl), _dp]
        _lib.oracle_constraint_terms.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp, _dp, _dp]
        _lib.oracle_scale.argtypes = [_dp, _dp, _dp]
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dynamics(y, u, tf, cst, flags=0):
    y, u, cst = _c(y), _c(u), _c(cst)
    out = np.zeros(7)
    rc = lib().oracle_dynamics(_p(y), _p(u), tf, _p(cst), flags, _p(out))
    return out, rc


def A_func(x, u, tf, cst, flags=0):
    x, u, cst = _c(x), _c(u), _c(cst)
    out = np.zeros((7, 7))
    lib().oracle_A_func(_p(x), _p(u), tf, _p(cst), flags, _p(out))
    return out


def B_func(x, u, tf, cst):
    x, u, cst = _c(x), _c(u), _c(cst)
    out = np.zeros((7, 3))
    lib().oracle_B_func(_p(x), _p(u), tf, _p(cst), _p(out))
    return out


def xi_func(x, u, tf, cst, flags=0):
    x, u, cst = _c(x), _c(u), _c(cst)
    out = np.zeros(7)
    lib().oracle_xi_func(_p(x), _p(u), tf, _p(cst), flags, _p(out))
    return out


def u_foh(tau, u):
    u = _c(u)
    out = np.zeros(3)
    rc = lib().oracle_u_foh(float(tau), _p(u), u.shape[1], _p(out))
    return out, rc


def discretize(x, u, tf, cst, flags=0, max_step=1e-2, dump_nodes=False, uniform_steps=0):
    """uniform_steps = integrator_steps of Discretizer.use_uniform_steps (0: the default adaptive quadrature nodes)"""
    x, u, cst = _c(x), _c(u), _c(cst)
    K, Ku = x.shape[1], u.shape[1]
    A = np.zeros((K - 1, 7, 7)); Bp = np.zeros((K - 1, 7, 3)); Bn = np.zeros((K - 1, 7, 3))
    Sig = np.zeros((7, K - 1)); xi = np.zeros((7, K - 1))
    cnt = np.zeros(K - 1, dtype=np.int32); nfev = np.zeros(K - 1, dtype=np.int32)
    cap = max(64, int(uniform_steps) + 1) * (K - 1) if dump_nodes else 0
    nt = np.zeros(max(cap, 1)); ny = np.zeros((max(cap, 1), 56))
    rc = lib().oracle_discretize_mode(K, Ku, _p(x), _p(u), float(tf), _p(cst), flags, max_step, int(uniform_steps), _p(A),
                                 _p(Bp), _p(Bn), _p(Sig), _p(xi), cnt.ctypes.data_as(_ip),
                                 nfev.ctypes.data_as(_ip), _p(nt) if dump_nodes else None,
                                 _p(ny) if dump_nodes else None, cap)
    out = dict(A=A, Bp=Bp, Bn=Bn, Sigma=Sig, xi=xi, node_counts=cnt, node_nfev=nfev, status=rc)
    if dump_nodes:
        n = int(cnt.sum())
        out["node_t"] = nt[:n]; out["node_y"] = ny[:n]
    return out


def make_ctrl(kind, thrust=(0.0, 0.0, 0.0), useq=None, end_tau=1.0):
    c = OracleCtrl()
    c.kind = kind
    for i in range(3):
        c.thrust[i] = float(thrust[i])
    keep = None
    if useq is not None:
        keep = _c(useq)
        c.useq = _p(keep); c.Ku = keep.shape[1]
    c.end_tau = float(end_tau)
    c._keep = keep
    return c


def propagate(y0, tf, cst, ctrl, n_eval, flags=0, max_step=1e-3):
    y0, cst = _c(y0), _c(cst)
    out = np.zeros((7, n_eval))
    ns = np.zeros(1, dtype=np.int32)
    rc = lib().oracle_propagate(_p(y0), float(tf), _p(cst), flags, C.byref(ctrl), n_eval, max_step,
                                _p(out), ns.ctypes.data_as(_ip))
    return out, rc, int(ns[0])


def extract_uk(x, t, ctrl):
    x, t = _c(x), _c(t)
    K = x.shape[1]
    u = np.zeros((3, K))
    lib().oracle_extract_uk(K, _p(x), _p(t), C.byref(ctrl), _p(u))
    return u


def constraint_terms(x, u, mu):
    x, u = _c(x), _c(u)
    K = x.shape[1]
    rbar = np.zeros((3, K - 1)); ubar = np.zeros((3, K)); T = np.zeros(CT_NTERMS)
    with np.errstate(all="ignore"):
        lib().oracle_constraint_terms(K, _p(x), _p(u), float(mu), _p(rbar), _p(ubar), _p(T))
    out = {"rbar_hat": rbar, "ubar_hat": ubar}
    for k, s in CT_SLICES.items():
        out[k] = T[s].copy() if isinstance(s, slice) else T[s]
    return out


def scale(state):
    state = _c(state)
    sc = np.zeros(7); cst = np.zeros(8)
    lib().oracle_scale(_p(state), _p(sc), _p(cst))
    return sc, cst
