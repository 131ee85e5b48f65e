"""The C-ABI library: it loads, exports every symbol include/mpcx.h declares, the ctypes binding covers
them all, and on a machine without an MI355X the product fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mpcx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mpcx_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from mpconstellation_amd import _ffi, build
    build.build()                       # hipcc cross-compiles gfx950 without a GPU
    lib = C.CDLL(_ffi.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mpcx.h but not exported"
    assert set(_ffi.exported_symbols()) == set(names)     # the Python binding binds exactly the header
    assert _ffi.load().mpcx_version() == 500


def test_struct_layout_and_defaults():
    from mpconstellation_amd import _ffi
    o = _ffi.make_solve_opts({"u_lim": [0, 3.0], "r_lim": [0.95, 4.0], "eps_r": 1e-3, "tf_max": 2.0}, max_iter=50)
    assert C.sizeof(_ffi.SolveOpts) == 13 * 8 + 4 * 4
    assert (o.u_max, o.r_min, o.r_max, o.eps_r, o.tf_max, o.max_iter) == (3.0, 0.95, 4.0, 1e-3, 2.0, 50)
    assert (o.min_mass, o.eps_vr, o.eps_vn, o.eps_vt, o.w_nu, o.w_tr, o.tol) == (0.1, 1e-5, 1e-5, 1e-5, 1000.0, 0.002, 1e-8)   # optimizer.py:178-188
    lib = _ffi.load()
    assert lib.mpcx_mpc_step_workspace_bytes(64, 30) > lib.mpcx_solve_workspace_bytes(64, 30) > 0


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback():
    from mpconstellation_amd import _ffi, Discretizer, mpc_step_batch
    lib = _ffi.load()
    h = C.c_void_p()
    assert lib.mpcx_create(0, C.byref(h)) == -1            # MPCX_E_NODEVICE
    assert b"no CPU fallback" in lib.mpcx_last_error(None)
    x = np.zeros((1, 7, 5)); u = np.zeros((1, 3, 5))
    with pytest.raises(_ffi.MpcxError):
        mpc_step_batch(x, u, [1.0], np.zeros((1, 8)), [1.0])

    class Cst:
        def as_vector(self): return np.zeros(8)
    def satellite_dynamics(): pass
    with pytest.raises(_ffi.MpcxError):
        Discretizer(Cst()).discretize(satellite_dynamics, x[0], u[0], 1.0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mpconstellation_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle_lib" not in txt and "nlp_ipm" not in txt and "liboracle" not in txt, f
