"""Where a host-pointer call (mpcx_mpc_step_batch, numpy in / numpy out) spends its time at S = 4096, K = 30:
the C call alone with result arrays that are (a) page-locked, (b) ordinary and already touched, (c) freshly allocated
for every call (what mpc_step_batch does), against the device-resident step."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first HIP runtime in the process, as in bench.py)
import bench
from mpconstellation_amd import _ffi, mpc_step_batch

S, K = 4096, 30
run = bench.Runner("S4096_K30", 0, 1, 0, n_variants=1)
h = run.host
lib, ctx = _ffi.load(), _ffi.context(0)
opts = _ffi.make_solve_opts({})


def call(x, u, tf, c, rd, X, U, NU, tfo, st, it, kk):
    rc = lib.mpcx_mpc_step_batch(ctx, S, K, _ffi.dptr(x), _ffi.dptr(u), _ffi.dptr(tf), _ffi.dptr(c), _ffi.dptr(rd), 0, 1e-2,
                                 C.byref(opts), _ffi.dptr(X), _ffi.dptr(U), _ffi.dptr(NU), _ffi.dptr(tfo), _ffi.iptr(st), _ffi.iptr(it),
                                 _ffi.dptr(kk))
    assert rc == 0


def outs(pinned=False):
    mk = (lambda s, d=np.float64: _ffi.pinned_empty(s, d)) if pinned else (lambda s, d=np.float64: np.empty(s, d))
    return [mk((S, 7, K)), mk((S, 3, K)), mk((S, 7, K)), mk((S,)), mk((S,), np.int32), mk((S,), np.int32), mk((S,))]


def timeit(f, n=5):
    f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3


ins = [h[k] for k in ("xbar", "ubar", "tfbar", "consts", "r_des")]
pins = [_ffi.pinned_copy(a) for a in ins]
o_pin = outs(True); o_page = outs(False)
for a in o_page: a[...] = 0
print(f"device-resident step (bench.Runner.step + sync)   {timeit(lambda: (run.step(), torch.cuda.synchronize())):8.3f} ms")
print(f"C call, inputs + results page-locked              {timeit(lambda: call(*pins, *o_pin)):8.3f} ms")
print(f"C call, pageable inputs, page-locked results      {timeit(lambda: call(*ins, *o_pin)):8.3f} ms")
print(f"C call, pageable inputs, touched pageable results {timeit(lambda: call(*ins, *o_page)):8.3f} ms")
print(f"C call, pageable inputs, FRESH pageable results   {timeit(lambda: call(*ins, *outs())):8.3f} ms")
print(f"mpc_step_batch (Python wrapper, fresh results)    {timeit(lambda: mpc_step_batch(*ins)):8.3f} ms")
print(f"mpc_step_batch(pinned inputs, pinned_results)     {timeit(lambda: mpc_step_batch(*pins, pinned_results=True)):8.3f} ms")
keep = []
print(f"mpc_step_batch keeping every result alive         {timeit(lambda: keep.append(mpc_step_batch(*ins))):8.3f} ms")
