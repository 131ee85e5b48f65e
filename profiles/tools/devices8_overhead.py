"""Host time of a multi-device call OUTSIDE the library (north star: satellites shard across the GPUs of a node behind the drop-in
API).  mpc_step_batch(..., devices=[0] * 8) at 65 536 satellites x 30 nodes (BASELINE configs[4]) on the one GPU of the box: eight
contexts, eight host threads, the device calls serialised by the one device.  What can be measured here is what the HOST adds
around them: wall time of the Python call minus the span of the library calls (first entry to last return, taken inside ctypes
wrappers of the entry point) -- allocation of result arrays, slicing, and (until round 4) the np.concatenate of the blocks'
results.  "Unmeasured on hardware" stays true of the eight-device execution itself.
usage: python profiles/tools/devices8_overhead.py [package_parent_dir]   (another copy of the Python package, e.g. the round-4 one
under profiles/tools/_ab/pkg_r04, against the in-tree library: MPCX_LIB)"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg_parent = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else ROOT
os.environ.setdefault("MPCX_LIB", os.path.join(ROOT, "mpconstellation_amd", "libmpcx.so"))
sys.path.insert(0, pkg_parent)
sys.path.insert(1, os.path.join(ROOT, "tests"))
import mpconstellation_amd as M                                   # noqa: E402
from mpconstellation_amd import _ffi, mpc_step_batch              # noqa: E402
assert os.path.dirname(os.path.dirname(os.path.abspath(M.__file__))) == pkg_parent

S, K, NDEV = int(os.environ.get("S", 65536)), 30, 8
from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust      # noqa: E402
from mpconstellation_amd.simulator import propagate_batch         # noqa: E402
y0, consts = normalize_batch(constellation_states(S))
xbar = np.empty((S, 7, K))
for b in range(0, S, 8192):                                        # (set-up rollouts in blocks)
    xbar[b:b + 8192] = propagate_batch(y0[b:b + 8192], np.ones(min(8192, S - b)), consts[b:b + 8192], (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), K)[0]
ubar = np.ascontiguousarray(tangential_thrust(xbar, 0.5))
r_des = np.linalg.norm(xbar[:, :3, -1], axis=1); tf = np.ones(S)

lib = _ffi.load()
spans = []; lock = threading.Lock()
inner = lib.mpcx_mpc_step_batch


def timed(*a):
    t0 = time.perf_counter(); rc = inner(*a); t1 = time.perf_counter()
    with lock: spans.append((t0, t1))
    return rc


lib.mpcx_mpc_step_batch = timed
rows = []
res = None
for it in range(int(os.environ.get("CALLS", 8))):
    del res                                                         # (the caller drops the previous results, as a loop would)
    spans.clear()
    t0 = time.perf_counter()
    res = mpc_step_batch(xbar, ubar, tf, consts, r_des, devices=[0] * NDEV)
    t1 = time.perf_counter()
    lo, hi = min(s[0] for s in spans), max(s[1] for s in spans)
    rows.append((1e3 * (t1 - t0), 1e3 * (hi - lo), 1e3 * (lo - t0), 1e3 * (t1 - hi)))
assert (res.status == 0).all() and len(spans) == NDEV
one = mpc_step_batch(xbar[:8192], ubar[:8192], tf[:8192], consts[:8192], r_des[:8192])
assert np.array_equal(one.X, res.X[:8192]) and np.array_equal(one.tf, res.tf[:8192])       # bit-equal to the single-device call of the first block
print(f"package {pkg_parent}: S = {S}, K = {K}, devices = [0] * {NDEV}; per call: wall, library span, before the first library call, after the last (ms)")
for r in rows: print("  %8.2f %8.2f %7.2f %7.2f" % r)
tail = rows[2:]
print("host time outside the library calls, mean of the calls after the second: before %.2f ms + after %.2f ms = %.2f ms of %.1f ms" %
      (np.mean([r[2] for r in tail]), np.mean([r[3] for r in tail]), np.mean([r[2] + r[3] for r in tail]), np.mean([r[0] for r in tail])))
