"""profiling helper: find the occasional slow host-pointer call (BENCH_r03: one 53 ms call among 1.5 ms ones at 64 satellites;
40-50 ms for the second call of a process at 4096).  N consecutive mpc_step_batch calls per variant (pageable / page-locked
arrays), each timed around the whole Python wrapper and around the ctypes call alone, with Python's garbage collector
observed (gc.callbacks) -- and MPCX_HOST_TRACE=<ms> makes the library report where a slow call spent its time.
usage: MPCX_HOST_TRACE=5 python profiles/tools/host_stall_probe.py [S ...]"""
import gc, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch                                   # (bench.py's process has torch loaded: same heap of Python objects)
from test_full_size_gpu import workload
from mpconstellation_amd import mpc_step_batch, _ffi

gc_log = []
_t = [0.0]
def _cb(phase, info):
    if phase == "start": _t[0] = time.perf_counter()
    else: gc_log.append((info["generation"], (time.perf_counter() - _t[0]) * 1e3, time.perf_counter()))
gc.callbacks.append(_cb)

lib = _ffi.load()
orig = lib.mpcx_mpc_step_batch
inner = []
class Timed:
    def __call__(self, *a):
        t0 = time.perf_counter(); rc = orig(*a); inner.append((time.perf_counter() - t0) * 1e3); return rc
lib.mpcx_mpc_step_batch = Timed()

N = int(os.environ.get("N_CALLS", "50"))
for S in [int(a) for a in sys.argv[1:]] or [64, 4096]:
    xbar, ubar, consts, r_des = workload(4096, 30, first=0, count=S)
    tf = np.ones(S)
    hp = [_ffi.pinned_copy(a) for a in (xbar, ubar, tf, consts, r_des)]
    for name, f in (("pageable", lambda: mpc_step_batch(xbar, ubar, tf, consts, r_des)),
                    ("pinned", lambda: mpc_step_batch(*hp, pinned_results=True))):
        for use_gc in (True, False):
            (gc.enable if use_gc else gc.disable)()
            inner.clear(); gc_log.clear(); outer = []; stamps = []
            for i in range(N):
                t0 = time.perf_counter(); f(); t1 = time.perf_counter()
                outer.append((t1 - t0) * 1e3); stamps.append((t0, t1))
            o = np.array(outer); c = np.array(inner)
            slow = [i for i in range(N) if o[i] > 1.5 * np.median(o)]
            print(f"S {S:5d} {name:8s} gc {'on ' if use_gc else 'off'}: wrapper median {np.median(o):.3f} max {o.max():.3f} ms | "
                  f"ctypes call median {np.median(c):.3f} max {c.max():.3f} ms | slow calls {[(i, round(o[i], 2), round(c[i], 2)) for i in slow]}", flush=True)
            for g, ms, when in gc_log:
                if ms > 1.0:
                    idx = [i for i, (a, b) in enumerate(stamps) if a <= when <= b]
                    print(f"      gc generation {g}: {ms:.2f} ms during call {idx}", flush=True)
    gc.enable()
