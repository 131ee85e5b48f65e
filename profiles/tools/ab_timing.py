"""profiling helper: A/B timing of two builds of libmpcx.so on the same box, alternating runs
usage: python profiles/tools/ab_timing.py libA.so libB.so [workload ...]"""
import os, sys, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
nl = 3 if len(sys.argv) > 3 and sys.argv[3].endswith(".so") else 2
libs = [os.path.abspath(a) for a in sys.argv[1:1 + nl]]
wls = sys.argv[1 + nl:] or ["S64_K30", "S4096_K30"]
code = '''
import sys
sys.path.insert(0, "%s")
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = "%s"
import torch, bench
import re
for wl in %r:
    if wl not in bench.WORKLOADS:          # any S<satellites>_K<nodes>
        m = re.fullmatch(r"S(\\d+)_K(\\d+)", wl); bench.WORKLOADS[wl] = (int(m.group(1)), int(m.group(2)), 1)
    r = bench.Runner(wl, 0, 1, 0)
    r.opts.flags = int(__import__("os").environ.get("SOLVE_FLAGS", "0"))      # (64: the time-parallel kernel)
    best = 1e9
    for rep in range(3):
        el, ms = bench.measure(r, 5, 2, 1)
        best = min(best, ms)
    print("%s", wl, "solve_kernel best of 3x5: %%.3f ms" %% best)
    del r; torch.cuda.empty_cache()
'''
for rnd in range(2):
    for lib in libs:
        subprocess.check_call([sys.executable, "-c", code % (ROOT, lib, wls, os.path.basename(lib))])
