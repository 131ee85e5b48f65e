"""profiling helper: A/B of builds of libmpcx.so on the closed loop of bench.py (ConstellationMPC.run_segments, the reference's test_mpc
configuration: stiff terminal windows, iterations that refine), alternating child processes on one box
usage: python profiles/tools/ab_closed_loop.py libA.so libB.so [S ...]"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
libs = [os.path.abspath(a) for a in sys.argv[1:3]]
sizes = [int(a) for a in sys.argv[3:]] or [4096]
code = '''
import sys
sys.path.insert(0, "%s")
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = "%s"
import bench
for S in %r:
    best = max(bench.closed_loop(S, 0)["value"] for _ in range(3))
    print("%s closed loop S", S, "best of 3: %%.0f steps/s (%%.2f ms per segment)" %% (best, 1e3 * S / best))
'''
for rnd in range(2):
    for lib in libs:
        subprocess.check_call([sys.executable, "-c", code % (ROOT, lib, sizes, os.path.basename(lib))])
