"""diagnostic (not a test): effect of the longest-first launch order on the S4096 / S8192 workloads"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
for wl in ("S4096_K30", "S8192_K30", "S64_K30"):
    for flags in (1, 0):
        run = bench.Runner(wl, 0, 1, 0)
        run.opts.flags = flags
        el, ms = bench.measure(run, 4, 2, 1)
        print(wl, "index order" if flags else "longest first", "ms/step %.3f solve_kernel %.3f" % (el / 4 * 1e3, ms))
        del run; torch.cuda.empty_cache()
