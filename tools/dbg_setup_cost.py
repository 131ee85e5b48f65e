"""diagnostic (not a test): solve_kernel time at max_iter = 0, 1, 2 (set-up cost and cost per iteration)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch, bench
for wl in ("S64_K30", "S4096_K30"):
    r = bench.Runner(wl, 0, 1, 0)
    for mi in (0, 1, 2, 4, 200):
        r.opts.max_iter = mi
        r.solve_events = []
        el, ms = bench.measure(r, 4, 2, 1)
        print(f"{wl} max_iter {mi:3d}: step {el / 4 * 1e3:7.3f} ms  solve_kernel {ms:7.3f} ms", flush=True)
    del r; torch.cuda.empty_cache()
