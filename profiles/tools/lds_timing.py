"""profiling helper: solve kernel durations (HIP events, device-resident inputs: bench.Runner) of a small batch with the
LDS-resident build (default) against the global-workspace two-wave kernel (flag 32) and the one-wave kernel (16), alternating
in one process.  usage: python profiles/tools/lds_timing.py [S ...]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch, bench
for S in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    bench.WORKLOADS["probe"] = (S, 30, 1)
    r = bench.Runner("probe", 0, 1, 0)
    best = {0: 1e9, 32: 1e9, 16: 1e9}
    for rep in range(4):
        for fl in (0, 32, 16):
            r.opts.flags = fl; r.solve_events = []
            el, ms = bench.measure(r, 8, 2, 1)
            best[fl] = min(best[fl], ms)
    print(f"S {S:4d} K 30: solve kernel (mean of 8, best of 4 rounds)  LDS-resident {best[0]:.3f} ms | two waves, global workspace {best[32]:.3f} ms | one wave {best[16]:.3f} ms", flush=True)
    del r; torch.cuda.empty_cache()
