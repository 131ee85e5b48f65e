"""profiling helper: solve_kernel time against the number of satellites (1 .. 4 waves per SIMD's worth)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from mpconstellation_amd import _ffi
if os.environ.get("MPCX_LIB"): _ffi.LIB_PATH = os.path.abspath(os.environ["MPCX_LIB"])       # A/B against another build
import torch, bench
for S in [int(a) for a in sys.argv[1:]] or [64, 256, 512, 1024, 2048, 3072, 4096, 8192]:
    name = f"S{S}_K30"
    bench.WORKLOADS[name] = (S, 30, 1)
    r = bench.Runner(name, 0, 1, 0)
    r.opts.flags = int(os.environ.get("SOLVE_FLAGS", "0"))       # e.g. 32 = MPCX_SOLVE_NO_LDS, 16 = MPCX_SOLVE_ONE_WAVE
    el, ms = bench.measure(r, 3, 2, 1)
    st = r.solver_stats()
    print(f"S {S:5d}  ms/step {el / 3 * 1e3:8.3f}  solve_kernel {ms:8.3f} ms   {S / (el / 3):10.0f} steps/s   iters mean {st[1].mean():.2f} max {st[1].max()} ok {(st[0] == 0).sum()}", flush=True)
    del r; torch.cuda.empty_cache()
