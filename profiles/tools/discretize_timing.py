"""profiling helper: duration of discretize_kernel (HIP events on its stream) on the benchmark workloads"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import ctypes as C
from mpconstellation_amd import _ffi
if os.environ.get("MPCX_LIB"): _ffi.LIB_PATH = os.path.abspath(os.environ["MPCX_LIB"])       # A/B against another build
import torch, bench
for wl in sys.argv[1:] or ["S64_K30", "S4096_K30", "S4096_K100_scp2"]:
    r = bench.Runner(wl, 0, 1, 0)
    p = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(r.stream)
    ms = []
    for rep in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        _ffi.check(r.lib.mpcx_discretize_stages_dev(r.ctx, r.S, r.K, r.K, p(r.d_x), p(r.d_u), p(r.d_tf), p(r.d_c), 0, 1e-2, p(r.d_stage), p(r.d_dst), st), r.ctx, "discretize")
        e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    print(f"{wl}: discretize_kernel {min(ms):.3f} ms (min of 5), status ok {int((r.d_dst == 0).sum())}/{r.S}", flush=True)
    del r; torch.cuda.empty_cache()
