"""diagnostic (not a test): discretize_kernel at 1 vs 2 waves per SIMD (register bound 360 vs 256)"""
import os, sys, subprocess, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpconstellation_amd import build as b
import bench
for waves in (1, 2):
    lib = f"/tmp/libmpcx_disc{waves}.so"
    subprocess.check_call([b.HIPCC] + b.FLAGS + [f"-DMPCX_DISC_WAVES={waves}", "-o", lib] + b.sources())
res = {}
for waves in (1, 2):
    code = f'''
import sys, ctypes as C
sys.path.insert(0, "{ROOT}")
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = "/tmp/libmpcx_disc{waves}.so"
import torch, bench
for wl in ("S64_K30", "S4096_K30"):
    r = bench.Runner(wl, 0, 1, 0)
    p = lambda t: C.c_void_p(t.data_ptr()); st = C.c_void_p(r.stream)
    def disc():
        _ffi.check(r.lib.mpcx_discretize_stages_dev(r.ctx, r.S, r.K, r.K, p(r.d_x), p(r.d_u), p(r.d_tf), p(r.d_c), 0, 1e-2, p(r.d_stage), p(r.d_dst), st), r.ctx, "d")
    disc(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): disc()
    e1.record(); torch.cuda.synchronize()
    print("waves/SIMD {waves}", wl, "discretize_kernel %.3f ms" % (e0.elapsed_time(e1) / 10), "checksum", float(r.d_stage.sum()))
'''
    subprocess.check_call([sys.executable, "-c", code])
