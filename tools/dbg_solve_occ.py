"""diagnostic (not a test): solve_kernel bounded for 2 vs 3 waves per SIMD"""
import os, sys, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from mpconstellation_amd import build as b
for waves in (2, 3):
    lib = f"/tmp/libmpcx_sw{waves}.so"
    subprocess.check_call([b.HIPCC] + b.FLAGS + [f"-DMPCX_SOLVE_WAVES={waves}", "-o", lib] + b.sources())
    code = f'''
import sys
sys.path.insert(0, "{ROOT}")
from mpconstellation_amd import _ffi
_ffi.LIB_PATH = "{lib}"
import torch, bench
for wl in ("S64_K30", "S4096_K30"):
    r = bench.Runner(wl, 0, 1, 0)
    el, ms = bench.measure(r, 4, 2, 1)
    print("waves/SIMD {waves}", wl, "ms/step %.3f solve_kernel %.3f" % (el / 4 * 1e3, ms))
    del r; torch.cuda.empty_cache()
'''
    subprocess.check_call([sys.executable, "-c", code])
