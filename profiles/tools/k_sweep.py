"""solve_kernel's cost per (satellite x node x IPM iteration) against the horizon length K at a full device: the slot workspaces
(2048 slots x ws_doubles(K)) fit the 256 MB infinity cache below K ~ 19 and not above.  usage: python profiles/tools/k_sweep.py [S]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from mpconstellation_amd import _ffi
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
lib = _ffi.load()
print(f"S = {S}; K, slot workspace MB (2048 slots), solve_kernel ms (best of 3 x 5), mean IPM iterations, ns per satellite-node-iteration")
for K in (8, 12, 16, 20, 24, 30, 40, 60):
    name = f"S{S}_K{K}"
    bench.WORKLOADS[name] = (S, K, 1)
    r = bench.Runner(name, 0, 1, 0)
    best = 1e9
    for rep in range(3):
        el, ms = bench.measure(r, 5, 2, 1); best = min(best, ms)
    it = r.d_it.cpu().numpy(); st = r.d_st.cpu().numpy()
    ws = lib.mpcx_solve_workspace_bytes(S, K) / 1e6
    print(f"K {K:3d}  ws {ws:7.1f} MB  kernel {best:7.3f} ms  iters {it.mean():5.2f}  ok {int(np.isin(st, (0, 7)).sum())}/{S}  {best * 1e6 / (S * K * it.mean()):7.2f} ns", flush=True)
    del r; torch.cuda.empty_cache()
