"""profiling helper: duration of propagate_kernel (HIP events on its stream) for the rollouts of the MPC loop:
tangential reference rollout and first-order-hold re-rollout, n_eval 30 / 100, 64 .. 8192 satellites"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpconstellation_amd import _ffi
from mpconstellation_amd.constellation import constellation_states, normalize_batch
lib = _ffi.load(); ctx = _ffi.context(0)
dev = torch.device("cuda", 0)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for S in (64, 4096, 8192):
    y0, consts = normalize_batch(constellation_states(S))
    T = lambda a, dt=torch.float64: torch.tensor(a, dtype=dt, device=dev)
    d_y0, d_c, d_tf, d_mag, d_one = T(y0), T(consts), T(np.ones(S)), T(np.full(S, 0.5)), T(np.ones(S))
    for n_eval in (30, 100):
        d_y = torch.empty((S, 7, n_eval), dtype=torch.float64, device=dev)
        d_st = torch.empty(S, dtype=torch.int32, device=dev); d_ns = torch.empty(S, dtype=torch.int32, device=dev)
        rng = np.random.default_rng(1); useq = T(0.5 * rng.standard_normal((S, 3, n_eval)))
        for name, kind, vec, Ku, et in (("tangential", _ffi.CTRL_TANGENTIAL, d_mag, 0, None), ("sequence", _ffi.CTRL_SEQUENCE, useq, n_eval, d_one)):
            ms = []
            for rep in range(4):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
                rc = lib.mpcx_propagate_batch_dev(ctx, S, n_eval, p(d_y0), p(d_tf), p(d_c), 0, kind, p(vec), Ku,
                                                  p(et) if et is not None else None, 1e-3, p(d_y), p(d_st), p(d_ns), st)
                e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
                assert rc == 0
            print(f"S {S:5d} n_eval {n_eval:3d} {name:10s}: {min(ms):7.3f} ms  steps {int(d_ns.max())}  status ok {int((d_st == 0).sum())}/{S}", flush=True)
