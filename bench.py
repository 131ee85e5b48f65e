#!/usr/bin/env python3
"""bench.py -- satellite-MPC-steps/s of the fused discretize+solve hot path on MI355X.

One "step" = one pass of the hot path (Discretizer.discretize + get_constraint_terms + solve_OPT of
the reference, one SCP iteration) over every satellite of the batch, inputs resident in HBM.
Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement"."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {            # BASELINE.json configs
    "S64_K30": (64, 30), "S4096_K30": (4096, 30), "S4096_K100": (4096, 100), "S8192_K30": (8192, 30),
}
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md chip table


def algorithmic_bytes(K):
    """compulsory traffic of one fused satellite-MPC-step: read xbar(7K) ubar(3K) tf consts(5),
    write x(7K) u(3K) nu(7K) tf status  =  8(27K+7)+8 bytes."""
    return 8 * (27 * K + 7) + 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="S64_K30", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=6, help="satellites solved by the CPU oracle")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libmpcx has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from mpconstellation_amd import _ffi
    from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
    from mpconstellation_amd.simulator import propagate_batch
    lib = _ffi.load(); ctx = _ffi.context(local_rank)

    S, K = WORKLOADS[args.workload]      # per GPU (weak scaling: satellites shard with no exchange)
    S_total = S * world
    # ---- synthetic inputs (setup, untimed): this rank's block of the S_total-satellite constellation ----
    states = constellation_states(S_total, first=rank * S, count=S)
    y0, consts = normalize_batch(states)
    tfbar = np.ones(S)
    xbar, st, _ = propagate_batch(y0, tfbar, consts, (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), K,
                                  device=local_rank)
    assert (st == 0).all()
    ubar = tangential_thrust(xbar, 0.5)
    r_des = np.linalg.norm(xbar[:, 0:3, -1], axis=1)

    t64 = dict(dtype=torch.float64, device=dev)
    d_x = torch.tensor(xbar, **t64); d_u = torch.tensor(np.ascontiguousarray(ubar), **t64)
    d_tf = torch.tensor(tfbar, **t64); d_c = torch.tensor(consts, **t64); d_rd = torch.tensor(r_des, **t64)
    d_X = torch.empty((S, 7, K), **t64); d_U = torch.empty((S, 3, K), **t64); d_NU = torch.empty((S, 7, K), **t64)
    d_tfo = torch.empty(S, **t64); d_kkt = torch.empty(S, **t64)
    d_st = torch.empty(S, dtype=torch.int32, device=dev); d_it = torch.empty(S, dtype=torch.int32, device=dev)
    ws_bytes = lib.mpcx_mpc_step_workspace_bytes(S, K)
    d_ws = torch.empty(ws_bytes // 8 + 8, **t64)
    opts = _ffi.make_solve_opts({})
    stream = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr())

    def step():
        rc = lib.mpcx_mpc_step_batch_dev(ctx, S, K, p(d_x), p(d_u), p(d_tf), p(d_c), p(d_rd), 0, 1e-2, C.byref(opts),
                                         p(d_X), p(d_U), p(d_NU), p(d_tfo), p(d_st), p(d_it), p(d_kkt), p(d_ws),
                                         C.c_void_p(stream))
        _ffi.check(rc, ctx, "mpcx_mpc_step_batch_dev")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in ev:        # events on the stream the kernels are launched on
        e0.record(); step(); e1.record()
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], **t64); dist.all_reduce(tt, op=dist.ReduceOp.MAX); elapsed = float(tt.item())
    step_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))

    status = d_st.cpu().numpy(); iters = d_it.cpu().numpy(); kkt = d_kkt.cpu().numpy()
    conv = int(((status == 0) | (status == 7)).sum())
    stats = torch.tensor([conv, S], dtype=torch.float64, device=dev)
    if world > 1: dist.all_reduce(stats)

    if rank == 0:
        value = S_total * args.steps / elapsed
        B = algorithmic_bytes(K)
        achieved = S * B / (step_ms * 1e-3) / 1e9          # per GPU, whole fused step (discretize + solve kernels)
        out = {
            "metric": "satellite-MPC-steps/sec (whole constellation; 1 step = discretize + constraint terms + solve, 1 SCP iteration)",
            "value": value, "unit": "satellite-MPC-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "satellites_per_gpu": S, "satellites_total": S_total, "nodes_K": K,
                       "scp_iterations_per_step": 1, "parallelism": f"satellite-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "note": "algorithmic bytes/step = 8(27K+7)+8 per satellite; fused step = discretize_kernel + solve_kernel, "
                                 "duration from HIP events on the launch stream; the path is fp64-VALU/latency bound, not HBM bound"},
            "solver": {"converged": int(stats[0].item()), "of": int(stats[1].item()),
                       "ipm_iterations_mean": float(iters.mean()), "ipm_iterations_max": int(iters.max()),
                       "kkt_max": float(kkt.max())},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(xbar, ubar, tfbar, consts, r_des, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def cpu_baseline(xbar, ubar, tfbar, consts, r_des, n):
    """The CPU oracle (oracle/: C discretize + numpy interior point) timed on one host core on the first n
    satellites of the same workload.  Reported baseline only; pyomo+ipopt are not installed on this image."""
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib as O
    import nlp_ipm as N
    n = min(n, xbar.shape[0])
    t0 = time.perf_counter(); ok = 0
    for i in range(n):
        d = O.discretize(xbar[i], ubar[i], float(tfbar[i]), consts[i])
        terms = O.constraint_terms(xbar[i], ubar[i], consts[i][0])
        P = N.MpcProblem(xbar[i], ubar[i], float(tfbar[i]), consts[i][0], d, terms, {"r_des": float(r_des[i])})
        r = N.solve(P)
        ok += int(r["status"] in (0, 7))
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "satellite-MPC-steps/s", "cores": 1, "kind": "port",
            "sample": f"first {n} satellites of the workload, oracle/ (C discretize + numpy IPM), {ok}/{n} converged, {dt:.1f} s"}


if __name__ == "__main__":
    main()
