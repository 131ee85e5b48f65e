"""profiling helper: the slow second host-pointer call of a process (bench.py: calls_ms.warmup [32, 49] ms at 4096 satellites
against 8.7 ms in the steady state; MPCX_HOST_TRACE puts the time into the wait for the device).  Reproduces bench.py's
sequence -- device-resident steps through a torch Runner first, then mpc_step_batch with numpy arrays -- and is meant to run
under `rocprofv3 --kernel-trace --memory-copy-trace`, whose per-dispatch durations say whether the device work itself is slow.
usage: rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/warm -- python3 profiles/tools/host_warmup_trace.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from mpconstellation_amd import mpc_step_batch
run = bench.Runner("S4096_K30", 0, 1, 0)
bench.measure(run, 5, 2, 1)
h = run.host
for i in range(6):
    t0 = time.perf_counter()
    mpc_step_batch(h["xbar"], h["ubar"], h["tfbar"], h["consts"], h["r_des"])
    print(f"host-pointer call {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
