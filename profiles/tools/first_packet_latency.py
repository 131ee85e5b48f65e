"""profiling helper: when does a stream take 20-30 ms to execute the FIRST packet of a host-pointer call?  (MPCX_HOST_TRACE's
"q:first-marker": a bare event recorded on the context's stream at the start of the call, polled until complete -- in
bench.py's warm-up calls it alone accounts for the calls' 40 ms.)  The same 4096-satellite call after different histories.
usage: MPCX_HOST_TRACE=0 python profiles/tools/first_packet_latency.py 2>&1 | grep -v amdgpu"""
import os, sys, time, re, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from test_full_size_gpu import workload
from mpconstellation_amd import mpc_step_batch, _ffi
big = workload(4096, 30, first=0, count=4096)
tf = np.ones(4096)

def calls(label, n=3):
    sys.stderr.flush()
    print(f"== {label}", file=sys.stderr, flush=True)
    for _ in range(n):
        t0 = time.perf_counter(); mpc_step_batch(big[0], big[1], tf, big[2], big[3]); dt = (time.perf_counter() - t0) * 1e3
        print(f"   call {dt:.2f} ms", file=sys.stderr, flush=True)

calls("fresh context (allocations in the first call)")
calls("again, back to back")
time.sleep(0.5); calls("after 0.5 s of sleep (device idle)")
a = np.random.default_rng(0).random((3000, 3000)); t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5: a @ a
calls("after 0.5 s of host numpy work (device idle, host busy)")
x = torch.randn(8192, 8192, device="cuda"); torch.cuda.synchronize(); t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5: y = x @ x
torch.cuda.synchronize()
calls("after 0.5 s of torch matmuls on torch's stream")
t = torch.empty(1 << 28, dtype=torch.float64, device="cuda"); t.fill_(1.0); torch.cuda.synchronize()
calls("after torch allocated and filled a 2 GB tensor")
del t; torch.cuda.empty_cache()
calls("after torch released it to the driver")
h = [torch.tensor(b, dtype=torch.float64, device="cuda") for b in big]; torch.cuda.synchronize()
calls("after torch uploaded the 17 MB of inputs (pageable H2D copies)")
r = h[0].cpu(); 
calls("after torch downloaded 7 MB (D2H copy)")
