"""Distribution of the host-pointer call's trace fields (include/mpcx.h: mpcx_trace_enable) over many calls: what
tests/test_full_size_gpu.py::test_host_pointer_calls_have_no_stragglers asserts on 50 calls, measured on N (default 1000) at 64
and 4096 satellites -- percentiles of every field, the calls whose wall time exceeds 1.5 x the median with their breakdown, and
how many calls would break the test's bounds (device span <= 1.25 x median, host work after the first packet <= 1.5 x median)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpconstellation_amd import mpc_step_batch, _ffi      # noqa: E402
from test_full_size_gpu import workload                    # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for S in (64, 4096):
    xbar, ubar, consts, r_des = workload(4096, 30, first=0, count=S)
    tf = np.ones(S)
    for _ in range(3): mpc_step_batch(xbar, ubar, tf, consts, r_des)
    _ffi.trace_enable(True)
    recs = []
    for _ in range(N if S == 64 else max(N // 4, 50)):
        mpc_step_batch(xbar, ubar, tf, consts, r_des); recs.append(_ffi.last_call_trace())
    _ffi.trace_enable(False)
    f = {k: np.array([t[k] for t in recs]) for k in _ffi.TRACE_FIELDS}
    after = f["host_stage_ms"] + f["host_wait_ms"] + f["host_copyout_ms"]
    f["after_first_packet"] = after
    print(f"S = {S}, {len(recs)} traced calls; ms: median / p90 / p99 / max")
    for k in ("wall_ms", "first_marker_ms", "after_first_packet", "host_stage_ms", "host_wait_ms", "host_copyout_ms", "dev_span_ms", "dev_kernels_ms"):
        v = f[k]
        print(f"  {k:20s} {np.median(v):8.3f} {np.percentile(v, 90):8.3f} {np.percentile(v, 99):8.3f} {v.max():8.3f}")
    print(f"  calls beyond the test's bounds: device span > 1.25 x median: {(f['dev_span_ms'] > 1.25 * np.median(f['dev_span_ms'])).sum()}, "
          f"host after first packet > 1.5 x median: {(after > 1.5 * np.median(after)).sum()}, wall > 1.5 x median: {(f['wall_ms'] > 1.5 * np.median(f['wall_ms'])).sum()}")
    for i in np.nonzero(f["wall_ms"] > 1.5 * np.median(f["wall_ms"]))[0][:12]:
        print(f"   slow call {i}: " + ", ".join(f"{k} {recs[i][k]:.3f}" for k in _ffi.TRACE_FIELDS[:-1]))
