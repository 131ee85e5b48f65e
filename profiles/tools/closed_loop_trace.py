"""profiling helper: the closed loop (bench.py's closed_loop leg: ConstellationMPC.run_segments, test_mpc configuration) under
rocprofv3 --kernel-trace: which kernels its device time consists of.
usage: rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cl -- python3 profiles/tools/closed_loop_trace.py [S]
       python3 profiles/tools/closed_loop_trace.py --summarize gpurun_out/cl"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    import csv, glob
    from collections import defaultdict
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        # the LAST two segments of the process (the timed ones): everything after the last-but-two propagate_kernel<2,0> launches
        starts = [i for i, r in enumerate(rows) if "propagate_kernel<2" in r["Kernel_Name"]]
        rows = rows[starts[-2]:]
        t0 = int(rows[0]["Start_Timestamp"]); acc = defaultdict(float)
        for r in rows:
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if d > 0.05: print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:9.3f} ms  {d:8.3f} ms  {name}')
            acc[name] += d
        span = (int(rows[-1]["End_Timestamp"]) - t0) / 1e6
        print(f"two segments: {span:.2f} ms from the first rollout's start to the last kernel's end; kernel time {sum(acc.values()):.2f} ms")
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1]): print(f"   {v:8.3f} ms  {k}")
    sys.exit(0)
import numpy as np
from mpconstellation_amd import Satellite, ConstellationMPC
from mpconstellation_amd.constellation import constellation_states
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
st = constellation_states(S)
make = lambda: [Satellite(s[:3].copy(), s[3:6].copy(), float(s[6])) for s in st]
kw = dict(base_res=30, tf_horizon=2, tf_interval=1, r_des=1.5, sim_base_res=100)
ConstellationMPC(make(), **kw).run_segments(tf=2, num_segments=1)
mpc = ConstellationMPC(make(), **kw)
for seg in range(2):
    mpc.run_segment(1)
    print(f"segment {seg}: interior-point iterations per SCP iteration (mean, max):", [(round(float(a.mean()), 2), int(a.max())) for a in mpc.last_iters],
          "nodes of the second iteration:", int(mpc.plan_K.min()), "..", int(mpc.plan_K.max()), "not converged:", int((mpc.last_status != 0).sum()))
