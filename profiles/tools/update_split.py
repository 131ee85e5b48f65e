"""mpcx_mpc_update_batch as one chain or as two (the halves of the batch on two streams: MPCX_UPDATE_SPLIT = 0 / 1 / 2, csrc/solve_api.hip):
the closed loop of bench.py (ConstellationMPC.run_segments, the reference's test_mpc configuration) at several constellation sizes,
the modes alternating in one process; seconds per segment, best and median of the repetitions.
usage: python profiles/tools/update_split.py [sizes ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpconstellation_amd import Satellite, ConstellationMPC                 # noqa: E402
from mpconstellation_amd.constellation import constellation_states            # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]
REPS = int(os.environ.get("REPS", 5))
kw = dict(base_res=30, tf_horizon=2, tf_interval=1, r_des=1.5, sim_base_res=100)
for S in sizes:
    st = constellation_states(S)
    make = lambda: [Satellite(s[:3].copy(), s[3:6].copy(), float(s[6])) for s in st]
    times = {"0": [], "1": [], "2": []}
    final = {}
    for rep in range(REPS + 1):
        for mode in times:
            os.environ["MPCX_UPDATE_SPLIT"] = mode
            mpc = ConstellationMPC(make(), **kw)
            t0 = time.perf_counter(); mpc.run_segments(tf=2, num_segments=2); dt = time.perf_counter() - t0
            if rep: times[mode].append(dt / 2)
            final[mode] = np.array([s.get_state_vector() for s in mpc.sats])
    same = all(np.array_equal(final["0"], final[m]) for m in ("1", "2"))
    print(f"S = {S}: ms per segment (best / median of {REPS}) and constellation-MPC-steps/s at the best; flown states bit-identical across the modes: {same}")
    for mode, label in (("0", "one chain"), ("1", "two chains"), ("2", "two chains, second delayed to the first's solve")):
        t = np.array(times[mode]) * 1e3
        print(f"   MPCX_UPDATE_SPLIT={mode} ({label}): {t.min():8.2f} / {np.median(t):8.2f} ms   {S / (t.min() * 1e-3):10.0f} steps/s")
os.environ.pop("MPCX_UPDATE_SPLIT", None)
