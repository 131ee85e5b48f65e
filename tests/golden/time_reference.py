#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path for the parts of one satellite-MPC-step that can run here: Discretizer.discretize as
shipped (mp.Pool(cpu_count()) per call, linearize_discretize.py:377-380) and serially in-process (the get_matrices loop,
:8-82), and the nonlinear rollout Simulator.run (simulator.py:29-48, solve_ivp max_step 1e-3) -- per satellite, at K = 30 and
K = 100.  The solve (pyomo + ipopt, optimizer.py:254-603) cannot be timed: neither is installed.

Runs ONLY in the build container (needs /root/reference; imports it the way make_golden.py does).  Writes
tests/golden/reference_cpu_timing.json (numbers only); bench.py echoes it as cpu_baseline.reference_discretize, labelled
"build container, not this box" -- the reference cannot travel to the GPU box."""
import json
import os
import platform
import sys
import time
import types

os.environ.setdefault("MPLBACKEND", "Agg")
for _name in ["pyomo", "pyomo.environ", "pyomo.core", "pyomo.core.base", "pyomo.core.base.expression"]:
    sys.modules[_name] = types.ModuleType(_name)
sys.modules["pyomo.core.base.expression"].ScalarExpression = object
sys.path.insert(0, "/root/reference")

import numpy as np
import scipy
from functools import partial

from simulator import Simulator                                    # noqa: E402
from satellite import Satellite                                    # noqa: E402
from satellite_scale import SatelliteScale                         # noqa: E402
import linearize_discretize as LD                                  # noqa: E402
from control import ConstantTangentialThrustController             # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def best(f, reps):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); out.append(time.perf_counter() - t0)
    return min(out), float(np.median(out))


def main():
    sat = Satellite(np.array([5371.4806, -4133.1393, 1399.9594]) * 1000.0, np.array([4.6921, 4.9848, -3.2752]) * 1000.0, 12200.0)
    scale = SatelliteScale(sat=sat)
    const = scale.get_normalized_constants()
    res = {}
    for K in (30, 100):
        c = ConstantTangentialThrustController([sat], 0.5)                     # control.py:178-179
        sim = Simulator(sats=[sat], controller=c, scale=scale, base_res=K, include_drag=False, include_J2=False)
        t_roll = best(lambda: sim.run(tf=1), 3)
        sim.run(tf=1)
        x = sim.sim_data[sat.id]; t = sim.sim_time[sat.id]
        d = LD.Discretizer(const, include_drag=False, include_J2=False)
        u = d.extract_uk(x, t, c)
        f = Simulator.satellite_dynamics
        t_pool = best(lambda: d.discretize(f, x, u, 1.0), 3)

        def serial():
            # Discretizer.discretize with its pool.map replaced by a plain loop over the same worker (linearize_discretize.py:354-390)
            dd = LD.Discretizer(const, include_drag=False, include_J2=False)
            orig = LD.mp.Pool

            class Inline:
                def __init__(self, *a, **k): pass
                def map(self, g, it): return [g(i) for i in it]
                def close(self): pass
                def join(self): pass
                def __enter__(self): return self
                def __exit__(self, *a): return False
            LD.mp.Pool = Inline
            try:
                return dd.discretize(f, x, u, 1.0)
            finally:
                LD.mp.Pool = orig
        t_ser = best(serial, 3)
        res[f"K{K}"] = {"rollout_s": {"best": t_roll[0], "median": t_roll[1]},
                        "discretize_pool_s": {"best": t_pool[0], "median": t_pool[1]},
                        "discretize_serial_s": {"best": t_ser[0], "median": t_ser[1]}}
        print(K, res[f"K{K}"], flush=True)
    out = {"what": "the reference's own Discretizer.discretize (as shipped: mp.Pool(cpu_count()) per call; and the same work serially "
                   "in-process) and Simulator.run per satellite, Hubble fixture, tangential-0.5 reference, tf = 1, drag / J2 off; "
                   "seconds, best and median of 3",
           "where": "build container, not the GPU box (the reference cannot travel)", "host": platform.processor() or platform.machine(),
           "nproc": os.cpu_count(), "python": platform.python_version(), "numpy": np.__version__, "scipy": scipy.__version__,
           "not_timed": "pyomo model build + ipopt solve (optimizer.py:254-603): pyomo and ipopt are not installed", "results": res}
    json.dump(out, open(os.path.join(HERE, "reference_cpu_timing.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
