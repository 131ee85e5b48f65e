"""The two-segment (time-parallel) form of the reduced solve, kept checked on the CPU oracle: DESIGN.md section 8,
tests/tools/partitioned_riccati.py.  The device does not run it (yet); this pins the derivation a kernel would implement."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))


import pytest


@pytest.mark.parametrize("segments", [2, 4])
def test_partitioned_solve_equals_the_sequential_one(segments):
    import partitioned_riccati as PR
    PR.SEGMENTS = segments; PR._cache.clear()
    N = PR.N
    (_, P), = PR.problems_benchmark(1)
    a = N.solve(P)
    PR.STATS["cond"].clear(); PR.STATS["err"].clear(); PR.STATS["spd"].clear()
    N.riccati_channel = PR.partitioned_channel
    try:
        b = N.solve(P)
    finally:
        N.riccati_channel = PR.SEQ_CHANNEL
    assert a["status"] == 0 and b["status"] == 0 and a["iters"] == b["iters"]
    assert np.abs(a["X"] - b["X"]).max() < 1e-11 and np.abs(a["U"] - b["U"]).max() < 1e-11 and abs(a["tf"] - b["tf"]) < 1e-12
    # every channel of every iteration: the partitioned sweeps against the sequential ones for the same right-hand side
    assert len(PR.STATS["err"]) >= 8 * a["iters"] and max(PR.STATS["err"]) < 1e-9 and np.median(PR.STATS["err"]) < 1e-12
