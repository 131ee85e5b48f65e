import cProfile, pstats, sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from mpconstellation_amd import Satellite, ConstellationMPC
from mpconstellation_amd.constellation import constellation_states
S = 4096
st = constellation_states(S)
make = lambda: [Satellite(s[:3].copy(), s[3:6].copy(), float(s[6])) for s in st]
ConstellationMPC(make(), base_res=30, tf_horizon=2, tf_interval=1, r_des=1.5, sim_base_res=100).run_segments(tf=2, num_segments=1)
mpc = ConstellationMPC(make(), base_res=30, tf_horizon=2, tf_interval=1, r_des=1.5, sim_base_res=100)
pr = cProfile.Profile(); pr.enable()
mpc.run_segments(tf=2, num_segments=2)
pr.disable()
ps = pstats.Stats(pr); ps.sort_stats("cumulative").print_stats(28)
