import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mpconstellation_amd import mpc_step_batch
from test_full_size_gpu import workload
S, K = int(sys.argv[1]), int(sys.argv[2])
xbar, ubar, consts, r_des = workload(4096, K, first=0, count=S)
b = mpc_step_batch(xbar, ubar, np.ones(S), consts, r_des, flags=64)
print("status", b.status.tolist(), "iters", b.iters.tolist(), "kkt", b.kkt.tolist())
for i in range(S):
    if b.kkt[i] == -1.0: print("sat", i, "mailbox", b.NU[i].ravel()[:32].astype(int).tolist())
