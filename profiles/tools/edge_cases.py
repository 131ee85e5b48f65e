"""profiling helper: solver status / iteration counts on edge-case option sets and inputs (256 satellites each)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from mpconstellation_amd import mpc_step_batch, _ffi
from mpconstellation_amd.constellation import constellation_states, normalize_batch, tangential_thrust
from mpconstellation_amd.simulator import propagate_batch
S, K = 256, 30
y0, consts = normalize_batch(constellation_states(4096, first=0, count=S))
xbar, st, _ = propagate_batch(y0, np.ones(S), consts, (_ffi.CTRL_TANGENTIAL, np.array([0.5]), 0, None), K)
ubar = np.ascontiguousarray(tangential_thrust(xbar, 0.5)); rd = np.linalg.norm(xbar[:, :3, -1], axis=1)
def run(name, **kw):
    o = kw.pop('options', {}); x = kw.pop('x', xbar); u = kw.pop('u', ubar); r = kw.pop('r_des', rd); tf = kw.pop('tf', np.ones(S))
    t0 = time.time()
    res = mpc_step_batch(x, u, tf, consts, r, options=o, **kw)
    a, c = np.unique(res.status, return_counts=True)
    print(f"{name:34s} status {dict(zip(a.tolist(), c.tolist()))} iters mean {res.iters.mean():.1f} max {res.iters.max()} kkt max {np.nanmax(res.kkt):.1e}  {1e3*(time.time()-t0):.0f} ms", flush=True)
    return res
run('default')
run('u_max 0.3 (thrust saturated)', options={'u_lim': [0, 0.3]})
run('u_max 0.05', options={'u_lim': [0, 0.05]})
run('r_min 1.0 (plane active)', options={'r_lim': [1.0, 5]})
run('r_min 1.01 (start infeasible)', options={'r_lim': [1.01, 5]})
run('eps_vr = eps_vn = 0', options={'eps_vr': 0.0, 'eps_vn': 0.0})
run('eps_r = 0', options={'eps_r': 0.0})
run('tf_max 0.5 (< tf_bar)', options={'tf_max': 0.5})
run('w_tr 0.2', options={'w_tr': 0.2})
run('w_nu 10', options={'w_nu': 10})
run('min_mass 0.999', options={'min_mass': 0.999})
run('r_des 3', r_des=np.full(S, 3.0))
run('r_des 0.9 (below)', r_des=np.full(S, 0.9))
xn = xbar.copy(); xn[3, 2, 7] = np.nan
run('NaN in one xbar', x=xn)
un = ubar.copy(); un[5] = 0.0
run('zero ubar for one sat', u=un)
run('linear vt', linear_vt=True)
run('linear vt eps_vt 1e-8', linear_vt=True, options={'eps_vt': 1e-8})
run('r_des 5.5 (> r_max: empty set)', r_des=np.full(S, 5.5))
r1 = rd.copy(); r1[5] = 7.0
run('one satellite with r_des 7', r_des=r1)
run('eps_vr -1e-3 (empty window)', options={'eps_vr': -1e-3})
