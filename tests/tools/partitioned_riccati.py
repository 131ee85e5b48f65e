"""Feasibility study (CPU oracle, lives under tests/: it uses oracle/): a TIME-PARALLEL form of the reduced solve of one
interior-point iteration -- the horizon cut at node m = K // 2 into two segments whose recursions do not wait for each other.

Sequential form (oracle/nlp_ipm.py riccati_factor / riccati_channel, csrc/solve_riccati.hpp): one backward recursion over the
K nodes, one forward sweep: 2 K dependent node steps.  Partitioned form, in the recursion's shifted state y_k:

  segment 2 (nodes m .. K-1): the same backward recursion as now -- its cost-to-go at the cut is V_m(y) = 1/2 y'P_m y + p_m'y;
  segment 1 (nodes 0 .. m-1): the same recursion started from a ZERO cost-to-go behind node m-1, and, besides each channel's
      own right-hand side, seven unit channels whose only datum is a linear terminal cost e_i'y_m.  Its forward sweep from
      y_0 = 0 gives y_m as an affine function of the terminal price l:  y_m(l) = y_m0 - N l,  N >= 0 (7 x 7, one per
      factorisation, shared by all channels);
  interface: l must be the gradient of segment 2's cost-to-go at the point segment 1 arrives at, l = P_m z + p_m with
      z = y_m(l):   (I + N P_m) z = y_m0 - N p_m   -- one 7 x 7 solve per channel;
  then segment 1's trajectory = its local one + sum_i l_i (unit trajectory i), segment 2's forward sweep starts from y_m = z.

Both backward recursions run side by side (m and K - m nodes), then both forward sweeps: K dependent node steps instead of
2 K, at the price of 7 more channels in segment 1 and the interface solve.  This script checks, on interior-point iterates
of the benchmark constellation and of the closed loop's stiff-window problems, (a) that the partitioned solve returns the
sequential one's direction and to how many digits, (b) that whole solves run with it take the same iterations.

(csrc/solve_tp.hip / solve_tp.hpp are this algebra on the device, four segments of 7 / 7 / 7 / 9 thirtieths of the horizon.)
usage: [SEGMENTS=2|3|4] python tests/tools/partitioned_riccati.py [n_satellites] [closed_loop_cache.pkl]
(the cache is what closed_loop_start_rules.py gen writes; without it only the benchmark set runs)"""
import os, pickle, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import numpy as np
import oracle_lib as O, nlp_ipm as N
from mpconstellation_amd.constellation import constellation_states, normalize_batch

SEQ_CHANNEL = N.riccati_channel


def factor_range(P, nb, k_lo, k_hi):
    """riccati_factor's node step for k = k_hi-1 .. k_lo with nothing behind node k_hi-1 (zero cost-to-go there); k_hi < K"""
    K = P.K; Wx, Wu, D = nb["Wx0"], nb["Wu0"], nb["D"]
    F = dict(P=np.zeros((K + 1, 7, 7)), Minv=np.zeros((K, 7, 7)), G=np.zeros((K, 7, 7)), Pt=np.zeros((K, 7, 7)),
             Qi=np.zeros((K, 3, 3)), Kg=np.zeros((K, 3, 7)), Bh=np.zeros((K, 7, 3)))
    for k in range(k_hi - 1, k_lo - 1, -1):
        Bpm = P.Bp[k - 1] if k >= 1 else np.zeros((7, 3))
        Pn = F["P"][k + 1]
        X1, X2, rd = N.ldl_solve7(np.diag(D[:, k]) + Pn, Pn)
        C1 = rd[:, None] * X1; C2 = rd[:, None] * X2
        Pt = Pn - X1.T @ C1; Pt = 0.5 * (Pt + Pt.T)
        F["G"][k] = X1.T @ C2; F["Minv"][k] = X2.T @ C2
        Ah = P.A[k]; Bh = P.A[k] @ Bpm + P.Bn[k]
        Quu = Wu[k] + Bpm.T @ Wx[k] @ Bpm + Bh.T @ Pt @ Bh
        Quy = Bpm.T @ Wx[k] + Bh.T @ Pt @ Ah
        np.linalg.cholesky(Quu)
        Qi = np.linalg.inv(Quu); Kg = Qi @ Quy
        Pk = Wx[k] + Ah.T @ Pt @ Ah - Quy.T @ Kg
        for (cu, cy, ex) in nb["stiff"][k]:
            if cu is None: cu = Bpm.T @ cy
            t = Qi @ cu; om = 1.0 / (1.0 / ex + cu @ t); v = cy - Kg.T @ cu
            Pk = Pk + om * np.outer(v, v); Kg = Kg + om * np.outer(t, v); Qi = Qi - om * np.outer(t, t)
        F["P"][k] = 0.5 * (Pk + Pk.T); F["Pt"][k] = Pt; F["Qi"][k] = Qi; F["Kg"][k] = Kg; F["Bh"][k] = Bh
    return F


def sweep_segment(P, nb, F, k_lo, k_hi, p_end, y_start, gx, gu, rho, aff):
    """backward sweep over nodes k_hi-1 .. k_lo with the linear cost-to-go p_end behind node k_hi-1, then the forward sweep
    from y_{k_lo} = y_start; the arithmetic of riccati_channel.  Returns the trajectories of the nodes, y behind the last
    node and the linear term p_{k_lo} of the segment's cost-to-go."""
    K = P.K; D = nb["D"]
    p = np.zeros((K + 1, 7)); qu = np.zeros((K, 3)); p[k_hi] = p_end
    for k in range(k_hi - 1, k_lo - 1, -1):
        Bpm = P.Bp[k - 1] if k >= 1 else np.zeros((7, 3))
        if k <= K - 2:
            t = p[k + 1] - F["G"][k] @ (rho[:, k] + p[k + 1]) + F["Pt"][k] @ aff[:, k]; Ah = P.A[k]
        else:
            t = np.zeros(7); Ah = np.zeros((7, 7))
        qu[k] = gu[:, k] + Bpm.T @ gx[:, k] + F["Bh"][k].T @ t
        p[k] = gx[:, k] + Ah.T @ t - F["Kg"][k].T @ qu[k]
    X = np.zeros((7, K)); U = np.zeros((3, K)); NU = np.zeros((7, K - 1)); LAM = np.zeros((7, K - 1))
    y = y_start.copy()
    for k in range(k_lo, k_hi):
        Bpm = P.Bp[k - 1] if k >= 1 else np.zeros((7, 3))
        u = -(F["Kg"][k] @ y) - F["Qi"][k] @ qu[k]
        U[:, k] = u; X[:, k] = y + Bpm @ u
        if k <= K - 2:
            yh = P.A[k] @ y + F["Bh"][k] @ u + aff[:, k]
            nu = -(F["G"][k].T @ yh) - F["Minv"][k] @ (rho[:, k] + p[k + 1])
            NU[:, k] = nu; LAM[:, k] = D[:, k] * nu + rho[:, k]
            y = yh + nu
    return (X, U, NU, LAM), y, p[k_lo]


_cache = {}
STATS = {"cond": [], "err": [], "spd": []}
SEGMENTS = int(os.environ.get("SEGMENTS", "2"))


def spd_solve(Nm, W, r):
    """(I + N W)^-1 r through the similar symmetric positive definite I + L'N L, W = L L' (no pivoting: what a kernel can do in
    registers); the pivoted LAPACK solve beside it is recorded in STATS["spd"]"""
    Ns = 0.5 * (Nm + Nm.T); Ws = 0.5 * (W + W.T)
    try:
        L = np.linalg.cholesky(Ws)
        Ls = np.linalg.cholesky(np.eye(7) + L.T @ Ns @ L)
        z = np.linalg.solve(L.T, np.linalg.solve(Ls.T, np.linalg.solve(Ls, L.T @ r)))
    except np.linalg.LinAlgError:
        # the cost-to-go behind the cut is not positive definite (the tangential equality's curvature lam_vt H_v in the early
        # iterations; the sequential recursion asks for D + P > 0, not for P > 0): the other similar symmetric matrix, with
        # N = C C' -- N contains D^-1 of the segment's last node -- I + C'W C, whose Cholesky factorisation is the cut's share of
        # the recursion's pivot test.  (csrc/solve_tp.hpp: tp_iface_factor, mode 1)
        C = np.linalg.cholesky(Ns)
        Ls = np.linalg.cholesky(np.eye(7) + C.T @ Ws @ C)
        z = C @ np.linalg.solve(Ls.T, np.linalg.solve(Ls, np.linalg.solve(C, r)))
    if r.ndim == 1:
        z_lu = np.linalg.solve(np.eye(7) + Nm @ W, r)
        STATS["spd"].append(np.abs(z - z_lu).max() / max(np.abs(z_lu).max(), 1e-300))
    return z


def partitioned_channel(P, nb, F, gx, gu, rho, aff):
    """the reduced solve of one channel over SEGMENTS segments (cuts[j-1] .. cuts[j]-1, j = 1 .. S): every segment but the last
    has its own recursion from a zero cost-to-go (the last one is the tail of the sequential factorisation F), a local
    trajectory from the start state 0 and price 0, seven responses to a unit start state (segments 2 .. S) and seven to a unit
    terminal price (segments 1 .. S-1).  With a_j the start state and l_j the terminal price of segment j:
        a_{j+1} = y0_j + Phi_j a_j - N_j l_j ,     l_j = W_{j+1} a_{j+1} + p0_{j+1} + Psi_{j+1} l_{j+1}      (a_1 = 0, l_S = 0)
    -- a coarse problem of S - 1 interfaces, solved by its own backward recursion What_j, qhat_j (7 x 7 blocks, S - 2 steps, the
    matrices once per factorisation) and a forward pass per channel."""
    K = P.K; S = SEGMENTS
    cuts = [round(j * K / S) for j in range(S + 1)]
    Z7 = np.zeros((7, K)); Z3 = np.zeros((3, K)); Zn = np.zeros((7, K - 1)); z7 = np.zeros(7)
    key = id(F)
    if key not in _cache:
        _cache.clear()
        segs = []
        for j in range(1, S + 1):
            lo, hi = cuts[j - 1], cuts[j]
            Fj = F if j == S else factor_range(P, nb, lo, hi)
            up = []; ua = []; Nm = np.zeros((7, 7)); Psi = np.zeros((7, 7)); Phi = np.zeros((7, 7))
            for i in range(7):
                e = np.zeros(7); e[i] = 1.0
                if j < S:                                  # unit terminal price
                    tr, ye, ps = sweep_segment(P, nb, Fj, lo, hi, e, z7, Z7, Z3, Zn, Zn)
                    up.append(tr); Nm[:, i] = -ye; Psi[:, i] = ps
                if j > 1:                                  # unit start state
                    tr, ye, _ = sweep_segment(P, nb, Fj, lo, hi, z7, e, Z7, Z3, Zn, Zn)
                    ua.append(tr); Phi[:, i] = ye
            segs.append(dict(F=Fj, lo=lo, hi=hi, up=up, ua=ua, N=Nm, Psi=Psi, Phi=Phi, W=Fj["P"][lo]))
        # coarse backward recursion (channel independent part): What_j for j = S .. 2
        What = [None] * (S + 2); What[S] = segs[S - 1]["W"]
        for j in range(S - 1, 1, -1):
            sj = segs[j - 1]
            EPhi = spd_solve(sj["N"], What[j + 1], sj["Phi"])
            What[j] = sj["W"] + sj["Psi"] @ What[j + 1] @ EPhi
            What[j] = 0.5 * (What[j] + What[j].T)
        for j in range(1, S): STATS["cond"].append(np.linalg.cond(np.eye(7) + segs[j - 1]["N"] @ What[j + 1]))
        _cache[key] = (segs, What, F)
    segs, What, _ = _cache[key]
    # per channel: the segments' local sweeps (independent of each other) ...
    loc = []; y0 = []; p0 = []
    for j in range(1, S + 1):
        sj = segs[j - 1]
        tr, ye, ps = sweep_segment(P, nb, sj["F"], sj["lo"], sj["hi"], z7, z7, gx, gu, rho, aff)
        loc.append(tr); y0.append(ye); p0.append(ps)
    # ... the coarse backward pass qhat_j, then forward a_j, l_j
    qhat = [None] * (S + 2); qhat[S] = p0[S - 1]
    for j in range(S - 1, 1, -1):
        sj = segs[j - 1]
        t = spd_solve(sj["N"], What[j + 1], y0[j - 1] - sj["N"] @ qhat[j + 1])
        qhat[j] = p0[j - 1] + sj["Psi"] @ (What[j + 1] @ t + qhat[j + 1])
    a = [None] * (S + 2); ell = [None] * (S + 2); a[1] = z7
    for j in range(1, S):
        sj = segs[j - 1]
        a[j + 1] = spd_solve(sj["N"], What[j + 1], y0[j - 1] + sj["Phi"] @ a[j] - sj["N"] @ qhat[j + 1])
        ell[j] = What[j + 1] @ a[j + 1] + qhat[j + 1]
    out = [np.zeros_like(loc[0][q]) for q in range(4)]
    for j in range(1, S + 1):
        sj = segs[j - 1]
        for q in range(4):
            v = loc[j - 1][q].copy()
            if j > 1: v += sum(a[j][i] * sj["ua"][i][q] for i in range(7))
            if j < S: v += sum(ell[j][i] * sj["up"][i][q] for i in range(7))
            out[q] += v                                    # (disjoint node ranges)
    # how far from the sequential sweeps' result for the same right-hand side (relative to the largest entry of each array)
    ref = SEQ_CHANNEL(P, nb, F, gx, gu, rho, aff)
    STATS["err"].append(max(np.abs(out[q] - ref[q]).max() / max(np.abs(ref[q]).max(), 1e-300) for q in range(4)))
    return tuple(out)


def problems_benchmark(n):
    st = constellation_states(4096, first=0, count=n)
    y0, cst = normalize_batch(st)
    probs = []
    for i in range(n):
        ctrl = O.make_ctrl(2, thrust=(0.5, 0.0, 0.0))
        K = 30
        x = O.propagate(y0[i], 1.0, cst[i], ctrl, K)[0]; t = np.linspace(0, 1, K)
        u = O.extract_uk(x, t, ctrl)
        d = O.discretize(x, u, 1.0, cst[i])
        probs.append(("bench", N.MpcProblem(x, u, 1.0, cst[i][0], d, O.constraint_terms(x, u, cst[i][0]), {"r_des": 1.5})))
    return probs


def problems_closed_loop(path, n):
    with open(path, "rb") as f: lst = pickle.load(f)
    out = []
    for q in lst[:n]:
        d = dict(q["d"]);
        out.append((f"loop seg{q['seg']} it{q['it']}", N.MpcProblem(q["x"], q["u"], q["tf"], q["mu"], d, q["terms"], q["opts"])))
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    probs = problems_benchmark(n)
    if len(sys.argv) > 2: probs += problems_closed_loop(sys.argv[2], 4 * n)
    print(f"whole solves: sequential | partitioned into {SEGMENTS} segments (iterations, status), distance of the solutions")
    tot = [0, 0]
    for label, P in probs:
        N.riccati_channel = SEQ_CHANNEL; a = N.solve(P)
        STATS["cond"].clear(); STATS["err"].clear(); STATS["spd"].clear()
        N.riccati_channel = partitioned_channel
        try: b = N.solve(P)
        finally: N.riccati_channel = SEQ_CHANNEL
        dx = np.abs(a["X"] - b["X"]).max(); du = np.abs(a["U"] - b["U"]).max()
        tot[0] += a["iters"]; tot[1] += b["iters"]
        print(f"{label:16s} K {P.K:3d}  seq {a['iters']:3d} st {a['status']}  part {b['iters']:3d} st {b['status']}  |dX| {dx:.2e} |dU| {du:.2e} |dtf| {abs(a['tf'] - b['tf']):.2e}"
              f"  cond(I + N P_m) max {max(STATS['cond']):.2e}  channel results against the sequential sweeps': median {np.median(STATS['err']):.1e} max {max(STATS['err']):.1e}  interface: symmetric form against LU max {max(STATS['spd']):.1e}", flush=True)
    print(f"iterations in total: sequential {tot[0]}, partitioned {tot[1]}")
