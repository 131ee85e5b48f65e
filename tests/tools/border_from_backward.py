"""Groundwork for DESIGN.md section 8(d): the 7 x 7 border of the solver's linear system from BACKWARD-sweep data alone.
The border entries are functionals of the channels' full solutions (a_i . x_K, Sigma . lam); they equal the bilinear form
of the optimal value's constant term,
    B(c', c) = sum_k [ - qu'_k . Qi_k qu_k - w'_k . Minv_k w_k + aff'_k . Pt_k aff_k + t'_k(c') . aff_k + t'_k(c) . aff'_k ],
    w_k = rho_k + p_{k+1},  t'_k = p_{k+1} - G_k w_k,
with a_i . x_K^c = B(2+i, c) and Sigma . lam^c = -B(1, c).  This script checks that on interior-point iterates of the oracle
(lives under tests/: it uses oracle/).  usage: IT=<iterations before the check> python tests/tools/border_from_backward.py"""
import sys, os, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_lib as O, nlp_ipm as N
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'disc_tan_K30_tf1.npz')); x, u, tf, cst = d["x"], d["u"], float(d["tf"]), d["const"]
stage = {k: d[k] for k in ("A", "Bp", "Bn", "Sigma", "xi")}
P = N.MpcProblem(x, u, tf, cst[0], stage, O.constraint_terms(x, u, cst[0]), {"r_des": float(np.linalg.norm(x[:3, -1]))})
r = N.solve(P, max_iter=int(os.environ.get("IT", "5"))); it = r["iterate"]
mu = 1e-4
nb = N.newton_blocks(P, it, mu, 0.0); nb["lam_vt_cur"] = it.lam_vt
F = N.riccati_factor(P, nb); K = P.K
zero = dict(X=np.zeros((7, K)), U=np.zeros((3, K)), NU=np.zeros((7, K - 1)), tf=0.0, lam=-it.lam.copy(), lam_vt=-it.lam_vt, zeta=np.zeros(len(nb["term"])))
rhs = N.reduced_residual(P, nb, it, zero, F["win"])
Z7 = np.zeros((7, K)); Z3 = np.zeros((3, K)); Zn = np.zeros((7, K - 1))
avt = nb["avt"]; term = nb["term"]
gx0 = rhs["gx"].copy(); gx0[:, K - 1] -= F["gam"] * rhs["rvt"] * avt
R = [(gx0, rhs["gu"], rhs["rho"], rhs["aff"]), (Z7, Z3, Zn, P.Sig)]
vecs = [avt] + [a for (a, w, gh) in term]
for a in vecs:
    g1 = Z7.copy(); g1[:, K - 1] = a; R.append((g1, Z3, Zn, Zn))
chans = [N.riccati_channel(P, nb, F, *rr) for rr in R]
nch = len(R)
# forward-sweep functionals
Ffwd = np.zeros((nch, nch))     # row: functional index (channel whose rhs defines it), col: channel
for c in range(nch):
    Ffwd[1, c] = -(P.Sig * chans[c][3]).sum()                       # -(Sig . LAM_c)  = B(1, c)
    for i, a in enumerate(vecs): Ffwd[2 + i, c] = a @ chans[c][0][:, K - 1]
# backward data per channel
def backward(gx, gu, rho, aff):
    p = np.zeros((K + 1, 7)); qu = np.zeros((K, 3)); w = np.zeros((K, 7)); tp = np.zeros((K, 7))
    for k in range(K - 1, -1, -1):
        Bpm = P.Bp[k - 1] if k >= 1 else np.zeros((7, 3))
        if k <= K - 2:
            w[k] = rho[:, k] + p[k + 1]; tp[k] = p[k + 1] - F["G"][k] @ w[k]
            t = tp[k] + F["Pt"][k] @ aff[:, k]; Ah = P.A[k]
        else:
            t = np.zeros(7); Ah = np.zeros((7, 7))
        qu[k] = gu[:, k] + Bpm.T @ gx[:, k] + F["Bh"][k].T @ t
        p[k] = gx[:, k] + Ah.T @ t - F["Kg"][k].T @ qu[k]
    return p, qu, w, tp
bw = [backward(*rr) for rr in R]
B = np.zeros((nch, nch))
for c1 in range(nch):
    for c2 in range(nch):
        s = 0.0
        for k in range(K):
            s -= bw[c1][1][k] @ F["Qi"][k] @ bw[c2][1][k]
            if k <= K - 2:
                a1, a2 = R[c1][3][:, k], R[c2][3][:, k]
                s += -bw[c1][2][k] @ F["Minv"][k] @ bw[c2][2][k] + a1 @ F["Pt"][k] @ a2 + bw[c1][3][k] @ a2 + bw[c2][3][k] @ a1
        B[c1, c2] = s
print("functional rows 1.. vs bilinear form: max abs diff", np.abs(Ffwd[1:] - B[1:]).max(), " scale", np.abs(Ffwd[1:]).max())
print("symmetry of B:", np.abs(B - B.T).max())
