"""profiling helper: builds packed_probe.hip, feeds it the Newton blocks and right-hand sides of
real interior-point iterates (from the CPU oracle), checks P_k, Kg, Quu^-1 and the backward-swept channel vectors against a
dense numpy recursion, and reports cycles per node per wave (4 satellites, 32 channels) next to solve_kernel's figure for one
satellite.  usage: python profiles/tools/packed_probe/packed_probe.py [n_ipm_iterations_before_the_snapshot]"""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.join(HERE, "..", "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_lib as O
import nlp_ipm as N
from mpconstellation_amd.constellation import constellation_states, normalize_batch

REC_N, NCH = 156, 8


def snapshot(i, K, n_it):
    st = constellation_states(4096, first=i, count=1); y0, cs = normalize_batch(st)
    tan = O.make_ctrl(O.CTRL_TANGENTIAL, (0.5, 0, 0))
    x, _, _ = O.propagate(y0[0], 1.0, cs[0], tan, K)
    u = O.extract_uk(x, np.linspace(0, 1, K), tan)
    d = O.discretize(x, u, 1.0, cs[0])
    P = N.MpcProblem(x, u, 1.0, cs[0][0], d, O.constraint_terms(x, u, cs[0][0]), {"r_des": float(np.linalg.norm(x[:3, -1]))})
    it = N.solve(P, max_iter=n_it)["iterate"] if n_it else N.initial_iterate(P, "ref", N.FAST)
    mu = 0.1 * sum((it.s[k] * it.z[k]).sum() for k in it.s) / sum(v.size for v in it.s.values())
    nb = N.newton_blocks(P, it, mu, 0.0); nb["lam_vt_cur"] = it.lam_vt
    F = N.riccati_factor(P, nb)
    zero = dict(X=np.zeros((7, K)), U=np.zeros((3, K)), NU=np.zeros((7, K - 1)), tf=0.0, lam=-it.lam.copy(), lam_vt=-it.lam_vt,
                zeta=np.zeros(len(nb["term"])))
    rhs = N.reduced_residual(P, nb, it, zero, F["win"])
    rec = np.zeros((K, REC_N)); ch = np.zeros((K, NCH, 24))
    for k in range(K):
        A = P.A[k] if k <= K - 2 else np.zeros((7, 7)); Bn = P.Bn[k] if k <= K - 2 else np.zeros((7, 3))
        Bpm = P.Bp[k - 1] if k >= 1 else np.zeros((7, 3))
        Wx = F["WxK"] if k == K - 1 else nb["Wx0"][k]
        D = nb["D"][:, k] if k <= K - 2 else np.ones(7)
        rec[k] = np.concatenate([A.ravel(), (A @ Bpm + Bn).ravel(), Bpm.ravel(), Wx.ravel(), nb["Wu0"][k].ravel(), D])
        gx0 = rhs["gx"][:, k].copy()
        if k == K - 1: gx0 -= F["gam"] * rhs["rvt"] * nb["avt"]
        ch[k, 0, :7] = gx0; ch[k, 0, 7:10] = rhs["gu"][:, k]
        if k <= K - 2: ch[k, 0, 10:17] = rhs["rho"][:, k]; ch[k, 0, 17:24] = rhs["aff"][:, k]; ch[k, 1, 17:24] = P.Sig[:, k]
    vecs = [nb["avt"]] + [a for (a, w, gh) in nb["term"]]
    for j, a in enumerate(vecs[:6]): ch[K - 1, 2 + j, :7] = a
    stiff = sum(len(s) for s in nb["stiff"])
    return rec, ch, F, stiff


def reference(rec, ch):
    K = rec.shape[0]
    Pn = np.zeros((7, 7)); pn = np.zeros((NCH, 7))
    outP = np.zeros((K, 7, 7)); outKg = np.zeros((K, 3, 7)); outQi = np.zeros((K, 3, 3)); outp = np.zeros((K, NCH, 7)); outqu = np.zeros((K, NCH, 3))
    for k in range(K - 1, -1, -1):
        r = rec[k]
        A = r[0:49].reshape(7, 7); Bh = r[49:70].reshape(7, 3); Bpm = r[70:91].reshape(7, 3); Wx = r[91:140].reshape(7, 7)
        Wu = r[140:149].reshape(3, 3); D = r[149:156]
        Mi = np.linalg.inv(np.diag(D) + Pn)
        G = Pn @ Mi; Pt = Pn - Pn @ Mi @ Pn
        Quu = Wu + Bpm.T @ Wx @ Bpm + Bh.T @ Pt @ Bh
        Quy = Bpm.T @ Wx + Bh.T @ Pt @ A
        Qi = np.linalg.inv(Quu); Kg = Qi @ Quy
        Pk = Wx + A.T @ Pt @ A - Quy.T @ Kg
        for c in range(NCH):
            gx, gu, rho, aff = ch[k, c, :7], ch[k, c, 7:10], ch[k, c, 10:17], ch[k, c, 17:24]
            t = pn[c] - G @ (rho + pn[c]) + Pt @ aff
            qu = gu + Bpm.T @ gx + Bh.T @ t
            pn[c] = gx + A.T @ t - Kg.T @ qu
            outp[k, c] = pn[c]; outqu[k, c] = qu
        Pn = Pk
        outP[k] = Pk; outKg[k] = Kg; outQi[k] = Qi
    return outP, outKg, outQi, outp, outqu


def reference_forward(rec, ch, outP, outKg, outQi, outp, outqu):
    K = rec.shape[0]
    X = np.zeros((K, NCH, 7)); U = np.zeros((K, NCH, 3)); NU = np.zeros((K, NCH, 7)); LAM = np.zeros((K, NCH, 7))
    for c in range(NCH):
        y = np.zeros(7)
        for k in range(K):
            r = rec[k]
            A = r[0:49].reshape(7, 7); Bh = r[49:70].reshape(7, 3); Bpm = r[70:91].reshape(7, 3); D = r[149:156]
            u = -(outKg[k] @ y) - outQi[k] @ outqu[k, c]
            U[k, c] = u; X[k, c] = y + Bpm @ u
            if k <= K - 2:
                rho, aff = ch[k, c, 10:17], ch[k, c, 17:24]
                yh = A @ y + Bh @ u + aff
                nu = -np.linalg.solve(np.diag(D) + outP[k + 1], outP[k + 1] @ yh + rho + outp[k + 1, c])
                NU[k, c] = nu; LAM[k, c] = D * nu + rho
                y = yh + nu
    return X, U, NU, LAM


def main():
    import torch
    n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    K = 30
    for waves in (1, 2):
        lib_path = f"/tmp/libpacked_probe_w{waves}.so"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared",
                               f"-DPROBE_WAVES={waves}", "-o", lib_path, os.path.join(HERE, "packed_probe.hip")])
    snaps = [snapshot(i, K, n_it) for i in (0, 517, 1033, 2049, 3071, 4000, 77, 1999)]
    refs = [reference(r, c) for r, c, _, _ in snaps]
    frefs = [reference_forward(sn[0], sn[1], *rf) for sn, rf in zip(snaps, refs)]
    # the dense recursion of this file against the oracle's own factorisation (where no stiff stage term is active)
    for (r, c, F, stiff), ref in zip(snaps, refs):
        if stiff == 0:
            e = max(np.abs(ref[0] - F["P"]).max() / np.abs(F["P"]).max(), np.abs(ref[1] - F["Kg"]).max() / np.abs(F["Kg"]).max())
            assert e < 1e-9, e
    dev = torch.device("cuda", 0); t64 = dict(dtype=torch.float64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    for waves in ((1, 2) if os.environ.get("PROBE_BOTH") else (1,)):     # (the 256-register build spills: 4-7x slower, measured)
        lib = C.CDLL(f"/tmp/libpacked_probe_w{waves}.so")
        for S in (8, 4096, 8192, 16384):
            idx = np.arange(S) % len(snaps)
            rec = torch.tensor(np.stack([snaps[i][0] for i in idx]), **t64); chin = torch.tensor(np.stack([snaps[i][1] for i in idx]), **t64)
            oP = torch.zeros((S, K, 49), **t64); oKg = torch.zeros((S, K, 21), **t64); oQi = torch.zeros((S, K, 9), **t64)
            op = torch.zeros((S, K, NCH, 7), **t64); oqu = torch.zeros((S, K, NCH, 3), **t64)
            cyc = torch.zeros((S + 3) // 4, dtype=torch.int64, device=dev)
            oL = torch.zeros((S, K, 28), **t64)
            ms = C.c_float(0)
            rc = lib.packed_probe_run(S, K, p(rec), p(chin), p(oP), p(oKg), p(oQi), p(oL), p(op), p(oqu), p(cyc), 5, C.byref(ms))
            assert rc == 0, rc
            fX = torch.zeros((S, K, NCH, 7), **t64); fU = torch.zeros((S, K, NCH, 3), **t64); fN = torch.zeros((S, K, NCH, 7), **t64); fL = torch.zeros((S, K, NCH, 7), **t64)
            fcyc = torch.zeros((S + 3) // 4, dtype=torch.int64, device=dev); fms = C.c_float(0)
            rc = lib.packed_forward_run(S, K, p(rec), p(chin), p(oP), p(oKg), p(oQi), p(oL), p(op), p(oqu), p(fX), p(fU), p(fN), p(fL), p(fcyc), 5, C.byref(fms))
            assert rc == 0, rc
            ferr = []
            for s in range(min(S, 16)):
                fr = frefs[idx[s]]
                fg = (fX[s].cpu().numpy(), fU[s].cpu().numpy(), fN[s].cpu().numpy(), fL[s].cpu().numpy())
                ferr.append([np.abs(g - r).max() / max(np.abs(r).max(), 1e-300) for g, r in zip(fg, fr)])
            ferr = np.max(np.array(ferr), axis=0); fcn = fcyc.cpu().numpy() / K
            errs = []
            for s in range(min(S, 16)):
                ref = refs[idx[s]]
                got = (oP[s].cpu().numpy().reshape(K, 7, 7), oKg[s].cpu().numpy().reshape(K, 3, 7), oQi[s].cpu().numpy().reshape(K, 3, 3),
                       op[s].cpu().numpy(), oqu[s].cpu().numpy())
                errs.append([np.abs(g - r).max() / max(np.abs(r).max(), 1e-300) for g, r in zip(got, ref)])
            errs = np.max(np.array(errs), axis=0)
            cn = cyc.cpu().numpy() / K
            print(f"waves/SIMD {waves}  S {S:6d} ({(S + 3) // 4:5d} waves): {ms.value:8.3f} ms per pass  "
                  f"cycles per node per wave (4 satellites, 32 channels) median {np.median(cn):7.0f} max {cn.max():7.0f}   "
                  f"max rel err P {errs[0]:.1e} Kg {errs[1]:.1e} Qi {errs[2]:.1e} p {errs[3]:.1e} qu {errs[4]:.1e}", flush=True)
            print(f"   forward sweep of the 32 channels:        {fms.value:8.3f} ms per pass  cycles per node per wave median {np.median(fcn):7.0f} max {fcn.max():7.0f}   "
                  f"max rel err x {ferr[0]:.1e} u {ferr[1]:.1e} nu {ferr[2]:.1e} lam {ferr[3]:.1e}", flush=True)
    print("solve_kernel today (profiles/r02/phase_timing.txt): factorisation ~8 450 cycles per node for ONE satellite alone on its SIMD, ~10 800 with two "
          "waves per SIMD; forward sweep ~2 700 / ~4 300")


if __name__ == "__main__":
    main()
