"""numpy restatement of SequenceController's first-order hold over a ragged batch (test infrastructure: the checker of
mpcx_resample_sequence_dev and of the rollouts' u_out; the product does this on the device)."""
import numpy as np


def foh_resample_ragged(u, Ku, n):
    """The same for a ragged batch, every satellite at once: table u (S,3,Kmax) with Ku[s] columns in use, evaluated at
    linspace(0, 1, n[s]) -> (S,3,max(n)), zero past a satellite's last node.  Extract_uk of the reference
    (linearize_discretize.py:393-411) for SequenceController(u_s, tf_u, tf_sim = tf_u): np.linspace's nodes (i * step, the
    last one exactly 1), k = int(tau // dtau) (numpy's float floor_divide is CPython's algorithm), tau_k = k / (K-1),
    the blend as written in control.py:122-126."""
    u = np.ascontiguousarray(u, dtype=np.float64)
    S, _, Kmax = u.shape
    Ku = np.asarray(Ku).reshape(S, 1); nn = np.asarray(n).reshape(S, 1)
    nmax = int(nn.max())
    i = np.arange(nmax, dtype=np.float64)[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        step = 1.0 / (nn - 1.0)
        tau = i * step
        tau[np.broadcast_to(nn <= 1, tau.shape)] = 0.0
        at1 = (i == nn - 1) & (nn > 1)              # np.linspace's last node is exactly 1
        tau[at1] = 1.0
        km1 = (Ku - 1).astype(np.float64)
        dtau = 1 / km1
        k = np.floor_divide(tau, dtau)
    k = np.clip(k, 0, Ku - 2).astype(np.int64)
    k[at1] = 0
    tau_k = k / km1; tau_kp1 = (k + 1) / km1
    den = tau_kp1 - tau_k
    lam_n = (tau_kp1 - tau) / den; lam_p = (tau - tau_k) / den
    keep = i < nn
    out = np.zeros((S, 3, nmax))
    base = np.arange(S, dtype=np.int64)[:, None] * (3 * Kmax)
    flat = u.reshape(-1)
    last = (Ku - 1).astype(np.int64)
    for c in range(3):                          # (flat gathers: much cheaper than take_along_axis on a broadcast index)
        off = base + c * Kmax
        val = lam_n * flat[off + k] + lam_p * flat[off + k + 1]
        val[at1] = np.broadcast_to(flat[off + last], val.shape)[at1]
        oc = out[:, c, :]
        oc[keep] = val[keep]
    return out
