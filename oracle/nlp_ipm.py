"""oracle/nlp_ipm.py -- CPU restatement of the solve half of the hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and
only as the checker; the product (mpconstellation_amd/, libmpcx.so) never does.

What is restated
----------------
* The NLP that Optimizer.solve_OPT transcribes with pyomo (reference optimizer.py:254-603):
  variables :267-270,287; objective :300-325,355; dynamics/initial state :327-345,357,361; final
  mass :351-352,363; thrust ball :379-381; radial min/max :384-395; final radius :398-403; radial /
  normal velocity windows :406-416,432-446,466-467; exact tangential-velocity equality :492-517,577;
  L1 slack of the virtual control :579-585; tf range :588; option defaults :178-188.
* The solver it hands that NLP to is ipopt (third-party, unpinned, not installed here).  What is
  restated of ipopt is its published primal-dual interior-point scheme (Waechter & Biegler, Math.
  Prog. 106, 2006): slack form of the inequalities, relaxation of every bound by 1e-8*max(1,|b|)
  (bound_relax_factor), slack initialisation s = max(-g, bound_push max(1,|b|)) with bound_push 1e-4 (ipopt: 1e-2; the
  reference trajectory is nearly feasible and a smaller push keeps the start on it: 12.5 -> 10.9 iterations on the
  benchmark constellation, nothing lost on the scenario sweeps), multipliers z = mu_init/s (ipopt's
  bound_mult_init_method = mu-based; its default 'constant 1' left 7 of 4096 benchmark satellites crawling
  along one boundary for 90-165 iterations), mu_init = 1 and the L1 slack pairs started dual feasible
  (z = w_nu/2, s = t = mu/z): mean iterations 42 -> 32,
  fraction-to-the-boundary rule tau = max(0.99, 1-mu), the scaled optimality error E_0 with
  s_max = 100 and tol = 1e-8, multiplier safeguard z <= kappa mu/s (upper side only, see below).  The barrier
  parameter is adaptive (ipopt: mu_strategy adaptive), not ipopt's default monotone schedule -- see "barrier
  parameter" below.

PARITY UNPINNED at this boundary: the reference's own tests hold no numbers for solve_OPT and
ipopt cannot be run here, so nothing below is checked against ipopt output.  It is checked by
KKT residuals, by an independent dense Newton/KKT solve, and against solutions of the full
(un-eliminated, polynomial) NLP computed by scipy's trust-constr from a second, independent transcription
of optimizer.py (tests/golden/make_nlp_xcheck.py -> xcheck_*.npz: the reference's own A/B matrices, default
and OptimalController option sets, the constant-thrust reference whose optimum needs virtual control, the
convex linearised-vt variant, ipopt's all-zero start) -- see tests/test_oracle_solver.py.  The same fixtures
check the HIP kernel directly (tests/test_solve_xcheck_gpu.py).

Deliberate differences from ipopt's path (none changes the NLP or its KKT points)
* start: the reference trajectory (x_bar, u_bar, nu=0, tf_bar); ipopt starts from zeros because
  pyomo Vars carry no initial value.
* the quartic equality (v.t)^2 = vt_des^2 |t|^2, t = (r x v) x r, equals |h|^2 (|h|^2 - vt_des^2 |r|^2)
  with h = r x v; it is imposed as |v|^2 - (r.v)^2/|r|^2 - vt_des^2 = 0, which has the same zero set
  for h != 0, r != 0 (see vt_poly / vt_reduced below and the test that compares them).
* globalisation: backtracking on the 2-norm of the perturbed KKT residual with one step length for
  primal and dual variables plus a N_-inf(1e-8) centrality neighbourhood, instead of ipopt's filter.  The
  backtracking stops once the next trial would fall below ALPHA_FLOOR = 0.25 and that last trial is taken even
  if it fails the decrease test (a strict merit rejects too many good steps of this non-convex problem: on 1024
  satellites of the benchmark constellation the floor and the loose neighbourhood cut the mean iteration count
  32.7 -> 21.6 and the maximum 48 -> 30 with every problem still converging; with no line search at all the
  K = 60, tf = 2 fixture breaks down).
* kappa_Sigma = 100 instead of ipopt's 1e10, and only its upper side z <= kappa mu / s: with one step length for
  primal and dual variables a loose safeguard lets (s_i, z_i) pairs jam against the boundary (1 of the 64 benchmark
  satellites stalled at mu = 0.1); pulling oversized multipliers back after every step removed every stall.  The
  lower side (z >= mu / (kappa s)) is dropped: raising the multipliers of inactive constraints after every step only
  adds dual infeasibility (measured: 39 -> 32 iterations on long-arc K = 60 references, 51 -> 46 with the MPC loop's
  option set, 21.7 -> 21.2 on the benchmark constellation; the saturated-thrust scenarios pay 26 -> 28).
* barrier parameter: every iteration aims at mu = max(tol/10, SIGMA mean(s z), MU_ERR E_0) with SIGMA = 0.1, the
  path-following rule of primal-dual methods (ipopt's mu_strategy = adaptive is the same idea with an oracle choosing
  sigma; its default is the monotone Fiacco-McCormick schedule mu <- max(tol/10, min(0.2 mu, mu^1.5)) once
  E_mu <= 10 mu, which this oracle followed first and which spent several short-stepped iterations on every level).
  Measured (CPU, this file): benchmark constellation 20.8 -> 12.2 iterations on average, 29 -> 18 at most; short arcs
  (K = 20, tf = 0.5) 20 -> 13; long arcs (K = 60, tf = 2) 24 -> 15; the MPC loop's option set 42 -> 25.  Mehrotra's
  predictor-corrector (sigma from an affine probe plus the second-order term) needs 10.4 / 12.8 / 10.9 / 16.5 but
  every iteration then costs a second pair of sequential sweeps (1.5-1.7x on the device) and, left unguarded, its
  target collapses to tol/10 on 3 of the 4096 benchmark satellites while they are still infeasible; sigma = 0.05 or
  0.2, LOQO's centrality rule or a sigma tied to the last step length are all worse than 0.1.
* x_0 is eliminated (it is fixed by an equality), nu_{K-1}, t_{K-1} (which enter no dynamics row) are
  reported as 0.
* linear algebra: stage-wise Riccati recursion (see riccati_factor / riccati_solve) instead of MUMPS.  Barrier
  weights z/s reach 1e14..1e16 on active constraints at mu = 1e-9; three devices keep the structured solve as
  accurate as a pivoted factorisation of the un-condensed system (with them the MPC option set of control.py:192-197
  and optima with non-zero virtual control converge like the dense LAPACK path; without them they stalled):
  - terminal rank-1 terms above TERM_CAP leave the recursion as border unknowns zeta (1/w form, no huge entry);
  - the border unknowns (dtf, vt multiplier, zetas) are part of the direction that iterative refinement corrects:
    the residual carries zeta explicitly, so a refinement pass solves for small corrections and removes the
    cancellation error of the first pass's channel combination (reduced_residual);
  - stage terms above STAGE_CAP (an active r_min plane, radius or thrust ball) enter the recursion through a
    Sherman-Morrison update of (Q_uu^-1, gain, P_k) in which the weight appears only as its reciprocal.

Variants and modes (tests only; the device implements variant exact/linvt in the FAST mode from the reference start)
* MpcProblem(variant="linvt"): the linearised tangential pair of optimizer.py:471-489 instead of the quartic: convex.
* solve(start=...): "ref" (default), "zero" (pyomo's unset Vars: ipopt's start; only where the reduced tangential
  form is defined, i.e. variant linvt), or explicit arrays.
* solve(mode="ipopt_default"): ipopt's documented defaults for the barrier strategy and initialisation (IPOPT_DEFAULT
  below), frozen; the FAST parameters are the tuned ones.  Both must end at the same point (tested).
"""
import numpy as np

DEFAULT_OPTIONS = dict(min_mass=0.1, u_lim=[0, 5], r_lim=[0.99, 5], r_des=1, eps_r=0.01, eps_vr=0.00001,
                       eps_vn=0.00001, eps_vt=0.00001, tf_max=5, w_nu=1000, w_tr=0.002)   # optimizer.py:178-188

ST_OK, ST_MAXITER, ST_NUMERIC, ST_ACCEPTABLE, ST_INFEASIBLE = 0, 5, 6, 7, 8
BOUND_RELAX = 1e-8
BOUND_PUSH = 1e-4
KAPPA_SIGMA = 100.0
GAMMA_NBHD = 1e-8
DW_FIRST, DW_MIN, DW_MAX = 1e-4, 1e-20, 1e40     # ipopt first_hessian_perturbation, min_/max_hessian_perturbation
ALPHA_FLOOR = 0.25    # backtracking never takes the step below this (unless the fraction to the boundary does)
MU_INIT = 1.0
SIGMA = 0.1           # every iteration aims at mu = SIGMA * mean(s z)
# A "clean" start -- the reference trajectory strictly inside the stage constraints and the tf range (no slack of the thrust
# ball, of the r_max ball, of the r_min plane or of 0 <= tf <= tf_max had to be pushed) and ending no further than
# CLEAN_RADIUS window half-widths eps_r from the target radius (the terminal velocity windows are not asked: a reference
# never ends inside them) -- begins at MU_INIT_CLEAN and lets mu fall superlinearly, mu = min(SIGMA m, m^1.5) with
# m = mean(s z) (the exponent of ipopt's theta_mu).  Benchmark problems: 10.8 -> 8.8 iterations at K = 30, 10.8 -> 7.1 at
# K = 100 (their second SCP iteration, whose re-rollout misses the window by up to 0.7 eps_r: 10.1 -> 6.5), slowest satellite
# 15 -> 11 / 14 -> 9.  Every other start keeps MU_INIT and the SIGMA rule, because there the small start value costs: starts
# outside their stage constraints (thrust limit 0.3 below the reference thrust 0.5: 78 -> 330 regularised iterations over
# 96 problems, tail 58 -> 91 iterations); references that still have to travel to the target radius (the reference's
# test_mpc, r_des = 1.5: 14.6 -> 20.0 iterations) or that miss a tight radius window by many widths after the re-rollout
# (second SCP iteration of test_mpc, eps_r = 1e-6: 14.2 -> 18.5; the steps then jam between the two sides of the window).
CLEAN_RADIUS = 3.0
MU_INIT_CLEAN = 0.01
# ... but never below MU_ERR times the iterate's total error E_0 (infeasibilities included): the mean complementarity can
# collapse to the tol/10 floor while the iterate is still 1e-3 from feasible; the fraction-to-the-boundary rule then cuts
# every step to 0 and the multipliers run away (seen with a thrust limit of 0.3, one of 256 satellites).  Healthy paths
# keep mu / E_0 above 2e-4, so the bound never binds on them (benchmark iteration paths unchanged)
MU_ERR = 1e-6
# Second safeguard: FB_N consecutive accepted steps shorter than FB_ALPHA mean the iterate is jammed against its bounds
# (a start far outside the constraints: a thrust limit a tenth of the reference thrust, tf_max below the reference time);
# mu is then lifted to FB_BOOST * mean(s z) and follows ipopt's monotone Fiacco-McCormick rule from there on (mu moves on
# only when the barrier problem is solved to E_mu <= 10 mu).  With the MU_ERR bound in place the fallback is no longer
# what makes these problems converge; it makes them converge sooner (tf_max 0.5: 44 instead of 65 iterations on
# average).  FB_N = 8 and not less: benchmark problems at K = 100 take up to seven short regularised steps in a row and
# then recover by themselves in 16 / 25 iterations, where the monotone rule needs 32 / 41 -- and a launch ends with its
# slowest satellite.
FB_ALPHA, FB_N, FB_BOOST = 0.1, 8, 10.0
TERM_CAP = 1e4        # share of a terminal barrier weight kept inside the Riccati recursion
REFINE_TW = 1e10      # iterative refinement only once a barrier weight z/s (terminal terms, stage balls/planes, tf) exceeds this
                      # (1e9 until round 3: the benchmark's terminal windows cross 1e9 in their last one or two iterations, where a
                      #  direction good to 1e-6 is plenty; the stiff sets -- weights 1e12 .. 1e16 -- refine as before)
STAGE_CAP = 1e8       # share of a stage barrier weight kept inside the Hessian blocks of the recursion
N_TERM = 5            # rank-1 terminal barrier directions: rf_min, vr, vn, mass, |r|^2


def vt_poly(r, v, vt_des):
    """optimizer.py:492-517 as written: (v.t)^2 - vt_des^2 |t|^2 with h = r x v, t = h x r."""
    h = np.cross(r, v); t = np.cross(h, r)
    return (v @ t) ** 2 - vt_des ** 2 * (t @ t)


def vt_reduced(r, v, vt_des):
    """c~ = |v|^2 - (r.v)^2/|r|^2 - vt_des^2 with gradient (6,) and Hessian (6,6)."""
    q = r @ r; rv = r @ v
    c = v @ v - rv * rv / q - vt_des ** 2
    gr = -2 * rv * v / q + 2 * rv * rv * r / q ** 2
    gv = 2 * v - 2 * rv * r / q
    I = np.eye(3)
    Hvv = 2 * I - 2 * np.outer(r, r) / q
    Hrr = (-2 * np.outer(v, v) / q + 4 * rv * (np.outer(v, r) + np.outer(r, v)) / q ** 2
           + 2 * rv * rv * I / q ** 2 - 8 * rv * rv * np.outer(r, r) / q ** 3)
    Hrv = -2 * np.outer(v, r) / q - 2 * rv * I / q + 4 * rv * np.outer(r, r) / q ** 2
    return c, np.concatenate([gr, gv]), np.block([[Hrr, Hrv], [Hrv.T, Hvv]])


class MpcProblem:
    """One satellite's SCP subproblem.  stage = dict(A (K-1,7,7), Bp, Bn (K-1,7,3), Sigma, xi (7,K-1));
    terms = output of Optimizer.get_constraint_terms for this satellite (optimizer.py:80-170)."""

    def __init__(self, xbar, ubar, tfbar, mu_grav, stage, terms, options=None, variant="exact", fixed_tf=None):
        """variant "exact": the quartic tangential-velocity equality the reference enables (optimizer.py:577);
        "linvt": the linearised pair it keeps commented out (:471-489, :575-576) instead -- a convex problem."""
        o = {**DEFAULT_OPTIONS, **(options or {})}
        self.o = o; self.variant = variant
        assert variant in ("exact", "linvt")
        # fixed_tf: the final time is not a variable but held at this value (its range constraint :588 and its
        # stationarity row drop out); the inner problem of the shared-tf decomposition (solve_shared_tf below)
        self.fixed_tf = None if fixed_tf is None else float(fixed_tf)
        self.xbar = np.array(xbar, dtype=float); self.ubar = np.array(ubar, dtype=float)
        self.tfbar = float(tfbar)
        self.K = K = self.xbar.shape[1]
        self.A, self.Bp, self.Bn = stage["A"], stage["Bp"], stage["Bn"]
        self.Sig, self.xi = stage["Sigma"], stage["xi"]
        self.vt_des = np.sqrt(mu_grav / o["r_des"])                   # optimizer.py:283
        rl = lambda b: b + BOUND_RELAX * max(1.0, abs(b))
        self.b_u = rl(o["u_lim"][1] ** 2)                             # :379-381
        self.b_rmax = rl(o["r_lim"][1] ** 2)                          # :393-395
        self.b_rmin = rl(-o["r_lim"][0])                              # :384-391  (-rhat.r <= -r_min)
        self.rbar_hat = np.array(terms["rbar_hat"])                   # (3,K-1)
        nT = 8 if variant == "linvt" else 6
        aT = np.zeros((nT, 7)); bT = np.zeros(nT)
        aT[0, :3] = -np.asarray(terms["rf_hat"]); bT[0] = rl(-(o["r_des"] - o["eps_r"]))      # :398-402
        for row, (V, D, Db, eps) in zip((1, 3), (("Vr", "DrVr_DvVr", "DrVr_DvVr_bar", "eps_vr"),
                                                 ("Vn", "DrVn_DvVn", "DrVn_DvVn_bar", "eps_vn"))):
            g = np.asarray(terms[D]); c0 = terms[V] - terms[Db]
            aT[row, :6] = g; bT[row] = rl(o[eps] - c0)                # :406-410 / :436-440
            aT[row + 1, :6] = -g; bT[row + 1] = rl(o[eps] + c0)       # :412-416 / :442-446
        aT[5, 6] = -1.0; bT[5] = rl(-o["min_mass"])                   # :351-352
        if variant == "linvt":
            # |Vt_lin(x_K) - Vc_lin(r_K)| <= eps_vt: min_tan_vel_rule (:480-489) row 6, max_tan_vel_rule (:471-479) row 7
            g = np.array(terms["DrVt_DvVt"], dtype=float).copy(); g[:3] -= np.asarray(terms["DrVc"])
            c0 = terms["Vt"] - terms["DrVt_DvVt_bar"] - terms["Vc"] + terms["DrVc_rbar"]
            aT[6, :6] = g; bT[6] = rl(o["eps_vt"] - c0)
            aT[7, :6] = -g; bT[7] = rl(o["eps_vt"] + c0)
        self.aT, self.bT = aT, bT
        self.b_rfmax = rl((o["r_des"] + o["eps_r"]) ** 2)             # :403
        self.b_tf = np.array([rl(0.0), rl(o["tf_max"])])              # :588
        self.w_tr, self.w_nu = o["w_tr"], o["w_nu"]

    def structural_violation(self):
        """> 0 when the constraint set is empty whatever the dynamics (the virtual control nu makes every x_1..x_K
        reachable, so nothing else can make the NLP infeasible): the fixed start node violates its own radius
        constraints (x_0 = xbar_0 is an equality, optimizer.py:344-345, and :384-395 apply at k = 0 too), the terminal
        radius window lies outside the r_max ball (:393-403), r_min > r_max, an empty velocity window (eps < 0) or an
        empty tf range (:588).  ipopt ends such a problem in its restoration phase ("converged to a point of local
        infeasibility"); here it is seen before the first iteration.  Returns the largest violation (relaxed bounds)."""
        K = self.K
        r0 = self.xbar[:3, 0]
        v = [r0 @ r0 - self.b_rmax]                                                  # start node outside the r_max ball
        if K >= 3: v.append(-np.sqrt(r0 @ r0) - self.b_rmin)                         # ... or below the r_min plane (k = 0 < K-1)
        rK_max = np.sqrt(min(self.b_rmax, self.b_rfmax))
        v.append(-self.bT[0] - rK_max)                                               # r_hat.r_K >= r_des - eps_r out of reach
        if K >= 3: v.append(-self.b_rmin - np.sqrt(self.b_rmax))                     # r_min plane outside the r_max ball
        for lo in (1, 3) + ((6,) if len(self.bT) == 8 else ()):                      # windows a.x <= b+, -a.x <= b-
            v.append(-(self.bT[lo] + self.bT[lo + 1]))
        if self.fixed_tf is None: v.append(-(self.b_tf[0] + self.b_tf[1]))           # 0 < tf <= tf_max
        return max(v)

    # ---- NLP functions ------------------------------------------------------------------
    def objective(self, X, U, T, tf):                                 # :300-325
        return tf + self.w_nu * T.sum() + self.w_tr * (((X - self.xbar) ** 2).sum()
                                                       + ((U - self.ubar) ** 2).sum() + (tf - self.tfbar) ** 2)

    def dyn_residual(self, X, U, NU, tf):                             # :327-342
        K = self.K
        e = np.zeros((7, K - 1))
        for k in range(K - 1):
            e[:, k] = X[:, k + 1] - (self.A[k] @ X[:, k] + self.Bn[k] @ U[:, k] + self.Bp[k] @ U[:, k + 1]
                                     + self.Sig[:, k] * tf + self.xi[:, k] + NU[:, k])
        return e

    def ineq(self, X, U, NU, T, tf):
        """all inequality constraints in g(w) <= 0 form (relaxed bounds), keyed by family"""
        K = self.K
        g = {
            "u": (U ** 2).sum(0) - self.b_u,                                             # k = 0..K-1
            "rmax": (X[:3, 1:] ** 2).sum(0) - self.b_rmax,                               # k = 1..K-1
            "rmin": -(self.rbar_hat[:, 1:] * X[:3, 1:K - 1]).sum(0) - self.b_rmin,       # k = 1..K-2
            "term": self.aT @ X[:, K - 1] - self.bT,
            "rfmax": np.array([(X[:3, K - 1] ** 2).sum() - self.b_rfmax]),
            "tp": NU - T, "tn": -NU - T,                                                 # :579-585
            "tf": np.array([-tf, tf]) - self.b_tf,
        }
        if self.fixed_tf is not None: del g["tf"]
        return g


class Iterate:
    def copy(self):
        n = Iterate(); n.__dict__.update(self.__dict__); return n


FAST = dict(mu_strategy="adaptive", mu_init=MU_INIT, bound_push=BOUND_PUSH, kappa_sigma=KAPPA_SIGMA, kappa_two_sided=False,
            z_init="mu", centred_l1=True, clean_start=True)
# ipopt's documented defaults for the same knobs: monotone Fiacco-McCormick barrier update, mu_init 0.1, bound_push 1e-2,
# bound_mult_init_val 1, kappa_sigma 1e10 on both sides.  FROZEN: speed work changes FAST only; tests/test_oracle_solver.py
# requires both modes to end at the same solution, so that tuning cannot move the answer.
IPOPT_DEFAULT = dict(mu_strategy="monotone", mu_init=0.1, bound_push=1e-2, kappa_sigma=1e10, kappa_two_sided=True,
                     z_init="one", centred_l1=False)


def initial_iterate(P, start="ref", prm=FAST):
    """start "ref": the reference trajectory (x_bar, u_bar, tf_bar); "zero": every variable 0 as pyomo hands the model
    to ipopt (Vars without initial values, optimizer.py:267-270, 287; x_0 is eliminated and stays x_bar_0);
    or a dict(X, U, tf[, NU]) of arrays."""
    K = P.K; it = Iterate()
    if isinstance(start, dict):
        it.X = np.array(start["X"], dtype=float); it.U = np.array(start["U"], dtype=float); it.tf = float(start["tf"])
        it.X[:, 0] = P.xbar[:, 0]
    elif start == "zero":
        it.X = np.zeros_like(P.xbar); it.X[:, 0] = P.xbar[:, 0]; it.U = np.zeros_like(P.ubar); it.tf = 0.0
    else:
        it.X = P.xbar.copy(); it.U = P.ubar.copy(); it.tf = P.tfbar
    if P.fixed_tf is not None: it.tf = P.fixed_tf
    it.NU = np.zeros((7, K - 1)); it.T = np.zeros((7, K - 1))
    if isinstance(start, dict) and "NU" in start: it.NU = np.array(start["NU"], dtype=float)[:, :K - 1].copy()
    it.lam = np.zeros((7, K - 1)); it.lam_vt = 0.0
    g = P.ineq(it.X, it.U, it.NU, it.T, it.tf)
    bnd = {"u": P.b_u, "rmax": P.b_rmax, "rmin": P.b_rmin, "term": P.bT, "rfmax": P.b_rfmax, "tp": 0.0,
           "tn": 0.0, "tf": P.b_tf}
    it.s = {k: np.maximum(-v, prm["bound_push"] * np.maximum(1.0, np.abs(bnd[k]))) for k, v in g.items()}
    it.clean = False; it.mu0 = prm["mu_init"]
    if prm.get("clean_start") and P.fixed_tf is None:      # (fixed-tf solves feed a root search with their g_tf: left as they were)
        it.clean = all((-g[k] >= prm["bound_push"] * np.maximum(1.0, np.abs(bnd[k]))).all() for k in ("u", "rmax", "rmin", "tf") if k in g)
        it.clean = it.clean and bool(abs(np.linalg.norm(it.X[:3, K - 1]) - P.o["r_des"]) <= CLEAN_RADIUS * P.o["eps_r"])
        if it.clean: prm = dict(prm, mu_init=MU_INIT_CLEAN); it.mu0 = MU_INIT_CLEAN
    if prm["z_init"] == "mu": it.z = {k: prm["mu_init"] / it.s[k] for k in g}        # ipopt bound_mult_init_method = mu-based
    else: it.z = {k: np.ones_like(it.s[k]) for k in g}                                # ... = constant, bound_mult_init_val 1
    if prm["centred_l1"]:
        # L1 slack pairs start dual feasible and centred: z+ = z- = w_nu/2 (stationarity in t), t = |nu| + mu/z
        zl = P.w_nu / 2.0
        it.T = np.abs(it.NU) + prm["mu_init"] / zl
        for k, sg in (("tp", 1.0), ("tn", -1.0)):
            it.s[k] = it.T - sg * it.NU; it.z[k] = np.full_like(it.s[k], zl)
    return it


def lagrangian_gradient(P, it, zz, with_lambda=True):
    """gradient of f + lam^T c + zz^T g with respect to (x_k, u_k, tf, nu_k, t_k)"""
    K = P.K; X, U = it.X, it.U
    gx = 2 * P.w_tr * (X - P.xbar); gu = 2 * P.w_tr * (U - P.ubar); gtf = 1 + 2 * P.w_tr * (it.tf - P.tfbar)
    if P.variant == "exact": cv, gv, Hv = vt_reduced(X[:3, K - 1], X[3:6, K - 1], P.vt_des)
    else: cv, gv, Hv = 0.0, np.zeros(6), np.zeros((6, 6))       # no equality row: the tangential pair sits in aT
    if with_lambda:
        lam = it.lam
        for k in range(K - 1):
            gx[:, k + 1] += lam[:, k]; gx[:, k] -= P.A[k].T @ lam[:, k]
            gu[:, k] -= P.Bn[k].T @ lam[:, k]; gu[:, k + 1] -= P.Bp[k].T @ lam[:, k]
            gtf -= P.Sig[:, k] @ lam[:, k]
        gx[:6, K - 1] += it.lam_vt * gv
    gu += 2 * U * zz["u"][None, :]
    gx[:3, 1:] += 2 * X[:3, 1:] * zz["rmax"][None, :]
    gx[:3, 1:K - 1] += -P.rbar_hat[:, 1:] * zz["rmin"][None, :]
    gx[:, K - 1] += P.aT.T @ zz["term"]
    gx[:3, K - 1] += 2 * X[:3, K - 1] * zz["rfmax"][0]
    if "tf" in zz: gtf += -zz["tf"][0] + zz["tf"][1]
    gnu = zz["tp"] - zz["tn"] - (it.lam if with_lambda else 0.0)
    gt = P.w_nu - zz["tp"] - zz["tn"]
    return gx, gu, gtf, gnu, gt, cv, gv, Hv


def residual_vectors(P, it, mu):
    """the perturbed KKT residual F_mu as a list of arrays"""
    gx, gu, gtf, gnu, gt, cv, _, _ = lagrangian_gradient(P, it, it.z)
    g = P.ineq(it.X, it.U, it.NU, it.T, it.tf)
    e = P.dyn_residual(it.X, it.U, it.NU, it.tf)
    dual = [gx[:, 1:], gu, np.array([gtf if P.fixed_tf is None else 0.0]), gnu, gt]
    prim = [e, np.array([cv])] + [g[k] + it.s[k] for k in g]
    comp = [it.s[k] * it.z[k] - mu for k in g]
    return dual, prim, comp


def optimality_error(P, it, mu, s_max=100.0):
    """ipopt's scaled E_mu (Waechter-Biegler eq. (5)-(6))"""
    dual, prim, comp = residual_vectors(P, it, mu)
    zsum = sum(np.abs(v).sum() for v in it.z.values()); nz = sum(v.size for v in it.z.values())
    lsum = np.abs(it.lam).sum() + abs(it.lam_vt); nl = it.lam.size + (1 if P.variant == "exact" else 0)
    sd = max(s_max, (zsum + lsum) / (nz + nl)) / s_max
    sc = max(s_max, zsum / nz) / s_max
    d = max(np.abs(v).max() for v in dual); p = max(np.abs(v).max() for v in prim)
    c = max(np.abs(v).max() for v in comp)
    return max(d / sd, p, c / sc), d, p, c


def residual_norm(P, it, mu):
    dual, prim, comp = residual_vectors(P, it, mu)
    return np.sqrt(sum((v ** 2).sum() for v in dual + prim + comp))


# ------------------------------------------------------------------------------------------
# Newton system: blocks, Riccati solve, dense cross-check
# ------------------------------------------------------------------------------------------
def newton_blocks(P, it, mu, delta_w=0.0):
    """Reduced Newton/KKT system after eliminating slacks, inequality multipliers and t.
    Unknowns: dx_k (k>=1), du_k, dtf, dnu_k, new multipliers lam_k and lam_vt."""
    K = P.K; X, U = it.X, it.U
    g = P.ineq(it.X, it.U, it.NU, it.T, it.tf); s, z = it.s, it.z
    sig = {k: z[k] / s[k] for k in g}
    zhat = {k: mu / s[k] + sig[k] * (g[k] + s[k]) for k in g}
    gx, gu, gtf, gnu, gt, cv, gv, Hv = lagrangian_gradient(P, it, zhat, with_lambda=False)
    Wx = np.zeros((K, 7, 7)); Wu = np.zeros((K, 3, 3))
    for k in range(K):
        Wx[k] = (2 * P.w_tr + delta_w) * np.eye(7)
        Wu[k] = (2 * P.w_tr + delta_w + 2 * z["u"][k]) * np.eye(3) + sig["u"][k] * 4 * np.outer(U[:, k], U[:, k])
    for k in range(1, K - 1):
        r = X[:3, k]; rh = P.rbar_hat[:, k]
        Wx[k][:3, :3] += (2 * z["rmax"][k - 1] * np.eye(3) + sig["rmax"][k - 1] * 4 * np.outer(r, r)
                          + sig["rmin"][k - 1] * np.outer(rh, rh))
    # Stiff stage terms (an active r_min plane or radius / thrust ball: z/s reaches 1e14 at mu = 1e-9) would wipe out the
    # trust-region curvature 2 w_tr of the other directions if they were summed into the 7x7 / 3x3 blocks (1e14 * 1e-16 >
    # 0.004): the recursion keeps only min(sigma, STAGE_CAP) inside W and applies the excess as a rank-1 update of
    # (Q_uu^-1, gain, P_k) in which it appears only as 1/sigma_ex (riccati_factor).  Wx, Wu keep the full weights for
    # the residuals of the refinement.
    Wx0 = Wx.copy(); Wu0 = Wu.copy(); stiff = [[] for _ in range(K)]
    for k in range(1, K - 1):
        # at most one of the two position terms can be stiff (r_min < r_max): the one with the larger excess leaves the
        # block, the other stays whole
        ex_max = sig["rmax"][k - 1] - STAGE_CAP; ex_min = sig["rmin"][k - 1] - STAGE_CAP
        cy = np.zeros(7)
        if ex_max > 0 and ex_max >= ex_min: cy[:3] = 2 * X[:3, k]; ex = ex_max
        elif ex_min > 0: cy[:3] = P.rbar_hat[:, k]; ex = ex_min
        else: continue
        Wx0[k] -= ex * np.outer(cy, cy); stiff[k].append((None, cy, ex))
    for k in range(K):
        ex = sig["u"][k] - STAGE_CAP
        if ex > 0:
            cu = 2 * U[:, k]; Wu0[k] -= ex * np.outer(cu, cu); stiff[k].append((cu, np.zeros(7), ex))
    r = X[:3, K - 1]
    WxK_soft = Wx[K - 1].copy()
    WxK_soft[:3, :3] += 2 * (z["rmax"][K - 2] + z["rfmax"][0]) * np.eye(3)
    WxK_soft[:6, :6] += it.lam_vt * Hv
    a2r = np.zeros(7); a2r[:3] = 2 * r
    # terminal rank-1 barrier terms (direction, weight, gradient coefficient)
    term = [(P.aT[0], sig["term"][0], zhat["term"][0]),
            (P.aT[1], sig["term"][1] + sig["term"][2], zhat["term"][1] - zhat["term"][2]),
            (P.aT[3], sig["term"][3] + sig["term"][4], zhat["term"][3] - zhat["term"][4]),
            (P.aT[5], sig["term"][5], zhat["term"][5]),
            (a2r, sig["rmax"][K - 2] + sig["rfmax"][0], zhat["rmax"][K - 2] + zhat["rfmax"][0])]
    if P.variant == "linvt":
        term.append((P.aT[6], sig["term"][6] + sig["term"][7], zhat["term"][6] - zhat["term"][7]))
    # gx so far contains the full gradient coefficient of every terminal barrier term; take it out
    gxK_soft = gx[:, K - 1].copy()
    for a, w, gh in term: gxK_soft -= gh * a
    a_ = sig["tp"] + sig["tn"]; b_ = sig["tn"] - sig["tp"]
    D = 4 * sig["tp"] * sig["tn"] / a_
    avt = np.zeros(7); avt[:6] = gv
    return dict(Wx=Wx, Wu=Wu, Wx0=Wx0, Wu0=Wu0, stiff=stiff, WxK_soft=WxK_soft, gxK_soft=gxK_soft, term=term, Wtf=2 * P.w_tr + delta_w + (sig["tf"].sum() if "tf" in sig else 0.0),
                D=D, rho=gnu - (b_ / a_) * gt, gx=gx, gu=gu, gtf=gtf, e=P.dyn_residual(it.X, it.U, it.NU, it.tf),
                cv=cv, avt=avt, Hv=Hv, a_=a_, b_=b_, gt=gt, g=g, sig=sig, zhat=zhat)


def ldl_solve7(M, B):
    """M = Lt diag(d) Lt^T (unit lower Lt, no pivoting, SPD M).  Returns X1 = Lt^-1 B, X2 = Lt^-1 and 1/d, so that
    M^-1 = X2^T diag(1/d) X2.  Written entry by entry in the order the device kernel uses."""
    n = M.shape[0]
    W = M.copy(); Lt = np.eye(n); rd = np.zeros(n)
    for p in range(n):
        d = W[p, p]
        if not d > 0.0: raise np.linalg.LinAlgError("not positive definite")
        rd[p] = 1.0 / d
        col = W[p + 1:, p].copy()
        Lt[p + 1:, p] = col * rd[p]
        W[p + 1:, p + 1:] -= np.outer(col * rd[p], col)
    X = np.hstack([B, np.eye(n)]).astype(float)
    for p in range(n):
        for q in range(p):
            X[p] -= Lt[p, q] * X[q]
    return X[:, :n], X[:, n:], rd


def riccati_factor(P, nb):
    """Backward Riccati sweep in the shifted state y_k = x_k - Bp_{k-1} u_k (absorbs the first-order
    hold), nu_k eliminated per stage through M = D + P_{k+1} (LDL^T, explicit inverse so that the
    linear-term sweeps are pure matrix-vector products).  Terminal Hessian: soft part + capped share of
    the rank-1 barrier weights + augmented-Lagrangian term gamma a_vt a_vt^T (exact, see riccati_solve)."""
    K = P.K; Wx, Wu, D = nb["Wx0"], nb["Wu0"], nb["D"]
    WxK = nb["WxK_soft"].copy()
    win = []
    for a, w, gh in nb["term"]:
        wi = min(w, TERM_CAP); win.append(wi)
        WxK += wi * np.outer(a, a)
    avt = nb["avt"]
    # augmented-Lagrangian weight of the tangential equality inside the recursion.  Any value gives the same direction (the
    # row itself is a border equality); it must be large enough for the recursion's pivots to stay positive whenever the
    # reduced Hessian is positive definite.  Sized like a capped terminal weight: a weight of the order of lam_vt |H_v|
    # covers the terminal node only, and the negative curvature lam_vt H_v leaves at x_K grows a thousandfold on its way
    # back through 30 stages of orbital dynamics (seen with r_des = 3: spurious breakdowns at node 1, delta_w = 0.1..0.5
    # in every iteration of the endgame, MAXITER)
    gam = (TERM_CAP + 10.0 * abs(nb["lam_vt_cur"]) * np.linalg.norm(nb["Hv"])) / (avt @ avt) if P.variant == "exact" else 0.0
    WxK += gam * np.outer(avt, avt)
    F = dict(P=np.zeros((K, 7, 7)), Minv=np.zeros((K, 7, 7)), G=np.zeros((K, 7, 7)), Pt=np.zeros((K, 7, 7)),
             Qi=np.zeros((K, 3, 3)), Kg=np.zeros((K, 3, 7)), Bh=np.zeros((K, 7, 3)), win=win, gam=gam, WxK=WxK)
    I7 = np.eye(7)
    for k in range(K - 1, -1, -1):
        Bpm = P.Bp[k - 1] if k >= 1 else np.zeros((7, 3))
        Wxk = WxK if k == K - 1 else Wx[k]
        if k <= K - 2:
            Pn = F["P"][k + 1]
            X1, X2, rd = ldl_solve7(np.diag(D[:, k]) + Pn, Pn)       # raises LinAlgError when not PD
            C1 = rd[:, None] * X1; C2 = rd[:, None] * X2
            Pt = Pn - X1.T @ C1; Pt = 0.5 * (Pt + Pt.T)
            F["G"][k] = X1.T @ C2                                    # Pn M^-1
            F["Minv"][k] = X2.T @ C2
            Ah = P.A[k]; Bh = P.A[k] @ Bpm + P.Bn[k]
        else:
            Pt = np.zeros((7, 7)); Ah = np.zeros((7, 7)); Bh = np.zeros((7, 3))
        Quu = Wu[k] + Bpm.T @ Wxk @ Bpm + Bh.T @ Pt @ Bh
        Quy = Bpm.T @ Wxk + Bh.T @ Pt @ Ah
        np.linalg.cholesky(Quu)
        Qi = np.linalg.inv(Quu)
        Kg = Qi @ Quy
        Pk = Wxk + Ah.T @ Pt @ Ah - Quy.T @ Kg
        for (cu, cy, ex) in nb["stiff"][k]:
            # Q += ex c c^T with c = (c_u, c_y) in the (u_k, y_k) coordinates (x_k = y_k + Bpm u_k): Sherman-Morrison
            if cu is None: cu = Bpm.T @ cy
            t = Qi @ cu
            om = 1.0 / (1.0 / ex + cu @ t)
            v = cy - Kg.T @ cu
            Pk = Pk + om * np.outer(v, v); Kg = Kg + om * np.outer(t, v); Qi = Qi - om * np.outer(t, t)
        F["P"][k] = 0.5 * (Pk + Pk.T); F["Pt"][k] = Pt; F["Qi"][k] = Qi; F["Kg"][k] = Kg; F["Bh"][k] = Bh
    return F


def riccati_channel(P, nb, F, gx, gu, rho, aff):
    """one linear-term sweep (backward + forward) for given gradients / affine dynamics terms; only
    matrix-vector products with the stored stage matrices"""
    K = P.K; D = nb["D"]
    p = np.zeros((K, 7)); qu = np.zeros((K, 3))
    for k in range(K - 1, -1, -1):
        Bpm = P.Bp[k - 1] if k >= 1 else np.zeros((7, 3))
        if k <= K - 2:
            t = p[k + 1] - F["G"][k] @ (rho[:, k] + p[k + 1]) + F["Pt"][k] @ aff[:, k]
            Ah = P.A[k]
        else:
            t = np.zeros(7); Ah = np.zeros((7, 7))
        qu[k] = gu[:, k] + Bpm.T @ gx[:, k] + F["Bh"][k].T @ t
        p[k] = gx[:, k] + Ah.T @ t - F["Kg"][k].T @ qu[k]
    X = np.zeros((7, K)); U = np.zeros((3, K)); NU = np.zeros((7, K - 1)); LAM = np.zeros((7, K - 1))
    y = np.zeros(7)
    for k in range(K):
        Bpm = P.Bp[k - 1] if k >= 1 else np.zeros((7, 3))
        u = -(F["Kg"][k] @ y) - F["Qi"][k] @ qu[k]
        U[:, k] = u; X[:, k] = y + Bpm @ u
        if k <= K - 2:
            yh = P.A[k] @ y + F["Bh"][k] @ u + aff[:, k]
            nu = -(F["G"][k].T @ yh) - F["Minv"][k] @ (rho[:, k] + p[k + 1])
            NU[:, k] = nu; LAM[:, k] = D[:, k] * nu + rho[:, k]
            y = yh + nu
    return X, U, NU, LAM


def border_ldl_solve(Mb, rb, wex, gex, n_eq=1):
    """The bordered system in the order (vt, zeta_1..5, dtf) -- Mb, rb come in channel order (dtf, vt, zetas) --
    with the zeta rows in their 1/w_excess form (a zeta without excess weight is decoupled: pivot -1, value 0).
    In that order the matrix is symmetric, its constraint-type block is negative definite and dtf's Schur complement
    is positive exactly when the reduced KKT matrix has the inertia of a convex problem: an L D L^T without pivoting
    gives the solution AND, by Sylvester, the inertia ipopt reads off its linear solver (all constraint pivots
    negative, the last positive).  A wrong inertia raises LinAlgError like a breakdown of the recursion and is
    regularised by delta_w; without the check the iteration can alternate between a descent and an ascent direction
    in tf (seen on short-arc references).  Same arithmetic as border_factor / border_solve of the device kernel."""
    n = Mb.shape[0]
    order = list(range(1, n)) + [0]
    S = Mb[np.ix_(order, order)].copy(); r = rb[order].copy()
    for t in range(len(wex)):
        q = n_eq + t
        if wex[t] > 0.0:
            S[q, q] -= 1.0 / wex[t]; r[q] -= gex[t] / wex[t]
        else:
            S[q, :] = 0.0; S[:, q] = 0.0; S[q, q] = -1.0; r[q] = 0.0
    L = np.eye(n); rd = np.zeros(n)
    for p in range(n):
        d = S[p, p]
        if (p < n - 1 and not d < 0.0) or (p == n - 1 and not d > 0.0):
            raise np.linalg.LinAlgError("wrong inertia of the border system")
        rd[p] = 1.0 / d
        for i in range(p + 1, n):
            m = S[i, p] * rd[p]
            S[i, p + 1:] -= m * S[p, p + 1:]
            L[i, p] = m
    S0 = Mb[np.ix_(order, order)].copy()
    for t in range(len(wex)):
        q = n_eq + t
        if wex[t] > 0.0: S0[q, q] -= 1.0 / wex[t]
        else: S0[q, :] = 0.0; S0[:, q] = 0.0; S0[q, q] = -1.0

    def ldl_solve(v):
        v = v.copy()
        for p in range(n):
            v[p + 1:] -= L[p + 1:, p] * v[p]
        v *= rd
        for p in range(n - 1, -1, -1):
            v[p] -= L[p + 1:, p] @ v[p + 1:]
        return v
    x = ldl_solve(r)
    x = x + ldl_solve(r - S0 @ x)                                   # one step of iterative refinement
    sol = np.zeros(n); sol[0] = x[n - 1]; sol[1:] = x[:n - 1]
    return sol


def riccati_solve(P, nb, F, rhs):
    """Solve the reduced KKT system for a right-hand side
    rhs = dict(gx (7,K), gu (3,K), rho (7,K-1), aff (7,K-1), gtf, rvt, gterm (N_TERM,)).
    The border unknowns are dtf, the multiplier of the vt row and, per terminal rank-1 barrier term,
    zeta_j = w_excess_j a_j.dx_K + gterm_excess_j."""
    K = P.K
    Z7 = np.zeros((7, K)); Z3 = np.zeros((3, K)); Zn = np.zeros((7, K - 1))
    term = nb["term"]; win = F["win"]; avt = nb["avt"]
    gx0 = rhs["gx"].copy()
    wex = [w - win[j] for j, (a, w, _) in enumerate(term)]
    # zeta rows in residual form: a.dx_K - dzeta / wex = -rz  (rz = residual of the row at the current direction)
    gex = [rhs["rz"][j] * wex[j] for j in range(len(term))]
    n_eq = 1 if P.variant == "exact" else 0
    gx0[:, K - 1] -= F["gam"] * rhs["rvt"] * avt                      # AL term: gamma (a.dx - rvt) a
    chans = [riccati_channel(P, nb, F, gx0, rhs["gu"], rhs["rho"], rhs["aff"]),
             riccati_channel(P, nb, F, Z7, Z3, Zn, P.Sig)]
    vecs = ([avt] if n_eq else []) + [a for (a, w, gh) in term]
    for a in vecs:
        g1 = Z7.copy(); g1[:, K - 1] = a
        chans.append(riccati_channel(P, nb, F, g1, Z3, Zn, Zn))
    nbd = 1 + len(vecs)
    Mb = np.zeros((nbd, nbd)); rb = np.zeros(nbd)
    Mb[0, 0] = nb["Wtf"]
    for c in range(1, 1 + nbd): Mb[0, c - 1] -= (P.Sig * chans[c][3]).sum()
    rb[0] = -rhs["gtf"] + (P.Sig * chans[0][3]).sum()
    if P.fixed_tf is not None:                         # dtf = 0: its row and column leave the border (pivot +1)
        Mb[0, :] = 0.0; Mb[:, 0] = 0.0; Mb[0, 0] = 1.0; rb[0] = 0.0
    for i, a in enumerate(vecs):
        for c in range(1, 1 + nbd): Mb[1 + i, c - 1] += a @ chans[c][0][:, K - 1]
        rb[1 + i] = -(a @ chans[0][0][:, K - 1])
    if n_eq: rb[1] += rhs["rvt"]
    sol = border_ldl_solve(Mb, rb, wex, gex, n_eq)
    comb = lambda i: chans[0][i] + sum(sol[c - 1] * chans[c][i] for c in range(1, 1 + nbd))
    return dict(X=comb(0), U=comb(1), NU=comb(2), lam=comb(3), tf=sol[0], lam_vt=sol[1] if n_eq else 0.0,
                zeta=sol[1 + n_eq:].copy())


def reduced_residual(P, nb, it, d, win):
    """right-hand side minus reduced-KKT-matrix times d, in the layout riccati_solve takes"""
    K = P.K
    dX, dU, dtf, dNU, dl, dlv = d["X"], d["U"], d["tf"], d["NU"], d["lam"], d["lam_vt"]
    Wx, Wu = nb["Wx"], nb["Wu"]
    gx = np.zeros((7, K)); gu = np.zeros((3, K))
    for k in range(1, K):
        if k == K - 1:
            v = nb["WxK_soft"] @ dX[:, k] + nb["avt"] * (it.lam_vt + dlv)
            grad = nb["gxK_soft"]
        else:
            v = Wx[k] @ dX[:, k]; grad = nb["gx"][:, k]
        v = v + (it.lam[:, k - 1] + dl[:, k - 1])
        if k <= K - 2: v = v - P.A[k].T @ (it.lam[:, k] + dl[:, k])
        gx[:, k] = grad + v
    # terminal rank-1 terms: the share win a a^T stays a Hessian term, the excess lives in zeta (kept as an unknown of the
    # linear solve): x_K row  += (gh share + win a.dx + zeta) a ;  zeta row  a.dx - zeta / wex + gh / w  (all O(1) entries)
    rz = np.zeros(len(nb["term"])); zeta = d["zeta"]
    for j, (a, w, gh) in enumerate(nb["term"]):
        adx = a @ dX[:, K - 1]
        wex = w - win[j]
        share = win[j] / w if w > 0 else 1.0
        gx[:, K - 1] += (gh * share + win[j] * adx + (zeta[j] if wex > 0 else 0.0)) * a
        rz[j] = adx - zeta[j] / wex + gh / w if wex > 0 else 0.0
    gtf = nb["gtf"] + nb["Wtf"] * dtf
    for k in range(K):
        v = Wu[k] @ dU[:, k]
        if k <= K - 2: v = v - P.Bn[k].T @ (it.lam[:, k] + dl[:, k])
        if k >= 1: v = v - P.Bp[k - 1].T @ (it.lam[:, k - 1] + dl[:, k - 1])
        gu[:, k] = nb["gu"][:, k] + v
    rho = nb["rho"] + nb["D"] * dNU - (it.lam + dl)
    aff = np.zeros((7, K - 1))
    for k in range(K - 1):
        gtf -= P.Sig[:, k] @ (it.lam[:, k] + dl[:, k])
        aff[:, k] = -nb["e"][:, k] - (dX[:, k + 1] - P.A[k] @ dX[:, k] - P.Bn[k] @ dU[:, k]
                                      - P.Bp[k] @ dU[:, k + 1] - P.Sig[:, k] * dtf - dNU[:, k])
    rvt = -nb["cv"] - nb["avt"] @ dX[:, K - 1]
    return dict(gx=gx, gu=gu, rho=rho, aff=aff, gtf=gtf, rvt=rvt, rz=rz)


def newton_direction(P, it, mu, delta_w=0.0, n_refine=1):
    nb = newton_blocks(P, it, mu, delta_w)
    nb["lam_vt_cur"] = it.lam_vt
    K = P.K
    F = riccati_factor(P, nb)
    zero = dict(X=np.zeros((7, K)), U=np.zeros((3, K)), NU=np.zeros((7, K - 1)), tf=0.0,
                lam=-it.lam.copy(), lam_vt=-it.lam_vt, zeta=np.zeros(len(nb["term"])))   # so that lam + dlam = 0: rhs has no multipliers
    d = zero
    # refinement once a barrier weight z/s (terminal rank-1 terms, stage balls and planes, tf bounds) costs digits
    stiff = max(max(w for (a, w, gh) in nb["term"]), max(nb["sig"][k].max() for k in ("u", "rmax", "rmin", "tf") if k in nb["sig"]))
    passes = 1 + (n_refine if stiff > REFINE_TW else 0)
    for _ in range(passes):
        rhs = reduced_residual(P, nb, it, d, F["win"])
        c = riccati_solve(P, nb, F, rhs)
        d = dict(X=d["X"] + c["X"], U=d["U"] + c["U"], NU=d["NU"] + c["NU"], tf=d["tf"] + c["tf"],
                 lam=d["lam"] + c["lam"], lam_vt=d["lam_vt"] + c["lam_vt"], zeta=d["zeta"] + c["zeta"])
    return finish_direction(P, it, nb, d)


def finish_direction(P, it, nb, d):
    """back-substitution for dt, ds, dz"""
    K = P.K; X, U = it.X, it.U
    d["T"] = (-nb["gt"] - nb["b_"] * d["NU"]) / nb["a_"]
    dg = {"u": (2 * U * d["U"]).sum(0),
          "rmax": (2 * X[:3, 1:] * d["X"][:3, 1:]).sum(0),
          "rmin": -(P.rbar_hat[:, 1:] * d["X"][:3, 1:K - 1]).sum(0),
          "term": P.aT @ d["X"][:, K - 1],
          "rfmax": np.array([(2 * X[:3, K - 1] * d["X"][:3, K - 1]).sum()]),
          "tp": d["NU"] - d["T"], "tn": -d["NU"] - d["T"],
          "tf": np.array([-d["tf"], d["tf"]])}
    if "tf" not in nb["g"]: del dg["tf"]
    g, sig, zhat = nb["g"], nb["sig"], nb["zhat"]
    d["s"] = {k: -(g[k] + it.s[k]) - dg[k] for k in g}
    d["z"] = {k: zhat[k] + sig[k] * dg[k] - it.z[k] for k in g}
    return d


def newton_direction_dense(P, it, mu, delta_w=0.0):
    """Independent cross-check of newton_direction: assemble the reduced KKT matrix densely
    (terminal barrier terms inside the Hessian, no border, no AL term) and call LAPACK."""
    nb = newton_blocks(P, it, mu, delta_w)
    K = P.K
    Wx = nb["Wx"].copy(); Wx[K - 1] = nb["WxK_soft"].copy()
    gx = nb["gx"].copy()
    for a, w, gh in nb["term"]: Wx[K - 1] += w * np.outer(a, a)
    nx = 7 * (K - 1); nu_ = 3 * K; npr = nx + nu_ + 1; nd = 7 * (K - 1) + 1
    M = np.zeros((npr + nd, npr + nd)); rhs = np.zeros(npr + nd)
    ix = lambda k: slice(7 * (k - 1), 7 * k)
    iu = lambda k: slice(nx + 3 * k, nx + 3 * k + 3)
    itf = nx + nu_
    il = lambda k: slice(npr + 7 * k, npr + 7 * k + 7)
    ivt = npr + 7 * (K - 1)
    for k in range(1, K): M[ix(k), ix(k)] = Wx[k]; rhs[ix(k)] = -gx[:, k]
    for k in range(K): M[iu(k), iu(k)] = nb["Wu"][k]; rhs[iu(k)] = -nb["gu"][:, k]
    M[itf, itf] = nb["Wtf"]; rhs[itf] = -nb["gtf"]
    fixed = P.fixed_tf is not None
    for k in range(K - 1):
        J = np.zeros((7, npr)); J[:, ix(k + 1)] = np.eye(7)
        if k >= 1: J[:, ix(k)] = -P.A[k]
        J[:, iu(k)] = -P.Bn[k]; J[:, iu(k + 1)] = -P.Bp[k]; J[:, itf] = -P.Sig[:, k]
        M[il(k), :npr] = J; M[:npr, il(k)] = J.T
        M[il(k), il(k)] = -np.diag(1.0 / nb["D"][:, k])
        rhs[il(k)] = -nb["e"][:, k] - nb["rho"][:, k] / nb["D"][:, k]
    if P.variant == "exact":
        M[ivt, ix(K - 1)] = nb["avt"]; M[ix(K - 1), ivt] = nb["avt"]; rhs[ivt] = -nb["cv"]
    else:
        M[ivt, ivt] = -1.0                                          # no equality row: decoupled placeholder (keeps the inertia count)
    if fixed:
        M[itf, :] = 0.0; M[:, itf] = 0.0; M[itf, itf] = 1.0; rhs[itf] = 0.0
    sol = np.linalg.solve(M, rhs)
    d = dict(X=np.zeros((7, K)), U=sol[nx:nx + nu_].reshape(K, 3).T.copy(), tf=sol[itf])
    d["X"][:, 1:] = sol[:nx].reshape(K - 1, 7).T
    lam_new = sol[npr:npr + 7 * (K - 1)].reshape(K - 1, 7).T
    d["NU"] = (lam_new - nb["rho"]) / nb["D"]
    d["lam"] = lam_new - it.lam; d["lam_vt"] = sol[ivt] - it.lam_vt
    ev = np.linalg.eigvalsh(M)
    d["inertia_ok"] = bool((ev > 0).sum() == npr and (ev < 0).sum() == nd)
    return finish_direction(P, it, nb, d)


# ------------------------------------------------------------------------------------------
# Interior-point iteration
# ------------------------------------------------------------------------------------------
def step(it, d, a):
    n = it.copy()
    n.X = it.X + a * d["X"]; n.U = it.U + a * d["U"]; n.tf = it.tf + a * d["tf"]
    n.NU = it.NU + a * d["NU"]; n.T = it.T + a * d["T"]
    n.lam = it.lam + a * d["lam"]; n.lam_vt = it.lam_vt + a * d["lam_vt"]
    n.s = {k: it.s[k] + a * d["s"][k] for k in it.s}
    n.z = {k: it.z[k] + a * d["z"][k] for k in it.z}
    return n


def candidate(P, it, d, a, mu_clip, prm=FAST):
    """the iterate a step of length a would give: it + a d with the slacks reset to s >= -g and the multipliers
    pulled back to z <= kappa mu / s (upper side only, see the header).  The line search tests this point."""
    n = step(it, d, a)
    g = P.ineq(n.X, n.U, n.NU, n.T, n.tf)
    for k in n.s:
        n.s[k] = np.maximum(n.s[k], -g[k])
        n.z[k] = np.minimum(n.z[k], prm["kappa_sigma"] * mu_clip / n.s[k])
        if prm["kappa_two_sided"]: n.z[k] = np.maximum(n.z[k], mu_clip / (prm["kappa_sigma"] * n.s[k]))
    return n


def solve(P, tol=1e-8, max_iter=200, acceptable_tol=1e-6, acceptable_iter=15, n_refine=1, dense=False,
          verbose=False, start="ref", mode="fast"):
    """Returns dict(X (7,K), U (3,K), NU (7,K), tf, status, iters, kkt, objective, n_regularised = number of
    iterations whose factorisation broke down and needed delta_w > 0, first_regularised = index of the first, -1 if none)."""
    prm = FAST if mode == "fast" else IPOPT_DEFAULT
    viol = P.structural_violation()
    if viol > 0.0:       # empty constraint set: reported at once, the reference trajectory handed back unchanged
        Z = np.zeros((7, P.K))
        return dict(X=P.xbar.copy(), U=P.ubar.copy(), NU=Z, T=Z.copy(), tf=P.tfbar if P.fixed_tf is None else P.fixed_tf,
                    status=ST_INFEASIBLE, iters=0, n_regularised=0, first_regularised=-1, kkt=viol, objective=np.nan,
                    iterate=None, g_tf=0.0)
    it = initial_iterate(P, start, prm)
    mu = it.mu0
    n_acc = 0; status = ST_MAXITER; k_it = 0
    mono = prm["mu_strategy"] != "adaptive"; n_small = 0
    dw_last = 0.0; n_reg = 0; first_reg = -1
    for k_it in range(max_iter + 1):
        E0 = optimality_error(P, it, 0.0)[0]
        if verbose: print(f"it {k_it:3d} E0 {E0:.2e} tf {it.tf:.8f}")
        if not np.isfinite(E0): status = ST_NUMERIC; break
        if E0 <= tol: status = ST_OK; break
        n_acc = n_acc + 1 if E0 <= acceptable_tol else 0
        if n_acc >= acceptable_iter: status = ST_ACCEPTABLE; break
        if k_it == max_iter: status = ST_ACCEPTABLE if E0 <= acceptable_tol else ST_MAXITER; break
        mu_cur = sum((it.s[k] * it.z[k]).sum() for k in it.s) / sum(v.size for v in it.s.values())
        if not mono and n_small >= FB_N:
            mono = True
            mu = max(tol / 10, min(MU_INIT, FB_BOOST * mu_cur))
        if not mono:
            mu_t = min(SIGMA * mu_cur, mu_cur * np.sqrt(mu_cur)) if it.clean else SIGMA * mu_cur
            mu = max(mu_t, tol / 10, MU_ERR * E0)
        else:
            # Fiacco-McCormick: the barrier problem is solved to E_mu <= kappa_eps mu (kappa_eps = 10) before mu moves on
            # to max(tol / 10, min(kappa_mu mu, mu^theta_mu)), kappa_mu = 0.2, theta_mu = 1.5 (Waechter & Biegler eq. (7))
            while mu > tol / 10 and optimality_error(P, it, mu)[0] <= 10.0 * mu:
                mu = max(tol / 10, min(0.2 * mu, mu ** 1.5))
        # Hessian regularisation on breakdown: ipopt's inertia-correction schedule (Waechter & Biegler 2006, Alg. IC):
        # delta_w = 0 first, then a third of the last successful value (1e-4 the first time), growing by 8
        # (by 100 until some value has worked), giving up above 1e40
        d = None; dw = 0.0
        while True:
            try:
                d = newton_direction_dense(P, it, mu, dw) if dense else newton_direction(P, it, mu, dw, n_refine if dw == 0 else 0)
                if all(np.isfinite(d[k]).all() for k in ("X", "U", "NU")) and np.isfinite(d["tf"]): break
                d = None
            except np.linalg.LinAlgError:
                d = None
            if dw == 0.0: dw = DW_FIRST if dw_last == 0.0 else max(DW_MIN, dw_last / 3.0)
            else: dw *= 100.0 if dw_last == 0.0 else 8.0
            if dw > DW_MAX: break
        if d is None: status = ST_NUMERIC; break
        if dw > 0.0:
            dw_last = dw; n_reg += 1
            if first_reg < 0: first_reg = k_it
        tau = max(0.99, 1 - mu)
        a = 1.0
        for v, dv in ((it.s, d["s"]), (it.z, d["z"])):
            for k in v:
                neg = dv[k] < 0
                if neg.any(): a = min(a, (-tau * v[k][neg] / dv[k][neg]).min())
        r0 = residual_norm(P, it, mu)
        mu_clip = max(mu, mu_cur)
        n = None
        for ls in range(30):
            if 0.5 * a < ALPHA_FLOOR: n = None; break   # a rejection could not shorten the step any more: take it
            n = candidate(P, it, d, a, mu_clip, prm)
            prod = np.concatenate([(n.s[k] * n.z[k]).ravel() for k in n.s])
            if residual_norm(P, n, mu) <= (1 - 1e-4 * a) * r0 and prod.min() >= GAMMA_NBHD * min(mu, prod.mean()):
                break
            a *= 0.5
        n_small = n_small + 1 if a < FB_ALPHA else 0
        if verbose: print(f"       mu {mu:.2e} step {a:.4f} delta_w {dw:.1e}{' (monotone)' if mono else ''}")
        it = n if n is not None else candidate(P, it, d, a, mu_clip, prm)
    K = P.K
    NU = np.zeros((7, K)); NU[:, :K - 1] = it.NU
    T = np.zeros((7, K)); T[:, :K - 1] = it.T
    # this satellite's term of the tf stationarity row: d/dtf of its optimal value at fixed tf (envelope theorem)
    g_tf = 2 * P.w_tr * (it.tf - P.tfbar) - sum(P.Sig[:, k] @ it.lam[:, k] for k in range(K - 1))
    return dict(X=it.X, U=it.U, NU=NU, T=T, tf=it.tf, status=status, iters=k_it, n_regularised=n_reg, first_regularised=first_reg,
                kkt=optimality_error(P, it, 0.0)[0], objective=P.objective(it.X, it.U, it.T, it.tf), iterate=it, g_tf=g_tf)


def shared_tf_root(G, tf_max, tf0, gtol=1e-7, xtol=2e-8):
    """The scalar outer problem of the shared-tf decomposition: the root of the tf stationarity row G(tf) = 1 + sum_s g_s(tf)
    on (0, tf_max] (G increasing: tf enters convexly), or tf_max when G(tf_max) <= 0 (the range constraint active).
    Starts at the reference final time, walks towards the root with doubling steps until the sign changes (the inner
    problems are linearised around tf_bar: far from it they are hard and never needed), then a bracketing secant
    (Illinois).  Each G costs one batched inner solve; its accuracy is that of the inner multipliers (~1e-6), hence gtol.
    Returns (tf, evaluations [(tf, G)]).  The device host code (optimizer.py) runs the same procedure."""
    ev = []
    def g(t):
        v = G(t); ev.append((t, v)); return v
    a = min(tf0, tf_max); ga = g(a)
    if abs(ga) <= gtol or (a == tf_max and ga <= 0.0): return a, ev
    h = 0.05 * a
    while True:
        b = a - h if ga > 0.0 else a + h
        b = min(max(b, 0.05 * a), tf_max)
        gb = g(b)
        if abs(gb) <= gtol: return b, ev
        if (ga > 0.0) != (gb > 0.0): break
        if b == tf_max and gb <= 0.0: return tf_max, ev
        a, ga = b, gb; h *= 2.0
        if len(ev) > 40: return b, ev
    lo, glo, hi, ghi = (a, ga, b, gb) if ga < 0.0 else (b, gb, a, ga)
    side = 0; t = 0.5 * (lo + hi)
    for _ in range(40):
        if hi - lo <= xtol: break
        t_prev = t
        t = (lo * ghi - hi * glo) / (ghi - glo)
        if abs(t - t_prev) <= 1e-9 * max(1.0, abs(t)): break       # G carries the noise of the inner multipliers: no finer root
        gt = g(t)
        if abs(gt) <= gtol: return t, ev
        if gt > 0.0:
            hi, ghi = t, gt
            if side == 1: glo *= 0.5
            side = 1
        else:
            lo, glo = t, gt
            if side == -1: ghi *= 0.5
            side = -1
    return t, ev


def solve_shared_tf(problems, tf_max, **kw):
    """Several satellites in one Optimizer share ONE final time (optimizer.py:287,311,322,336; get_solved_tf ignores s,
    :199-203).  The NLP then separates given tf: min_tf [ tf + sum_s V_s(tf) ], V_s = satellite s's problem at fixed tf
    (its trust-region term w_tr (tf - tf_bar)^2 included), V_s'(tf) = g_tf of the inner solution.  The KKT conditions of
    the monolithic NLP are exactly: every inner problem's KKT conditions + 1 + sum_s g_s = 0 (or tf on its bound)."""
    import copy
    def G(t):
        tot = 1.0
        for P in problems:
            Q = copy.copy(P); Q.fixed_tf = float(t)
            r = solve(Q, **kw)
            assert r["status"] in (ST_OK, ST_ACCEPTABLE), r["status"]
            tot += r["g_tf"]
        return tot
    tf, ev = shared_tf_root(G, tf_max, problems[0].tfbar)
    out = []
    for P in problems:
        Q = copy.copy(P); Q.fixed_tf = float(tf)
        out.append(solve(Q, **kw))
    return tf, out, ev
