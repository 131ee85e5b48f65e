// solve_driver.hpp -- one satellite from its problem constants to its results (solve_satellite: the interior-point
// iteration that calls the phases of solve_phases.hpp and solve_riccati.hpp) and the kernels' LDS working set.
#pragma once

#ifdef MPCX_WS_LDS
extern __shared__ double mpcx_ws_lds[];      // the workgroup's dynamic LDS: the satellite's working set (sat_view)
#endif

namespace MPCX_NS {
#ifdef MPCX_PHASE_TIMING
#define PT_DECL unsigned long long pt_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, pt0_ = 0; unsigned pc_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; \
    const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime(), mt0_ = __builtin_amdgcn_s_memtime();
#define PT_BEGIN pt0_ = __builtin_amdgcn_s_memtime();
#define PT_END(i) { pt_[i] += __builtin_amdgcn_s_memtime() - pt0_; pc_[i]++; }
#else
#define PT_DECL
#define PT_BEGIN
#define PT_END(i)
#endif

#ifndef MPCX_SOLVE_WAVES
#define MPCX_SOLVE_WAVES 2     // waves per SIMD the register allocation is bounded for (256 registers; 3 was measured slower)
#endif

// The two kernels' LDS working set: ONE pair of module-scope objects, so that it sits at the same LDS address in both and
// the out-of-line phase functions (which take it by reference) keep addressing it with compile-time offsets -- with a
// pair per kernel the addresses reach them as run-time pointers (measured: solve_kernel 6.85 -> 8.4 ms at S4096).
#ifndef MPCX_TP          // (the time-parallel build defines them ahead of its own functions: solve_tp.hpp)
__shared__ SatData g_sd;
__shared__ Scratch g_w;
// ... and the view of the satellite's problem and workspace (round 5): until then a struct on the driver's stack that every phase
// function received by reference -- its ~50 dwords were read back with flat loads (10-15 per call, some 135 per iteration) and
// the driver's pointer swaps went to scratch.  In LDS, at one address in every kernel, the phase functions address it with
// compile-time offsets like the other two objects.
__shared__ Sat g_s;
#endif

// View of satellite `sat`'s problem and of workspace slot `slot` (K: its node count, Kmax: the row length of the arrays)
__device__ __forceinline__ Sat sat_view(const SolveArgs &a, const int sat, const int slot, const int K, const int Kmax)
{
    Sat s;
    s.K = K; s.ldk = Kmax;
    s.stage = (cgf64 *)a.stage + (size_t)sat * (Kmax - 1) * MPCX_STAGE_DOUBLES;
    s.xbar = (cgf64 *)a.xbar + (size_t)sat * 7 * Kmax;
    s.ubar = (cgf64 *)a.ubar + (size_t)sat * 3 * Kmax;
    const int KP = padded_nodes(K);
    s.KP = KP;
#ifdef MPCX_WS_LDS
    // LDS-resident build: iterate, candidate, direction, r-hat, Newton / factor / channel records and the globals are carved
    // from the workgroup's dynamic LDS (lds_ws_doubles(K): 134 KB at K = 30); the stage copy, the Newton scalars and the channel
    // trajectories keep their places in the slot's global workspace (whose layout is the other builds')
    wf64 *wl = (wf64 *)mpcx_ws_lds;
    s.ws = wl;
    s.it = wl; wl += (size_t)KP * IT_N;
    s.dr = wl; wl += (size_t)KP * IT_N;
    s.itB = wl; wl += (size_t)KP * IT_N;
    s.rbh = wl; wl += (size_t)KP * 3;
    s.nb = wl; wl += (size_t)K * NB_N;
    s.fac = wl; wl += (size_t)K * FAC_N;
    s.ch = wl; wl += (size_t)K * CH_N;
    s.itg = (lf64 *)wl; wl += GL_N;
    s.drg = (lf64 *)wl; wl += GL_N;
    s.itgB = (lf64 *)wl; wl += GL_N;
    s.sink = wl;
    s.o_fac = (int)(KP * (3 * IT_N + 3) + K * NB_N);
    s.o_ch = s.o_fac + K * FAC_N; s.o_sink = s.o_ch + K * CH_N + 3 * GL_N;
    gf64 *ws = (gf64 *)a.ws + (size_t)slot * a.ws_stride;
    s.wsg = ws;
    s.nbs = ws + (size_t)KP * 3 * IT_N;
    s.stT = s.nbs + (size_t)KP * NS_N;
    const int o_ch_g = (int)(KP * (3 * IT_N + NS_N + MPCX_STAGE_DOUBLES + 3) + K * NB_N) + K * FAC_N;
    s.o_traj = o_ch_g + K * CH_N; s.o_sinkg = s.o_traj + K * NCH * TR_N + 3 * GL_N;
    s.traj = ws + s.o_traj;
#else
    gf64 *ws = (gf64 *)a.ws + (size_t)slot * a.ws_stride;
    s.ws = ws;
    s.it = ws; ws += (size_t)KP * IT_N;
    s.dr = ws; ws += (size_t)KP * IT_N;
    s.itB = ws; ws += (size_t)KP * IT_N;
    s.nbs = ws; ws += (size_t)KP * NS_N;
    s.stT = ws; ws += (size_t)KP * MPCX_STAGE_DOUBLES;
    s.rbh = ws; ws += (size_t)KP * 3;
    s.nb = ws; ws += (size_t)K * NB_N;
    s.fac = ws; ws += (size_t)K * FAC_N;
    s.ch = ws; ws += (size_t)K * CH_N;
    s.traj = ws; ws += (size_t)K * NCH * TR_N;
    // (the global part of iterate / direction / candidate lives in LDS, SatData::gl; its three slots in the workspace stay
    //  where they were, unused, so that every offset of the layout is the other rounds')
    s.itg = (lf64 *)g_sd.gl[0]; s.drg = (lf64 *)g_sd.gl[1]; s.itgB = (lf64 *)g_sd.gl[2];
    ws += 3 * GL_N;
    s.sink = ws;
    // (offsets as integers computed from the layout, not as pointer differences: the compiler would fold base + (sink -
    //  base) back into a second pointer and emit a branch with one store per path)
    s.o_fac = (int)(KP * (3 * IT_N + NS_N + MPCX_STAGE_DOUBLES + 3) + K * NB_N);
    s.o_ch = s.o_fac + K * FAC_N; s.o_traj = s.o_ch + K * CH_N; s.o_sink = s.o_traj + K * NCH * TR_N + 3 * GL_N;
#ifdef MPCX_TP
    // behind the other kernels' layout for the call's row length (ws_doubles_tp): the extra backward record, the second
    // trajectory bank, the mailbox of the satellite's workgroups and their exchange records
    s.chx = s.ws + tp_extras_offset(Kmax); s.trajx = s.chx + (size_t)Kmax * CHX_N;
    s.o_trajx = (int)(tp_extras_offset(Kmax) + (size_t)Kmax * CHX_N); s.o_chx = (int)tp_extras_offset(Kmax);
    s.mail = (int *)(s.ws + tp_mail_offset(Kmax)); s.xch = s.ws + tp_mail_offset(Kmax) + TP_MAIL_N;
#endif
#endif
    return s;
}

// Shared tf: one residual evaluation of the whole launch from the satellites' own (grid_reduce) plus the rows and pairs
// that belong to the launch: tf's stationarity row 1 + sum_s g_s - z_0 + z_1 and the two sides of its range constraint
// (optimizer.py:588) with slacks gs and multipliers gz.
__device__ __forceinline__ void shared_fold(GridSync &g, ResAcc &r, double tf, const double (&b_tf)[2], const double (&gs)[2], const double (&gz)[2],
                                            double mu, int lane)
{
    double v[GR_N];
    gr_clear(v);
    v[0] = r.sq; v[1] = r.zsum; v[2] = r.lsum; v[3] = r.prod_sum; v[4] = r.g_tf;
    v[GR_SUM] = r.dual_max; v[GR_SUM + 1] = r.prim_max; v[GR_SUM + 2] = r.prod_max;
    v[GR_SUM + GR_MAX] = r.prod_min;
    grid_reduce(g, v, lane);
    r.sq = v[0]; r.zsum = v[1]; r.lsum = v[2]; r.prod_sum = v[3]; r.g_tf = v[4];
    r.dual_max = v[GR_SUM]; r.prim_max = v[GR_SUM + 1]; r.prod_max = v[GR_SUM + 2]; r.prod_min = v[GR_SUM + GR_MAX];
    const double gtf = 1.0 + r.g_tf - gz[0] + gz[1];
    r.dual_max = fmax(r.dual_max, fabs(gtf)); r.sq += gtf * gtf;
    const double gv[2] = {-tf - b_tf[0], tf - b_tf[1]};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double pr = gv[j] + gs[j], sz = gs[j] * gz[j], q = sz - mu;
        r.prim_max = fmax(r.prim_max, fabs(pr)); r.sq += pr * pr + q * q;
        r.zsum += fabs(gz[j]); r.prod_min = fmin(r.prod_min, sz); r.prod_max = fmax(r.prod_max, sz); r.prod_sum += sz;
    }
}

// One satellite from the problem constants to its results; `slot` selects the workspace (see solve_kernel).
// SHARED (solve_shared_kernel): the satellites of the launch share ONE final time (several satellites in one reference
// Optimizer, optimizer.py:287,311,322,336).  Every workgroup runs this same iteration in lock step: barrier parameter, step
// length, line-search decisions, regularisation and the convergence test come from launch-wide reductions (grid_reduce),
// the tf row of the Newton system is assembled across the launch (border_solve_shared), and the launch-wide variables --
// tf's range-constraint slacks and multipliers -- are carried identically by every workgroup (gs, gz below).
template <bool SHARED>
__device__ __forceinline__ void solve_satellite(const SolveArgs &a, const int sat, const int slot, SatData &sd, Scratch &w, const int lane, GridSync *gsync = nullptr)
{
    PT_DECL
    const int Kmax = a.K;
    const int K = uni(a.Ks ? a.Ks[sat] : Kmax);   // (wave-uniform: one satellite per workgroup)
    if (K < 3 || K > Kmax) {                      // ragged batch with a node count the solver cannot take
        // defined results all the same (as on the INFEASIBLE exit): the reference rows back, no virtual control, tf_bar
        cgf64 *xb = (cgf64 *)a.xbar + (size_t)sat * 7 * Kmax, *ub = (cgf64 *)a.ubar + (size_t)sat * 3 * Kmax;
        for (int e = lane; e < 7 * Kmax; e += 64) { a.X[(size_t)sat * 7 * Kmax + e] = xb[e]; a.NU[(size_t)sat * 7 * Kmax + e] = 0.0; }
        for (int e = lane; e < 3 * Kmax; e += 64) a.U[(size_t)sat * 3 * Kmax + e] = ub[e];
        if (lane == 0) {
            a.tf_out[sat] = (a.o.fixed_tf && !SHARED) ? 0.0 : a.tfbar[sat];
            a.status[sat] = MPCX_ST_BADK; a.iters[sat] = 0; a.kkt[sat] = 0.0;
            if (a.nreg) { a.nreg[2 * sat] = 0; a.nreg[2 * sat + 1] = -1; }
        }
        return;
    }
    { const Sat sv = sat_view(a, sat, slot, K, Kmax); if (lane == 0) g_s = sv; }
    WG_SYNC();
    Sat &s = g_s;                                 // (one copy per workgroup, in LDS: see g_s)
    const int KP = uni(s.KP);
    const SolveOpts &o = a.o;

    // ---- problem constants (constraint terms) and the initial iterate ----
    PT_BEGIN
    if (lane == 0) {
        double xK[7];
        for (int i = 0; i < 7; ++i) xK[i] = s.xbar[(size_t)i * Kmax + K - 1];
        build_terminal(xK, a.consts[(size_t)sat * MPCX_NCONST + MPCX_C_MU], a.r_des[sat], o, sd);
        sd.tfbar = a.tfbar[sat];
        double x0[3];
        for (int i = 0; i < 3; ++i) x0[i] = s.xbar[(size_t)i * Kmax];
        sd.infeas = structural_violation(x0, K, sd);
#ifdef MPCX_PHASE_TIMING
        for (int i = 0; i < 16; ++i) sd.fpt[i] = 0;
#endif
    }
    WG_SYNC();
    PT_END(11)                      // (timing build: the start-up in three parts -- terms | transposition | start point)
    PT_BEGIN
    double gr[GR_N];                // (shared tf: operands / results of the launch-wide reductions)
    if (SHARED) {
        // the launch is ONE problem: empty if any satellite's constraint set is, or tf's own range (which build_terminal
        // leaves out of the per-satellite check when tf is not that satellite's variable)
        gr_clear(gr);
        gr[GR_SUM] = fmax(sd.infeas, -(sd.b_tf[0] + sd.b_tf[1]));
        grid_reduce(*gsync, gr, lane);
        if (lane == 0) sd.infeas = gsync->aborted ? 1.0 : gr[GR_SUM];
        WG_SYNC();
    }
    if (sd.infeas > 0.0) {      // empty constraint set: the reference trajectory goes back unchanged, no iteration is spent
        for (int e = lane; e < 7 * Kmax; e += 64) { a.X[(size_t)sat * 7 * Kmax + e] = s.xbar[e]; a.NU[(size_t)sat * 7 * Kmax + e] = 0.0; }
        for (int e = lane; e < 3 * Kmax; e += 64) a.U[(size_t)sat * 3 * Kmax + e] = s.ubar[e];
        if (lane == 0) {
            if (!sd.fixed_tf || SHARED) a.tf_out[sat] = sd.tfbar; else a.tf_out[sat] = 0.0;      // (fixed tf: the slot returns g_s)
            a.status[sat] = (SHARED && gsync->aborted) ? MPCX_ST_NUMERIC : MPCX_ST_INFEASIBLE; a.iters[sat] = 0; a.kkt[sat] = sd.infeas;
            if (a.nreg) { a.nreg[2 * sat] = 0; a.nreg[2 * sat + 1] = -1; }
        }
        return;
    }
    // field-major copy of the stage records for the node-parallel phases (read every iteration, written once): 16 records
    // at a time through LDS -- read as one contiguous block, written field by field with 16 consecutive nodes in
    // consecutive lanes (straight from the record order it was an 8-byte store per cache line)
    // (round 5: the next block's loads issued before a block's stores and wave-level fences in place of the barriers -- one memory
    //  round trip instead of four -- changes nothing under load, 4.85 / 4.87 ms at S4096_K30: profiles/r05/streaming_phases_ab.txt)
    {
        double *stg = (double *)&w;
        static_assert(sizeof(Scratch) >= 16 * MPCX_STAGE_DOUBLES * sizeof(double), "stage transposition buffer");
        for (int k0 = 0; k0 < K - 1; k0 += 16) {
            const int nk = (K - 1 - k0 < 16) ? K - 1 - k0 : 16;
            cgf64 *rec = s.A(k0);
            for (int e = lane; e < nk * MPCX_STAGE_DOUBLES; e += 64) stg[e] = rec[e];
            WG_SYNC();
            for (int e = lane; e < 16 * MPCX_STAGE_DOUBLES; e += 64) {
                const int f = e >> 4, kl = e & 15;
                if (kl < nk) s.stT[f * KP + k0 + kl] = stg[kl * MPCX_STAGE_DOUBLES + f];
            }
            WG_SYNC();
        }
    }
    PT_END(9)
    PT_BEGIN
    bool pushed = false;
    for (int k = lane; k < K; k += 64) {
        double x[7], u[3];
        for (int i = 0; i < 7; ++i) x[i] = s.xbar[(size_t)i * s.ldk + k];
        for (int i = 0; i < 3; ++i) u[i] = s.ubar[(size_t)i * s.ldk + k];
        const double rn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        const auto rb = s.rbn(k);
        for (int i = 0; i < 3; ++i) rb[i] = x[i] / rn;           // optimizer.py:129-130
        const auto p = s.itn(k);
        // (only the rows of the iterate that nothing below writes are zeroed -- nu, lambda, and t of the terminal node -- and the
        //  direction not at all: until the first solve has written it, it is read only where a step length of 0 masks it
        //  (trial_value).  Rounds 1-4 stored all 2 x 66 rows of zeros here: 130 of the start-up's 190 store instructions)
        for (int i = 0; i < 7; ++i) { p[I_NU + i] = 0.0; p[I_LAM + i] = 0.0; }
        if (k == K - 1) for (int i = 0; i < 7; ++i) p[I_T + i] = 0.0;
        for (int i = 0; i < 7; ++i) p[I_X + i] = x[i];
        for (int i = 0; i < 3; ++i) p[I_U + i] = u[i];
        // slacks pushed into the interior (bound_push); the multipliers follow below, once the start value of mu is known
        const double pu = kBoundPush * fmax(1.0, fabs(sd.b_u)), su = -(u[0] * u[0] + u[1] * u[1] + u[2] * u[2] - sd.b_u);
        const double pmax = kBoundPush * fmax(1.0, fabs(sd.b_rmax)), smax = -(rn * rn - sd.b_rmax);
        const double pmin = kBoundPush * fmax(1.0, fabs(sd.b_rmin)), smin = -(-(rb[0] * x[0] + rb[1] * x[1] + rb[2] * x[2]) - sd.b_rmin);
        p[I_SU] = fmax(su, pu); p[I_SRMAX] = fmax(smax, pmax); p[I_SRMIN] = fmax(smin, pmin);
        // (the constraints proper: thrust ball k = 0..K-1, r_max ball k = 1..K-1, r_min plane k = 1..K-2)
        if (su < pu || (k >= 1 && smax < pmax) || (k >= 1 && k <= K - 2 && smin < pmin)) pushed = true;
    }
    // A clean start ("Solver algorithm"): the reference strictly inside its stage constraints and the tf range
    // begins at mu = kMuInitClean and lets mu fall superlinearly; any other start, a fixed-tf solve and the shared-tf launch
    // (one mu for all its satellites) keep kMuInit and the kSigma rule.
    int &clean = sd.dv.clean;
    clean = !SHARED && !sd.fixed_tf && !__any(pushed);      // (fixed-tf solves feed a host root search with their g_tf: left as they were)
    if (clean) {
        const double tf = sd.tfbar;
        if (-(-tf - sd.b_tf[0]) < kBoundPush * fmax(1.0, fabs(sd.b_tf[0])) || -(tf - sd.b_tf[1]) < kBoundPush * fmax(1.0, fabs(sd.b_tf[1])))
            clean = false;
        // ... and the reference ends within kCleanRadius half-widths of the terminal radius window
        const double xr[3] = {s.xbar[K - 1], s.xbar[(size_t)Kmax + K - 1], s.xbar[(size_t)2 * Kmax + K - 1]};
        if (!(fabs(sqrt(xr[0] * xr[0] + xr[1] * xr[1] + xr[2] * xr[2]) - a.r_des[sat]) <= kCleanRadius * o.eps_r)) clean = false;
    }
#ifdef MPCX_NO_CLEAN_START      // measurement builds only (profiles/tools): every start treated as it was before round 3
    clean = false;
#endif
    const double mu0 = uni(clean ? kMuInitClean : kMuInit);
    for (int k = lane; k < K; k += 64) {
        const auto p = s.itn(k);
        // L1 slack pairs start dual feasible and centred: z+ = z- = w_nu/2, s = t = mu/z
        if (k <= K - 2) for (int i = 0; i < 7; ++i) { const double zl = sd.w_nu / 2.0, sl = mu0 / zl; p[I_T + i] = sl; p[I_STP + i] = sl; p[I_STN + i] = sl; p[I_ZTP + i] = zl; p[I_ZTN + i] = zl; }
        else for (int i = 0; i < 7; ++i) { p[I_STP + i] = 1.0; p[I_STN + i] = 1.0; p[I_ZTP + i] = 1.0; p[I_ZTN + i] = 1.0; }
        p[I_ZU] = mu0 / p[I_SU]; p[I_ZRMAX] = mu0 / p[I_SRMAX]; p[I_ZRMIN] = mu0 / p[I_SRMIN];
    }
    if (lane == 0) {
        for (int i = 0; i < GL_N; ++i) { s.itg[i] = 0.0; s.drg[i] = 0.0; }
        double xK[7];
        for (int i = 0; i < 7; ++i) xK[i] = s.xbar[(size_t)i * Kmax + K - 1];
        for (int j = 0; j < sd.nT; ++j) {
            double gj = -sd.bT[j];
            for (int i = 0; i < 7; ++i) gj += sd.aT[j][i] * xK[i];
            s.itg[gs_term(j)] = fmax(-gj, kBoundPush * fmax(1.0, fabs(sd.bT[j]))); s.itg[gz_term(j)] = mu0 / s.itg[gs_term(j)];
        }
        const double r2 = xK[0] * xK[0] + xK[1] * xK[1] + xK[2] * xK[2];
        s.itg[G_SRF] = fmax(-(r2 - sd.b_rfmax), kBoundPush * fmax(1.0, fabs(sd.b_rfmax))); s.itg[G_ZRF] = mu0 / s.itg[G_SRF];
        const double tf = (sd.fixed_tf && !SHARED) ? a.tf_out[sat] : sd.tfbar;      // (fixed: the value to hold comes in through tf_out)
        s.itg[G_TF] = tf;
        s.itg[G_STF] = fmax(-(-tf - sd.b_tf[0]), kBoundPush * fmax(1.0, fabs(sd.b_tf[0]))); s.itg[G_ZTF] = mu0 / s.itg[G_STF];
        s.itg[G_STF + 1] = fmax(-(tf - sd.b_tf[1]), kBoundPush * fmax(1.0, fabs(sd.b_tf[1]))); s.itg[G_ZTF + 1] = mu0 / s.itg[G_STF + 1];
    }
    WG_SYNC();

    // (the loop state lives in LDS, SatData::dv: see there)
    double &mu = sd.dv.mu, &dw_last = sd.dv.dw_last;            // mu: this iteration's complementarity target
    mu = mu0; dw_last = 0.0;
    // (shared tf: the counts of the whole launch -- S satellites without their own tf rows plus tf's two range inequalities)
    const int nzc = uni(SHARED ? a.S * n_ineq(K, sd.nT, 1) + 2 : n_ineq(K, sd.nT, sd.fixed_tf));
    const int nlc = uni((SHARED ? a.S : 1) * (7 * (K - 1) + (sd.nT == 6 ? 1 : 0)));
    // shared tf: slacks / multipliers of 0 <= tf <= tf_max, their trial values, and the launch's part of the tf row
    double gs[2] = {0.0, 0.0}, gz[2] = {0.0, 0.0}, gst[2] = {0.0, 0.0}, gzt[2] = {0.0, 0.0}, gds[2] = {0.0, 0.0}, gdz[2] = {0.0, 0.0};
    if (SHARED) {
        const double tf = sd.tfbar;
        gs[0] = fmax(-(-tf - sd.b_tf[0]), kBoundPush * fmax(1.0, fabs(sd.b_tf[0]))); gz[0] = kMuInit / gs[0];
        gs[1] = fmax(-(tf - sd.b_tf[1]), kBoundPush * fmax(1.0, fabs(sd.b_tf[1]))); gz[1] = kMuInit / gs[1];
    }
    const double b_tf2[2] = {sd.b_tf[0], sd.b_tf[1]};
    int &n_acc = sd.dv.n_acc, &status = sd.dv.status, &it_count = sd.dv.it_count, &n_reg = sd.dv.n_reg, &first_reg = sd.dv.first_reg;
    n_acc = 0; status = MPCX_ST_MAXITER; it_count = 0; n_reg = 0; first_reg = -1;
    // second safeguard of the adaptive barrier rule (the first is the kMuErr bound below): after kFbN consecutive accepted
    // steps shorter than kFbAlpha -- the iterate is jammed against its bounds -- mu is lifted to kFbBoost * mean(s z) and
    // follows ipopt's monotone Fiacco-McCormick rule from then on.  kFbN = 8: benchmark problems at K = 100 take up to seven
    // short regularised steps in a row and recover by themselves in 16 / 25 iterations (the monotone rule: 32 / 41).
    int &mono = sd.dv.mono, &refined_prev = sd.dv.refined_prev, &n_small = sd.dv.n_small;      // refined_prev: the last iteration's unregularised solve ran refinement passes
    double &E0 = sd.dv.E0;
    mono = 0; refined_prev = 0; n_small = 0; E0 = 0.0;
    // residual of the start point; afterwards the accepted trial of the line search is the next iteration's evaluation
    // (sq in its mu = 0 form: it serves E_0 and, for any mu, the line search's ||F_mu||)
    ResAcc &r0 = sd.racc[0], &rt = sd.racc[1];       // (in LDS: eval_residual<false> writes the first, <true> the second)
    PT_END(6)
    PT_BEGIN
    eval_residual<false>(s, sd, 0.0, 0.0, 0.0, lane);
    PT_END(0)
    if (SHARED) shared_fold(*gsync, r0, sd.tfbar, b_tf2, gs, gz, 0.0, lane);
    int &iter = sd.dv.iter;
    for (iter = 0;; ++iter) {
        it_count = iter;
        E0 = uni(scaled_error_n(r0, nzc, nlc, 0.0));
        if (SHARED && gsync->aborted) { status = MPCX_ST_NUMERIC; break; }
        if (!(E0 == E0) || !(E0 < 1e300)) { status = MPCX_ST_NUMERIC; break; }
        if (E0 <= o.tol) { status = MPCX_ST_OK; break; }
        n_acc = (E0 <= o.acc_tol) ? n_acc + 1 : 0;
        if (n_acc >= o.acc_iter) { status = MPCX_ST_ACCEPTABLE; break; }
        if (iter >= o.max_iter) { status = (E0 <= o.acc_tol) ? MPCX_ST_ACCEPTABLE : MPCX_ST_MAXITER; break; }
        // adaptive barrier parameter: a fixed fraction of the iterate's mean complementarity ("Solver algorithm")
        double &mu_cur = sd.dv.mu_cur;
        mu_cur = uni(r0.prod_sum / (double)nzc);
        if (!mono && n_small >= kFbN) {
            mono = 1;
            mu = uni(fmax(o.tol / 10.0, fmin(kMuInit, kFbBoost * mu_cur)));
        }
        // (never below kMuErr * E_0: the mean complementarity may collapse while the iterate is still infeasible)
        if (!mono) mu = uni(fmax(fmax(clean ? fmin(kSigma * mu_cur, mu_cur * sqrt(mu_cur)) : kSigma * mu_cur, o.tol / 10.0), kMuErr * E0));
        else {
            // mu moves on only when the barrier problem is solved to E_mu <= 10 mu: mu <- max(tol/10, min(0.2 mu, mu^1.5))
            for (int lv = 0; lv < 64 && mu > o.tol / 10.0 && scaled_error_n(r0, nzc, nlc, mu) <= 10.0 * mu; ++lv)
                mu = uni(fmax(o.tol / 10.0, fmin(0.2 * mu, mu * sqrt(mu))));
        }
        // Newton direction, with Hessian regularisation retries on breakdown
        int &have_dir = sd.dv.have_dir;
        double &delta_w = sd.dv.delta_w, &alpha = sd.dv.alpha;
        have_dir = 0; delta_w = 0.0; alpha = 1.0;
#ifdef MPCX_ITER_LOG
        int fail_mask = 0;     // decimal digits: factor, border, finite-check failures of this iteration
#endif
        double &tau = sd.dv.tau;
        tau = uni(fmax(0.99, 1.0 - mu));
        // Hessian regularisation on breakdown follows ipopt's inertia-correction schedule: 0 first, then a third of
        // the last value that worked (1e-4 the first time), growing by 8 (by 100 until some value has worked), up to 1e40
        while (!have_dir && delta_w <= kDwMax) {
            PT_BEGIN
            // (an iteration that follows a refining one almost always refines too -- the barrier weights grow as mu falls -- and then
            //  needs the Newton scalars kept: newton_blocks<true> at once, instead of <false> now and <true> again below.  Same
            //  records either way: <true> is <false> plus the stores of the scalars)
            const bool ns_kept = refined_prev && delta_w == 0.0;
            if (ns_kept) newton_blocks<true>(s, sd, (double *)&w, mu, delta_w, lane);
            else newton_blocks<false>(s, sd, (double *)&w, mu, delta_w, lane);
            PT_END(1)
            first_rhs_scalars(sd);    // (sd.rs_*; the node records of the first right-hand side: newton_blocks)
#ifndef MPCX_TP                 // (the shared-tf launch has its own kernel, solve.hip)
            if (SHARED) {
                // ---- the same direction computation in lock step with the other satellites of the launch ----
                GridSync &g = *gsync;
                if (g.aborted) break;
                // the launch's own part of the tf row: the 1 of the objective, the barrier terms of 0 <= tf <= tf_max, delta_w
                double W_glob = delta_w, g_glob = 1.0, sig_tf = 0.0;
                const double tfc = s.itg[G_TF];
                const double gvv[2] = {-tfc - sd.b_tf[0], tfc - sd.b_tf[1]};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const double sig = gz[j] / gs[j], zh = mu / gs[j] + sig * (gvv[j] + gs[j]);
                    W_glob += sig; g_glob += (j == 0 ? -zh : zh); sig_tf = fmax(sig_tf, sig);
                }
                double twmax = fmax(sd.sigmax, sig_tf);
                for (int t = 0; t < NTERM; ++t) twmax = fmax(twmax, sd.tw[t]);
                gr_clear(gr); gr[GR_SUM] = twmax;
                grid_reduce(g, gr, lane);                         // every satellite refines, or none
                const int passes = 1 + ((delta_w == 0.0 && gr[GR_SUM] > kRefineTw) ? o.n_refine : 0);
                if (passes > 1 && !ns_kept) newton_blocks<true>(s, sd, (double *)&w, mu, delta_w, lane);
                refined_prev = passes > 1;
                bool okl = riccati_factor(s, sd, w, lane, true, passes > 1);     // (a local breakdown is reported through the border's reduction)
                bool ok = true;
                for (int pass = 0; pass < passes && ok; ++pass) {
                    if (okl && pass > 0) { reduced_residual(s, sd, (double *)&w, lane); sweep_backward(s, sd, w, 0, 1, lane); }
                    if (okl) {
                        sweep_forward(s, sd, w, 0, (pass == 0) ? NCH : 1, lane);
                        if (pass == 0) okl = border_factor_shared(sd, lane);
                    }
                    const double dtf_cur = (pass == 0) ? 0.0 : s.drg[G_TF];
                    ok = border_solve_shared(sd, g, W_glob, -(g_glob + W_glob * dtf_cur), !okl, lane);
                    if (!ok) break;
                    combine_channels(s, sd, (double *)&w, lane, pass == 0);
                }
                if (ok) {
                    alpha = uni(finish_direction(s, sd, mu, tau, lane));
                    const bool fin = sd.dir_finite != 0;
                    const double dtf = s.drg[G_TF];
                    const double dgv[2] = {-dtf, dtf};
#pragma unroll
                    for (int j = 0; j < 2; ++j) {       // the range constraint's pairs: direction and fraction to the boundary
                        const PairDir q = pair_dir(gs[j], gz[j], gvv[j], dgv[j], mu);
                        gds[j] = q.ds; gdz[j] = q.dz;
                        if (q.ds < 0.0) alpha = fmin(alpha, -tau * gs[j] / q.ds);
                        if (q.dz < 0.0) alpha = fmin(alpha, -tau * gz[j] / q.dz);
                    }
                    gr_clear(gr); gr[0] = fin ? 0.0 : 1.0; gr[GR_SUM + GR_MAX] = alpha;
                    grid_reduce(g, gr, lane);                     // one step length for the whole launch
                    alpha = gr[GR_SUM + GR_MAX];
                    ok = (gr[0] == 0.0) && !g.aborted;
                }
                if (ok) have_dir = 1;
                else if (g.aborted) break;
                else if (delta_w == 0.0) delta_w = (dw_last == 0.0) ? kDwFirst : fmax(kDwMin, dw_last / 3.0);
                else delta_w *= (dw_last == 0.0) ? 100.0 : 8.0;
                continue;
            }
#endif
            // iterative refinement only once a barrier weight (terminal rank-1 terms, stage balls and planes, the tf
            // bounds) is stiff enough to cost digits
            double twmax = sd.sigmax;
            for (int t = 0; t < NTERM; ++t) twmax = fmax(twmax, sd.tw[t]);
            const int passes = 1 + ((delta_w == 0.0 && twmax > kRefineTw) ? o.n_refine : 0);
            if (passes > 1 && !ns_kept) newton_blocks<true>(s, sd, (double *)&w, mu, delta_w, lane);      // (the scalars reduced_residual reads)
            if (delta_w == 0.0) refined_prev = passes > 1;
            PT_BEGIN
#ifdef MPCX_TP
            bool ok = tp_cmd_factor(s, sd, g_tp, lane, passes > 1);       // every segment's factorisation + fused backward sweeps, side by side
            if (g_tp.dead) break;                                         // (a workgroup of the satellite did not answer: MPCX_ST_NUMERIC)
#else
            bool ok = riccati_factor(s, sd, w, lane, true, passes > 1);   // factorisation + backward sweep of all 8 channels
#endif
            PT_END(2)
#ifdef MPCX_ITER_LOG
            if (!ok) fail_mask += 1;
#endif
            if (ok) {
                // the direction starts from (0, ..., -lam, -lam_vt) so that the first right-hand side carries no
                // multipliers; combine_channels writes it with that starting value (no separate reset pass)
                for (int pass = 0; pass < passes && ok; ++pass) {
                    if (pass > 0) {
                        PT_BEGIN
                        reduced_residual(s, sd, (double *)&w, lane);
                        PT_END(5)
                    }
                    // pass 0: all 8 channels (right-hand side + the 7 border columns); refinement: channel 0 only
                    const int c1 = (pass == 0) ? NCH : 1;
#ifdef MPCX_TP
                    // the segments' sweeps side by side (all waves), then the coarse problem over the cuts: x_K and Sigma . lam of
                    // every channel for the border
                    PT_BEGIN
                    (void)c1;
                    if (pass > 0) tp_cmd_sweeps(s, sd, g_tp, lane, false);      // (the first pass's sweeps ran behind the factorisation)
                    PT_END(3)
                    if (g_tp.dead) { ok = false; break; }
                    PT_BEGIN
                    ok = tp_coarse(s, sd, g_tp, lane, pass == 0);
                    if (ok && pass == 0) ok = border_factor(sd, lane);
                    PT_END(4)
#else
                    if (pass > 0) {
                        PT_BEGIN
                        sweep_backward(s, sd, w, 0, c1, lane);
                        PT_END(3)
                    }
                    PT_BEGIN
                    sweep_forward(s, sd, w, 0, c1, lane);
                    if (pass == 0) ok = border_factor(sd, lane);
                    PT_END(4)
#endif
#ifdef MPCX_ITER_LOG
                    if (!ok) fail_mask += 100;
#endif
                    if (!ok) break;
                    PT_BEGIN
                    border_solve(sd, lane);
#if defined(MPCX_ITER_LOG) && defined(MPCX_LOG_IT)
                    // diagnostic build only: the border system of one chosen iteration into this satellite's NU block
                    if (iter == MPCX_LOG_IT && pass == 0 && lane == 0) {
                        double *lg = a.NU + (size_t)sat * 7 * Kmax; int n = 0;
                        for (int j = 0; j < NBD; ++j) lg[n++] = sd.sol[j];
                        for (int j = 0; j < NTERM; ++j) lg[n++] = sd.tw[j];
                        for (int j = 0; j < NTERM; ++j) lg[n++] = sd.twin[j];
                        for (int j = 0; j < NCH; ++j) lg[n++] = sd.siglam[j];
                        for (int c = 0; c < NCH; ++c) for (int j = 0; j < 7; ++j) lg[n++] = sd.xK[c][j];
                        lg[n++] = sd.rs_gtf; lg[n++] = sd.rs_rvt;
                        for (int j = 0; j < NTERM; ++j) lg[n++] = sd.rs_gex[j];
                        lg[n++] = sd.Wtf; lg[n++] = sd.gam; lg[n++] = delta_w;
                    }
#endif
#ifdef MPCX_TP
                    tp_cmd_combine(s, sd, g_tp, (double *)&w, lane, pass == 0);
#else
                    combine_channels(s, sd, (double *)&w, lane, pass == 0);
#endif
                    PT_END(7)
                }
            }
            if (ok) {
                // dt, ds, dz, the fraction-to-the-boundary step and the finite check on the direction
                PT_BEGIN
                alpha = uni(finish_direction(s, sd, mu, tau, lane));
                ok = sd.dir_finite != 0;
                PT_END(8)
#ifdef MPCX_ITER_LOG
                if (!ok) fail_mask += 10000;
#endif
            }
            if (ok) have_dir = 1;
            else if (delta_w == 0.0) delta_w = (dw_last == 0.0) ? kDwFirst : fmax(kDwMin, dw_last / 3.0);
            else delta_w *= (dw_last == 0.0) ? 100.0 : 8.0;
        }
        if (have_dir && delta_w > 0.0) { dw_last = delta_w; if (n_reg++ == 0) first_reg = iter; }
        if (!have_dir) {
#ifdef MPCX_ITER_LOG
            if (lane == 0 && 5 * iter + 4 < 7 * K) { double *lg = a.X + (size_t)sat * 7 * Kmax + 5 * iter; lg[0] = mu; lg[1] = E0; lg[2] = -1.0; lg[3] = delta_w; lg[4] = (double)fail_mask; }
#endif
            status = MPCX_ST_NUMERIC; break;
        }
        // backtracking on ||F_mu||_2 with the N_-inf(gamma) neighbourhood
        // ||F_mu||^2 of the iterate from the mu = 0 evaluation: sum (s z - mu)^2 = sum (s z)^2 - 2 mu sum s z + n mu^2
        double &rn0 = sd.dv.rn0;
        rn0 = uni(sqrt(fmax(0.0, r0.sq - 2.0 * mu * r0.prod_sum + (double)nzc * mu * mu)));
        // every trial is evaluated as the iterate it would become (slack reset and multiplier safeguard applied) and
        // left in the second iterate buffer
        double &mu_clip = sd.dv.mu_clip;
        mu_clip = uni(fmax(mu, mu_cur));
        // shared tf: the trial values of the range constraint's pairs (slack reset and multiplier safeguard like every
        // other pair), then the launch's residual from the satellites' (a reduction: every workgroup decides alike)
        auto shared_trial = [&]() {
            const double tft = s.itg[G_TF] + alpha * s.drg[G_TF];
            const double gvt[2] = {-tft - sd.b_tf[0], tft - sd.b_tf[1]};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                gst[j] = fmax(gs[j] + alpha * gds[j], -gvt[j]);
                gzt[j] = fmin(gz[j] + alpha * gdz[j], kKappaSigma * (mu_clip * rcp_pos(gst[j])));
            }
            shared_fold(*gsync, rt, tft, b_tf2, gst, gzt, mu, lane);
        };
        int &have_trial = sd.dv.have_trial, &ls = sd.dv.ls;
        have_trial = 0;
        for (ls = 0; ls < 30; ++ls) {
            if (0.5 * alpha < kAlphaFloor) break;      // a rejection could not shorten the step any more: take it
            PT_BEGIN
            eval_residual<true>(s, sd, alpha, mu, mu_clip, lane);
            PT_END(10)
            if (SHARED) shared_trial();
            const bool dec = sqrt(rt.sq) <= (1.0 - 1e-4 * alpha) * rn0;
            const bool cen = rt.prod_min >= kGammaNbhd * fmin(mu, rt.prod_sum / (double)nzc);
#ifdef MPCX_ITER_LOG
            // diagnostic build only: the first trial's margins into this satellite's U block
            if (ls == 0 && lane == 0 && 3 * iter + 2 < 3 * K) { double *lg = a.U + (size_t)sat * 3 * Kmax + 3 * iter; lg[0] = alpha; lg[1] = sqrt(rt.sq) / rn0; lg[2] = rt.prod_min / (kGammaNbhd * fmin(mu, rt.prod_sum / (double)nzc)); }
#endif
            if (dec && cen) { have_trial = 1; break; }
            alpha = uni(alpha * 0.5);
        }
        if (!have_trial) {                              // the step taken untested
            PT_BEGIN
            eval_residual<true>(s, sd, alpha, mu, mu_clip, lane);
            PT_END(10)
            if (SHARED) shared_trial();
        }
#ifdef MPCX_ITER_LOG
        // diagnostic build only: iteration log (mu, E0, accepted step, regularisation) into this satellite's X block
        if (lane == 0 && 5 * iter + 4 < 7 * K) { double *lg = a.X + (size_t)sat * 7 * Kmax + 5 * iter; lg[0] = mu; lg[1] = E0; lg[2] = alpha; lg[3] = delta_w; lg[4] = (double)fail_mask; }
#endif
        n_small = (alpha < kFbAlpha) ? n_small + 1 : 0;
        // accept: the candidate becomes the iterate, its residual (sq back in the mu = 0 form) the next iteration's
        if (lane == 0) { wf64 *q = s.it; s.it = s.itB; s.itB = q; lf64 *g = s.itg; s.itg = s.itgB; s.itgB = g; }
        WG_SYNC();
        if (SHARED) { gs[0] = gst[0]; gs[1] = gst[1]; gz[0] = gzt[0]; gz[1] = gzt[1]; }
        {   // (both records are in LDS: the next iteration's evaluation is a copy of the accepted trial's, by the wave's first lanes)
            const double sq0 = rt.sq + 2.0 * mu * rt.prod_sum - (double)nzc * mu * mu;
            const ResAcc t = rt;
            WG_SYNC();
            r0 = t; r0.sq = sq0;
            WG_SYNC();
        }
    }

    // ---- results in the reference's shapes: X (7,K), U (3,K), NU (7,K) ----
    for (int k = lane; k < K; k += 64) {
        const auto p = s.itn(k);
        for (int i = 0; i < 7; ++i) {
#ifndef MPCX_ITER_LOG
            a.X[(size_t)sat * 7 * Kmax + (size_t)i * Kmax + k] = p[I_X + i];
#endif
#if !(defined(MPCX_ITER_LOG) && defined(MPCX_LOG_IT))
            a.NU[(size_t)sat * 7 * Kmax + (size_t)i * Kmax + k] = (k <= K - 2) ? p[I_NU + i] : 0.0;
#endif
        }
#ifndef MPCX_ITER_LOG
        for (int i = 0; i < 3; ++i) a.U[(size_t)sat * 3 * Kmax + (size_t)i * Kmax + k] = p[I_U + i];
#endif
    }
    for (int k = K + lane; k < Kmax; k += 64) {          // ragged batch: the unused columns of this satellite's rows
        for (int i = 0; i < 7; ++i) { a.X[(size_t)sat * 7 * Kmax + (size_t)i * Kmax + k] = 0.0; a.NU[(size_t)sat * 7 * Kmax + (size_t)i * Kmax + k] = 0.0; }
        for (int i = 0; i < 3; ++i) a.U[(size_t)sat * 3 * Kmax + (size_t)i * Kmax + k] = 0.0;
    }
    if (lane == 0) {
        a.tf_out[sat] = (sd.fixed_tf && !SHARED) ? r0.g_tf : s.itg[G_TF];
        a.status[sat] = status;
        a.iters[sat] = it_count;
        a.kkt[sat] = E0;
        if (a.nreg) { a.nreg[2 * sat] = n_reg; a.nreg[2 * sat + 1] = first_reg; }
#ifdef MPCX_PHASE_TIMING
        // diagnostic build only: cycle sums per phase into the NU block of this satellite (never shipped)
        double *dbg = a.NU + (size_t)sat * 7 * Kmax;
        for (int i = 0; i < 12; ++i) { dbg[2 * i] = (double)pt_[i]; dbg[2 * i + 1] = (double)pc_[i]; }
        for (int i = 0; i < 16; ++i) dbg[24 + i] = (double)sd.fpt[i];
        // calibration: the satellite's life in s_memrealtime ticks (constant 100 MHz) and in s_memtime ticks
        dbg[40] = (double)(__builtin_amdgcn_s_memrealtime() - rt0_); dbg[41] = (double)(__builtin_amdgcn_s_memtime() - mt0_);
#ifdef MPCX_TP
        for (int i = 0; i < 16; ++i) dbg[48 + i] = (double)s.mail[16 + i];       // the segments' cycles of the last factorisation command (solve_tp.hpp)
#endif
#endif
    }
}

}  // namespace MPCX_NS
