// propagate.hip -- batched nonlinear rollout on gfx950.
//
// Replaces Simulator.get_trajectory_ODE (reference simulator.py:164-189): scipy solve_ivp(RK45,
// rtol 1e-3, atol 1e-6, max_step, t_eval = linspace(0,1,n_eval)) of Simulator.satellite_dynamics
// (simulator.py:116-161) under one of the reference's thrust laws (control.py:8-143).  Samples come
// from the RK45 dense-output interpolant exactly as scipy produces them.
//
// Four lanes per satellite: the 7-state system and its 7 RK stages live in the registers of a quad (Ctrl below);
// satellites of a wave take their own adaptive step sequences under predicate.  It is a setup / SCP re-linearisation
// kernel (7 x ~1000 steps per satellite), not bandwidth relevant.
#include "mpcx_device.hpp"
#include "mpcx_host.hpp"

namespace mpcx {

struct PropArgs {
    int S, n_eval, flags, ctrl_kind, Ku;     // n_eval, Ku: output points / table columns of every satellite, or row lengths
    const int32_t *n_evals, *Kus;            // ragged batch: per-satellite counts (<= n_eval, <= Ku); nullptr: all the same
    double max_step;
    const double *y0, *tf, *consts;
    const double *ctrl_vec;   // CONSTANT: [S][3]; TANGENTIAL: [S] magnitudes; SEQUENCE: [S][3][Ku]
    const double *end_tau;    // SEQUENCE: [S]
    double *y_out;            // [S][7][n_eval]
    double *u_out;            // [S][3][n_eval] or nullptr: the thrust law at the output points (Discretizer.extract_uk)
    int32_t *status, *nsteps;
};

// Four lanes per satellite (a quad): lane c < 3 owns position and velocity component c, every lane carries the mass; lane 3
// mirrors lane 2 and is left out of the sums.  What the right-hand side needs across components travels by quad_perm DPP
// moves: |r|^2, |h|^2, |u|^2 and the error norms as quad sums, the cross products through the two rotations of the triple.
// One lane per satellite (rounds 1-2) was bound by the issue rate of its ~810 instructions per RK45 step (the whole
// 7-state system and its stage combinations in one lane); a lane of the quad issues ~500.
struct Ctrl {
    int kind, Ku, ldu;        // ldu: row length of the thrust table in memory (Ku of its columns in use)
    double vc, vn;            // constant thrust: this lane's component and |v|; tangential: vc = magnitude
    const double *useq;       // sequence: this lane's ROW of the table
    double end_tau, inv_end_tau;
    // first-order hold: the interval in use (an RK45 step is at most max_step = 1e-3 long against intervals of 1/(Ku-1):
    // the six stages of a step and many steps in a row read the same two table columns -- kept in registers)
    int kc;
    double uk, uk1, tau_k, tau_kp1, id;
};

// 1/d and 1/sqrt(d) for d > 0 well inside the normal range: hardware seed + two Newton steps (half an ulp, measured:
// profiles/tools/rcp_accuracy.hip) instead of the IEEE division / square-root sequences (scaling, fix-up: ~3x the
// instructions).  The rollout is ~1000 sequential RK45 steps x 6 right-hand sides: its time is the number of
// instructions of the right-hand side.
__device__ __forceinline__ double rcp_fast(double d)
{
    double r = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int n = 0; n < 2; ++n) { const double e = fma(-d, r, 1.0); r = fma(r, e, r); }
    return r;
}
__device__ __forceinline__ double rsq_fast(double d)
{
    double r = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int n = 0; n < 2; ++n) { const double e = fma(-d * r, r, 1.0); r = fma(0.5 * r, e, r); }
    return r;
}

// quad exchange: sum over the four lanes (bitwise identical on all of them: both steps add the same two operands), the two
// rotations of the (0,1,2) triple (lane 3 keeps its own value), and the value of lane 2
__device__ __forceinline__ double quad_sum(double v) { v += dpp64<0xB1>(v); v += dpp64<0x4E>(v); return v; }
__device__ __forceinline__ double rot1(double v) { return dpp64<0xC9>(v); }     // quad_perm [1,2,0,3]: lane c <- lane (c+1) mod 3
__device__ __forceinline__ double rot2(double v) { return dpp64<0xD2>(v); }     // quad_perm [2,0,1,3]: lane c <- lane (c+2) mod 3
__device__ __forceinline__ double lane2(double v) { return dpp64<0xAA>(v); }    // quad_perm [2,2,2,2]

// First-order hold of this lane's row of a (3,Ku) table at tau in [0,1] (control.py:104-126), same node index as foh3:
// k = int(tau // dtau) is the floor of the exact quotient (Python's float floor division goes through an exact fmod);
// floor(tau * (Ku-1)) is that number unless the product sits within rounding of an integer, and only then the exact routine
// is needed.
__device__ __forceinline__ double foh_cached(double tau, Ctrl &c, int &err)
{
    const int Ku = c.Ku;
    const double *__restrict__ u = c.useq;
    // well inside the interval in use (away from its ends by more than any rounding of the index computation): same k.
    // Everything else -- a new interval, an end point, tau == 1 -- is behind this one rarely taken branch: the right-hand
    // side is evaluated six times per step and every branch in it costs the in-order wave ~40 cycles.
    if (!(tau > c.tau_k + 1e-12 && tau < c.tau_kp1 - 1e-12)) {
    if (tau == 1.0) return u[Ku - 1];
    const double km1 = (double)(Ku - 1);
    const double q = tau * km1;
    int k = (fabs(q - rint(q)) > 1e-9 * fmax(1.0, q)) ? (int)floor(q) : (int)py_floordiv(tau, 1.0 / km1);
    if (k < 0 || k + 1 >= Ku) {          // the reference raises IndexError here
        err = MPCX_ST_FOH;
        k = k < 0 ? 0 : Ku - 2;
        if (Ku < 2) return 0.0;
    }
    if (k != c.kc) {
        c.kc = k;
        c.tau_k = (double)k / km1; c.tau_kp1 = (double)(k + 1) / km1;
        c.id = 1.0 / (c.tau_kp1 - c.tau_k);
        c.uk = u[k]; c.uk1 = u[k + 1];
    }
    }
    const double lam_n = (c.tau_kp1 - tau) * c.id, lam_p = (tau - c.tau_k) * c.id;
    return lam_n * c.uk + lam_p * c.uk1;
}

// Simulator.satellite_dynamics (simulator.py:116-161) under the controller's thrust law (control.py), times tf, for the
// component this lane owns (r, v: its position / velocity component, m: the mass; on: lane < 3) -- the same quantities as
// dynamics_unscaled / ctrl_eval (which the discretizer keeps using, division for division as numpy evaluates them),
// arranged around one reciprocal each of |r|, |h|, m instead of a division per component: results differ from those forms
// by rounding only (rollouts agree with the reference's to 1e-12, the accepted step sequence is the same -- max_step
// clips every step).  inv_gi = 1 / (g0 Isp).
template <int KIND, int FLAGS>
__device__ __forceinline__ void prop_rhs(Ctrl &c, const SatConst &cst, double inv_gi, double tf, double tau, int comp, bool on,
                                         double r, double v, double m, double &dr, double &dv, double &dm, int &err)
{
    constexpr int flags = FLAGS;
    const double r2 = quad_sum(r * r);            // (lane 3 of the quad carries r = v = 0: nothing to mask in the sums)
    const double irn = rsq_fast(r2), irn2 = irn * irn;
    if (m <= 0.0) err = MPCX_ST_MASS;
    const double im = rcp_fast(m > 0.0 ? m : 1.0);
    double u, un;
    if (KIND == MPCX_CTRL_CONSTANT) {
        u = c.vc; un = c.vn;
    } else if (KIND == MPCX_CTRL_TANGENTIAL) {
        // control.py:66-84: u = mag * t_hat, t_hat = h_hat x r_hat with h = r x v.  (r x v) x r = v |r|^2 - r (r.v) is that
        // direction without a cross product (h is normal to r, so |h x r| = |h| |r|): one quad sum for r.v, one for the
        // norm, no rotations of the triple
        const double rv = quad_sum(r * v);
        const double w = v * r2 - r * rv;
        const double iwn = rsq_fast(quad_sum(w * w));
        u = c.vc * (w * iwn);
        un = fabs(c.vc);
    } else if (KIND == MPCX_CTRL_SEQUENCE) {                  // control.py:132-142
        // tau / end_tau: reciprocal, product and one correction step (the quotient as the division sequence rounds it).
        // Past end_tau the thrust is zero: no branch, the table is read at the middle of the interval in use (always a
        // valid argument) and the result discarded.
        const bool live = tau <= c.end_tau;
        double tn = tau * c.inv_end_tau;
        tn = fma(fma(-tn, c.end_tau, tau), c.inv_end_tau, tn);
        tn = live ? tn : ((c.kc >= 0) ? 0.5 * (c.tau_k + c.tau_kp1) : 0.5);
        const double uf = foh_cached(tn, c, err);
        const double uu = quad_sum(on ? uf * uf : 0.0);
        un = live ? uu * rsq_fast(fmax(uu, 1e-300)) : 0.0;    // |u| (0 for u = 0)
        u = (live && on) ? uf : 0.0;
    } else { u = 0.0; un = 0.0; }
    const double kg = -cst.mu * (irn2 * irn);
    dr = v;
    dv = kg * r + u * im;
    if (flags & MPCX_FLAG_DRAG) {                            // simulator.py:150-153
        const double vn = sqrt(quad_sum(v * v));
        const double coef = -0.5 * kCd * cst.s * im * (kRho500 / cst.rho) * vn;
        dv += coef * v;
    }
    if (flags & MPCX_FLAG_J2) {                              // simulator.py:154-158
        const double rz = lane2(r);
        const double q2 = (rz * rz) * irn2;
        const double coef = 1.5 * cst.j2 * cst.mu * (cst.re * cst.re) * (irn2 * irn2 * irn);
        dv += coef * ((5.0 * q2 - (comp == 2 ? 3.0 : 1.0)) * r);
    }
    dm = -un * inv_gi;
    dr = tf * dr; dv = tf * dv; dm = tf * dm;
}

// One instantiation per thrust law and truth-model flag set (both are launch constants): the right-hand side appears
// eight times in the step and every variant it does not need would sit in each copy.
template <int KIND, int FLAGS>
__global__ __launch_bounds__(64) void propagate_kernel(PropArgs a)
{
    const int sat = blockIdx.x * 16 + (threadIdx.x >> 2);
    const int lane4 = threadIdx.x & 3;
    if (sat >= a.S) return;                                      // (quad-uniform: the exchanges never leave a quad)
    const bool on = lane4 < 3;
    const int comp = on ? lane4 : 2;                             // lane 3 mirrors lane 2 and is left out of sums and stores
    SatConst cst; cst.load(a.consts + (size_t)sat * MPCX_NCONST);
    const double tf = a.tf[sat];
    Ctrl c; c.kind = a.ctrl_kind; c.Ku = a.Kus ? a.Kus[sat] : a.Ku; c.ldu = a.Ku; c.useq = nullptr; c.end_tau = 1.0; c.vc = 0.0; c.vn = 0.0;
    c.kc = -1; c.tau_k = 2.0; c.tau_kp1 = -1.0; c.uk = c.uk1 = 0.0; c.id = 0.0;
    if (a.ctrl_kind == MPCX_CTRL_CONSTANT) {
        const double *v3 = a.ctrl_vec + (size_t)sat * 3;
        c.vc = on ? v3[comp] : 0.0; c.vn = sqrt(v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2]);
    } else if (a.ctrl_kind == MPCX_CTRL_TANGENTIAL) c.vc = a.ctrl_vec[sat];
    else if (a.ctrl_kind == MPCX_CTRL_SEQUENCE) { c.useq = a.ctrl_vec + (size_t)sat * 3 * a.Ku + (size_t)comp * a.Ku; c.end_tau = a.end_tau[sat]; }
    c.inv_end_tau = 1.0 / c.end_tau;
    const int ld = a.n_eval;                                     // row length of y_out
    int n_eval = a.n_evals ? a.n_evals[sat] : a.n_eval;
    if (n_eval < 1 || n_eval > ld || (a.ctrl_kind == MPCX_CTRL_SEQUENCE && (c.Ku < 2 || c.Ku > a.Ku))) {
        if (lane4 == 0) { a.status[sat] = MPCX_ST_BADK; a.nsteps[sat] = 0; }
        return;
    }
    const double inv_gi = 1.0 / (cst.g0 * cst.isp);
    const double rtol = 1e-3, atol = 1e-6, t_bound = 1.0;
    int err = 0;
    // rms over the 7 components of the system from this lane's three (the mass is counted by lane 0)
    auto rms7 = [&](double pr, double pv, double pm) {
        return sqrt(quad_sum(pr * pr + pv * pv + (lane4 == 0 ? pm * pm : 0.0))) / sqrt(7.0);
    };
    // (the quad's fourth lane carries a zero position / velocity component: its products vanish from every quad sum)
    double yr = on ? a.y0[(size_t)sat * 7 + comp] : 0.0, yv = on ? a.y0[(size_t)sat * 7 + 3 + comp] : 0.0, ym = a.y0[(size_t)sat * 7 + 6];
    double fr, fv, fm;
    double t = 0.0;
    prop_rhs<KIND, FLAGS>(c, cst, inv_gi, tf, t, comp, on, yr, yv, ym, fr, fv, fm, err);
    // select_initial_step (scipy common.py:68-134)
    double h_abs;
    {
        const double scr = atol + fabs(yr) * rtol, scv = atol + fabs(yv) * rtol, scm = atol + fabs(ym) * rtol;
        const double d0 = rms7(yr / scr, yv / scv, ym / scm), d1 = rms7(fr / scr, fv / scv, fm / scm);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = fmin(h0, 1.0);
        double f1r, f1v, f1m;
        prop_rhs<KIND, FLAGS>(c, cst, inv_gi, tf, t + h0, comp, on, yr + h0 * fr, yv + h0 * fv, ym + h0 * fm, f1r, f1v, f1m, err);
        const double d2 = rms7((f1r - fr) / scr, (f1v - fv) / scv, (f1m - fm) / scm) / h0;
        const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 0.2);
        h_abs = fmin(fmin(100.0 * h0, h1), fmin(1.0, a.max_step));
    }
    const double estep = (n_eval > 1) ? 1.0 / (double)(n_eval - 1) : 0.0;
    int ei = 0, nsteps = 0;
    bool rejected = false;
    double *yo = a.y_out + (size_t)sat * 7 * ld;
    for (int iter = 0; iter < 4000000; ++iter) {
        if (t == t_bound) break;
        // scipy: min_step = 10 |nextafter(t, inf) - t| <= 10 ulp(1) = 2.3e-15 for t in [0, 1]; it only guards against
        // vanishing steps, so it is evaluated only when h_abs could be anywhere near it
        const double min_step = (h_abs > 1e-12) ? 0.0 : 10.0 * fabs(nextafter(t, INFINITY) - t);
        if (!rejected) {
            if (h_abs > a.max_step) h_abs = a.max_step;
            else if (h_abs < min_step) h_abs = min_step;
        }
        if (!(h_abs >= min_step)) { err = MPCX_ST_STEP; break; }
        double h = h_abs, t_new = t + h;
        if (t_new - t_bound > 0.0) t_new = t_bound;
        h = t_new - t;
        const double h_try = fabs(h);
        // the six stages, each of this lane's three components with the arithmetic of the one-lane form
        double Kr[6], Kv[6], Km[6];
#define MPCX_STAGE(S, EXPR_R, EXPR_V, EXPR_M)                                                                            \
        prop_rhs<KIND, FLAGS>(c, cst, inv_gi, tf, t + RK_C[S] * h, comp, on, yr + (EXPR_R) * h, yv + (EXPR_V) * h, ym + (EXPR_M) * h, \
                              Kr[S - 1], Kv[S - 1], Km[S - 1], err);
        MPCX_STAGE(1, fr * RK_A[1][0], fv * RK_A[1][0], fm * RK_A[1][0])
        MPCX_STAGE(2, fr * RK_A[2][0] + Kr[0] * RK_A[2][1], fv * RK_A[2][0] + Kv[0] * RK_A[2][1], fm * RK_A[2][0] + Km[0] * RK_A[2][1])
        MPCX_STAGE(3, fr * RK_A[3][0] + Kr[0] * RK_A[3][1] + Kr[1] * RK_A[3][2], fv * RK_A[3][0] + Kv[0] * RK_A[3][1] + Kv[1] * RK_A[3][2],
                   fm * RK_A[3][0] + Km[0] * RK_A[3][1] + Km[1] * RK_A[3][2])
        MPCX_STAGE(4, fr * RK_A[4][0] + Kr[0] * RK_A[4][1] + Kr[1] * RK_A[4][2] + Kr[2] * RK_A[4][3],
                   fv * RK_A[4][0] + Kv[0] * RK_A[4][1] + Kv[1] * RK_A[4][2] + Kv[2] * RK_A[4][3],
                   fm * RK_A[4][0] + Km[0] * RK_A[4][1] + Km[1] * RK_A[4][2] + Km[2] * RK_A[4][3])
        MPCX_STAGE(5, fr * RK_A[5][0] + Kr[0] * RK_A[5][1] + Kr[1] * RK_A[5][2] + Kr[2] * RK_A[5][3] + Kr[3] * RK_A[5][4],
                   fv * RK_A[5][0] + Kv[0] * RK_A[5][1] + Kv[1] * RK_A[5][2] + Kv[2] * RK_A[5][3] + Kv[3] * RK_A[5][4],
                   fm * RK_A[5][0] + Km[0] * RK_A[5][1] + Km[1] * RK_A[5][2] + Km[2] * RK_A[5][3] + Km[3] * RK_A[5][4])
#undef MPCX_STAGE
        const double ynr = yr + h * (fr * RK_B[0] + Kr[0] * RK_B[1] + Kr[1] * RK_B[2] + Kr[2] * RK_B[3] + Kr[3] * RK_B[4] + Kr[4] * RK_B[5]);
        const double ynv = yv + h * (fv * RK_B[0] + Kv[0] * RK_B[1] + Kv[1] * RK_B[2] + Kv[2] * RK_B[3] + Kv[3] * RK_B[4] + Kv[4] * RK_B[5]);
        const double ynm = ym + h * (fm * RK_B[0] + Km[0] * RK_B[1] + Km[1] * RK_B[2] + Km[2] * RK_B[3] + Km[3] * RK_B[4] + Km[4] * RK_B[5]);
        prop_rhs<KIND, FLAGS>(c, cst, inv_gi, tf, t + h, comp, on, ynr, ynv, ynm, Kr[5], Kv[5], Km[5], err);
        const double ehr = (fr * RK_E[0] + Kr[0] * RK_E[1] + Kr[1] * RK_E[2] + Kr[2] * RK_E[3] + Kr[3] * RK_E[4] + Kr[4] * RK_E[5] + Kr[5] * RK_E[6]) * h;
        const double ehv = (fv * RK_E[0] + Kv[0] * RK_E[1] + Kv[1] * RK_E[2] + Kv[2] * RK_E[3] + Kv[3] * RK_E[4] + Kv[4] * RK_E[5] + Kv[5] * RK_E[6]) * h;
        const double ehm = (fm * RK_E[0] + Km[0] * RK_E[1] + Km[1] * RK_E[2] + Km[2] * RK_E[3] + Km[3] * RK_E[4] + Km[4] * RK_E[5] + Km[5] * RK_E[6]) * h;
        const double ssq = quad_sum(ehr * ehr + ehv * ehv + (lane4 == 0 ? ehm * ehm : 0.0));
        // scipy's error norm divides each component by atol + max(|y|, |y_new|) rtol >= atol.  If even the bound
        // rms(e h) / atol is below 0.4 the exact norm is below 0.5 (and below 1) whatever it is, which is all the
        // controller asks of it when the step was max_step long (see below): no divisions, no square roots then.
        const bool surely_small = (ssq < 7.0 * (0.4 * atol) * (0.4 * atol)) && h_try >= a.max_step;
        double en = 0.25;
        if (!surely_small)
            en = rms7(ehr / (atol + fmax(fabs(yr), fabs(ynr)) * rtol), ehv / (atol + fmax(fabs(yv), fabs(ynv)) * rtol),
                      ehm / (atol + fmax(fabs(ym), fabs(ynm)) * rtol));
        if (en < 1.0) {
            // scipy: factor = min(MAX_FACTOR, SAFETY * en^-0.2) (1 if the step before was rejected), h_abs = h_try * factor,
            // then h_abs is clipped to max_step at the top of the next step.  When this step already was max_step long
            // and en <= 0.5 < SAFETY^5, the factor is >= 1 and the clip undoes it whatever its value: same h_abs,
            // bit for bit, without the pow (the usual case: max_step = 1e-3 keeps en around 1e-9).
            if (en <= 0.5 && h_try >= a.max_step) h_abs = h_try;
            else {
                double factor = (en == 0.0) ? RK_MAX_FACTOR : fmin(RK_MAX_FACTOR, RK_SAFETY * pow(en, -0.2));
                if (rejected) factor = fmin(1.0, factor);
                h_abs = h_try * factor;
            }
            rejected = false;
            // dense output at the t_eval points in (t, t_new]  (scipy RkDenseOutput, rk.py:552-574)
            while (ei < n_eval) {
                const double te = (ei == n_eval - 1 && n_eval > 1) ? 1.0 : (double)ei * estep + 0.0;
                if (te > t_new) break;
                const double x = (te - t) / h;
                const double pw[4] = {x, x * x, (x * x) * x, ((x * x) * x) * x};
                auto dense = [&](double f0, const double (&K)[6], double y0) {
                    const double kk[7] = {f0, K[0], K[1], K[2], K[3], K[4], K[5]};
                    double acc = 0.0;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        double q = 0.0;
#pragma unroll
                        for (int j = 0; j < 7; ++j) q += kk[j] * RK_P[j][cc];
                        acc += q * pw[cc];
                    }
                    return h * acc + y0;
                };
                const double orr = dense(fr, Kr, yr), ov = dense(fv, Kv, yv), om = dense(fm, Km, ym);
                if (on) { yo[(size_t)comp * ld + ei] = orr; yo[(size_t)(3 + comp) * ld + ei] = ov; }
                if (lane4 == 0) yo[(size_t)6 * ld + ei] = om;
                if (a.u_out) {
                    // extract_uk (linearize_discretize.py:393-411): u_func(x_k, t_k) at the output point -- the reference thrust
                    // of the next linearisation, which the host would otherwise recompute from the trajectory it gets back
                    double uo = 0.0;
                    if (KIND == MPCX_CTRL_CONSTANT) uo = c.vc;
                    else if (KIND == MPCX_CTRL_TANGENTIAL) {
                        const double q2 = quad_sum(on ? orr * orr : 0.0), qv = quad_sum(on ? orr * ov : 0.0);
                        const double wv = ov * q2 - orr * qv;
                        uo = c.vc * (wv * rsq_fast(quad_sum(on ? wv * wv : 0.0)));
                    } else if (KIND == MPCX_CTRL_SEQUENCE) {
                        if (te <= c.end_tau) {                       // control.py:132-142, the hold as foh3 evaluates it (divisions)
                            double o3[3];
                            foh3(te / c.end_tau, a.ctrl_vec + (size_t)sat * 3 * a.Ku, c.Ku, a.Ku, o3, err);
                            uo = comp == 0 ? o3[0] : (comp == 1 ? o3[1] : o3[2]);
                        }
                    }
                    if (on) a.u_out[((size_t)sat * 3 + comp) * ld + ei] = uo;
                }
                ++ei;
            }
            t = t_new;
            yr = ynr; yv = ynv; ym = ynm; fr = Kr[5]; fv = Kv[5]; fm = Km[5];
            ++nsteps;
        } else {
            h_abs = h_try * fmax(RK_MIN_FACTOR, RK_SAFETY * pow(en, -0.2));
            rejected = true;
        }
    }
    if (t != t_bound && err == 0) err = MPCX_ST_STEP;
    if (lane4 == 0) { a.status[sat] = err; a.nsteps[sat] = nsteps; }
}

// Discretizer.extract_uk (linearize_discretize.py:393-411) of a SequenceController played over its own horizon
// (SequenceController(u, tf_u, tf_sim = tf_u), control.py:217-221: end_tau = 1) at the nodes linspace(0, 1, n_s): the
// reference thrust of the next SCP iteration.  One lane per (satellite, output node).
__global__ __launch_bounds__(256) void resample_sequence_kernel(int S, int Ku, const int32_t *Kus, const double *u, int n, const int32_t *ns,
                                                                double *out, int32_t *status)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)S * n) return;
    const int s = (int)(idx / n), i = (int)(idx - (long)s * n);
    const int ku = Kus ? Kus[s] : Ku, nn = ns ? ns[s] : n;
    double o[3] = {0.0, 0.0, 0.0};
    if (ku < 2 || ku > Ku || nn < 1 || nn > n) { if (i == 0) status[s] = MPCX_ST_BADK; }
    else if (i < nn) {
        // np.linspace(0, 1, nn)[i]: i * step, the last point exactly 1
        const double tau = (i == nn - 1 && nn > 1) ? 1.0 : (double)i * (1.0 / (double)(nn > 1 ? nn - 1 : 1)) + 0.0;
        int err = 0;
        foh3(tau, u + (size_t)s * 3 * Ku, ku, Ku, o, err);
        if (err) atomicMax(&status[s], err);
    }
    for (int r = 0; r < 3; ++r) out[(size_t)s * 3 * n + (size_t)r * n + i] = o[r];      // (zeros past the satellite's last node)
}

}  // namespace mpcx

using namespace mpcx;

extern "C" int mpcx_resample_sequence_dev(mpcx_ctx *ctx, int S, int Ku, const int32_t *Kus, const double *u, int n,
                                          const int32_t *ns, double *u_out, int32_t *status, void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || Ku < 2 || n < 1 || !u || !u_out || !status) return ctx_fail(ctx, MPCX_E_BADARG, "resample_sequence: need S>=1, Ku>=2, n>=1 and all arrays");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    MPCX_HIP(ctx, hipMemsetAsync(status, 0, sizeof(int32_t) * S, (hipStream_t)stream));
    const long total = (long)S * n;
    hipLaunchKernelGGL(resample_sequence_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S, Ku, Kus, u, n, ns,
                       u_out, status);
    MPCX_HIP(ctx, hipGetLastError());
    return MPCX_OK;
}

extern "C" int mpcx_propagate_thrust_batch_ragged_dev(mpcx_ctx *ctx, int S, int n_eval, const int32_t *n_evals, const double *y0,
                                                      const double *tf, const double *consts, int flags, int ctrl_kind,
                                                      const double *ctrl_vec, int Ku, const int32_t *Kus, const double *end_tau,
                                                      double max_step, double *y_out, double *u_out, int32_t *status,
                                                      int32_t *nsteps, void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || n_eval < 1 || !(max_step > 0.0)) return ctx_fail(ctx, MPCX_E_BADARG, "propagate: need S>=1, n_eval>=1, max_step>0");
    if (ctrl_kind < MPCX_CTRL_ZERO || ctrl_kind > MPCX_CTRL_SEQUENCE) return ctx_fail(ctx, MPCX_E_BADARG, "propagate: unknown thrust law");
    if (ctrl_kind == MPCX_CTRL_SEQUENCE && (Ku < 2 || !end_tau || !ctrl_vec)) return ctx_fail(ctx, MPCX_E_BADARG, "propagate: sequence needs Ku>=2, table and end_tau");
    if ((ctrl_kind == MPCX_CTRL_CONSTANT || ctrl_kind == MPCX_CTRL_TANGENTIAL) && !ctrl_vec) return ctx_fail(ctx, MPCX_E_BADARG, "propagate: thrust parameters missing");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    PropArgs a{S, n_eval, flags, ctrl_kind, Ku, n_evals, Kus, max_step, y0, tf, consts, ctrl_vec, end_tau, y_out, u_out, status, nsteps};
    const dim3 grid((S + 15) / 16), block(64);            // 16 satellites (quads) per wave
    hipStream_t st = (hipStream_t)stream;
#define MPCX_PROP_LAUNCH(KIND, FLAGS) hipLaunchKernelGGL((propagate_kernel<KIND, FLAGS>), grid, block, 0, st, a)
#define MPCX_PROP_FLAGS(KIND)                                                                                         \
    switch (flags & 3) {                                                                                              \
    case 0: MPCX_PROP_LAUNCH(KIND, 0); break;                                                                         \
    case 1: MPCX_PROP_LAUNCH(KIND, 1); break;                                                                         \
    case 2: MPCX_PROP_LAUNCH(KIND, 2); break;                                                                         \
    default: MPCX_PROP_LAUNCH(KIND, 3); break;                                                                        \
    }
    switch (ctrl_kind) {
    case MPCX_CTRL_ZERO: MPCX_PROP_FLAGS(MPCX_CTRL_ZERO); break;
    case MPCX_CTRL_CONSTANT: MPCX_PROP_FLAGS(MPCX_CTRL_CONSTANT); break;
    case MPCX_CTRL_TANGENTIAL: MPCX_PROP_FLAGS(MPCX_CTRL_TANGENTIAL); break;
    default: MPCX_PROP_FLAGS(MPCX_CTRL_SEQUENCE); break;
    }
#undef MPCX_PROP_FLAGS
#undef MPCX_PROP_LAUNCH
    MPCX_HIP(ctx, hipGetLastError());
    return MPCX_OK;
}

extern "C" int mpcx_propagate_batch_ragged_dev(mpcx_ctx *ctx, int S, int n_eval, const int32_t *n_evals, const double *y0,
                                               const double *tf, const double *consts, int flags, int ctrl_kind,
                                               const double *ctrl_vec, int Ku, const int32_t *Kus, const double *end_tau,
                                               double max_step, double *y_out, int32_t *status, int32_t *nsteps, void *stream)
{
    return mpcx_propagate_thrust_batch_ragged_dev(ctx, S, n_eval, n_evals, y0, tf, consts, flags, ctrl_kind, ctrl_vec, Ku, Kus, end_tau,
                                                  max_step, y_out, nullptr, status, nsteps, stream);
}

extern "C" int mpcx_propagate_batch_dev(mpcx_ctx *ctx, int S, int n_eval, const double *y0, const double *tf,
                                        const double *consts, int flags, int ctrl_kind, const double *ctrl_vec,
                                        int Ku, const double *end_tau, double max_step, double *y_out,
                                        int32_t *status, int32_t *nsteps, void *stream)
{
    return mpcx_propagate_batch_ragged_dev(ctx, S, n_eval, nullptr, y0, tf, consts, flags, ctrl_kind, ctrl_vec, Ku, nullptr, end_tau,
                                           max_step, y_out, status, nsteps, stream);
}

extern "C" int mpcx_propagate_thrust_batch_ragged(mpcx_ctx *ctx, int S, int n_eval, const int32_t *n_evals, const double *y0,
                                                  const double *tf, const double *consts, int flags, int ctrl_kind,
                                                  const double *ctrl_vec, int Ku, const int32_t *Kus, const double *end_tau,
                                                  double max_step, double *y_out, double *u_out, int32_t *status, int32_t *nsteps)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || n_eval < 1) return ctx_fail(ctx, MPCX_E_BADARG, "propagate: need S>=1, n_eval>=1");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    DeviceArena ar(ctx);
    double *dy0 = ar.upload(y0, (size_t)S * 7), *dtf = ar.upload(tf, S), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST);
    size_t nv = 0;
    if (ctrl_kind == MPCX_CTRL_CONSTANT) nv = (size_t)S * 3;
    else if (ctrl_kind == MPCX_CTRL_TANGENTIAL) nv = S;
    else if (ctrl_kind == MPCX_CTRL_SEQUENCE) nv = (size_t)S * 3 * Ku;
    double *dv = (nv && ctrl_vec) ? ar.upload(ctrl_vec, nv) : nullptr;
    double *de = (ctrl_kind == MPCX_CTRL_SEQUENCE && end_tau) ? ar.upload(end_tau, S) : nullptr;
    int32_t *dne = n_evals ? ar.upload(n_evals, S) : nullptr, *dku = Kus ? ar.upload(Kus, S) : nullptr;
    double *dy = ar.alloc<double>((size_t)S * 7 * n_eval);
    double *du = u_out ? ar.alloc<double>((size_t)S * 3 * n_eval) : nullptr;
    int32_t *dst = ar.alloc<int32_t>(S), *dns = ar.alloc<int32_t>(S);
    if (ar.failed()) return ar.code();
    if (n_evals) {                                                                                             // the unused columns
        MPCX_HIP(ctx, hipMemsetAsync(dy, 0, (size_t)S * 7 * n_eval * sizeof(double), ctx->stream));
        if (du) MPCX_HIP(ctx, hipMemsetAsync(du, 0, (size_t)S * 3 * n_eval * sizeof(double), ctx->stream));
    }
    int rc = mpcx_propagate_thrust_batch_ragged_dev(ctx, S, n_eval, dne, dy0, dtf, dc, flags, ctrl_kind, dv, Ku, dku, de, max_step, dy,
                                                    du, dst, dns, ctx->stream);
    if (rc) return rc;
    ar.download(y_out, dy, (size_t)S * 7 * n_eval);
    if (du) ar.download(u_out, du, (size_t)S * 3 * n_eval);
    ar.download(status, dst, S); ar.download(nsteps, dns, S);
    return ar.finish();
}

extern "C" int mpcx_propagate_batch_ragged(mpcx_ctx *ctx, int S, int n_eval, const int32_t *n_evals, const double *y0,
                                           const double *tf, const double *consts, int flags, int ctrl_kind,
                                           const double *ctrl_vec, int Ku, const int32_t *Kus, const double *end_tau,
                                           double max_step, double *y_out, int32_t *status, int32_t *nsteps)
{
    return mpcx_propagate_thrust_batch_ragged(ctx, S, n_eval, n_evals, y0, tf, consts, flags, ctrl_kind, ctrl_vec, Ku, Kus, end_tau,
                                              max_step, y_out, nullptr, status, nsteps);
}

extern "C" int mpcx_propagate_batch(mpcx_ctx *ctx, int S, int n_eval, const double *y0, const double *tf,
                                    const double *consts, int flags, int ctrl_kind, const double *ctrl_vec, int Ku,
                                    const double *end_tau, double max_step, double *y_out, int32_t *status,
                                    int32_t *nsteps)
{
    return mpcx_propagate_batch_ragged(ctx, S, n_eval, nullptr, y0, tf, consts, flags, ctrl_kind, ctrl_vec, Ku, nullptr, end_tau,
                                       max_step, y_out, status, nsteps);
}
