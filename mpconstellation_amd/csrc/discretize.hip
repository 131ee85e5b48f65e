// discretize.hip -- batched linearise/discretise of the orbital dynamics on gfx950.
//
// Replaces Discretizer.discretize / get_matrices of the reference
// (linearize_discretize.py:8-82, 257-291, 334-390) for S satellites x (K-1) intervals.
//
// Mapping (CDNA4, wave64): one 8-lane group per (satellite, interval), 8 groups per wave.
// The 56-component ODE state [Phi (7x7) ; x (7)] is held column-wise: lane c < 7 owns column c
// of Phi, lane 7 owns x.  Every column obeys the same linear ODE d(col)/dtau = A(x,u) col, so
// the 8 lanes run the same instruction stream; the only cross-lane traffic per RHS evaluation
// is the broadcast of (r, m) from lane 7 (4 x two v_mov_b64_dpp, mpcx_device.hpp) and the 3-step butterfly
// of the RMS error norm (DPP as well: no LDS round trip in the step loop).  All groups of a wave take their own adaptive RK45 step sequence
// (scipy's controller, reproduced decision for decision); the loop is wave-uniform and a
// group that has reached its end point idles under predicate.  At every accepted node each
// lane forms one of the 8 quadrature columns [B lam-, B lam+, Sigma, xi], solves
// Phi(tau_i) g = column (6x6 elimination with partial pivoting, distributed over the group by columns: lu_solve_cols)
// and adds its trapezoid slice, so no node is ever stored.
#include "mpcx_device.hpp"
#include "mpcx_host.hpp"

namespace mpcx {

constexpr int kRec = 112;              
constexpr int kMaxRkIters = 200000;   // attempts per interval before giving up (status STEP)

enum { LAYOUT_STAGE = 0, LAYOUT_REF = 1 };

struct DiscArgs {
    int S, K, Ku, flags;                 // K, Ku: nodes / thrust-table columns of every satellite, or (Ks / Kus given) row lengths
    const int32_t *Ks, *Kus;             // ragged batch: per-satellite node and table-column counts; nullptr: all K / Ku
    double max_step;
    const double *xbar, *ubar, *tf, *consts;
    double *stage;                       // LAYOUT_STAGE
    double *A, *Bp, *Bn, *Sigma, *xi;    // LAYOUT_REF
    int32_t *status;
};

__device__ __forceinline__ double group_sum(double v)
{
    return group_sum8(v);   // bitwise identical on the 8 lanes (each level adds the same two operands); DPP, no LDS round trip
}

struct RhsCtx {
    const double *us;
    int Ku, ldu, flags, c;
    double tf, inv_ve;   // inv_ve = 1 / (g0 Isp)
    SatConst cst;
    FohCache foh;        // the thrust table's interval in use (see foh3_cached)
};

// 1/d and 1/sqrt(d) for d > 0 well inside the normal range: hardware seed + two Newton steps (half an ulp, measured:
// profiles/tools/rcp_accuracy.hip) instead of the IEEE division / square-root sequences (~3x the instructions)
__device__ __forceinline__ double rcp_nr(double d)
{
    double r = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int n = 0; n < 2; ++n) { const double e = fma(-d, r, 1.0); r = fma(r, e, r); }
    return r;
}
__device__ __forceinline__ double rsq_nr(double d)
{
    double r = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int n = 0; n < 2; ++n) { const double e = fma(-d * r, r, 1.0); r = fma(0.5 * r, e, r); }
    return r;
}

// The linearisation at one point (x, u): tf * G = tf * d a / d r, tf * gm = tf * d a / d m (jacobian_blocks), the acceleration and the
// mass flow of Simulator.satellite_dynamics (dynamics_unscaled, tf = 1).  The ~48 right-hand sides and ~9 quadrature nodes per
// interval are nearly all of the kernel's instructions, and a third of theirs were IEEE division / square-root sequences:
// here everything is arranged around one reciprocal each of |r|, m and |u| (round 3).  Results move by rounding only (a few
// ulp per evaluation, 1e-13 on A_k, B_k against the reference's arrays; the accepted RK45 nodes are the same).
struct Lin {
    double Gt[3][3], gmt[3], acc[3], mdot, im, iun, un;
};

__device__ __forceinline__ void lin_eval(const double (&r)[3], const double (&v)[3], double m, const double (&u)[3],
                                         const SatConst &c, int flags, double tf, double inv_ve, Lin &L)
{
    const double rx = r[0], ry = r[1], rz = r[2];
    const double r2 = rx * rx + ry * ry + rz * rz;
    const double irn = rsq_nr(r2), irn2 = irn * irn, ir3 = irn2 * irn, ir5 = ir3 * irn2;
    const double im = rcp_nr(m > 0.0 ? m : 1.0);
    const double c1 = -c.mu * ir3, c1t = tf * c1, c2t = tf * (3.0 * c.mu * ir5);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) L.Gt[i][j] = (i == j ? c1t : 0.0) + c2t * (r[i] * r[j]);
    double j2a[3] = {0.0, 0.0, 0.0};
    if (flags & MPCX_FLAG_J2) {
        const double kJ2 = 1.5 * c.j2 * c.mu * (c.re * c.re);
        const double q2 = (rz * rz) * irn2;
        const double g[3] = {5.0 * q2 - 1.0, 5.0 * q2 - 1.0, 5.0 * q2 - 3.0};
        const double ir4 = irn2 * irn2, ir7 = ir5 * irn2;
        double ddr[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) ddr[j] = 5.0 * (rz * rz) * (-2.0 * (r[j] * ir4));
        ddr[2] += (5.0 * irn2) * (2.0 * rz);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double t = ((kJ2 * g[i]) * r[i]) * (-5.0 * r[j] * ir7) + kJ2 * ir5 * (r[i] * ddr[j]);
                if (i == j) t += kJ2 * ir5 * g[i];
                L.Gt[i][j] += tf * t;
            }
            j2a[i] = (kJ2 * ir5) * (g[i] * r[i]);
        }
    }
    const double gs = -(tf * (im * im));
#pragma unroll
    for (int i = 0; i < 3; ++i) { L.gmt[i] = gs * u[i]; L.acc[i] = c1 * r[i] + u[i] * im + j2a[i]; }
    if (flags & MPCX_FLAG_DRAG) {                            // simulator.py:150-153 (the Python Discretizer never sets it)
        const double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const double coef = -0.5 * kCd * c.s * im * (kRho500 / c.rho) * vn;
#pragma unroll
        for (int i = 0; i < 3; ++i) L.acc[i] += coef * v[i];
    }
    const double uu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
    L.iun = rsq_nr(fmax(uu, 1e-300));
    L.un = uu * L.iun;
    L.mdot = -L.un * inv_ve;
    L.im = im;
}

// One evaluation of dPhi (linearize_discretize.py:262-290) for this lane's column.
__device__ __forceinline__ void rhs_eval(RhsCtx &p, const double (&ys)[7], double ts,
                                         double (&out)[7], int &err)
{
    double u[3];
    foh3_cached(ts, p.us, p.Ku, p.ldu, p.foh, u, err);
    const double r[3] = {bcast8<7>(ys[0]), bcast8<7>(ys[1]), bcast8<7>(ys[2])};
    const double v[3] = {ys[3], ys[4], ys[5]};               // drag acts in the x column only: lane 7's own velocity
    const double m = bcast8<7>(ys[6]);
    const double tf = p.tf;
    Lin L;
    lin_eval(r, v, m, u, p.cst, p.flags, tf, p.inv_ve, L);
    const bool isx = (p.c == 7);
    if (isx && m <= 0.0) err = MPCX_ST_MASS;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        out[i] = tf * ys[3 + i];                             // both the Phi columns ([0 I 0] rows of Dxf) and x
        double a = L.Gt[i][0] * ys[0];                       // Phi column: (tf * Dxf) @ col
        a += L.Gt[i][1] * ys[1];
        a += L.Gt[i][2] * ys[2];
        a += L.gmt[i] * ys[6];
        out[3 + i] = isx ? tf * L.acc[i] : a;                // x column: tf * f(x, u)
    }
    out[6] = isx ? tf * L.mdot : 0.0;
}

// Solve P z = b for the 8 right-hand sides of a group at once, P distributed by columns: lane j < 6 holds column j in
// col[], every lane its own right-hand side in b[] (lanes 6, 7 run the same instructions on columns nobody reads).
// Gaussian elimination with partial pivoting, the operations of a LAPACK-style solve in the same order: at step p the
// owner of column p finds the pivot row and the multipliers, both go to the group by DPP broadcast, every lane swaps and
// eliminates in its own column and right-hand side.  No lane factorises the whole matrix (8x fewer elimination
// operations than the replicated 6x6 of round 2, no LDS staging of Phi, 60 registers less).  Returns false on a zero pivot.
__device__ __forceinline__ bool lu_solve_cols(double (&col)[6], double (&b)[6])
{
    bool ok = true;
    double invd[6];
#define MPCX_LU_STEP(P_)                                                                              \
    {                                                                                                 \
        constexpr int p = P_;                                                                         \
        int piv = p;                                                                                  \
        double best = fabs(col[p]);                                                                   \
        _Pragma("unroll") for (int i = p + 1; i < 6; ++i) {                                           \
            const double v = fabs(col[i]);                                                            \
            if (v > best) { best = v; piv = i; }                                                      \
        }                                                                                             \
        piv = bcast8i<p>(piv);                                                                        \
        if (bcast8<p>(best) == 0.0) ok = false;                                                       \
        _Pragma("unroll") for (int i = p + 1; i < 6; ++i) {                                           \
            const bool sw = (piv == i);                                                               \
            const double a = col[p], bb = col[i];                                                     \
            col[p] = sw ? bb : a; col[i] = sw ? a : bb;                                               \
            const double ra = b[p], rb = b[i];                                                        \
            b[p] = sw ? rb : ra; b[i] = sw ? ra : rb;                                                 \
        }                                                                                             \
        invd[p] = bcast8<p>(rcp_nr(col[p]));                                                          \
        _Pragma("unroll") for (int i = p + 1; i < 6; ++i) {                                           \
            const double mlt = bcast8<p>(col[i]) * invd[p];                                           \
            col[i] -= mlt * col[p];                                                                   \
            b[i] -= mlt * b[p];                                                                       \
        }                                                                                             \
    }
    MPCX_LU_STEP(0) MPCX_LU_STEP(1) MPCX_LU_STEP(2) MPCX_LU_STEP(3) MPCX_LU_STEP(4) MPCX_LU_STEP(5)
#undef MPCX_LU_STEP
    // back substitution: U[i][j] lives on lane j
    b[5] *= invd[5];
    b[4] = (b[4] - bcast8<5>(col[4]) * b[5]) * invd[4];
    b[3] = (b[3] - bcast8<4>(col[3]) * b[4] - bcast8<5>(col[3]) * b[5]) * invd[3];
    b[2] = (b[2] - bcast8<3>(col[2]) * b[3] - bcast8<4>(col[2]) * b[4] - bcast8<5>(col[2]) * b[5]) * invd[2];
    b[1] = (b[1] - bcast8<2>(col[1]) * b[2] - bcast8<3>(col[1]) * b[3] - bcast8<4>(col[1]) * b[4] - bcast8<5>(col[1]) * b[5]) * invd[1];
    b[0] = (b[0] - bcast8<1>(col[0]) * b[1] - bcast8<2>(col[0]) * b[2] - bcast8<3>(col[0]) * b[3] - bcast8<4>(col[0]) * b[4] -
            bcast8<5>(col[0]) * b[5]) * invd[0];
    return ok;
}

// Quadrature integrand column of this lane at an accepted node (linearize_discretize.py:60-75):
// g = Phi(t)^-1 [B lam-, B lam+, Sigma, xi][:, c]
__device__ __forceinline__ void node_integrand(RhsCtx &p, const double (&y)[7],
                                               double t, double tau_k, double tau_kp1,
                                               double (&g)[7], int &err)
{
    const int c = p.c;
    double x[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = bcast8<7>(y[i]);
    double u[3];
    foh3_cached(t, p.us, p.Ku, p.ldu, p.foh, u, err);
    const double lam_n = (tau_kp1 - t) / (tau_kp1 - tau_k);      // exact 1 and 0 at the interval's ends
    const double lam_p = (t - tau_k) / (tau_kp1 - tau_k);
    const double tf = p.tf;

    const double r[3] = {x[0], x[1], x[2]}, v[3] = {x[3], x[4], x[5]};
    Lin L;
    lin_eval(r, v, x[6], u, p.cst, p.flags, tf, p.inv_ve, L);

    // B column (B_func :186-215), Sigma (:239-254), xi (:218-236)
    const int j = (c < 3) ? c : c - 3;
    const double uj = (j == 0) ? u[0] : (j == 1 ? u[1] : u[2]);
    const double lam = (c < 3) ? lam_n : lam_p;
    const double Bm = (tf * L.im) * lam;
    const bool nou = L.un <= kEps;                        // B_func's guard: no mass-flow sensitivity at zero thrust
    const double b6s = -(tf * (p.inv_ve * L.iun));        // tf * d mdot / d u_j = b6s * u_j
    const double B6 = nou ? 0.0 : (b6s * uj) * lam;
    double xi[7];
    double bu6 = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        xi[i] = -(tf * x[3 + i]);
        double ax = L.Gt[i][0] * x[0];
        ax += L.Gt[i][1] * x[1];
        ax += L.Gt[i][2] * x[2];
        ax += L.gmt[i] * x[6];
        xi[3 + i] = -(ax + (tf * L.im) * u[i]);
        bu6 += nou ? 0.0 : (b6s * u[i]) * u[i];
    }
    xi[6] = -(0.0 + bu6);
    const double sg[7] = {x[3], x[4], x[5], L.acc[0], L.acc[1], L.acc[2], L.mdot};      // Sigma = f(.; tf = 1)

    double R[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        double rb = 0.0;
        if (i == 3 + 0) rb = (j == 0) ? Bm : 0.0;
        if (i == 3 + 1) rb = (j == 1) ? Bm : 0.0;
        if (i == 3 + 2) rb = (j == 2) ? Bm : 0.0;
        if (i == 6) rb = B6;
        R[i] = (c < 6) ? rb : (c == 6 ? sg[i] : xi[i]);
    }
    // Phi = [[P q],[0 1]]  =>  g6 = R6 ; P g' = R' - q R6      (column j of P is lane j's y, q lane 6's)
    double col[6], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { col[i] = y[i]; b[i] = R[i] - bcast8<6>(y[i]) * R[6]; }
    if (!lu_solve_cols(col, b)) err = MPCX_ST_SINGULAR;
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = b[i];
    g[6] = R[6];
}

// UNIFORM: Discretizer.use_uniform_steps (linearize_discretize.py:27-30, 50-53): the quadrature nodes are integrator_steps
// uniform points per interval taken from the RK45 dense-output interpolant (scipy's t_eval branch), A_k the interpolant
// at the last of them; a separate instantiation, the default path is untouched.
// METHOD: 45 -- scipy's 'RK45', the reference's default (linearize_discretize.py:105) -- or 23: 'RK23' (ivp_solver goes to
// solve_ivp's `method`, :40): the same controller around the Bogacki-Shampine tableau, separate instantiations.
template <int LAYOUT, bool UNIFORM, int METHOD = 45>
#ifndef MPCX_DISC_WAVES
#define MPCX_DISC_WAVES 1      // waves per SIMD the register allocation is bounded for (360 registers at 1)
#endif
__global__ __launch_bounds__(64, MPCX_DISC_WAVES) void discretize_kernel(DiscArgs a)
{
    __shared__ double lds[8 * kRec];
    const int lane = threadIdx.x;
    const int grp = lane >> 3, c = lane & 7;
    double *rec = lds + grp * kRec;

    const int Km1 = a.K - 1;                        // slots per satellite in the output arrays
    const long total = (long)a.S * Km1;
    const long item0 = (long)blockIdx.x * 8;
    const long item = item0 + grp;
    const bool in_range = item < total;
    const long it = in_range ? item : total - 1;      // tail groups shadow the last item, write nothing
    const int s = (int)(it / Km1), kslot = (int)(it % Km1);
    // ragged batch: this satellite's own node count; the slots past its last interval shadow that interval (their output
    // lands in slots nobody reads)
    int Ks = a.Ks ? a.Ks[s] : a.K;
    const bool badk = Ks < 2 || Ks > a.K;
    Ks = badk ? a.K : Ks;
    const int Km1s = Ks - 1;
    const bool valid = in_range && kslot < Km1s;
    const int k = kslot < Km1s ? kslot : Km1s - 1;

    RhsCtx p;
    p.us = a.ubar + (size_t)s * 3 * a.Ku;
    p.Ku = a.Kus ? a.Kus[s] : a.Ku; p.ldu = a.Ku;
    const bool badku = p.Ku < 2 || p.Ku > a.Ku;
    if (badku) p.Ku = a.Ku;
    p.flags = a.flags & (MPCX_FLAG_DRAG | MPCX_FLAG_J2); p.c = c; p.foh.reset();
    const int n_uni = UNIFORM ? (a.flags >> 8) : 0;                 // integrator_steps
    p.tf = a.tf[s];
    p.cst.load(a.consts + (size_t)s * MPCX_NCONST);
    p.inv_ve = 1.0 / (p.cst.g0 * p.cst.isp);
    const double *xs = a.xbar + (size_t)s * 7 * a.K;

    // np.linspace(0, 1, K)[k], [k+1]
    const double step = 1.0 / (double)Km1s;
    const double tau_k = (double)k * step + 0.0;
    const double tau_kp1 = (k + 1 == Km1s) ? 1.0 : (double)(k + 1) * step + 0.0;
    const double t_bound = tau_kp1;
    const double rtol = 1e-3, atol = 1e-6;
    constexpr double kErrExp = (METHOD == 23) ? -1.0 / 3.0 : -0.2;      // rk.py:93: -1 / (error_estimator_order + 1)
    int err = (badk || badku) ? MPCX_ST_BADK : 0;

    double y[7], f[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) y[i] = (c < 7) ? ((i == c) ? 1.0 : 0.0) : xs[(size_t)i * a.K + k];
    double t = tau_k;
    rhs_eval(p, y, t, f, err);

    // ---- scipy select_initial_step (common.py:68-134), direction +1, order 4 ----
    double h_abs;
    {
        const double interval = fabs(t_bound - t);
        double s0 = 0.0, s1 = 0.0, scale[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            scale[i] = atol + fabs(y[i]) * rtol;
            const double a0 = y[i] / scale[i], a1 = f[i] / scale[i];
            s0 += a0 * a0; s1 += a1 * a1;
        }
        const double inv_sqrt_n = 1.0 / sqrt(56.0);
        const double d0 = sqrt(group_sum(s0)) * inv_sqrt_n, d1 = sqrt(group_sum(s1)) * inv_sqrt_n;
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        h0 = fmin(h0, interval);
        double y1[7], f1[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) y1[i] = y[i] + h0 * f[i];
        rhs_eval(p, y1, t + h0, f1, err);
        double s2 = 0.0;
#pragma unroll
        for (int i = 0; i < 7; ++i) { const double d = (f1[i] - f[i]) / scale[i]; s2 += d * d; }
        const double d2 = sqrt(group_sum(s2)) * inv_sqrt_n / h0;
        double h1;
        if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
        else h1 = pow(0.01 / fmax(d1, d2), METHOD == 23 ? 1.0 / 3.0 : 1.0 / 5.0);       // 1 / (error_estimator_order + 1)
        h_abs = fmin(fmin(100.0 * h0, h1), fmin(interval, a.max_step));
        if (interval == 0.0) h_abs = 0.0;
    }

    // ---- quadrature state: trapezoid accumulators for this lane's column ----
    double acc[7], gprev[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = 0.0;
    node_integrand(p, y, t, tau_k, tau_kp1, gprev, err);

    // uniform mode: index of the next evaluation point, time of the previous one, the interpolated state at it
    const double ustep = UNIFORM ? (tau_kp1 - tau_k) / (double)(n_uni - 1) : 0.0;   // np.linspace(tau_k, tau_kp1, n)
    int ei = 1;                                                    // (point 0 = tau_k: the start state itself, done above)
    double te_prev = tau_k, ylast[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) ylast[i] = y[i];

    // ---- adaptive RK45 (rk.py:110-168); retry-after-reject folded into the same loop ----
    bool rejected = false;
    for (int iter = 0;; ++iter) {
        const bool active = (t != t_bound);
        if (!__any(active)) break;
        if (iter >= kMaxRkIters) {       // every wave drains: hard cap on attempts
            if (active) err = MPCX_ST_STEP;
            break;
        }
        const double min_step = 10.0 * fabs(nextafter(t, INFINITY) - t);
        if (!rejected) {
            if (h_abs > a.max_step) h_abs = a.max_step;
            else if (h_abs < min_step) h_abs = min_step;
        }
        bool fail = active && !(h_abs >= min_step);   // also catches a non-finite step
        double h = h_abs;
        double t_new = t + h;
        if (t_new - t_bound > 0.0) t_new = t_bound;
        h = t_new - t;
        const double h_try = fabs(h);

        double K1[7], K2[7], K3[7], K4[7], K5[7], K6[7], yt[7], yn[7];
        double se = 0.0;
        if constexpr (METHOD == 23) {
            // rk_step (rk.py:14-70) with the RK23 tableau: two inner stages, y_new, and f(y_new) -- kept in K6, the slot of the
            // last stage in both methods (first-same-as-last: it becomes f of the next step)
#pragma unroll
            for (int i = 0; i < 7; ++i) yt[i] = y[i] + (f[i] * RK23_A10) * h;
            rhs_eval(p, yt, t + RK23_C[1] * h, K1, err);
#pragma unroll
            for (int i = 0; i < 7; ++i) yt[i] = y[i] + (f[i] * 0.0 + K1[i] * RK23_A21) * h;
            rhs_eval(p, yt, t + RK23_C[2] * h, K2, err);
#pragma unroll
            for (int i = 0; i < 7; ++i) yn[i] = y[i] + h * (f[i] * RK23_B[0] + K1[i] * RK23_B[1] + K2[i] * RK23_B[2]);
            rhs_eval(p, yn, t + h, K6, err);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                K3[i] = 0.0; K4[i] = 0.0; K5[i] = 0.0;
                const double e = f[i] * RK23_E[0] + K1[i] * RK23_E[1] + K2[i] * RK23_E[2] + K6[i] * RK23_E[3];
                const double sc = atol + fmax(fabs(y[i]), fabs(yn[i])) * rtol;
                const double q = (e * h) * rcp_nr(sc);
                se += q * q;
            }
        } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) yt[i] = y[i] + (f[i] * RK_A[1][0]) * h;
        rhs_eval(p, yt, t + RK_C[1] * h, K1, err);
#pragma unroll
        for (int i = 0; i < 7; ++i) yt[i] = y[i] + (f[i] * RK_A[2][0] + K1[i] * RK_A[2][1]) * h;
        rhs_eval(p, yt, t + RK_C[2] * h, K2, err);
#pragma unroll
        for (int i = 0; i < 7; ++i)
            yt[i] = y[i] + (f[i] * RK_A[3][0] + K1[i] * RK_A[3][1] + K2[i] * RK_A[3][2]) * h;
        rhs_eval(p, yt, t + RK_C[3] * h, K3, err);
#pragma unroll
        for (int i = 0; i < 7; ++i)
            yt[i] = y[i] + (f[i] * RK_A[4][0] + K1[i] * RK_A[4][1] + K2[i] * RK_A[4][2] +
                            K3[i] * RK_A[4][3]) * h;
        rhs_eval(p, yt, t + RK_C[4] * h, K4, err);
#pragma unroll
        for (int i = 0; i < 7; ++i)
            yt[i] = y[i] + (f[i] * RK_A[5][0] + K1[i] * RK_A[5][1] + K2[i] * RK_A[5][2] +
                            K3[i] * RK_A[5][3] + K4[i] * RK_A[5][4]) * h;
        rhs_eval(p, yt, t + RK_C[5] * h, K5, err);
#pragma unroll
        for (int i = 0; i < 7; ++i)
            yn[i] = y[i] + h * (f[i] * RK_B[0] + K1[i] * RK_B[1] + K2[i] * RK_B[2] +
                                K3[i] * RK_B[3] + K4[i] * RK_B[4] + K5[i] * RK_B[5]);
        rhs_eval(p, yn, t + h, K6, err);

#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const double e = f[i] * RK_E[0] + K1[i] * RK_E[1] + K2[i] * RK_E[2] + K3[i] * RK_E[3] +
                             K4[i] * RK_E[4] + K5[i] * RK_E[5] + K6[i] * RK_E[6];
            const double sc = atol + fmax(fabs(y[i]), fabs(yn[i])) * rtol;
            const double q = (e * h) * rcp_nr(sc);
            se += q * q;
        }
        }
        const double error_norm = sqrt(group_sum(se)) / sqrt(56.0);
        bool accept = false;
        if (error_norm < 1.0) {
            double factor = (error_norm == 0.0) ? RK_MAX_FACTOR
                                                : fmin(RK_MAX_FACTOR, RK_SAFETY * pow(error_norm, kErrExp));
            if (rejected) factor = fmin(1.0, factor);
            if (active && !fail) { h_abs = h_try * factor; accept = true; }
        } else if (active && !fail) {
            // NaN error norms land here as in scipy (comparison false) and shrink the step
            h_abs = h_try * fmax(RK_MIN_FACTOR, RK_SAFETY * pow(error_norm, kErrExp));
            rejected = true;
        }
        if (fail) {            // scipy: TOO_SMALL_STEP -> solver fails; freeze this group
            err = MPCX_ST_STEP;
            t = t_bound;
        }
        double yold[7];
        const double t_old = t;
        if (accept) {
            rejected = false;
            t = t_new;
#pragma unroll
            for (int i = 0; i < 7; ++i) { yold[i] = y[i]; y[i] = yn[i]; }
        }
        if (UNIFORM) {
            // scipy ivp.py: t_eval points with t_old < te <= t get sol(te) = y_old + h Q p(x), Q = K^T P, x = (te - t_old) / h
            // (rk.py:552-574); each is a quadrature node.  Groups have different numbers of points in their step: the
            // loop runs while any group has one (node_integrand is a wave-uniform call), the others idle.
            if (__any(accept)) {
                double Q[7][4];
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    if constexpr (METHOD == 23) {
                        const double kk[4] = {f[i], K1[i], K2[i], K6[i]};       // (K of rk.py: f, the two inner stages, f(y_new))
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {
                            double q = 0.0;
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) q += kk[jj] * RK23_P[jj][cc];
                            Q[i][cc] = q;
                        }
                        Q[i][3] = 0.0;
                    } else {
                    const double kk[7] = {f[i], K1[i], K2[i], K3[i], K4[i], K5[i], K6[i]};
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        double q = 0.0;
#pragma unroll
                        for (int jj = 0; jj < 7; ++jj) q += kk[jj] * RK_P[jj][cc];
                        Q[i][cc] = q;
                    }
                    }
                }
                for (;;) {
                    const double te = (ei == n_uni - 1) ? tau_kp1 : (double)ei * ustep + tau_k;
                    const bool has = accept && ei < n_uni && !(te > t);
                    if (!__any(has)) break;
                    const double xx = (te - t_old) / h;
                    const double p1 = xx, p2 = p1 * xx, p3 = p2 * xx, p4 = p3 * xx;
                    double yd[7], g[7];
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        const double accq = (METHOD == 23) ? Q[i][0] * p1 + Q[i][1] * p2 + Q[i][2] * p3
                                                            : Q[i][0] * p1 + Q[i][1] * p2 + Q[i][2] * p3 + Q[i][3] * p4;
                        yd[i] = has ? h * accq + yold[i] : y[i];
                    }
                    node_integrand(p, yd, has ? te : t, tau_k, tau_kp1, g, err);
                    if (has) {
                        const double d = te - te_prev;
#pragma unroll
                        for (int i = 0; i < 7; ++i) { acc[i] += d * (g[i] + gprev[i]) / 2.0; gprev[i] = g[i]; ylast[i] = yd[i]; }
                        te_prev = te; ++ei;
                    }
                }
            }
        }
        if (accept) {
#pragma unroll
            for (int i = 0; i < 7; ++i) f[i] = K6[i];
        }
        // node quadrature (wave-uniform call; only accepting groups commit)
        if (!UNIFORM && __any(accept)) {
            double g[7];
            node_integrand(p, y, t, tau_k, tau_kp1, g, err);
            if (accept) {
                const double d = h;          // ts[i+1] - ts[i]
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    acc[i] += d * (g[i] + gprev[i]) / 2.0;
                    gprev[i] = g[i];
                }
            }
        }
    }

    // ---- A_k = Phi(tau_k+1); B_k-, B_k+, Sigma_k, xi_k = A_k @ trapz (:43-44, :77-80) ----
    if (c < 7) {
#pragma unroll
        for (int i = 0; i < 7; ++i) rec[i * 7 + c] = UNIFORM ? ylast[i] : y[i];
    }
    __syncthreads();
    double out[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        double sacc = 0.0;
#pragma unroll
        for (int l = 0; l < 7; ++l) sacc += rec[i * 7 + l] * acc[l];
        out[i] = sacc;
    }
    // record = [A 49 | Bn 21 | Bp 21 | Sigma 7 | xi 7]
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        int off;
        if (c < 3) off = 49 + i * 3 + c;
        else if (c < 6) off = 70 + i * 3 + (c - 3);
        else if (c == 6) off = 91 + i;
        else off = 98 + i;
        rec[off] = out[i];
    }
    __syncthreads();

    if (LAYOUT == LAYOUT_STAGE) {
        const long base = item0 * MPCX_STAGE_DOUBLES;
        const long lim = total * MPCX_STAGE_DOUBLES;
        for (int e = lane; e < 8 * MPCX_STAGE_DOUBLES; e += 64) {
            const int gi = e / MPCX_STAGE_DOUBLES, el = e - gi * MPCX_STAGE_DOUBLES;
            if (base + e < lim) a.stage[base + e] = lds[gi * kRec + el];
        }
    } else {
        for (int e = lane; e < 8 * 49; e += 64) {
            const int gi = e / 49, el = e - gi * 49;
            if (item0 + gi < total) a.A[item0 * 49 + e] = lds[gi * kRec + el];
        }
        for (int e = lane; e < 8 * 21; e += 64) {
            const int gi = e / 21, el = e - gi * 21;
            if (item0 + gi < total) {
                a.Bn[item0 * 21 + e] = lds[gi * kRec + 49 + el];
                a.Bp[item0 * 21 + e] = lds[gi * kRec + 70 + el];
            }
        }
        if (valid && c < 7) {
            const size_t o = (size_t)s * 7 * Km1 + (size_t)c * Km1 + k;
            a.Sigma[o] = rec[91 + c];
            a.xi[o] = rec[98 + c];
        }
    }

    // per-satellite status: worst code over its intervals
    int e8 = err;
    e8 = max(e8, __shfl_xor(e8, 1, 8));
    e8 = max(e8, __shfl_xor(e8, 2, 8));
    e8 = max(e8, __shfl_xor(e8, 4, 8));
    if (valid && c == 0 && e8 != 0) atomicMax(&a.status[s], e8);
}

}  // namespace mpcx

using namespace mpcx;

static int launch_discretize(mpcx_ctx *ctx, int layout, DiscArgs a, hipStream_t st)
{
    if (a.S < 1 || a.K < 2 || a.Ku < 2) return ctx_fail(ctx, MPCX_E_BADARG, "discretize: need S>=1, K>=2, Ku>=2");
    if (!(a.max_step > 0.0)) return ctx_fail(ctx, MPCX_E_BADARG, "discretize: max_step must be > 0");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    MPCX_HIP(ctx, hipMemsetAsync(a.status, 0, sizeof(int32_t) * a.S, st));
    const long total = (long)a.S * (a.K - 1);
    const unsigned blocks = (unsigned)((total + 7) / 8);
    const bool uni = (a.flags & MPCX_FLAG_UNIFORM_STEPS) != 0;
    if (uni && (a.flags >> 8) < 2) return ctx_fail(ctx, MPCX_E_BADARG, "discretize: uniform steps need MPCX_UNIFORM_STEPS(n), n >= 2");
    const bool rk23 = (a.flags & MPCX_FLAG_RK23) != 0;
    if (!uni) a.flags &= (MPCX_FLAG_DRAG | MPCX_FLAG_J2);
#define MPCX_DISC_LAUNCH(L, U, M) hipLaunchKernelGGL((discretize_kernel<L, U, M>), dim3(blocks), dim3(64), 0, st, a)
    if (layout == LAYOUT_STAGE) {
        if (rk23) { if (uni) MPCX_DISC_LAUNCH(LAYOUT_STAGE, true, 23); else MPCX_DISC_LAUNCH(LAYOUT_STAGE, false, 23); }
        else if (uni) MPCX_DISC_LAUNCH(LAYOUT_STAGE, true, 45);
        else MPCX_DISC_LAUNCH(LAYOUT_STAGE, false, 45);
    } else {
        if (rk23) { if (uni) MPCX_DISC_LAUNCH(LAYOUT_REF, true, 23); else MPCX_DISC_LAUNCH(LAYOUT_REF, false, 23); }
        else if (uni) MPCX_DISC_LAUNCH(LAYOUT_REF, true, 45);
        else MPCX_DISC_LAUNCH(LAYOUT_REF, false, 45);
    }
#undef MPCX_DISC_LAUNCH
    MPCX_HIP(ctx, hipGetLastError());
    return MPCX_OK;
}

extern "C" int mpcx_discretize_batch_dev(mpcx_ctx *ctx, int S, int K, int Ku, const double *xbar,
                                         const double *ubar, const double *tf, const double *consts,
                                         int flags, double max_step, double *A, double *Bp,
                                         double *Bn, double *Sigma, double *xi, int32_t *status,
                                         void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    DiscArgs a{S, K, Ku, flags, nullptr, nullptr, max_step, xbar, ubar, tf, consts, nullptr, A, Bp, Bn, Sigma, xi, status};
    return launch_discretize(ctx, LAYOUT_REF, a, (hipStream_t)stream);
}

extern "C" int mpcx_discretize_stages_ragged_dev(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, int Ku, const int32_t *Kus,
                                                 const double *xbar, const double *ubar, const double *tf,
                                                 const double *consts, int flags, double max_step, double *stage,
                                                 int32_t *status, void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    DiscArgs a{S, K, Ku, flags, Ks, Kus, max_step, xbar, ubar, tf, consts, stage, nullptr, nullptr, nullptr, nullptr, nullptr, status};
    return launch_discretize(ctx, LAYOUT_STAGE, a, (hipStream_t)stream);
}

extern "C" int mpcx_discretize_stages_dev(mpcx_ctx *ctx, int S, int K, int Ku, const double *xbar,
                                          const double *ubar, const double *tf, const double *consts,
                                          int flags, double max_step, double *stage, int32_t *status,
                                          void *stream)
{
    return mpcx_discretize_stages_ragged_dev(ctx, S, K, nullptr, Ku, nullptr, xbar, ubar, tf, consts, flags, max_step, stage, status, stream);
}

extern "C" int mpcx_discretize_batch(mpcx_ctx *ctx, int S, int K, int Ku, const double *xbar,
                                     const double *ubar, const double *tf, const double *consts,
                                     int flags, double max_step, double *A, double *Bp, double *Bn,
                                     double *Sigma, double *xi, int32_t *status)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 2 || Ku < 2) return ctx_fail(ctx, MPCX_E_BADARG, "discretize: need S>=1, K>=2, Ku>=2");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)S * (K - 1);
    DeviceArena ar(ctx);
    double *dx = ar.upload(xbar, (size_t)S * 7 * K), *du = ar.upload(ubar, (size_t)S * 3 * Ku);
    double *dtf = ar.upload(tf, S), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST);
    double *dA = ar.alloc<double>(n * 49), *dBp = ar.alloc<double>(n * 21), *dBn = ar.alloc<double>(n * 21);
    double *dS = ar.alloc<double>(n * 7), *dX = ar.alloc<double>(n * 7);
    int32_t *dst = ar.alloc<int32_t>(S);
    if (ar.failed()) return ar.code();
    int rc = mpcx_discretize_batch_dev(ctx, S, K, Ku, dx, du, dtf, dc, flags, max_step, dA, dBp, dBn,
                                       dS, dX, dst, ctx->stream);
    if (rc) return rc;
    ar.download(A, dA, n * 49); ar.download(Bp, dBp, n * 21); ar.download(Bn, dBn, n * 21);
    ar.download(Sigma, dS, n * 7); ar.download(xi, dX, n * 7); ar.download(status, dst, S);
    return ar.finish();
}
