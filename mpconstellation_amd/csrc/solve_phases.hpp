// solve_phases.hpp -- the node-parallel phases of the interior-point iteration (one lane per temporal node: KKT residual and
// line-search trials, Newton blocks, refinement residual, fraction to the boundary, channel combination) and the problem
// constants a satellite keeps in LDS.  Included by solve.hip (namespace mpcx, one wave per satellite) and by solve2w.hip
// (namespace mpcx2w, two waves per satellite): see solve.hip.
#pragma once

namespace MPCX_NS {
#ifdef MPCX_TWO_WAVE
using namespace mpcx;          // (the device helpers of mpcx_device.hpp, the layout of solve_layout.hpp)
#endif

// ---- per-satellite constant data kept in LDS --------------------------------------------
struct ResAcc {   // accumulators of one residual evaluation
    double dual_max, prim_max, sq, zsum, lsum, prod_min, prod_max, prod_sum;
    double g_tf;      // the satellite's term of the tf stationarity row, 2 w_tr (tf - tf_bar) - sum_k Sigma_k . lam_k
};

struct SatData {
    double aT[8][7], bT[8];
    int nT, linvt;           // terminal inequality rows (6, or 8 with the linearised tangential pair); convex variant flag
    int fixed_tf;            // tf is a constant of the problem: no range constraint, no stationarity row, dtf = 0
    int shared;              // tf is ONE variable shared by the satellites of the launch: its row is assembled across workgroups
    double tS, rS;           // shared tf: this satellite's share of the tf pivot (local Schur complement) and of its right-hand side
    double w_vt, gh_vt, zeta_vt;   // convex variant: weight, gradient coefficient and border unknown of the tangential pair
    double b_u, b_rmax, b_rmin, b_rfmax, b_tf[2], vt_des, w_tr, w_nu, tfbar;
    // Newton-step globals
    double WxK[49];          // terminal Hessian used inside the recursion (soft + capped + AL)
    double WxKsoft[49], gxKsoft[7];
    double ta[NTERM][7], tw[NTERM], tgh[NTERM], twin[NTERM];
    double avt[7], Hv[36], cv, gam, Wtf, gtf, sigmax;
    double Mb[NBD][NBD];     // border matrix, then its L D L^T factors (unit lower part, 1/d on the diagonal)
    double Sb[NBD][NBD];     // the border matrix itself (for the residual of the refinement step in border_solve)
    double siglam[NCH], xK[NCH][7];
    double sol[NBD];
    double zeta[NTERM];      // border unknowns of the terminal terms accumulated over the passes of one linear solve
    double red[8];
    // the "global" part of iterate, direction and candidate iterate (GL_N doubles each: slacks / multipliers of the terminal rows
    // and of the tf range, tf, lam_vt).  Round 5: they live HERE, in LDS, in every build -- until then they were three small
    // arrays of the satellite's global workspace, and the terminal-node sections of the node-parallel phases read and wrote them
    // one row at a time in loops the compiler cannot batch (a store to itgB between two loads of itg): seven dependent memory
    // round trips per trial evaluation and per step-limit pass, which is what those phases waited for under load.
    double gl[3][GL_N];
    // What the phases hand back to the driver (round 5: here, in LDS; they were reference parameters into the driver's stack --
    // flat stores in the phase functions, scratch reloads in the driver): the residual evaluation of the iterate ([0]) and of
    // the line search's trial ([1]); the scalars of a right-hand side (first_rhs_scalars / reduced_residual -> border_solve);
    // the finite flag of finish_direction.
    ResAcc racc[2];
    double rs_gtf, rs_rvt, rs_gex[NTERM];
    int dir_finite;
    // ... and the driver's own loop state (solve_satellite): every lane holds the same values, and as local variables they lived
    // in vector registers, which every phase function clobbers -- a scratch store before and a scratch load behind each call.
    // Here a call costs them nothing and a use is an LDS read.
    struct Drv {
        double mu, dw_last, E0, delta_w, alpha, tau, mu_cur, rn0, mu_clip;
        int n_acc, status, it_count, n_reg, first_reg, mono, refined_prev, n_small, iter, have_dir, have_trial, clean, ls;
    } dv;
    double infeas;           // > 0: the constraint set is empty whatever the dynamics (structural_violation)
    int flag;
#ifdef MPCX_PHASE_TIMING
    unsigned long long fpt[16];   // diagnostic build only: cycle sums of the recursion's inner phases
#endif
};

__device__ __forceinline__ double relax(double b) { return b + kBoundRelax * fmax(1.0, fabs(b)); }

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// gshfl8(v, q): value of lane q of the caller's 8-lane group, q a constant after unrolling -> bcast8<q> (mpcx_device.hpp: two
// v_mov_b64_dpp) instead of the ds_bpermute_b32 pair __shfl(v, q, 8) compiles to.  The sweeps exchange ~70 doubles per node
// this way; as ds_bpermute they were 40 % of the kernel's LDS instructions, and at two waves per SIMD the CU's LDS pipe
// (shared by its four SIMDs) is the resource the kernel saturates first (profiles/r02/pmc_sq.json: SQ_ACTIVE_INST_LDS).
__device__ __forceinline__ double gshfl8(double v, int q)
{
    switch (q) {
    case 0: return bcast8<0>(v);
    case 1: return bcast8<1>(v);
    case 2: return bcast8<2>(v);
    case 3: return bcast8<3>(v);
    case 4: return bcast8<4>(v);
    case 5: return bcast8<5>(v);
    case 6: return bcast8<6>(v);
    default: return bcast8<7>(v);
    }
}

// Barrier for the single-wave workgroups of this kernel when lanes exchange data through LDS only: DS operations of
// one wave execute in issue order, so it is enough to stop the compiler from moving LDS accesses across this point.
// Unlike WG_SYNC() it does not drain outstanding global loads (the node-ahead prefetch stays in flight).
__device__ __forceinline__ void wsync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// A value that is the same in all 64 lanes, moved to scalar registers (two v_readfirstlane for a double).  The driver of a solve
// (solve_satellite) keeps its loop state -- mu, step length, error, counters -- across the calls of the phase functions, which
// clobber every vector register: as vector values each of them is a scratch store before a call and a scratch load behind it.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// 1/d for d > 0 well inside the normal range: hardware seed + two Newton steps (the full IEEE division sequence
// with its scaling / fix-up is not needed for pivots, slacks and determinants).  Measured on gfx950 over 1e-40..1e40
// (profiles/tools/rcp_accuracy.hip): seed 4.5e-8 relative, one step 2.1e-15, two steps 1.1e-16 = half an ulp.
__device__ __forceinline__ double rcp_pos(double d)
{
    double r = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int n = 0; n < 2; ++n) { const double e = fma(-d, r, 1.0); r = fma(r, e, r); }
    return r;
}

// c~ = |v|^2 - (r.v)^2/|r|^2 - vt_des^2 (same zero set as the quartic of optimizer.py:492-517)
__device__ void vt_reduced(const double *x, double vt_des, double &c, double *g6, double *H36)
{
    const double *r = x, *v = x + 3;
    const double q = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    const double rv = r[0] * v[0] + r[1] * v[1] + r[2] * v[2];
    const double iq = 1.0 / q, iq2 = iq * iq, iq3 = iq2 * iq;       // one division, the powers of 1/q by products
    c = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] - rv * rv * iq - vt_des * vt_des;
    if (!g6) return;
    for (int i = 0; i < 3; ++i) {
        g6[i] = -2.0 * rv * v[i] * iq + 2.0 * rv * rv * r[i] * iq2;
        g6[3 + i] = 2.0 * v[i] - 2.0 * rv * r[i] * iq;
    }
    if (!H36) return;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double I = (i == j) ? 1.0 : 0.0;
            const double Hvv = 2.0 * I - 2.0 * r[i] * r[j] * iq;
            const double Hrr = -2.0 * v[i] * v[j] * iq + 4.0 * rv * (v[i] * r[j] + r[i] * v[j]) * iq2 +
                               2.0 * rv * rv * I * iq2 - 8.0 * rv * rv * r[i] * r[j] * iq3;
            const double Hrv = -2.0 * v[i] * r[j] * iq - 2.0 * rv * I * iq + 4.0 * rv * r[i] * r[j] * iq2;
            H36[i * 6 + j] = Hrr;
            H36[(3 + i) * 6 + 3 + j] = Hvv;
            H36[i * 6 + 3 + j] = Hrv;
            H36[(3 + j) * 6 + i] = Hrv;
        }
}

// Optimizer.get_constraint_terms (optimizer.py:80-170) for the terminal node, incl. the
// operator-precedence form of Dv_h_hat (:122); builds the six terminal linear inequalities.
__device__ __noinline__ void build_terminal(const double *xK, double mu_grav, double r_des, const SolveOpts &o, SatData &sd)
{
    double r[3] = {xK[0], xK[1], xK[2]}, v[3] = {xK[3], xK[4], xK[5]}, h[3], rh[3], hh[3];
    const double rn = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    h[0] = r[1] * v[2] - r[2] * v[1]; h[1] = r[2] * v[0] - r[0] * v[2]; h[2] = r[0] * v[1] - r[1] * v[0];
    const double hn = sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]);
    for (int i = 0; i < 3; ++i) { rh[i] = r[i] / rn; hh[i] = h[i] / hn; }
    const double ihn = 1.0 / hn, ihn3 = 1.0 / (hn * hn * hn), irn = 1.0 / rn, irn3 = 1.0 / (rn * rn * rn);
    double Ph[9], hh3[9], nSv[9], Sr[9], Dr_h[9], Dv_h[9], Dr_r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            hh3[i * 3 + j] = ihn3 * (h[i] * h[j]);
            Ph[i * 3 + j] = (i == j ? ihn : 0.0) - hh3[i * 3 + j];
            Dr_r[i * 3 + j] = (i == j ? irn : 0.0) - irn3 * (r[i] * r[j]);
        }
    // -skew(v), skew(r)
    nSv[0] = 0; nSv[1] = v[2]; nSv[2] = -v[1]; nSv[3] = -v[2]; nSv[4] = 0; nSv[5] = v[0]; nSv[6] = v[1]; nSv[7] = -v[0]; nSv[8] = 0;
    Sr[0] = 0; Sr[1] = -r[2]; Sr[2] = r[1]; Sr[3] = r[2]; Sr[4] = 0; Sr[5] = -r[0]; Sr[6] = -r[1]; Sr[7] = r[0]; Sr[8] = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0.0, b = 0.0;
            for (int l = 0; l < 3; ++l) { a += Ph[i * 3 + l] * nSv[l * 3 + j]; b += hh3[i * 3 + l] * Sr[l * 3 + j]; }
            Dr_h[i * 3 + j] = a;
            Dv_h[i * 3 + j] = (i == j ? ihn : 0.0) - b;
        }
    double DrVr[3], DrVn[3], DvVn[3];
    for (int j = 0; j < 3; ++j) {
        DrVr[j] = v[0] * Dr_r[j] + v[1] * Dr_r[3 + j] + v[2] * Dr_r[6 + j];
        DrVn[j] = v[0] * Dr_h[j] + v[1] * Dr_h[3 + j] + v[2] * Dr_h[6 + j];
        DvVn[j] = hh[j] + (v[0] * Dv_h[j] + v[1] * Dv_h[3 + j] + v[2] * Dv_h[6 + j]);
    }
    const double Vr = v[0] * rh[0] + v[1] * rh[1] + v[2] * rh[2];
    const double Vn = v[0] * hh[0] + v[1] * hh[1] + v[2] * hh[2];
    double gR[6] = {DrVr[0], DrVr[1], DrVr[2], rh[0], rh[1], rh[2]};
    double gN[6] = {DrVn[0], DrVn[1], DrVn[2], DvVn[0], DvVn[1], DvVn[2]};
    double gRbar = 0.0, gNbar = 0.0;
    for (int i = 0; i < 6; ++i) { gRbar += gR[i] * xK[i]; gNbar += gN[i] * xK[i]; }
    for (int i = 0; i < 8; ++i) { sd.bT[i] = 0.0; for (int j = 0; j < 7; ++j) sd.aT[i][j] = 0.0; }
    sd.linvt = o.linvt; sd.nT = o.linvt ? 8 : 6; sd.fixed_tf = o.fixed_tf | o.shared_tf; sd.shared = o.shared_tf;
    sd.w_vt = 0.0; sd.gh_vt = 0.0; sd.zeta_vt = 0.0;
    for (int j = 0; j < 3; ++j) sd.aT[0][j] = -rh[j];
    sd.bT[0] = relax(-(r_des - o.eps_r));
    for (int j = 0; j < 6; ++j) { sd.aT[1][j] = gR[j]; sd.aT[2][j] = -gR[j]; sd.aT[3][j] = gN[j]; sd.aT[4][j] = -gN[j]; }
    const double c0r = Vr - gRbar, c0n = Vn - gNbar;
    sd.bT[1] = relax(o.eps_vr - c0r); sd.bT[2] = relax(o.eps_vr + c0r);
    sd.bT[3] = relax(o.eps_vn - c0n); sd.bT[4] = relax(o.eps_vn + c0n);
    sd.aT[5][6] = -1.0; sd.bT[5] = relax(-o.min_mass);
    if (o.linvt) {
        // optimizer.py:119,124-125,146-153: t_hat = h_hat x r_hat, Dr_t = -skew(r_hat) Dr_h + skew(h_hat) Dr_r,
        // Dv_t = -skew(r_hat) Dv_h, Vt = v.t_hat, Vc = sqrt(mu/|r|), DrVc = -1/2 sqrt(mu) |r|^(-5/2) r;
        // rows 6 / 7: min_tan_vel_rule / max_tan_vel_rule (:480-489 / :471-479)
        double th[3] = {hh[1] * rh[2] - hh[2] * rh[1], hh[2] * rh[0] - hh[0] * rh[2], hh[0] * rh[1] - hh[1] * rh[0]};
        double nSrh[9] = {0, rh[2], -rh[1], -rh[2], 0, rh[0], rh[1], -rh[0], 0};      // -skew(r_hat)
        double Shh[9] = {0, -hh[2], hh[1], hh[2], 0, -hh[0], -hh[1], hh[0], 0};       // skew(h_hat)
        double Dr_t[9], Dv_t[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double a = 0.0, b = 0.0, c = 0.0;
                for (int l = 0; l < 3; ++l) { a += nSrh[i * 3 + l] * Dr_h[l * 3 + j]; b += Shh[i * 3 + l] * Dr_r[l * 3 + j]; c += nSrh[i * 3 + l] * Dv_h[l * 3 + j]; }
                Dr_t[i * 3 + j] = a + b; Dv_t[i * 3 + j] = c;
            }
        const double Vt = v[0] * th[0] + v[1] * th[1] + v[2] * th[2];
        const double Vc = sqrt(mu_grav / rn);
        const double kc = -0.5 * sqrt(mu_grav) * pow(rn, -2.5);
        double gT[6], gbar = 0.0, DrVc_r = 0.0;
        for (int j = 0; j < 3; ++j) {
            gT[j] = v[0] * Dr_t[j] + v[1] * Dr_t[3 + j] + v[2] * Dr_t[6 + j];
            gT[3 + j] = th[j] + (v[0] * Dv_t[j] + v[1] * Dv_t[3 + j] + v[2] * Dv_t[6 + j]);
        }
        for (int i = 0; i < 6; ++i) gbar += gT[i] * xK[i];
        for (int j = 0; j < 3; ++j) { DrVc_r += kc * r[j] * r[j]; gT[j] -= kc * r[j]; }
        const double c0t = Vt - gbar - Vc + DrVc_r;
        for (int j = 0; j < 6; ++j) { sd.aT[6][j] = gT[j]; sd.aT[7][j] = -gT[j]; }
        sd.bT[6] = relax(o.eps_vt - c0t); sd.bT[7] = relax(o.eps_vt + c0t);
    }
    sd.b_u = relax(o.u_max * o.u_max);
    sd.b_rmax = relax(o.r_max * o.r_max);
    sd.b_rmin = relax(-o.r_min);
    sd.b_rfmax = relax((r_des + o.eps_r) * (r_des + o.eps_r));
    sd.b_tf[0] = relax(0.0); sd.b_tf[1] = relax(o.tf_max);
    sd.vt_des = sqrt(mu_grav / r_des);
    sd.w_tr = o.w_tr; sd.w_nu = o.w_nu;
}

// > 0 when the constraint set is empty whatever the dynamics.  The virtual control makes every x_1..x_K reachable, so
// nothing else can make the reference's NLP infeasible: the fixed start node violates its own radius constraints (x_0 =
// xbar_0 is an equality, optimizer.py:344-345, and :384-395 apply at k = 0 too), the terminal radius window lies outside
// the r_max ball (:393-403), r_min > r_max, an empty velocity window (eps < 0), an empty tf range (:588).  ipopt ends
// such a problem in its restoration phase; here it is reported before the first iteration (MPCX_ST_INFEASIBLE) and the
// satellite leaves the launch at once.  Returns the largest violation of the relaxed bounds.
__device__ double structural_violation(const double *x0, int K, const SatData &sd)
{
    const double r2 = x0[0] * x0[0] + x0[1] * x0[1] + x0[2] * x0[2];
    double v = r2 - sd.b_rmax;
    if (K >= 3) v = fmax(v, -sqrt(r2) - sd.b_rmin);
    v = fmax(v, -sd.bT[0] - sqrt(fmin(sd.b_rmax, sd.b_rfmax)));
    if (K >= 3) v = fmax(v, -sd.b_rmin - sqrt(sd.b_rmax));      // node 1 is an inner node already at K = 3
    v = fmax(v, fmax(-(sd.bT[1] + sd.bT[2]), -(sd.bT[3] + sd.bT[4])));
    if (sd.nT == 8) v = fmax(v, -(sd.bT[6] + sd.bT[7]));
    if (!sd.fixed_tf) v = fmax(v, -(sd.b_tf[0] + sd.b_tf[1]));
    return v;
}

// ---- view of one satellite's problem + workspace -------------------------------------------
// Everything a satellite owns in HBM is addressed through pointers qualified with the global address space: the
// compiler then emits global_load/global_store (tracked by vmcnt only) instead of flat accesses, which also count
// against lgkmcnt and would make every LDS wait drain the node-ahead prefetch.
typedef __attribute__((address_space(1))) double gf64;
typedef const gf64 cgf64;
// wf64: the arrays of the satellite's workspace that the LDS-RESIDENT build keeps in LDS -- iterate, candidate, direction,
// Newton records, factor records, channel vectors (solve_lds.hip: MPCX_WS_LDS; batches of at most one satellite per compute
// unit, whose 160 KB of LDS then hold 134 KB of workspace at K = 30) -- and that live in the global workspace like the rest
// (stage copy, Newton scalars, channel trajectories: gf64) in the other two builds, where wf64 IS gf64.
#ifdef MPCX_WS_LDS
typedef __attribute__((address_space(3))) double wf64;
#else
typedef gf64 wf64;
#endif
typedef const wf64 cwf64;
typedef __attribute__((address_space(3))) double lf64;      // LDS in every build (SatData::gl: the global part of iterate / direction)
// the byte type of a pointer's address space (Col, ustore: base + byte offset)
template <typename T> struct as_bytes { typedef __attribute__((address_space(1))) char type; };
#ifdef MPCX_WS_LDS
template <> struct as_bytes<wf64> { typedef __attribute__((address_space(3))) char type; };
template <> struct as_bytes<cwf64> { typedef __attribute__((address_space(3))) char type; };
#endif

// Field-major view of one node's record: element i of node k lives at base[i * ld + k].  The node-parallel phases
// (one lane per node) read and write the same field of consecutive nodes in consecutive lanes, so every access is
// a couple of full cache lines instead of one line per lane.
// The base is made wave-uniform (SGPR pair) and the per-lane part is a 32-bit byte offset, so an access is one
// global_load/store with scalar base + vector offset and costs a single 32-bit VALU add for its address.
template <typename T>
__device__ __forceinline__ T *wave_uniform(T *p)
{
    if constexpr (sizeof(T *) == 4) {                     // (an LDS pointer: one 32-bit register)
        return (T *)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(size_t)p);
    } else {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (T *)(((unsigned long long)hi << 32) | lo);
    }
}

template <typename T>
struct Col {
    T *base;      // wave-uniform array base
    int k;        // element offset of this lane's node (and of the first field of the view)
    int ld;
    __device__ __forceinline__ T &operator[](int i) const
    {
        typedef typename as_bytes<T>::type bchar;
        return *(T *)((bchar *)base + (unsigned)((i * ld + k) * 8));
    }
    __device__ __forceinline__ Col operator+(int off) const { return Col{base, k + off * ld, ld}; }
    __device__ __forceinline__ Col node(int dk) const { return Col{base, k + dk, ld}; }   // same fields, node k + dk
};

// xld: a load of data that ANOTHER workgroup of the satellite has written (time-parallel build only: the Newton and
// right-hand-side records the first workgroup writes every iteration, the other segments' trajectories and exchange records).
// It goes past the compute unit's L1 to the L2 the satellite's workgroups share (a relaxed agent-scope atomic load:
// global_load ... sc1), so that no acquire has to invalidate caches -- the agent-scope invalidate (buffer_inv sc1) drops the
// XCD's whole L2, the read-only stage records of every satellite on it included.  Everywhere else it is a plain load.
// Measured (-DMPCX_TP_ACQ_MODE=4): correct, and NOT faster than the invalidate (88.4 against 88.4 k cycles for the
// factorisation command at 64 satellites, profiles/r04/time_parallel.txt) -- the default stays the invalidate, mode 2, which
// does not depend on every such load having been found.
#ifndef MPCX_TP_ACQ_MODE
#define MPCX_TP_ACQ_MODE 2
#endif
#if defined(MPCX_TP) && MPCX_TP_ACQ_MODE == 4
template <typename P> __device__ __forceinline__ double xld(P *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
template <typename P> __device__ __forceinline__ double xld(P *p) { return *p; }
#endif

// Store to element `e` of the satellite's workspace (wave-uniform base): scalar base + 32-bit vector offset.  The recursions issue
// their factor-record / trajectory stores for every lane (lanes with nothing to store aim at the satellite's sink): code
// without divergent store blocks is straight-line, so the compiler can count the stores issued after the node-ahead
// prefetch loads and waits for those loads with vmcnt(#stores) -- behind a branch it falls back to vmcnt(0), which puts
// the full HBM latency of the node's stores on the critical path of every node.
__device__ __forceinline__ void ustore(gf64 *ubase, int e, double v)
{
    typedef __attribute__((address_space(1))) char gchar;
    *(gf64 *)((gchar *)ubase + (unsigned)(e * 8)) = v;
}
#ifdef MPCX_WS_LDS
__device__ __forceinline__ void ustore(wf64 *ubase, int e, double v)
{
    typedef __attribute__((address_space(3))) char lchar;
    *(wf64 *)((lchar *)ubase + (unsigned)(e * 8)) = v;
}
#define WS_GLOBAL(s) ((s).wsg)      // base of the arrays that stay in the global workspace (channel trajectories) ...
#define SINK_GLOBAL(s) ((s).o_sinkg) // ... and the offset of the sink that belongs to it
#else
#define WS_GLOBAL(s) ((s).ws)
#define SINK_GLOBAL(s) ((s).o_sink)
#endif

struct Sat {
    int K, KP;
    int ldk;                      // row length of xbar / ubar (= K unless the batch is ragged)
    cgf64 *stage, *xbar, *ubar;   // stage (K-1,105) record per node; xbar (7,K); ubar (3,K)
    wf64 *it, *dr, *rbh;                        // field-major [field][KP]: iterate, direction, r-hat
    gf64 *nbs, *stT;                            // ... Newton scalars, stage copy (global in every build)
    lf64 *itg, *drg, *itgB;                     // global part of iterate, direction, candidate iterate: LDS (SatData::gl)
    wf64 *nb, *fac, *ch;                        // record-per-node arrays read by the recursion (one wave, one record)
    gf64 *traj;                                 // ... the channels' trajectories (global in every build)
    wf64 *itB;                                  // the candidate iterate of the line search (swapped with it -- and itgB with itg -- on acceptance)
    wf64 *sink;                                 // 64 doubles nobody reads: target of the lanes a branch-free store leaves idle
    wf64 *ws;                                   // base of the satellite's workspace and the element offsets of the arrays the
    int o_fac, o_ch, o_traj, o_sink;            // recursions store to (plain integers: see ustore)
#ifdef MPCX_WS_LDS
    gf64 *wsg;                                  // base of the global part (o_traj, o_sinkg count from here; o_fac, o_ch, o_sink from ws in LDS)
    int o_sinkg;
#endif
#ifdef MPCX_TP
    gf64 *chx, *trajx;                          // time-parallel build: the extra backward record (K x CHX_N), the second bank of
    int o_trajx, o_chx;                         // eight trajectory slots per node, the satellite's mailbox and exchange records
    int *mail; gf64 *xch;                       // (solve_tp.hpp)
#endif
    __device__ Col<wf64> itn(int k) const { return Col<wf64>{wave_uniform(it), k, KP}; }
    __device__ Col<wf64> itBn(int k) const { return Col<wf64>{wave_uniform(itB), k, KP}; }
    __device__ Col<wf64> drn(int k) const { return Col<wf64>{wave_uniform(dr), k, KP}; }
    __device__ Col<gf64> nsn(int k) const { return Col<gf64>{wave_uniform(nbs), k, KP}; }
    __device__ Col<wf64> rbn(int k) const { return Col<wf64>{wave_uniform(rbh), k, KP}; }
    // stage blocks for the node-parallel phases (field-major copy) ...
    __device__ Col<cgf64> At(int k) const { return Col<cgf64>{wave_uniform((cgf64 *)stT), k, KP}; }
    __device__ Col<cgf64> Bnt(int k) const { return At(k) + 49; }
    __device__ Col<cgf64> Bpt(int k) const { return At(k) + 70; }
    __device__ Col<cgf64> Sigt(int k) const { return At(k) + 91; }
    __device__ Col<cgf64> xit(int k) const { return At(k) + 98; }
    // ... and for the recursion (the discretizer's records)
    __device__ cgf64 *A(int k) const { return stage + (size_t)k * MPCX_STAGE_DOUBLES; }
    __device__ cgf64 *Sig(int k) const { return A(k) + 91; }
};

// Private copy of the view for the recursions with everything that is the same in all 64 lanes forced into scalar
// registers (node count, array bases): the compiler cannot see that these are wave-uniform -- they reach the function
// through a reference -- and otherwise keeps K, the loop counter derived from it and every 64-bit base in vector
// registers, which in riccati_factor meant spills reloaded inside the node loop behind an s_waitcnt vmcnt(0), i.e.
// behind every outstanding factor-record store of the node before.
__device__ __forceinline__ Sat uniform_view(const Sat &v)
{
    Sat s = v;
    s.K = __builtin_amdgcn_readfirstlane(v.K); s.KP = __builtin_amdgcn_readfirstlane(v.KP); s.ldk = __builtin_amdgcn_readfirstlane(v.ldk);
    s.stage = wave_uniform(v.stage); s.xbar = wave_uniform(v.xbar); s.ubar = wave_uniform(v.ubar);
    s.it = wave_uniform(v.it); s.dr = wave_uniform(v.dr); s.nbs = wave_uniform(v.nbs); s.stT = wave_uniform(v.stT); s.rbh = wave_uniform(v.rbh);
    s.itg = wave_uniform(v.itg); s.drg = wave_uniform(v.drg); s.nb = wave_uniform(v.nb); s.fac = wave_uniform(v.fac);
    s.ch = wave_uniform(v.ch); s.traj = wave_uniform(v.traj); s.itB = wave_uniform(v.itB); s.itgB = wave_uniform(v.itgB);
    s.sink = wave_uniform(v.sink); s.ws = wave_uniform(v.ws);
#ifdef MPCX_WS_LDS
    s.wsg = wave_uniform(v.wsg); s.o_sinkg = __builtin_amdgcn_readfirstlane(v.o_sinkg);
#endif
#ifdef MPCX_TP
    s.chx = wave_uniform(v.chx); s.trajx = wave_uniform(v.trajx); s.o_trajx = __builtin_amdgcn_readfirstlane(v.o_trajx); s.o_chx = __builtin_amdgcn_readfirstlane(v.o_chx);
    s.mail = wave_uniform(v.mail); s.xch = wave_uniform(v.xch);
#endif
    s.o_fac = __builtin_amdgcn_readfirstlane(v.o_fac); s.o_ch = __builtin_amdgcn_readfirstlane(v.o_ch);
    s.o_traj = __builtin_amdgcn_readfirstlane(v.o_traj); s.o_sink = __builtin_amdgcn_readfirstlane(v.o_sink);
    return s;
}

// Directions of the eliminated slack / multiplier pairs by back-substitution from the direction of the primal variables
// ("Linear solve").  They are not stored: finish_direction needs them once for the fraction-to-the-boundary
// step, every trial evaluation of the line search recomputes them from the iterate it reads anyway -- 41 field-major
// arrays less to write and to read back per iteration, against a handful of reciprocals per node.
// One inequality g + s = 0 with slack s, multiplier z: ds = -(g + s) - dg, dz = mu / s + (z / s) (g + s + dg) - z.
struct PairDir { double ds, dz; };
__device__ __forceinline__ PairDir pair_dir(double sv, double zv, double g, double dg, double mu)
{
    const double is = rcp_pos(sv), sig = zv * is, zh = mu * is + sig * (g + sv);
    return PairDir{-(g + sv) - dg, zh + sig * dg - zv};
}
// One component of the L1 pair nu - t <= 0, -nu - t <= 0 (t eliminated: optimizer.py:579-585): dt and the two pairs
struct L1Dir { double dt, dstp, dztp, dstn, dztn; };
__device__ __forceinline__ L1Dir l1_dir(double nu, double tt, double stp, double ztp, double stn, double ztn, double dnu, double mu, double w_nu)
{
    const double g1 = nu - tt, g2 = -nu - tt;
    const double ip = rcp_pos(stp), in = rcp_pos(stn);
    const double s1 = ztp * ip, s2 = ztn * in;
    const double zh1 = mu * ip + s1 * (g1 + stp), zh2 = mu * in + s2 * (g2 + stn);
    const double aa = s1 + s2, bb = s2 - s1, gt = w_nu - zh1 - zh2;
    L1Dir o;
    o.dt = (-gt - bb * dnu) * rcp_pos(aa);
    const double dg1 = dnu - o.dt, dg2 = -dnu - o.dt;
    o.dstp = -(g1 + stp) - dg1; o.dztp = zh1 + s1 * dg1 - ztp;
    o.dstn = -(g2 + stn) - dg2; o.dztn = zh2 + s2 * dg2 - ztn;
    return o;
}

// iterate + a * direction for one field, branch-free: both loads always issue (so they can all be in flight
// together); at a == 0 the direction value, which may be stale, is replaced by 0.
__device__ __forceinline__ double trial_value(const Col<wf64> &p, const Col<wf64> &d, int off, double a, bool z)
{
    const double dv = d[off], pv = p[off];
    return fma(a, z ? 0.0 : dv, pv);
}


// The node-parallel phases are written as chunks "loads -> arithmetic (-> stores)" separated by scheduling
// barriers: a chunk's loads are all in flight together (one memory latency per chunk instead of one per access,
// which is what interleaved may-alias stores would force), and the barrier keeps the scheduler from hoisting the
// loads of later chunks on top, which would spill.
#define CHUNK_END __builtin_amdgcn_sched_barrier(0);

// Two lanes per node in the node-parallel phases: lane (half, kl) = (lane >> 5, lane & 31) works on node kl (+32, ...);
// the part of a node's work that is a loop over the 7 state components is split between the two halves (components
// 4*half + r, r = 0..3, the eighth being a masked dummy), both halves running the same instructions; what cannot be
// split is computed by both and accounted once (half 0).  Partial sums meet through a lane ^ 32 shuffle.
#define HALF_OF(lane) ((lane) >> 5)
#define NODE_OF(lane) ((lane) & 31)

// Perturbed KKT residual F_mu at (iterate + a*direction): ipopt's scaled error pieces and the
// 2-norm used by the line search.  Results are wave-uniform.
// WRITE: the trial point is a candidate iterate -- slack reset s >= -g and multiplier safeguard z <= kappa mu_clip / s
// are applied to it first, the residual is that of the corrected point, and the point is stored in the second
// iterate buffer (s.itB, s.itgB): accepting the trial is a swap of the two buffers, and its residual is the next
// iteration's.
template <bool WRITE>
__device__ __noinline__ void eval_residual(const Sat &s, SatData &sd, double a, double mu, double mu_clip, int lane)
{
    ResAcc &out = sd.racc[WRITE ? 1 : 0];          // (the start point's evaluation: [0]; a trial's: [1])
#define POST(sv, zv, gval) { sv = fmax(sv, -(gval)); zv = fmin(zv, kKappaSigma * (mu_clip * rcp_pos(sv))); }
    const int K = s.K;
    double dual = 0.0, prim = 0.0, sq = 0.0, zsum = 0.0, lsum = 0.0, pmin = 1e300, pmax = -1e300, psum = 0.0;
    double gtf_part = 0.0;
    const double tf = s.itg[G_TF] + a * s.drg[G_TF];
    const double lvt = s.itg[G_LVT] + a * s.drg[G_LVT];
    const double w_tr = sd.w_tr, w_nu = sd.w_nu, b_u = sd.b_u, b_rmax = sd.b_rmax, b_rmin = sd.b_rmin;
    const bool z = (a == 0.0);
    const int half = HALF_OF(lane);
    const bool h0 = (half == 0);
#define ACC_D(v) { const double q_ = (v); dual = fmax(dual, fabs(q_)); sq += q_ * q_; }
#define ACC_P(v) { const double q_ = (v); prim = fmax(prim, fabs(q_)); sq += q_ * q_; }
#define ACC_C(sv, zv) { const double s_ = (sv), z_ = (zv), q_ = s_ * z_ - mu; sq += q_ * q_; \
                        zsum += fabs(z_); pmin = fmin(pmin, s_ * z_); pmax = fmax(pmax, s_ * z_); psum += s_ * z_; }
#define TRIAL(P, D, off) trial_value(P, D, off, a, z)
    for (int k = NODE_OF(lane); k < K; k += 32) {
        const auto p = s.itn(k), d = s.drn(k), w = s.itBn(k);
        const auto nsv = s.nsn(k);
        const bool has_prev = (k >= 1), dyn = (k <= K - 2);
        // ---- chunk 0 (both halves, accounted by half 0): states, thrust, ball slacks, objective gradient ----
        double x[7], u[3], gx[7], gu[3], un[3];
        double su, zu, srmax, zrmax, srmin, zrmin;
        const auto pn = p.node(dyn ? 1 : 0), dn = d.node(dyn ? 1 : 0);
        const auto rb = s.rbn(k);
        const double rb0 = rb[0], rb1 = rb[1], rb2 = rb[2];
        double xb[7], ub[3];       // reference row of the node: loaded with the first batch (behind it they were a memory round trip of their own)
        {
            double x0[7], dx[7], u0[3], du[3], bs[6];
#pragma unroll
            for (int i = 0; i < 7; ++i) { x0[i] = p[I_X + i]; dx[i] = d[I_X + i]; xb[i] = s.xbar[(size_t)i * s.ldk + k]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { u0[i] = p[I_U + i]; du[i] = d[I_U + i]; ub[i] = s.ubar[(size_t)i * s.ldk + k]; }
#pragma unroll
            for (int i = 0; i < 6; ++i) bs[i] = p[I_SU + i];
#pragma unroll
            for (int i = 0; i < 3; ++i) un[i] = TRIAL(pn, dn, I_U + i);
#pragma unroll
            for (int i = 0; i < 7; ++i) { dx[i] = z ? 0.0 : dx[i]; x[i] = fma(a, dx[i], x0[i]); }
#pragma unroll
            for (int i = 0; i < 3; ++i) { du[i] = z ? 0.0 : du[i]; u[i] = fma(a, du[i], u0[i]); }
            // the ball pairs' directions (pairs a node does not own keep their placeholder values)
            PairDir du_ = pair_dir(bs[0], bs[1], u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2] - b_u,
                                   2.0 * (u0[0] * du[0] + u0[1] * du[1] + u0[2] * du[2]), mu);
            PairDir dmax = pair_dir(bs[2], bs[3], x0[0] * x0[0] + x0[1] * x0[1] + x0[2] * x0[2] - b_rmax,
                                    2.0 * (x0[0] * dx[0] + x0[1] * dx[1] + x0[2] * dx[2]), mu);
            PairDir dmin = pair_dir(bs[4], bs[5], -(rb0 * x0[0] + rb1 * x0[1] + rb2 * x0[2]) - b_rmin,
                                    -(rb0 * dx[0] + rb1 * dx[1] + rb2 * dx[2]), mu);
            const bool on_max = !z && has_prev, on_min = !z && has_prev && dyn;
            su = fma(a, z ? 0.0 : du_.ds, bs[0]); zu = fma(a, z ? 0.0 : du_.dz, bs[1]);
            srmax = fma(a, on_max ? dmax.ds : 0.0, bs[2]); zrmax = fma(a, on_max ? dmax.dz : 0.0, bs[3]);
            srmin = fma(a, on_min ? dmin.ds : 0.0, bs[4]); zrmin = fma(a, on_min ? dmin.dz : 0.0, bs[5]);
        }
        const double g_u = u[0] * u[0] + u[1] * u[1] + u[2] * u[2] - b_u;
        const double g_rmax = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] - b_rmax;
        const double g_rmin = -(rb0 * x[0] + rb1 * x[1] + rb2 * x[2]) - b_rmin;
        if (WRITE) {
            POST(su, zu, g_u);
            if (has_prev) POST(srmax, zrmax, g_rmax);
            if (has_prev && dyn) POST(srmin, zrmin, g_rmin);
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) gx[i] = 2.0 * w_tr * (x[i] - xb[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) gu[i] = 2.0 * w_tr * (u[i] - ub[i]) + 2.0 * u[i] * zu;
        if (has_prev) {
#pragma unroll
            for (int i = 0; i < 3; ++i) gx[i] += 2.0 * x[i] * zrmax;
            if (dyn) { gx[0] -= rb0 * zrmin; gx[1] -= rb1 * zrmin; gx[2] -= rb2 * zrmin; }
        }
        if (h0) {
            ACC_P(g_u + su);                                                 // thrust ball, every node
            ACC_C(su, zu);
            if (has_prev) {
                ACC_P(g_rmax + srmax);
                ACC_C(srmax, zrmax);
                if (dyn) {
                    ACC_P(g_rmin + srmin);
                    ACC_C(srmin, zrmin);
                }
            }
            if (WRITE) {
#pragma unroll
                for (int i = 0; i < 7; ++i) w[I_X + i] = x[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) w[I_U + i] = u[i];
                w[I_SU] = su; w[I_ZU] = zu; w[I_SRMAX] = srmax; w[I_ZRMAX] = zrmax; w[I_SRMIN] = srmin; w[I_ZRMIN] = zrmin;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 7; ++i) gx[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) gu[i] = 0.0;
        }
        CHUNK_END
        // ---- four rounds: component i = 4*half + r of the dynamics row (optimizer.py:327-342), of its multiplier, of
        //      the L1 pair of nu_i, and of the previous row's multiplier (+lam_{k-1} on x_k, -Bp_{k-1}^T lam_{k-1} on u_k)
        {
            const auto A = s.At(dyn ? k : 0), Bn = s.Bnt(dyn ? k : 0), Bp = s.Bpt(dyn ? k : 0), Sg = s.Sigt(dyn ? k : 0), xi = s.xit(dyn ? k : 0);
            const auto Bm = s.Bpt(has_prev ? k - 1 : 0);
            const auto pm = p.node(has_prev ? -1 : 0), dm = d.node(has_prev ? -1 : 0);
            double sl = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int iv = 4 * half + r;
                const bool valid = iv < 7;
                const int i = valid ? iv : 6;
                double ar[7], bn[3], bp[3], bm[3];
#pragma unroll
                for (int j = 0; j < 7; ++j) ar[j] = A[i * 7 + j];
#pragma unroll
                for (int j = 0; j < 3; ++j) { bn[j] = Bn[i * 3 + j]; bp[j] = Bp[i * 3 + j]; bm[j] = Bm[i * 3 + j]; }
                const double sg = Sg[i], xv = xi[i];
                const double nu0 = p[I_NU + i], dnu_ = d[I_NU + i], tt0 = p[I_T + i], lam = TRIAL(p, d, I_LAM + i);
                const double stp0 = p[I_STP + i], ztp0 = p[I_ZTP + i], stn0 = p[I_STN + i], ztn0 = p[I_ZTN + i];
                const double xn = TRIAL(pn, dn, I_X + i);           // (with the round's other loads: one round trip per round, not two)
                const double lmv = TRIAL(pm, dm, I_LAM + i);
                const double dnu = z ? 0.0 : dnu_;
                const L1Dir ld = l1_dir(nu0, tt0, stp0, ztp0, stn0, ztn0, dnu, mu, w_nu);
                const bool lon = !z && dyn;                                   // (the terminal node has no virtual control)
                const double nu = fma(a, dnu, nu0), tt = fma(a, lon ? ld.dt : 0.0, tt0);
                double stp = fma(a, lon ? ld.dstp : 0.0, stp0), ztp = fma(a, lon ? ld.dztp : 0.0, ztp0);
                double stn = fma(a, lon ? ld.dstn : 0.0, stn0), ztn = fma(a, lon ? ld.dztn : 0.0, ztn0);
                if (WRITE && dyn) { POST(stp, ztp, nu - tt); POST(stn, ztn, -nu - tt); }
                const double lm = (valid && has_prev) ? lmv : 0.0;
                // previous row's multiplier
                gx[r] += half ? 0.0 : lm; gx[(4 + r) % 7] += (half && valid) ? lm : 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) gu[j] -= bm[j] * lm;
                if (valid && dyn) {
                    double acc = sg * tf + xv + nu;
#pragma unroll
                    for (int j = 0; j < 7; ++j) acc += ar[j] * x[j];
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc += bn[j] * u[j] + bp[j] * un[j];
                    ACC_P(xn - acc);
                    nsv[NS_E + i] = xn - acc;       // e_k of this point: newton_blocks takes it from here (see there)
                    sl += sg * lam;
                    lsum += fabs(lam);
                    // nu / t stationarity
                    ACC_D(ztp - ztn - lam);
                    ACC_D(w_nu - ztp - ztn);
                    ACC_P(nu - tt + stp);
                    ACC_P(-nu - tt + stn);
                    ACC_C(stp, ztp);
                    ACC_C(stn, ztn);
#pragma unroll
                    for (int j = 0; j < 7; ++j) gx[j] -= ar[j] * lam;
#pragma unroll
                    for (int j = 0; j < 3; ++j) gu[j] -= bn[j] * lam;
                }
                if (WRITE && valid) {
                    w[I_NU + i] = nu; w[I_T + i] = tt; w[I_LAM + i] = lam;
                    w[I_STP + i] = stp; w[I_ZTP + i] = ztp; w[I_STN + i] = stn; w[I_ZTN + i] = ztn;
                }
                CHUNK_END
            }
            gtf_part -= sl;
        }
        // the two halves' parts of the stationarity rows meet in half 0
#pragma unroll
        for (int i = 0; i < 7; ++i) gx[i] += __shfl_xor(gx[i], 32, 64);
#pragma unroll
        for (int i = 0; i < 3; ++i) gu[i] += __shfl_xor(gu[i], 32, 64);
        if (h0) {
            if (k == K - 1) {
                // terminal inequalities, final-radius ball, vt equality
                if (!sd.linvt) {
                    double cv, g6[6];
                    vt_reduced(x, sd.vt_des, cv, g6, nullptr);
                    ACC_P(cv);
#pragma unroll
                    for (int i = 0; i < 6; ++i) gx[i] += lvt * g6[i];
                    lsum += fabs(lvt);
                }
                const int nT = sd.nT;
                for (int j = 0; j < nT; ++j) {
                    const int js = gs_term(j), jz = gz_term(j);
                    double sj = s.itg[js] + a * s.drg[js];
                    double zj = s.itg[jz] + a * s.drg[jz];
                    double gj = -sd.bT[j];
                    for (int i = 0; i < 7; ++i) gj += sd.aT[j][i] * x[i];
                    if (WRITE) { POST(sj, zj, gj); s.itgB[js] = sj; s.itgB[jz] = zj; }
                    for (int i = 0; i < 7; ++i) gx[i] += sd.aT[j][i] * zj;
                    ACC_P(gj + sj);
                    ACC_C(sj, zj);
                }
                double srf = s.itg[G_SRF] + a * s.drg[G_SRF], zrf = s.itg[G_ZRF] + a * s.drg[G_ZRF];
                const double g_rf = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] - sd.b_rfmax;
                if (WRITE) { POST(srf, zrf, g_rf); s.itgB[G_SRF] = srf; s.itgB[G_ZRF] = zrf; s.itgB[G_LVT] = lvt; }
                ACC_P(g_rf + srf);
                ACC_C(srf, zrf);
                for (int i = 0; i < 3; ++i) gx[i] += 2.0 * x[i] * zrf;
            }
            if (has_prev) {
#pragma unroll
                for (int i = 0; i < 7; ++i) ACC_D(gx[i]);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) ACC_D(gu[i]);
        }
    }
    // tf stationarity and range constraints (lane 0 adds them after the reduction of gtf_part)
    double gtf = wave_sum(gtf_part);
    out.g_tf = gtf + 2.0 * sd.w_tr * (tf - sd.tfbar);
    if (lane == 0 && sd.fixed_tf) {
        if (WRITE) s.itgB[G_TF] = tf;
    } else if (lane == 0) {
        double s0 = s.itg[G_STF] + a * s.drg[G_STF], s1 = s.itg[G_STF + 1] + a * s.drg[G_STF + 1];
        double z0 = s.itg[G_ZTF] + a * s.drg[G_ZTF], z1 = s.itg[G_ZTF + 1] + a * s.drg[G_ZTF + 1];
        if (WRITE) {
            POST(s0, z0, -tf - sd.b_tf[0]); POST(s1, z1, tf - sd.b_tf[1]);
            s.itgB[G_TF] = tf; s.itgB[G_STF] = s0; s.itgB[G_STF + 1] = s1; s.itgB[G_ZTF] = z0; s.itgB[G_ZTF + 1] = z1;
        }
        gtf += 1.0 + 2.0 * sd.w_tr * (tf - sd.tfbar) - z0 + z1;
        ACC_D(gtf);
        ACC_P(-tf - sd.b_tf[0] + s0);
        ACC_P(tf - sd.b_tf[1] + s1);
        ACC_C(s0, z0);
        ACC_C(s1, z1);
    }
#undef ACC_D
#undef ACC_P
#undef ACC_C
#undef TRIAL
#undef POST
    out.dual_max = wave_max(dual); out.prim_max = wave_max(prim);
    out.sq = wave_sum(sq); out.zsum = wave_sum(zsum); out.lsum = wave_sum(lsum);
    out.prod_min = wave_min(pmin); out.prod_max = wave_max(pmax); out.prod_sum = wave_sum(psum);
    WG_SYNC();                       // the candidate iterate (WRITE) and the record are complete before anybody reads them
}

__device__ __forceinline__ int n_ineq(int K, int nT, int fixed_tf) { return K + (K - 1) + (K - 2) + nT + 1 + 14 * (K - 1) + (fixed_tf ? 0 : 2); }

// ipopt's scaled optimality error E_mu from one residual evaluation: max_i |s_i z_i - mu| = max(pmax - mu, mu - pmin)
__device__ double scaled_error_n(const ResAcc &r, int nz, int nl, double mu)
{
    const double smax = 100.0;
    const double sdl = fmax(smax, (r.zsum + r.lsum) / (double)(nz + nl)) / smax;
    const double sc = fmax(smax, r.zsum / (double)nz) / smax;
    const double comp = fmax(r.prod_max - mu, mu - r.prod_min);
    return fmax(fmax(r.dual_max / sdl, r.prim_max), comp / sc);
}
__device__ double scaled_error(const ResAcc &r, int K, int nT, int fixed_tf, double mu)
{
    const double smax = 100.0;
    const int nz = n_ineq(K, nT, fixed_tf), nl = 7 * (K - 1) + (nT == 6 ? 1 : 0);     // (the convex variant has no tangential equality)
    const double sdl = fmax(smax, (r.zsum + r.lsum) / (double)(nz + nl)) / smax;
    const double sc = fmax(smax, r.zsum / (double)nz) / smax;
    const double comp = fmax(r.prod_max - mu, mu - r.prod_min);
    return fmax(fmax(r.dual_max / sdl, r.prim_max), comp / sc);
}

// ---- Newton blocks (stage-parallel) ------------------------------------------------------------
// stg: LDS staging area of 32 * (NB_N + RHS_N) doubles (the recursion's scratch, idle during this phase).  A node's Newton and
// right-hand-side records are assembled there and the 32 records of a round go out as contiguous, coalesced blocks: written straight from the node
// lanes they were 8-byte stores scattered over 32 cache lines per instruction (measured: the 40 stores per node that the
// compact Hessian form removed were 6 % of the launch at S = 4096).
// KEEP_NS: also keep the node's gradient / rho / D scalars in the field-major Newton scalars -- only reduced_residual (the
// refinement passes of a stiff iteration) reads them: the driver runs newton_blocks<true> once more when it finds that the
// iteration refines (rare), the plain iteration does not write them.
template <bool KEEP_NS>
__device__ __noinline__ void newton_blocks(const Sat &s, SatData &sd, double *stg, double mu, double delta_w, int lane)
{
    const int K = s.K;
    const double w_tr = sd.w_tr, w_nu = sd.w_nu, b_u = sd.b_u, b_rmax = sd.b_rmax, b_rmin = sd.b_rmin;
    const int half = HALF_OF(lane);
    const bool h0 = (half == 0);
    double sigmax = 0.0;                                   // largest barrier weight z/s of the stage constraints
    for (int k0 = 0; k0 < K; k0 += 32) {
      const int k = k0 + NODE_OF(lane);
      if (k < K) {
        const auto p = s.itn(k);
        const auto ns = s.nsn(k);
        const auto rb = s.rbn(k);
        double *nb = stg + NODE_OF(lane) * NB_N;
        double *rhs = stg + 32 * NB_N + NODE_OF(lane) * RHS_LD;      // (odd stride: the 32 node lanes hit different banks)
        const bool dyn = (k <= K - 2), inner = (k >= 1 && k <= K - 2);
        // ---- chunk 0: objective, thrust ball, radius balls ----
        // (chunk 0 is computed by both halves and stored by half 0)
        double x[7], u[3], gx[7], gu[3], Wx3[9];
        double zh_rmax = 0.0, sig_rmax = 0.0, zrmax;
        double rv[4][7];           // the four rounds' inputs, loaded with chunk 0's: one memory round trip per node instead of five
        {
            double bs[6], xb[7], ub[3];
#pragma unroll
            for (int i = 0; i < 7; ++i) { x[i] = p[I_X + i]; xb[i] = s.xbar[(size_t)i * s.ldk + k]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { u[i] = p[I_U + i]; ub[i] = s.ubar[(size_t)i * s.ldk + k]; }
#pragma unroll
            for (int i = 0; i < 6; ++i) bs[i] = p[I_SU + i];
            const double rb0 = rb[0], rb1 = rb[1], rb2 = rb[2];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int iv = 4 * half + r;
                const int i = (iv < 7) ? iv : 6;
                rv[r][0] = p[I_NU + i]; rv[r][1] = p[I_T + i]; rv[r][2] = p[I_STP + i]; rv[r][3] = p[I_ZTP + i];
                rv[r][4] = p[I_STN + i]; rv[r][5] = p[I_ZTN + i]; rv[r][6] = ns[NS_E + i];
            }
            CHUNK_END
            const double rbv[3] = {rb0, rb1, rb2};
            const double su = bs[0], zu = bs[1], srmax = bs[2], srmin = bs[4], zrmin = bs[5];
            zrmax = bs[3];
#pragma unroll
            for (int i = 0; i < 7; ++i) gx[i] = 2.0 * w_tr * (x[i] - xb[i]);
            // thrust ball.  Only min(sigma, kStageCap) of a stage barrier weight goes into the Hessian blocks: summed into
            // a 3x3 / 7x7 block a weight of 1e14 (an active constraint at mu = 1e-9) would wipe out the trust-region
            // curvature 2 w_tr of the other directions; the excess reaches the recursion as a rank-1 update (riccati_factor)
            double Wu[9];
            {
                const double g = u[0] * u[0] + u[1] * u[1] + u[2] * u[2] - b_u;
                const double isu = rcp_pos(su), sig = zu * isu, zh = mu * isu + sig * (g + su);
                sigmax = fmax(sigmax, sig);
                const double sin_ = fmin(sig, kStageCap);
                if (h0) { nb[N_SX + SX_EU] = sig - sin_; nb[N_SX + SX_CU] = 2.0 * u[0]; nb[N_SX + SX_CU + 1] = 2.0 * u[1]; nb[N_SX + SX_CU + 2] = 2.0 * u[2]; }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    gu[i] = 2.0 * w_tr * (u[i] - ub[i]) + 2.0 * u[i] * zh;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        Wu[i * 3 + j] = (i == j ? 2.0 * w_tr + delta_w + 2.0 * zu : 0.0) + sin_ * 4.0 * u[i] * u[j];
                }
            }
            if (k >= 1) {
                const double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
                const double g = r2 - b_rmax;
                const double isr = rcp_pos(srmax);
                sig_rmax = zrmax * isr; zh_rmax = mu * isr + sig_rmax * (g + srmax);
                sigmax = fmax(sigmax, sig_rmax);
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) Wx3[i] = ((i & 3) == 0) ? 2.0 * w_tr + delta_w : 0.0;
            if (inner) {
                const double g = -(rb0 * x[0] + rb1 * x[1] + rb2 * x[2]) - b_rmin;
                const double isr = rcp_pos(srmin), sig = zrmin * isr, zh = mu * isr + sig * (g + srmin);
                sigmax = fmax(sigmax, sig);
                // at most one of the two position terms can be stiff (r_min < r_max): the one with the larger excess
                // leaves the block, the other stays whole
                const double ex_max = sig_rmax - kStageCap, ex_min = sig - kStageCap;
                const bool st_max = ex_max > 0.0 && ex_max >= ex_min, st_min = ex_min > 0.0 && !st_max;
                const double in_max = st_max ? kStageCap : sig_rmax, in_min = st_min ? kStageCap : sig;
                if (h0) {
                    nb[N_SX + SX_EX] = st_max ? ex_max : (st_min ? ex_min : 0.0);
#pragma unroll
                    for (int i = 0; i < 3; ++i) nb[N_SX + SX_A + i] = st_max ? 2.0 * x[i] : rbv[i];
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    gx[i] += 2.0 * x[i] * zh_rmax - rbv[i] * zh;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        Wx3[i * 3 + j] += (i == j ? 2.0 * zrmax : 0.0) + in_max * 4.0 * x[i] * x[j] + in_min * rbv[i] * rbv[j];
                }
            }
            if (h0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) nb[N_WU + i] = Wu[i];
                if (!inner) { nb[N_SX + SX_EX] = 0.0; nb[N_SX + SX_A] = 0.0; nb[N_SX + SX_A + 1] = 0.0; nb[N_SX + SX_A + 2] = 0.0; }
                if (KEEP_NS) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) ns[NS_GX + i] = gx[i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) ns[NS_GU + i] = gu[i];
                }
                // right-hand-side record of the iteration's first solve (= the Newton blocks; the zero direction carries
                // no multipliers): x_0 is fixed (no row), the terminal node's gradient is written with its Hessians below
#pragma unroll
                for (int i = 0; i < 3; ++i) rhs[R_GU + i] = gu[i];
                if (k != K - 1) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) rhs[R_GX + i] = (k == 0) ? 0.0 : gx[i];
                }
            }
            if (h0 && k != K - 1) {
                // stage Hessian of x: diagonal + the 3x3 position block (the terminal node's matrix goes to SatData below)
#pragma unroll
                for (int i = 0; i < 9; ++i) nb[N_W3 + i] = Wx3[i];
                nb[N_DIAG] = 2.0 * w_tr + delta_w; nb[N_ZERO] = 0.0;
            }
        }
        CHUNK_END
        // ---- four rounds: component i = 4*half + r of the virtual-control block (t eliminated, D and rho kept
        //      without multipliers).  The dynamics residual e_k of the iterate is already in the Newton scalars: the
        //      residual evaluation that produced this iterate (the accepted trial of the line search, or the start
        //      point's) stored it -- no second pass over the stage matrices here ----
        {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int iv = 4 * half + r;
                const bool valid = (iv < 7) && dyn;
                const int i = (iv < 7) ? iv : 6;
                const double nu = rv[r][0], tt = rv[r][1], stp = rv[r][2], ztp = rv[r][3];
                const double stn = rv[r][4], ztn = rv[r][5];
                const double g1 = nu - tt, g2 = -nu - tt;
                const double ip = rcp_pos(stp), in = rcp_pos(stn);      // slacks are positive: reciprocal + products
                const double s1 = ztp * ip, s2 = ztn * in;
                const double zh1 = mu * ip + s1 * (g1 + stp), zh2 = mu * in + s2 * (g2 + stn);
                const double aa = s1 + s2, bb = s2 - s1, gt = w_nu - zh1 - zh2;
                const double ia = rcp_pos(aa);
                const double dd = 4.0 * s1 * s2 * ia;
                const double ek = rv[r][6];
                const double rho = (zh1 - zh2) - (bb * ia) * gt;
                if (valid) {
                    nb[N_D + i] = dd;
                    if (KEEP_NS) { ns[NS_D + i] = dd; ns[NS_RHO + i] = rho; }
                }
                if (iv < 7) { rhs[R_RHO + i] = dyn ? rho : 0.0; rhs[R_AFF + i] = dyn ? -ek : 0.0; }
            }
            CHUNK_END
        }
        if (h0 && k == K - 1) {
            // terminal node: soft gradient, the five rank-1 barrier terms, the pieces of the terminal Hessians (the
            // 7x7 matrices themselves are assembled by 49 lanes after the loop)
            double g6[6];
            const double lvt = sd.linvt ? 0.0 : s.itg[G_LVT];
            double sig[8], zh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {               // (static indices: the arrays stay in registers)
                sig[j] = 0.0; zh[j] = 0.0;
                if (j >= sd.nT) continue;
                double gj = -sd.bT[j];
                for (int i = 0; i < 7; ++i) gj += sd.aT[j][i] * x[i];
                const double sj = s.itg[gs_term(j)], zj = s.itg[gz_term(j)];
                sig[j] = zj / sj; zh[j] = mu / sj + sig[j] * (gj + sj);
            }
            if (!sd.linvt) {
                double cv;
                vt_reduced(x, sd.vt_des, cv, g6, sd.Hv);
                sd.cv = cv;
            } else {
                // convex variant: the tangential pair is a rank-1 terminal term whose border unknown rides in the channel
                // of the (absent) equality's multiplier: direction a_vt = row 6, no curvature, no constraint value
                for (int i = 0; i < 6; ++i) g6[i] = sd.aT[6][i];
                for (int i = 0; i < 36; ++i) sd.Hv[i] = 0.0;
                sd.cv = 0.0;
                sd.w_vt = sig[6] + sig[7]; sd.gh_vt = zh[6] - zh[7];
            }
            for (int i = 0; i < 7; ++i) sd.avt[i] = (i < 6) ? g6[i] : 0.0;
            const double srf = s.itg[G_SRF], zrf = s.itg[G_ZRF];
            const double grf = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] - sd.b_rfmax;
            const double sigrf = zrf / srf, zhrf = mu / srf + sigrf * (grf + srf);
            sd.red[0] = 2.0 * (zrmax + zrf);          // diagonal of the two radius balls at the terminal node
            sd.red[1] = lvt;
            for (int i = 0; i < 7; ++i) sd.gxKsoft[i] = gx[i];
            const int rows[NTERM] = {0, 1, 3, 5, -1};
            for (int t = 0; t < NTERM; ++t) {
                if (rows[t] >= 0) for (int i = 0; i < 7; ++i) sd.ta[t][i] = sd.aT[rows[t]][i];
                else for (int i = 0; i < 7; ++i) sd.ta[t][i] = (i < 3) ? 2.0 * x[i] : 0.0;
            }
            sd.tw[0] = sig[0]; sd.tgh[0] = zh[0];
            sd.tw[1] = sig[1] + sig[2]; sd.tgh[1] = zh[1] - zh[2];
            sd.tw[2] = sig[3] + sig[4]; sd.tgh[2] = zh[3] - zh[4];
            sd.tw[3] = sig[5]; sd.tgh[3] = zh[5];
            sd.tw[4] = sig_rmax + sigrf; sd.tgh[4] = zh_rmax + zhrf;
            // capped share of the rank-1 weights kept inside the recursion, AL weight of the vt row
            for (int t = 0; t < NTERM; ++t) sd.twin[t] = fmin(sd.tw[t], kTermCap);
            double hn = 0.0, an = 0.0;
            for (int i = 0; i < 36; ++i) hn += sd.Hv[i] * sd.Hv[i];
            for (int i = 0; i < 6; ++i) an += g6[i] * g6[i];
            // (convex variant: the capped share of the pair's weight takes the place of the AL weight)
            sd.gam = sd.linvt ? fmin(sd.w_vt, kTermCap) : (kTermCap + 10.0 * fabs(lvt) * sqrt(hn)) / an;
            // terminal node's gradient of the first solve: soft gradient + capped share of the rank-1 gradient terms and
            // the AL shift (rvt = the vt row's right-hand side at the zero direction, see first_rhs_scalars)
            {
                const double rvt = sd.linvt ? -sd.gh_vt / sd.w_vt : -sd.cv;
                double gK[7];
                for (int i = 0; i < 7; ++i) gK[i] = gx[i];
                for (int t = 0; t < NTERM; ++t) {
                    const double share = (sd.tw[t] > 0.0) ? sd.twin[t] / sd.tw[t] : 1.0;
                    for (int i = 0; i < 7; ++i) gK[i] += sd.tgh[t] * share * sd.ta[t][i];
                }
                for (int i = 0; i < 7; ++i) rhs[R_GX + i] = gK[i] - sd.gam * rvt * sd.avt[i];
            }
            // (the terminal node's Hessian lives in SatData: its compact slots are unused, kept defined)
#pragma unroll
            for (int i = 0; i <= N_ZERO; ++i) nb[i] = 0.0;
        }
      }
      WG_SYNC();
      {
          const int ne = ((K - k0 < 32) ? K - k0 : 32) * NB_N;
          wf64 *dst = s.nb + (size_t)k0 * NB_N;
          for (int e = lane; e < ne; e += 64) dst[e] = stg[e];
          // ... and the right-hand-side records (24 contiguous doubles inside each node's channel record)
          const int nr = ((K - k0 < 32) ? K - k0 : 32) * RHS_N;
          wf64 *ch = s.ch + (size_t)k0 * CH_N + C_RHS;
          for (int e = lane; e < nr; e += 64) { const int kl = e / RHS_N, i = e - kl * RHS_N; ch[(size_t)kl * CH_N + i] = stg[32 * NB_N + kl * RHS_LD + i]; }
      }
      WG_SYNC();
    }
    sigmax = wave_max(sigmax);
    WG_SYNC();
    // terminal Hessians, one lane per element: soft part (objective, radius balls, lam_vt * Hessian of the vt row) for
    // the residuals; + capped rank-1 terms + AL term for the recursion, which reads it from the terminal node's slot
    if (lane < 49) {
        const int i = lane / 7, j = lane - 7 * i;
        double soft = (i == j) ? 2.0 * w_tr + delta_w + (i < 3 ? sd.red[0] : 0.0) : 0.0;
        if (i < 6 && j < 6) soft += sd.red[1] * sd.Hv[i * 6 + j];
        double full = soft;
#pragma unroll
        for (int t = 0; t < NTERM; ++t) full += sd.twin[t] * sd.ta[t][i] * sd.ta[t][j];
        full += sd.gam * sd.avt[i] * sd.avt[j];
        sd.WxKsoft[lane] = soft; sd.WxK[lane] = full;
    }
    if (lane == 0 && sd.shared) {
        // this satellite's share of the tf row: the trust-region term w_tr (tf - tf_bar)^2 (optimizer.py:311,322); the 1 of
        // the objective, the range constraint's barrier terms and delta_w belong to the launch as a whole (solve_satellite)
        sd.Wtf = 2.0 * sd.w_tr; sd.gtf = 2.0 * sd.w_tr * (s.itg[G_TF] - sd.tfbar); sd.sigmax = sigmax;
    } else if (lane == 0 && sd.fixed_tf) { sd.Wtf = 1.0; sd.gtf = 0.0; sd.sigmax = sigmax; }
    else if (lane == 0) {
        const double tf = s.itg[G_TF];
        double W = 2.0 * sd.w_tr + delta_w, g = 1.0 + 2.0 * sd.w_tr * (tf - sd.tfbar);
        const double gv[2] = {-tf - sd.b_tf[0], tf - sd.b_tf[1]};
        for (int j = 0; j < 2; ++j) {
            const double sj = s.itg[G_STF + j], zj = s.itg[G_ZTF + j];
            const double sig = zj / sj, zh = mu / sj + sig * (gv[j] + sj);
            W += sig; g += (j == 0 ? -zh : zh);
            sigmax = fmax(sigmax, sig);
        }
        sd.Wtf = W; sd.gtf = g; sd.sigmax = sigmax;
    }
    WG_SYNC();
}

// direction (+)= trajectory of channel 0 + sum_j sol[j] * trajectory of channel 1+j ; one lane per (node, component),
// four components per lane and round so that their loads are in flight together.  `first` (the plain solve of an
// iteration): the direction is written, with -lam as the starting value of the multiplier part (the first
// right-hand side carries no multipliers); otherwise (refinement) the correction is added.
__device__ __noinline__ void combine_channels(const Sat &s_in, SatData &sd, double *stg, int lane, bool first)
{
    const Sat s = uniform_view(s_in);
    const int K = s.K, KP = s.KP;
    double sol[NBD];
#pragma unroll
    for (int j = 0; j < NBD; ++j) sol[j] = sd.sol[j];
    wf64 *dr = wave_uniform(s.dr);
    cwf64 *it = wave_uniform((cwf64 *)s.it);
    cgf64 *traj = wave_uniform((cgf64 *)s.traj);
    // Rounds of 32 nodes.  The trajectories are read in their own order (node, channel, component: contiguous), the
    // combination goes through LDS (stg: the recursion's scratch, [component][node of the round]) and leaves in the
    // direction's field-major order, consecutive lanes on consecutive nodes: written straight from the reading lanes
    // the direction was 8-byte stores scattered over as many cache lines as lanes.
    for (int k0 = 0; k0 < K; k0 += 32) {
        const int nk = (K - k0 < 32) ? K - k0 : 32;
        const int n = nk * TR_N;
        // (eight entries per lane and step: the 8 x 8 trajectory loads of a step are in flight together -- a round of 30 nodes is
        //  one step, i.e. one memory round trip where four entries per step made two)
        constexpr int CQ = 8;
        for (int e0 = 0; e0 < n; e0 += 64 * CQ) {
            double v[CQ];
            int slot[CQ];
#pragma unroll
            for (int q = 0; q < CQ; ++q) {
                const int e = e0 + 64 * q + lane;
                const int ec = (e < n) ? e : 0;
                const int kl = ec / TR_N, i = ec - kl * TR_N;
                cgf64 *tr = traj + (size_t)(k0 + kl) * NCH * TR_N + i;
                double acc = tr[0];
#pragma unroll
                for (int j = 0; j < NBD; ++j)
                    acc += sol[j] * tr[(1 + j) * TR_N];
                v[q] = acc; slot[q] = (e < n) ? i * CMB_LD + kl : -1;
            }
#pragma unroll
            for (int q = 0; q < CQ; ++q) if (slot[q] >= 0) stg[slot[q]] = v[q];
        }
        WG_SYNC();
        // (the 24 x 32 entries of the round in 12 steps of 64 lanes: the three loads of ALL steps are issued first -- as a loop of
        //  load, load, load, wait, store this was twelve dependent memory round trips per round of nodes)
        constexpr int NQ = DIR_N * 32 / 64;
        double curv[NQ], Dv[NQ], rv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = lane + 64 * q;
            const int i = e >> 5, kl = e & 31, k = k0 + kl;
            const int off = (i < T_U) ? I_X + i : (i < T_NU ? I_U + (i - T_U) : (i < T_LAM ? I_NU + (i - T_NU) : I_LAM + (i - T_LAM)));
            const int dst = off * KP + (kl < nk ? k : k0);
            const int kc = (kl < nk) ? k : k0, j = (i >= T_LAM) ? i - T_LAM : 0;
            curv[q] = first ? it[dst] : dr[dst];
            Dv[q] = s.nb[(size_t)kc * NB_N + N_D + j]; rv[q] = s.ch[(size_t)kc * CH_N + C_RHS + R_RHO + j];
        }
        CHUNK_END
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = lane + 64 * q;
            const int i = e >> 5, kl = e & 31, k = k0 + kl;
            const bool act = kl < nk && !(k == K - 1 && i >= T_NU);
            const int off = (i < T_U) ? I_X + i : (i < T_NU ? I_U + (i - T_U) : (i < T_LAM ? I_NU + (i - T_NU) : I_LAM + (i - T_LAM)));
            const int dst = off * KP + (kl < nk ? k : k0);
            // starting value: -lam for the multiplier part of the first solve, the current direction when refining
            const double cur = curv[q];
            const double base = first ? ((i >= T_LAM) ? -cur : 0.0) : cur;
            // multiplier part: D_k nu_k + rho_k from the combined nu
            const int j = (i >= T_LAM) ? i - T_LAM : 0;
            const double val = (i >= T_LAM) ? fma(Dv[q], stg[(T_NU + j) * CMB_LD + kl], rv[q]) : stg[(i < T_LAM ? i : 0) * CMB_LD + kl];
            if (act) dr[dst] = base + val;
        }
        WG_SYNC();
    }
    if (lane == 0) {
        if (first) { s.drg[G_TF] = sd.sol[0]; s.drg[G_LVT] = sd.linvt ? 0.0 : -s.itg[G_LVT] + sd.sol[1]; }
        else { s.drg[G_TF] += sd.sol[0]; if (!sd.linvt) s.drg[G_LVT] += sd.sol[1]; }
        if (sd.linvt) sd.zeta_vt = (first ? 0.0 : sd.zeta_vt) + sd.sol[1];
        // the zetas of the stiff terminal terms are border unknowns like dtf: kept for the refinement's residual
        for (int t = 0; t < NTERM; ++t) sd.zeta[t] = (first ? 0.0 : sd.zeta[t]) + sd.sol[2 + t];
    }
    WG_SYNC();
}

// Residual of the reduced KKT system at the current direction -> rhs record of channel 0
// ("Linear solve").  Stage-parallel.  Returns gtf_rhs, rvt_rhs and gex[] = wex * (residual of the zeta rows).
// The border unknowns zeta_t of the stiff terminal terms are part of the direction being refined (sd.zeta): the x_K row
// carries zeta_t a_t itself and the zeta row reads a_t.dx_K - zeta_t / wex_t + gh_t / w_t -- every entry O(1) -- so a
// refinement pass solves for small corrections of all border unknowns and thereby removes the cancellation error the
// first pass's combination of O(1) channel trajectories into an O(1e-8) direction leaves in dx_K (which a terminal
// weight of 1e16 would turn into an O(1) error of the new multipliers).
// stg: LDS staging area (the recursion's scratch, idle here) of 64 right-hand-side records: they leave as coalesced blocks
// (straight from the node lanes they were 24 eight-byte stores per node, each to its own cache line: this phase was as long
// as a factorisation on problems that refine in most iterations -- the stiff terminal windows of OptimalController's options).
__device__ __noinline__ void reduced_residual(const Sat &s, SatData &sd, double *stg, int lane)
{
    double rvt_rhs, gex[NTERM];      // (-> sd.rs_rvt, sd.rs_gex at the end; sd.rs_gtf)
    const int K = s.K;
    double gtf_part = 0.0;
    const double dtf = s.drg[G_TF];
    // terminal-node completion terms (the rank-1 terms and the AL shift): from the direction at node K-1, known up front
    double gin[NTERM], rvt_x;
    {
        const auto dKc = s.drn(K - 1);
        double dK[7];              // the terminal node's direction, loaded once (the loops below re-read it per term)
#pragma unroll
        for (int i = 0; i < 7; ++i) dK[i] = dKc[I_X + i];
        double av = 0.0;
#pragma unroll
        for (int i = 0; i < 7; ++i) av += sd.avt[i] * dK[i];
        rvt_rhs = -sd.cv - av;
        rvt_x = rvt_rhs;            // what the x_K row's shift -gam * rvt_x * a_vt uses (the same value for the equality)
        if (sd.linvt) {
            // the tangential pair as a terminal rank-1 term: gam is its capped share, zeta_vt its border unknown
            const double wex = sd.w_vt - sd.gam;
            const bool on = wex > 0.0;
            rvt_x = -(sd.gh_vt * (sd.gam / sd.w_vt) + sd.gam * av + (on ? sd.zeta_vt : 0.0)) / sd.gam;
            rvt_rhs = on ? -(av - sd.zeta_vt / wex + sd.gh_vt / sd.w_vt) : 0.0;
        }
        for (int t = 0; t < NTERM; ++t) {   // coefficient of a_t in the x_K row: gh share + win a.dx + zeta
            double adx = 0.0;
#pragma unroll
            for (int i = 0; i < 7; ++i) adx += sd.ta[t][i] * dK[i];
            const double wex = sd.tw[t] - sd.twin[t];
            const double share = (sd.tw[t] > 0.0) ? sd.twin[t] / sd.tw[t] : 1.0;
            const bool on = wex > 0.0;
            gin[t] = sd.tgh[t] * share + sd.twin[t] * adx + (on ? sd.zeta[t] : 0.0);
            gex[t] = on ? (adx - sd.zeta[t] / wex + sd.tgh[t] / sd.tw[t]) * wex : 0.0;
        }
    }
    for (int k0 = 0; k0 < K; k0 += 64) {
      const int k = k0 + lane;
      double *rec = stg + lane * RHS_LD;
      if (k < K) {
        // Four batches of loads per node (round 5): the node's ~230 loads were issued one at a time, each waited for before the
        // next (the function holds its 248 registers' worth of live values and the scheduler sank every load to its use): ~150
        // dependent memory round trips per call -- the refinement passes of the stiff-window solves waited here longer than a
        // factorisation takes.  Loads from clamped (always valid) addresses, selected afterwards; every sum in its old order.
        cwf64 *nb = s.nb + (size_t)k * NB_N;
        const auto ns = s.nsn(k);
        const auto p = s.itn(k), d = s.drn(k);
        const bool hp = (k >= 1), dyn = (k <= K - 2), term = (k == K - 1);
        const auto pm = p.node(hp ? -1 : 0), dm = d.node(hp ? -1 : 0), dn = d.node(dyn ? 1 : 0);
        // ---- batch 1: multipliers of rows k and k-1 ----
        double lt[7], ltm[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const double a0 = p[I_LAM + i], a1 = d[I_LAM + i], b0 = pm[I_LAM + i], b1 = dm[I_LAM + i];
            lt[i] = dyn ? a0 + a1 : 0.0;        // total multipliers lam + dlam of rows k and k-1
            ltm[i] = hp ? b0 + b1 : 0.0;
        }
        CHUNK_END
        // ---- batch 2: the node's direction, Newton scalars and record ----
        double dx[7], du[3], dnu[7], nsgx[7], nsgu[3], w3[9], wu[9], sx[SX_N];
#pragma unroll
        for (int i = 0; i < 7; ++i) { dx[i] = d[I_X + i]; dnu[i] = d[I_NU + i]; nsgx[i] = ns[NS_GX + i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { du[i] = d[I_U + i]; nsgu[i] = ns[NS_GU + i]; }
#pragma unroll
        for (int i = 0; i < 9; ++i) { w3[i] = nb[N_W3 + i]; wu[i] = nb[N_WU + i]; }
#pragma unroll
        for (int i = 0; i < SX_N; ++i) sx[i] = nb[N_SX + i];
        const double dg = nb[N_DIAG];
        CHUNK_END
        double gx[7], gu[3];
        if (hp) {
            if (term) {
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    double acc = sd.gxKsoft[i] + ltm[i];
#pragma unroll
                    for (int j = 0; j < 7; ++j) acc += sd.WxKsoft[i * 7 + j] * dx[j];
                    gx[i] = acc;
                }
            } else {
                // stage Hessian in its compact form: the 3x3 position block, the common diagonal value elsewhere (the
                // entries left out are exact zeros: same sums as with the full matrix)
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    double acc = nsgx[i] + ltm[i];
                    if (i < 3) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) acc += w3[i * 3 + j] * dx[j];
                    } else acc += dg * dx[i];
                    gx[i] = acc;
                }
            }
            if (term) {
                const double lvt = s.itg[G_LVT] + s.drg[G_LVT];
#pragma unroll
                for (int i = 0; i < 7; ++i) gx[i] += sd.avt[i] * lvt;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 7; ++i) gx[i] = 0.0;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double acc = nsgu[i];
#pragma unroll
            for (int j = 0; j < 3; ++j) acc += wu[i * 3 + j] * du[j];
            gu[i] = acc;
        }
        {
            // the stiff stage terms' excess weight, which the blocks N_W3 / N_WU do not carry (newton_blocks)
            const double ex_x = sx[SX_EX], ex_u = sx[SX_EU];
            const double a0 = sx[SX_A], a1 = sx[SX_A + 1], a2 = sx[SX_A + 2];
            const double c0 = sx[SX_CU], c1 = sx[SX_CU + 1], c2 = sx[SX_CU + 2];
            const double px = ex_x * (a0 * dx[0] + a1 * dx[1] + a2 * dx[2]);
            const double pu = ex_u * (c0 * du[0] + c1 * du[1] + c2 * du[2]);
            if (hp && dyn) { gx[0] += px * a0; gx[1] += px * a1; gx[2] += px * a2; }
            gu[0] += pu * c0; gu[1] += pu * c1; gu[2] += pu * c2;
        }
        // ---- batch 3: B_kp of the interval before, Sigma, the next node's direction ----
        double sg[7], dnx[7], dnuu[3];
        {
            double bm[21];
            const auto Bm = s.Bpt(hp ? k - 1 : 0), Sg = s.Sigt(dyn ? k : 0);
#pragma unroll
            for (int e = 0; e < 21; ++e) bm[e] = Bm[e];
#pragma unroll
            for (int i = 0; i < 7; ++i) { sg[i] = Sg[i]; dnx[i] = dn[I_X + i]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) dnuu[i] = dn[I_U + i];
            CHUNK_END
            if (hp) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < 7; ++i) acc += bm[i * 3 + j] * ltm[i];
                    gu[j] -= acc;
                }
            }
        }
        // ---- batches 4, 5: A in two halves of rows (the column sums A^T lt run over the rows in order, across the halves) ----
        double aff[7], atl[7];      // aff: dn_x - Sigma dtf - dnu - A dx (- Bn du - Bp dn_u below); atl: A^T lt
#pragma unroll
        for (int j = 0; j < 7; ++j) atl[j] = 0.0;
        {
            const auto A = s.At(dyn ? k : 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r0 = h ? 4 : 0, r1 = h ? 7 : 4;
                double am[28];
#pragma unroll
                for (int i = r0; i < r1; ++i)
#pragma unroll
                    for (int j = 0; j < 7; ++j) am[(i - r0) * 7 + j] = A[i * 7 + j];
                CHUNK_END
#pragma unroll
                for (int i = r0; i < r1; ++i) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) atl[j] += am[(i - r0) * 7 + j] * lt[i];
                    double acc = dnx[i] - sg[i] * dtf - dnu[i];
#pragma unroll
                    for (int j = 0; j < 7; ++j) acc -= am[(i - r0) * 7 + j] * dx[j];
                    aff[i] = acc;
                }
            }
        }
        if (dyn && hp) {
#pragma unroll
            for (int j = 0; j < 7; ++j) gx[j] -= atl[j];
        }
        // ---- batch 6: B_kn, B_kp of this interval ----
        {
            double bn[21], bp[21];
            const auto Bn = s.Bnt(dyn ? k : 0), Bp = s.Bpt(dyn ? k : 0);
#pragma unroll
            for (int e = 0; e < 21; ++e) { bn[e] = Bn[e]; bp[e] = Bp[e]; }
            CHUNK_END
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < 7; ++i) acc += bn[i * 3 + j] * lt[i];
                if (dyn) gu[j] -= acc;
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                double acc = aff[i];
#pragma unroll
                for (int j = 0; j < 3; ++j) acc -= bn[i * 3 + j] * du[j] + bp[i * 3 + j] * dnuu[j];
                aff[i] = acc;
            }
        }
        // ---- batch 7: the node's rho, D, e ----
        {
            double rho[7], dd[7], ee[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) { rho[i] = ns[NS_RHO + i]; dd[i] = ns[NS_D + i]; ee[i] = ns[NS_E + i]; }
            CHUNK_END
            if (dyn) {
                double sl = 0.0;
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    rec[R_RHO + i] = rho[i] + dd[i] * dnu[i] - lt[i];
                    rec[R_AFF + i] = -ee[i] - aff[i];
                    sl += sg[i] * lt[i];
                }
                gtf_part -= sl;
            }
        }
        if (term) {
            // terminal-node completion: the rank-1 terms and the AL shift; its rho / aff slots are zero as in the first record
            for (int t = 0; t < NTERM; ++t) {
#pragma unroll
                for (int i = 0; i < 7; ++i) gx[i] += gin[t] * sd.ta[t][i];
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) gx[i] -= sd.gam * rvt_x * sd.avt[i];
#pragma unroll
            for (int i = 0; i < 7; ++i) { rec[R_RHO + i] = 0.0; rec[R_AFF + i] = 0.0; }
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) rec[R_GX + i] = gx[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) rec[R_GU + i] = gu[i];
      }
      WG_SYNC();
      {
          const int nr = ((K - k0 < 64) ? K - k0 : 64) * RHS_N;
          wf64 *ch = s.ch + (size_t)k0 * CH_N + C_RHS;
          for (int e = lane; e < nr; e += 64) { const int kl = e / RHS_N, i = e - kl * RHS_N; ch[(size_t)kl * CH_N + i] = stg[kl * RHS_LD + i]; }
      }
      WG_SYNC();
    }
    sd.rs_gtf = sd.gtf + sd.Wtf * dtf + wave_sum(gtf_part);
    sd.rs_rvt = rvt_rhs;
#pragma unroll
    for (int t = 0; t < NTERM; ++t) sd.rs_gex[t] = gex[t];
    WG_SYNC();
}

// Right-hand side of the first solve of an iteration: direction 0, total multipliers 0, i.e. the Newton blocks
// themselves (what reduced_residual returns for d = (0, -lam, -lam_vt)).  Lane k writes node k's record.
__device__ __forceinline__ void first_rhs_scalars(SatData &sd)
{
    sd.rs_gtf = sd.gtf;
    sd.rs_rvt = sd.linvt ? -sd.gh_vt / sd.w_vt : -sd.cv;       // (convex variant: the pair's zeta row at the zero direction)
    for (int t = 0; t < NTERM; ++t) {
        const double share = (sd.tw[t] > 0.0) ? sd.twin[t] / sd.tw[t] : 1.0;
        sd.rs_gex[t] = sd.tgh[t] * (1.0 - share);         // = wex * gh / w: the zeta row's residual at the zero direction, times wex
    }
    wsync();
}

// The fraction-to-the-boundary step of the direction and the finite check on it.  The directions of the eliminated pairs
// (dt, ds, dz by back-substitution: pair_dir, l1_dir) are formed here only to be measured against their variables; they are
// not stored -- every trial evaluation forms them again (eval_residual).  The handful of terminal / tf pairs live in the
// global part of the direction record, as before.
__device__ __noinline__ double finish_direction(const Sat &s, SatData &sd, double mu, double tau, int lane)
{
    const int K = s.K;
    double amax = 1.0, bad = 0.0;
#define CHK(v) { if (!(fabs(v) < 1e300)) bad = 1.0; }
    const double b_u = sd.b_u, b_rmax = sd.b_rmax, b_rmin = sd.b_rmin, w_nu = sd.w_nu;
#define LIM(v, dv) { const double v_ = (v), d_ = (dv); if (d_ < 0.0) amax = fmin(amax, -tau * v_ / d_); }
    const int half = HALF_OF(lane);
    const bool h0 = (half == 0);
    for (int k = NODE_OF(lane); k < K; k += 32) {
        const auto p = s.itn(k), d = s.drn(k);
        const auto rb = s.rbn(k);
        // ALL of the node's loads are issued together -- chunk 0's 29 and 8 for each of the four rounds (component i = 4*half + r of
        // the eliminated t and of the two L1 slack pairs): one memory round trip per node instead of five.  (Round 4 had a
        // scheduling barrier behind every round; under load this pass spent most of its time waiting for memory.)
        double x[7], dx[7], u[3], du[3], bs[6], rv[4][8];
#pragma unroll
        for (int i = 0; i < 7; ++i) { x[i] = p[I_X + i]; dx[i] = d[I_X + i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { u[i] = p[I_U + i]; du[i] = d[I_U + i]; }
#pragma unroll
        for (int i = 0; i < 6; ++i) bs[i] = p[I_SU + i];
        const double rb0 = rb[0], rb1 = rb[1], rb2 = rb[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iv = 4 * half + r;
            const int i = (iv < 7) ? iv : 6;
            rv[r][0] = p[I_NU + i]; rv[r][1] = p[I_T + i]; rv[r][2] = p[I_STP + i]; rv[r][3] = p[I_ZTP + i];
            rv[r][4] = p[I_STN + i]; rv[r][5] = p[I_ZTN + i]; rv[r][6] = d[I_NU + i]; rv[r][7] = d[I_LAM + i];
        }
        CHUNK_END
        // chunk 0 (both halves compute; the step limit and the finite flag are idempotent): the ball pairs
        {
#pragma unroll
            for (int i = 0; i < 7; ++i) CHK(dx[i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) CHK(du[i]);
            {
                const PairDir q = pair_dir(bs[0], bs[1], u[0] * u[0] + u[1] * u[1] + u[2] * u[2] - b_u,
                                           2.0 * (u[0] * du[0] + u[1] * du[1] + u[2] * du[2]), mu);
                LIM(bs[0], q.ds); LIM(bs[1], q.dz);
            }
            if (k >= 1) {
                const PairDir q = pair_dir(bs[2], bs[3], x[0] * x[0] + x[1] * x[1] + x[2] * x[2] - b_rmax,
                                           2.0 * (x[0] * dx[0] + x[1] * dx[1] + x[2] * dx[2]), mu);
                LIM(bs[2], q.ds); LIM(bs[3], q.dz);
            }
            if (k >= 1 && k <= K - 2) {
                const PairDir q = pair_dir(bs[4], bs[5], -(rb0 * x[0] + rb1 * x[1] + rb2 * x[2]) - b_rmin,
                                           -(rb0 * dx[0] + rb1 * dx[1] + rb2 * dx[2]), mu);
                LIM(bs[4], q.ds); LIM(bs[5], q.dz);
            }
        }
        {
            const bool dyn = (k <= K - 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int iv = 4 * half + r;
                const bool valid = (iv < 7) && dyn;
                const double nu = rv[r][0], tt = rv[r][1], stp = rv[r][2], ztp = rv[r][3];
                const double stn = rv[r][4], ztn = rv[r][5], dnu = rv[r][6], dlam = rv[r][7];
                const L1Dir q = l1_dir(nu, tt, stp, ztp, stn, ztn, dnu, mu, w_nu);
                if (valid) {
                    CHK(dnu); CHK(dlam); CHK(q.dt);
                    LIM(stp, q.dstp); LIM(ztp, q.dztp);
                    LIM(stn, q.dstn); LIM(ztn, q.dztn);
                }
            }
        }
        if (h0 && k == K - 1) {
            for (int j = 0; j < sd.nT; ++j) {
                const int js = gs_term(j), jz = gz_term(j);
                double gj = -sd.bT[j], dg = 0.0;
                for (int i = 0; i < 7; ++i) { gj += sd.aT[j][i] * x[i]; dg += sd.aT[j][i] * dx[i]; }
                const double sj = s.itg[js], zj = s.itg[jz];
                const double sig = zj / sj, zh = mu / sj + sig * (gj + sj);
                s.drg[js] = -(gj + sj) - dg; s.drg[jz] = zh + sig * dg - zj;
                LIM(sj, s.drg[js]); LIM(zj, s.drg[jz]);
            }
            const double srf = s.itg[G_SRF], zrf = s.itg[G_ZRF];
            const double g = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] - sd.b_rfmax;
            const double sig = zrf / srf, zh = mu / srf + sig * (g + srf);
            const double dg = 2.0 * (x[0] * dx[0] + x[1] * dx[1] + x[2] * dx[2]);
            s.drg[G_SRF] = -(g + srf) - dg; s.drg[G_ZRF] = zh + sig * dg - zrf;
            LIM(srf, s.drg[G_SRF]); LIM(zrf, s.drg[G_ZRF]);
        }
    }
    if (lane == 0 && !sd.fixed_tf) {
        const double tf = s.itg[G_TF], dtf = s.drg[G_TF];
        const double gv[2] = {-tf - sd.b_tf[0], tf - sd.b_tf[1]}, dgv[2] = {-dtf, dtf};
        for (int j = 0; j < 2; ++j) {
            const double sj = s.itg[G_STF + j], zj = s.itg[G_ZTF + j];
            const double sig = zj / sj, zh = mu / sj + sig * (gv[j] + sj);
            s.drg[G_STF + j] = -(gv[j] + sj) - dgv[j]; s.drg[G_ZTF + j] = zh + sig * dgv[j] - zj;
            LIM(sj, s.drg[G_STF + j]); LIM(zj, s.drg[G_ZTF + j]);
        }
    }
    if (!(fabs(s.drg[G_TF]) < 1e300)) bad = 1.0;
#undef LIM
#undef CHK
    amax = wave_min(amax);
    sd.dir_finite = (wave_max(bad) == 0.0) ? 1 : 0;
    WG_SYNC();
    return amax;
}

}  // namespace MPCX_NS
