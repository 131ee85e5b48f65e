// solve.hip -- batched per-satellite solve of the SCP subproblem on gfx950.
//
// Replaces Optimizer.get_constraint_terms + Optimizer.solve_OPT (reference optimizer.py:80-170,
// 219-613: the NLP that pyomo transcribes and ipopt solves) for S satellites at once.
//
// One wavefront (64 lanes) per satellite runs the whole primal-dual interior-point iteration:
//  * node-parallel phases (KKT residuals, barrier Hessian blocks, slack/multiplier updates, step
//    length, line-search trials) map one lane to one temporal node and work on field-major arrays
//    (coalesced), written as chunks "loads -> arithmetic -> stores" so that a chunk's loads are in flight together;
//  * the structure-exploiting linear solve is a Riccati recursion in the shifted state
//    y_k = x_k - Bp_{k-1} u_k (absorbs the first-order hold), 7x7 / 7x3 / 3x3 blocks staged in LDS
//    with one lane per matrix element and node k-1's operands prefetched while node k is worked on; the virtual
//    control nu_k is eliminated per stage by a 7x7 LDL^T (redundantly in the registers of every lane); the free
//    final time, the tangential-velocity equality and the five stiff rank-1 terminal barrier terms form a 7x7 border
//    solved by a symmetric quasi-definite LDL^T whose pivot signs also give the inertia; eight linear-term sweeps (1 right-hand side + 7 border columns) run side
//    by side in the 8 lane groups of the wave, the backward one fused into the factorisation loop; iterative
//    refinement on the reduced KKT system only once a terminal weight is stiff enough to cost digits.
// Per-satellite state lives in a global-memory workspace (ws_doubles: 215 KB at K = 30); no MFMA.
#include <cstddef>
#include "mpcx_device.hpp"
#include "mpcx_host.hpp"

// This file is compiled twice.  As itself it gives solve_kernel (one wave per satellite: every barrier of a phase function
// is that wave's) and the C entry points.  Included by solve2w.hip with MPCX_TWO_WAVE defined it gives, in its own
// namespace, the small-batch kernel whose workgroups have a second wave that shares the factorisation (riccati_factor
// below): there the phase functions still run on the first wave alone, so their "workgroup barriers" must not be hardware
// barriers (the second wave never executes them) -- WG_SYNC() is then the memory fence only, and the real two-wave
// barriers are spelled WG_BARRIER().
#ifdef MPCX_TWO_WAVE
#define MPCX_NS mpcx2w
#define WG_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define MPCX_NS mpcx
#define WG_SYNC() __syncthreads()
#endif
#define WG_BARRIER() __syncthreads()

namespace MPCX_NS {
#ifdef MPCX_TWO_WAVE
using namespace mpcx;          // (the device helpers of mpcx_device.hpp)
#endif

// ---- workspace layout (doubles) -------------------------------------------------------
// iterate / direction record per node
enum { I_X = 0, I_U = 7, I_NU = 10, I_T = 17, I_LAM = 24, I_STP = 31, I_ZTP = 38, I_STN = 45, I_ZTN = 52,
       I_SU = 59, I_ZU = 60, I_SRMAX = 61, I_ZRMAX = 62, I_SRMIN = 63, I_ZRMIN = 64, IT_N = 66 };
// global part of iterate / direction
enum { G_STERM = 0, G_ZTERM = 6, G_SRF = 12, G_ZRF = 13, G_STF = 14, G_ZTF = 16, G_TF = 18, G_LVT = 19, G_SVT = 20, G_ZVT = 22, GL_N = 24 };
// slack / multiplier slot of terminal inequality row j: rows 0..5 always, rows 6, 7 (the linearised tangential pair) in the
// convex variant only
__host__ __device__ inline int gs_term(int j) { return j < 6 ? G_STERM + j : G_SVT + (j - 6); }
__host__ __device__ inline int gz_term(int j) { return j < 6 ? G_ZTERM + j : G_ZVT + (j - 6); }
// Newton blocks per node: the part the recursion reads as one contiguous record per node ...
// (N_W3, N_DIAG, N_ZERO: the stage Hessian of x is diag(N_DIAG) with its 3x3 position block replaced by N_W3 -- stored in
//  that form, 11 doubles instead of 49 (N_ZERO holds 0.0: what the off-diagonal lanes of the expanding fetch read), and
//  expanded when the recursion fetches it into LDS; the terminal node's full matrix lives in SatData.  N_SX: the stage's stiff barrier terms -- excess weight above kStageCap and direction of the position
//  term (r_min plane or radius ball) and of the thrust ball; the blocks N_W3 / N_WU carry only the capped share, see
//  riccati_factor)
enum { N_W3 = 0, N_DIAG = 9, N_ZERO = 10, N_WU = 11, N_D = 20, N_SX = 27, NB_N = 35 };
enum { SX_EX = 0, SX_A = 1, SX_EU = 4, SX_CU = 5, SX_N = 8 };
// ... and the part only the node-parallel phases touch (field-major, see Col below)
enum { NS_AA = 0, NS_BB = 7, NS_GT = 14, NS_RHO = 21, NS_GX = 28, NS_GU = 35, NS_E = 38, NS_D = 45, NS_N = 52 };
// factorisation per node
// factorisation per node, stored in exactly the order the sweeps stage it through LDS (one contiguous block)
// (what the factorisation produces: G, Minv, Kg, Bh, Qi, and Pt for the refinement's backward sweep) ...
enum { F_G = 0, F_MINV = 49, F_KG = 98, F_BH = 119, F_QI = 140, F_PT = 149, FAC_USED = 198, FAC_N = 200 };
// ... followed, in the sweeps' LDS copy only, by the node's inputs fetched from where they already are: A (head of the
// stage record; the 15 doubles after it are B_kn, unused), Bpm (B_kp of the record before) and D (Newton record)
enum { F_A = 256, F_BPM = 320, F_D = 341, FLAT_N = 384 };
// channel vectors per node: 8 channels x (p 7, qu 3) then the rhs record (gx 7, gu 3, rho 7, aff 7)
enum { C_P = 0, C_QU = 56, C_RHS = 80, R_GX = 0, R_GU = 7, R_RHO = 10, R_AFF = 17, RHS_N = 24, CH_N = 104 };
// stored trajectory of one channel at one node
// (the multiplier part of a channel's trajectory is not stored: lam_k = D_k nu_k + rho_k -- sweep_forward -- is linear in
//  nu, so combine_channels forms it from the combined nu, D_k of the Newton record and the right-hand side's rho_k)
enum { T_X = 0, T_U = 7, T_NU = 10, TR_N = 17, T_LAM = TR_N, DIR_N = 24 };
constexpr int RHS_LD = RHS_N + 1, CMB_LD = 33;   // LDS strides of newton_blocks' rhs staging and combine_channels' transposition (bank-conflict free)
constexpr int NCH = 8;        // channel 0: rhs, 1: dtf, 2: vt multiplier, 3..7: terminal rank-1 terms
constexpr int NBD = 7;        // border unknowns
constexpr int NTERM = 5;
#ifndef MPCX_REFINE_TW
#define MPCX_REFINE_TW 1e10     // (1e9 until round 3: profiles/r03/refine_threshold.txt)
#endif
constexpr double kBoundRelax = 1e-8, kBoundPush = 1e-4, kKappaSigma = 100.0, kGammaNbhd = 1e-8, kAlphaFloor = 0.25, kDwFirst = 1e-4, kDwMin = 1e-20, kDwMax = 1e40, kTermCap = 1e4, kStageCap = 1e8, kRefineTw = MPCX_REFINE_TW, kMuInit = 1.0, kMuInitClean = 0.01, kCleanRadius = 3.0, kSigma = 0.1, kFbAlpha = 0.1, kFbBoost = 10.0, kMuErr = 1e-6;
constexpr int kFbN = 8;

struct SolveOpts {
    double min_mass, u_max, r_min, r_max, eps_r, eps_vr, eps_vn, eps_vt, tf_max, w_nu, w_tr, tol, acc_tol;
    int max_iter, acc_iter, n_refine, linvt;    // linvt: the linearised tangential pair (optimizer.py:471-489) instead of the quartic
    int fixed_tf, shared_tf;                    // fixed_tf: tf is held at the value passed in tf_out (MPCX_SOLVE_FIXED_TF);
                                                // shared_tf: ONE tf for all satellites of the launch (MPCX_SOLVE_SHARED_TF)
};

struct SolveArgs {
    int S, K;                 // K: node count of every satellite, or (Ks given) the row length of the arrays
    const int32_t *Ks;        // ragged batch: satellite s has Ks[s] <= K nodes in the first columns of its rows; nullptr: all K
    const double *stage, *xbar, *ubar, *tfbar, *consts, *r_des;
    SolveOpts o;
    double *X, *U, *NU, *tf_out, *kkt;
    int32_t *status, *iters;
    const int32_t *order;     // workgroup b solves satellite order[b]; nullptr = index order
    // shared-tf launches (solve_shared_kernel): per-block reduction slots [2][S][GR_N], arrival counter, abort flag
    double *red;
    int32_t *arrive, *abort_flag;
    int32_t *counter;         // work queue of the persistent workgroups: next position of the launch order (zeroed per launch)
    int32_t *nreg;            // [S][2]: iterations whose direction needed delta_w > 0, and the first of them (-1: none)
    double *ws;
    size_t ws_stride;
};

// padded node count: leading dimension of the field-major arrays (rows start on 128-byte boundaries)
__host__ __device__ inline int padded_nodes(int K) { return (K + 15) & ~15; }

__host__ __device__ inline size_t ws_doubles(int K)
{
    const size_t KP = (size_t)padded_nodes(K);
    const size_t n = KP * (3 * IT_N + NS_N + MPCX_STAGE_DOUBLES + 3) + (size_t)K * (NB_N + FAC_N + CH_N + NCH * TR_N) + 3 * GL_N + 64;
    return (n + 15) & ~(size_t)15;
}

// ---- per-satellite constant data kept in LDS --------------------------------------------
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

// Field-major view of one node's record: element i of node k lives at base[i * ld + k].  The node-parallel phases
// (one lane per node) read and write the same field of consecutive nodes in consecutive lanes, so every access is
// a couple of full cache lines instead of one line per lane.
// The base is made wave-uniform (SGPR pair) and the per-lane part is a 32-bit byte offset, so an access is one
// global_load/store with scalar base + vector offset and costs a single 32-bit VALU add for its address.
template <typename T>
__device__ __forceinline__ T *wave_uniform(T *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}

template <typename T>
struct Col {
    T *base;      // wave-uniform array base
    int k;        // element offset of this lane's node (and of the first field of the view)
    int ld;
    __device__ __forceinline__ T &operator[](int i) const
    {
        typedef __attribute__((address_space(1))) char gchar;
        return *(T *)((gchar *)base + (unsigned)((i * ld + k) * 8));
    }
    __device__ __forceinline__ Col operator+(int off) const { return Col{base, k + off * ld, ld}; }
    __device__ __forceinline__ Col node(int dk) const { return Col{base, k + dk, ld}; }   // same fields, node k + dk
};

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

struct Sat {
    int K, KP;
    int ldk;                      // row length of xbar / ubar (= K unless the batch is ragged)
    cgf64 *stage, *xbar, *ubar;   // stage (K-1,105) record per node; xbar (7,K); ubar (3,K)
    gf64 *it, *dr, *nbs, *stT, *rbh;            // field-major [field][KP]: iterate, direction, Newton scalars, stage copy, r-hat
    gf64 *itg, *drg, *nb, *fac, *ch, *traj;     // globals; record-per-node arrays read by the recursion (one wave, one record)
    gf64 *itB, *itgB;                           // the candidate iterate of the line search (swapped with it, itg on acceptance)
    gf64 *sink;                                 // 64 doubles nobody reads: target of the lanes a branch-free store leaves idle
    gf64 *ws;                                   // base of the satellite's workspace and the element offsets of the arrays the
    int o_fac, o_ch, o_traj, o_sink;            // recursions store to (plain integers: see ustore)
    __device__ Col<gf64> itn(int k) const { return Col<gf64>{wave_uniform(it), k, KP}; }
    __device__ Col<gf64> itBn(int k) const { return Col<gf64>{wave_uniform(itB), k, KP}; }
    __device__ Col<gf64> drn(int k) const { return Col<gf64>{wave_uniform(dr), k, KP}; }
    __device__ Col<gf64> nsn(int k) const { return Col<gf64>{wave_uniform(nbs), k, KP}; }
    __device__ Col<gf64> rbn(int k) const { return Col<gf64>{wave_uniform(rbh), k, KP}; }
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
__device__ __forceinline__ double trial_value(const Col<gf64> &p, const Col<gf64> &d, int off, double a, bool z)
{
    const double dv = d[off], pv = p[off];
    return fma(a, z ? 0.0 : dv, pv);
}

struct ResAcc {   // accumulators of one residual evaluation
    double dual_max, prim_max, sq, zsum, lsum, prod_min, prod_max, prod_sum;
    double g_tf;      // the satellite's term of the tf stationarity row, 2 w_tr (tf - tf_bar) - sum_k Sigma_k . lam_k
};

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
__device__ __noinline__ void eval_residual(const Sat &s, SatData &sd, double a, double mu, double mu_clip, int lane, ResAcc &out)
{
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
        const auto p = s.itn(k), d = s.drn(k), w = s.itBn(k), nsv = s.nsn(k);
        const bool has_prev = (k >= 1), dyn = (k <= K - 2);
        // ---- chunk 0 (both halves, accounted by half 0): states, thrust, ball slacks, objective gradient ----
        double x[7], u[3], gx[7], gu[3], un[3];
        double su, zu, srmax, zrmax, srmin, zrmin;
        const auto pn = p.node(dyn ? 1 : 0), dn = d.node(dyn ? 1 : 0);
        const auto rb = s.rbn(k);
        const double rb0 = rb[0], rb1 = rb[1], rb2 = rb[2];
        {
            double x0[7], dx[7], u0[3], du[3], bs[6];
#pragma unroll
            for (int i = 0; i < 7; ++i) { x0[i] = p[I_X + i]; dx[i] = d[I_X + i]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { u0[i] = p[I_U + i]; du[i] = d[I_U + i]; }
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
        for (int i = 0; i < 7; ++i) gx[i] = 2.0 * w_tr * (x[i] - s.xbar[(size_t)i * s.ldk + k]);
#pragma unroll
        for (int i = 0; i < 3; ++i) gu[i] = 2.0 * w_tr * (u[i] - s.ubar[(size_t)i * s.ldk + k]) + 2.0 * u[i] * zu;
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
                const double dnu = z ? 0.0 : dnu_;
                const L1Dir ld = l1_dir(nu0, tt0, stp0, ztp0, stn0, ztn0, dnu, mu, w_nu);
                const bool lon = !z && dyn;                                   // (the terminal node has no virtual control)
                const double nu = fma(a, dnu, nu0), tt = fma(a, lon ? ld.dt : 0.0, tt0);
                double stp = fma(a, lon ? ld.dstp : 0.0, stp0), ztp = fma(a, lon ? ld.dztp : 0.0, ztp0);
                double stn = fma(a, lon ? ld.dstn : 0.0, stn0), ztn = fma(a, lon ? ld.dztn : 0.0, ztn0);
                if (WRITE && dyn) { POST(stp, ztp, nu - tt); POST(stn, ztn, -nu - tt); }
                const double xn = TRIAL(pn, dn, I_X + i);
                const double lmv = TRIAL(pm, dm, I_LAM + i);
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
    if (WRITE) WG_SYNC();            // the candidate iterate is complete before anybody reads it
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
        const auto p = s.itn(k), ns = s.nsn(k);
        const auto rb = s.rbn(k);
        double *nb = stg + NODE_OF(lane) * NB_N;
        double *rhs = stg + 32 * NB_N + NODE_OF(lane) * RHS_LD;      // (odd stride: the 32 node lanes hit different banks)
        const bool dyn = (k <= K - 2), inner = (k >= 1 && k <= K - 2);
        // ---- chunk 0: objective, thrust ball, radius balls ----
        // (chunk 0 is computed by both halves and stored by half 0)
        double x[7], u[3], gx[7], gu[3], Wx3[9];
        double zh_rmax = 0.0, sig_rmax = 0.0, zrmax;
        {
            double bs[6], xb[7], ub[3];
#pragma unroll
            for (int i = 0; i < 7; ++i) { x[i] = p[I_X + i]; xb[i] = s.xbar[(size_t)i * s.ldk + k]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { u[i] = p[I_U + i]; ub[i] = s.ubar[(size_t)i * s.ldk + k]; }
#pragma unroll
            for (int i = 0; i < 6; ++i) bs[i] = p[I_SU + i];
            const double rb0 = rb[0], rb1 = rb[1], rb2 = rb[2];
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
                const double nu = p[I_NU + i], tt = p[I_T + i], stp = p[I_STP + i], ztp = p[I_ZTP + i];
                const double stn = p[I_STN + i], ztn = p[I_ZTN + i];
                const double g1 = nu - tt, g2 = -nu - tt;
                const double ip = rcp_pos(stp), in = rcp_pos(stn);      // slacks are positive: reciprocal + products
                const double s1 = ztp * ip, s2 = ztn * in;
                const double zh1 = mu * ip + s1 * (g1 + stp), zh2 = mu * in + s2 * (g2 + stn);
                const double aa = s1 + s2, bb = s2 - s1, gt = w_nu - zh1 - zh2;
                const double ia = rcp_pos(aa);
                const double dd = 4.0 * s1 * s2 * ia;
                const double ek = ns[NS_E + i];
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
          gf64 *dst = s.nb + (size_t)k0 * NB_N;
          for (int e = lane; e < ne; e += 64) dst[e] = stg[e];
          // ... and the right-hand-side records (24 contiguous doubles inside each node's channel record)
          const int nr = ((K - k0 < 32) ? K - k0 : 32) * RHS_N;
          gf64 *ch = s.ch + (size_t)k0 * CH_N + C_RHS;
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

// ---- tiny dense helpers on LDS matrices --------------------------------------------------------
// Inverse of a symmetric positive definite 3x3 through its LDL^T factorisation; false if a pivot is not positive.
// (Q_uu carries the thrust-ball barrier term sigma 4 u u^T, which reaches 1e12 when the ball is active: the cofactor
// formula and a determinant test lose every digit there and report breakdowns that are not; the pivots do not.)
__device__ __forceinline__ bool inv3_spd(const double *Q, double *Qi)
{
    const double a = Q[0], b = Q[1], c = Q[2], d = Q[4], e = Q[5], f = Q[8];
    const double d1 = a;
    const double r1 = rcp_pos(d1 > 0.0 ? d1 : 1.0);
    const double l21 = b * r1, l31 = c * r1;
    const double d2 = d - l21 * b;
    const double r2 = rcp_pos(d2 > 0.0 ? d2 : 1.0);
    const double t32 = e - l31 * b;
    const double l32 = t32 * r2;
    const double d3 = f - l31 * c - l32 * t32;
    const double r3 = rcp_pos(d3 > 0.0 ? d3 : 1.0);
    const bool ok = (d1 > 0.0) && (d2 > 0.0) && (d3 > 0.0);
    // rows of L^-1 (unit lower): m1 = (1, 0, 0), m2 = (-l21, 1, 0), m3 = (l21 l32 - l31, -l32, 1); Qi = sum_k m_k m_k^T / d_k
    const double m31 = l21 * l32 - l31, m32 = -l32, m21 = -l21;
    Qi[0] = r1 + m21 * m21 * r2 + m31 * m31 * r3;
    Qi[1] = m21 * r2 + m31 * m32 * r3;
    Qi[2] = m31 * r3;
    Qi[4] = r2 + m32 * m32 * r3;
    Qi[5] = m32 * r3;
    Qi[8] = r3;
    Qi[3] = Qi[1]; Qi[6] = Qi[2]; Qi[7] = Qi[5];
    return ok;
}

// Operands of one node, double-buffered in LDS.  A and Bh live side by side as F = [A | Bh] (7 x 10, row stride FS),
// Wx and Wx Bpm as G2 = [Wx | WxBp]: with them Pt F, Bpm^T G2 and F^T (Pt F) give every Q block in three rounds of
// dot products of one access pattern each (see the factorisation loop).
constexpr int FS = 10;
struct StageOps {
    double F[7 * FS], G2[7 * FS];
    double Bn[21], Bpm[21], Wu[9], D[7], SX[SX_N];        // the rest of the prefetched inputs (fetch order A Bn Bpm Wx Wu D SX)
    double G[49], Pt[49], Minv[49], Kg[21];
#ifdef MPCX_TWO_WAVE
    double Qi[9];                                         // Q_uu^-1 of the node: the second wave writes it to the factor record
#endif
};
constexpr int OPS_IN = 91 + 49 + 9 + 7 + SX_N;   // A 49 | Bn 21 | Bpm 21 | Wx 49 (expanded) | Wu 9 | D 7 | SX 8

struct Scratch {   // LDS working set of the recursion (and, between recursions, the staging area of newton_blocks)
    union {                        // the factorisation and the stand-alone sweeps never run at the same time
#ifdef MPCX_TWO_WAVE
        StageOps ops[3];           // (two waves: node k+1 is still being swept while node k-1's operands arrive)
#else
        StageOps ops[2];
#endif
        double flat[2][FLAT_N];    // sweep operands of one node, double-buffered (fac record + A, Bpm, D)
    };
#ifdef MPCX_TWO_WAVE
    double Pn2[2][49];             // P_{k+1} is read by the second wave while the first writes P_k
    double WlLi1[98];              // the second wave's own L^-1 [Pn | I]
    int cmd, cmd_arg, good_flag;   // command of the first wave to the second (solve2w.hip), breakdown flag of a node
#endif
    double Pn[49], WlLi[98], Qyy[49];
    double T[7 * FS];              // Pt F = [Pt A | Pt Bh]
    double sink[64];               // target of the lanes that have nothing to write in a branch-free phase
    double Quy[21];
    double Quu[9];
    double zero;                   // constant 0 (addend of the tasks that have none)
    double stage_pad[32 * (NB_N + RHS_LD) - 1129 > 0 ? 32 * (NB_N + RHS_LD) - 1129 : 1];   // newton_blocks stages 32 Newton + 32 rhs records here
};

static_assert(sizeof(Scratch) >= 32 * (NB_N + RHS_LD) * sizeof(double) && sizeof(Scratch) >= TR_N * CMB_LD * sizeof(double) && sizeof(Scratch) >= 64 * RHS_LD * sizeof(double), "newton_blocks stages 32 Newton and right-hand-side records in the recursion's scratch");

template <int N>
__device__ __forceinline__ double dotN(const double *a, int sa, const double *b, int sb)
{
    double x[N], y[N];
#pragma unroll
    for (int l = 0; l < N; ++l) { x[l] = a[l * sa]; y[l] = b[l * sb]; }
    double acc = 0.0;
#pragma unroll
    for (int l = 0; l < N; ++l) acc += x[l] * y[l];
    return acc;
}

#ifdef MPCX_PHASE_TIMING
#define FT_DECL unsigned long long ft0_ = __builtin_amdgcn_s_memtime(), ft1_;
#define FT_MARK(i) { ft1_ = __builtin_amdgcn_s_memtime(); if (lane == 0) sd.fpt[i] += ft1_ - ft0_; ft0_ = ft1_; }
#else
#define FT_DECL
#define FT_MARK(i)
#endif

struct ChanIn { double gx, gu, rho, aff; };

// inputs of component r of channel c at node k (channel 0: rhs record; 1: unit dtf; 2..: unit terminal gradients)
__device__ __forceinline__ ChanIn chan_inputs(const Sat &s, const SatData &sd, int k, int c, int r, bool act)
{
    ChanIn ci{0.0, 0.0, 0.0, 0.0};
    if (!act) return ci;
    const int K = s.K;
    const bool dyn = (k <= K - 2);
    if (c == 0) {
        cgf64 *ch = s.ch + (size_t)k * CH_N + C_RHS;
        ci.gx = ch[R_GX + r];
        if (r < 3) ci.gu = ch[R_GU + r];
        if (dyn) { ci.rho = ch[R_RHO + r]; ci.aff = ch[R_AFF + r]; }
    } else if (c == 1) { if (dyn) ci.aff = s.Sig(k)[r]; }
    else if (k == K - 1) ci.gx = (c == 2) ? sd.avt[r] : sd.ta[c - 3][r];
    return ci;
}

// Branch-free prefetch of the same inputs for a node k <= K-2: every lane loads from a valid address and
// chan_mask zeroes what its channel / component does not carry, so the loads stay in flight across the
// arithmetic of the node before (a load inside a divergent branch would be waited for at the branch's end).
struct ChanRaw { double gx, gu, rho, aff; };

__device__ __forceinline__ ChanRaw chan_fetch(const Sat &s, int k, int c, int rr, int r3)
{
    const int K = s.K;
    cgf64 *ch = s.ch + (size_t)k * CH_N + C_RHS;
    cgf64 *pa = (c == 1) ? s.Sig(k < K - 2 ? k : K - 2) + rr : ch + R_AFF + rr;
    ChanRaw cr;
    cr.gx = ch[R_GX + rr]; cr.gu = ch[R_GU + r3]; cr.rho = ch[R_RHO + rr]; cr.aff = *pa;
    return cr;
}

__device__ __forceinline__ ChanIn chan_mask(const ChanRaw &cr, int c, int r, bool act)
{
    ChanIn ci;
    const bool c0 = act && c == 0;
    ci.gx = c0 ? cr.gx : 0.0; ci.gu = (c0 && r < 3) ? cr.gu : 0.0; ci.rho = c0 ? cr.rho : 0.0;
    ci.aff = (act && c <= 1) ? cr.aff : 0.0;
    return ci;
}

// Stiff stage terms (excess weight ex above kStageCap of the position term, direction a, and of the thrust ball, direction
// c_u = 2u; newton_blocks left them out of Wx / Wu) enter the recursion as Q += ex c c^T with c = (c_u, c_y) in the
// (u_k, y_k) coordinates (x_k = y_k + Bpm u_k, so the position term has c_u = Bpm^T a, c_y = a), by Sherman-Morrison on
// the already inverted Q_uu:  t = Qi c_u, om = 1 / (1/ex + c_u.t), v = c_y - Quy^T t,  Qi -= om t t^T, Kg += om t v^T,
// P_k += om v v^T  -- the weight enters only through 1/ex, nothing of size ex is ever formed (condensed into the
// blocks, 1e14 r r^T would leave no digit of the trust-region curvature 2 w_tr in the other directions).  Position term
// first, thrust ball second (on the once-updated quantities).  Same arithmetic as the oracle's riccati_factor.
__device__ __noinline__ void stiff_stage_update(StageOps &o, Scratch &w, gf64 *fac, int lane)
{
    double Qi[9];
    (void)inv3_spd(w.Quu, Qi);
    const double ex_x = o.SX[SX_EX], ex_u = o.SX[SX_EU];
    double om1 = 0.0, om2 = 0.0, t1[3] = {0.0, 0.0, 0.0}, t2[3] = {0.0, 0.0, 0.0}, tc = 0.0;
    double ax[3], cu[3], c1[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) { ax[l] = o.SX[SX_A + l]; cu[l] = o.SX[SX_CU + l]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) c1[j] = o.Bpm[j] * ax[0] + o.Bpm[3 + j] * ax[1] + o.Bpm[6 + j] * ax[2];
    if (ex_x > 0.0) {
#pragma unroll
        for (int l = 0; l < 3; ++l) t1[l] = Qi[l * 3] * c1[0] + Qi[l * 3 + 1] * c1[1] + Qi[l * 3 + 2] * c1[2];
        om1 = 1.0 / (1.0 / ex_x + (c1[0] * t1[0] + c1[1] * t1[1] + c1[2] * t1[2]));
    }
    if (ex_u > 0.0) {
        double q2[3];
#pragma unroll
        for (int l = 0; l < 3; ++l) q2[l] = Qi[l * 3] * cu[0] + Qi[l * 3 + 1] * cu[1] + Qi[l * 3 + 2] * cu[2];
        tc = t1[0] * cu[0] + t1[1] * cu[1] + t1[2] * cu[2];
#pragma unroll
        for (int l = 0; l < 3; ++l) t2[l] = q2[l] - om1 * tc * t1[l];                  // Qi' c_u with Qi' = Qi - om1 t1 t1^T
        om2 = 1.0 / (1.0 / ex_u + (cu[0] * t2[0] + cu[1] * t2[1] + cu[2] * t2[2]));
    }
    // component j of v1 = a - Quy^T t1 and of v2 = -Kg'^T c_u = -(Quy^T q2) - om1 (t1.c_u) v1, from column j of Quy
    // (Quy^T q2 = Quy^T (t2 + om1 tc t1))
    auto sm_v = [&](const double (&qc)[3], int j, double &v1, double &v2) {
        const double cyj = (j < 3) ? o.SX[SX_A + j] : 0.0;
        const double qt1 = qc[0] * t1[0] + qc[1] * t1[1] + qc[2] * t1[2];
        v1 = (ex_x > 0.0) ? cyj - qt1 : 0.0;
        v2 = -(qc[0] * t2[0] + qc[1] * t2[1] + qc[2] * t2[2]) - om1 * tc * qt1 - om1 * tc * v1;
    };
    wsync();                                       // every lane has read what it needs of the un-updated values
    if (lane < 49) {
        const int mi = lane / 7, mj = lane - 7 * mi;
        const int lo = (mi < mj) ? mi : mj, hi = (mi < mj) ? mj : mi;
        double qi[3], qj[3];
#pragma unroll
        for (int l = 0; l < 3; ++l) { qi[l] = w.Quy[l * 7 + lo]; qj[l] = w.Quy[l * 7 + hi]; }
        double v1l, v2l, v1h, v2h;
        sm_v(qi, lo, v1l, v2l); sm_v(qj, hi, v1h, v2h);
        w.Pn[lane] += om1 * (v1l * v1h) + om2 * (v2l * v2h);
    }
    if (lane < 21) {
        const int r = lane / 7, c = lane - 7 * r;
        const double qc[3] = {w.Quy[c], w.Quy[7 + c], w.Quy[14 + c]};
        double v1, v2;
        sm_v(qc, c, v1, v2);
        const double kg = o.Kg[lane] + om1 * t1[r] * v1 + om2 * t2[r] * v2;
        o.Kg[lane] = kg; fac[F_KG + lane] = kg;
    }
    if (lane < 9) {
        const int r = lane / 3, c = lane - 3 * r;
        fac[F_QI + lane] = Qi[lane] - om1 * t1[r] * t1[c] - om2 * t2[r] * t2[c];
    }
}

#ifdef MPCX_TWO_WAVE
// (two-wave build: the same update on the node's LDS copies -- P_k's buffer, gain, Q_uu^-1 -- which the second wave stores)
__device__ __noinline__ void stiff_stage_update2(StageOps &o, Scratch &w, double *PnT, int lane)
{
    double Qi[9];
    (void)inv3_spd(w.Quu, Qi);
    const double ex_x = o.SX[SX_EX], ex_u = o.SX[SX_EU];
    double om1 = 0.0, om2 = 0.0, t1[3] = {0.0, 0.0, 0.0}, t2[3] = {0.0, 0.0, 0.0}, tc = 0.0;
    double ax[3], cu[3], c1[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) { ax[l] = o.SX[SX_A + l]; cu[l] = o.SX[SX_CU + l]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) c1[j] = o.Bpm[j] * ax[0] + o.Bpm[3 + j] * ax[1] + o.Bpm[6 + j] * ax[2];
    if (ex_x > 0.0) {
#pragma unroll
        for (int l = 0; l < 3; ++l) t1[l] = Qi[l * 3] * c1[0] + Qi[l * 3 + 1] * c1[1] + Qi[l * 3 + 2] * c1[2];
        om1 = 1.0 / (1.0 / ex_x + (c1[0] * t1[0] + c1[1] * t1[1] + c1[2] * t1[2]));
    }
    if (ex_u > 0.0) {
        double q2[3];
#pragma unroll
        for (int l = 0; l < 3; ++l) q2[l] = Qi[l * 3] * cu[0] + Qi[l * 3 + 1] * cu[1] + Qi[l * 3 + 2] * cu[2];
        tc = t1[0] * cu[0] + t1[1] * cu[1] + t1[2] * cu[2];
#pragma unroll
        for (int l = 0; l < 3; ++l) t2[l] = q2[l] - om1 * tc * t1[l];                  // Qi' c_u with Qi' = Qi - om1 t1 t1^T
        om2 = 1.0 / (1.0 / ex_u + (cu[0] * t2[0] + cu[1] * t2[1] + cu[2] * t2[2]));
    }
    // component j of v1 = a - Quy^T t1 and of v2 = -Kg'^T c_u = -(Quy^T q2) - om1 (t1.c_u) v1, from column j of Quy
    // (Quy^T q2 = Quy^T (t2 + om1 tc t1))
    auto sm_v = [&](const double (&qc)[3], int j, double &v1, double &v2) {
        const double cyj = (j < 3) ? o.SX[SX_A + j] : 0.0;
        const double qt1 = qc[0] * t1[0] + qc[1] * t1[1] + qc[2] * t1[2];
        v1 = (ex_x > 0.0) ? cyj - qt1 : 0.0;
        v2 = -(qc[0] * t2[0] + qc[1] * t2[1] + qc[2] * t2[2]) - om1 * tc * qt1 - om1 * tc * v1;
    };
    wsync();                                       // every lane has read what it needs of the un-updated values
    if (lane < 49) {
        const int mi = lane / 7, mj = lane - 7 * mi;
        const int lo = (mi < mj) ? mi : mj, hi = (mi < mj) ? mj : mi;
        double qi[3], qj[3];
#pragma unroll
        for (int l = 0; l < 3; ++l) { qi[l] = w.Quy[l * 7 + lo]; qj[l] = w.Quy[l * 7 + hi]; }
        double v1l, v2l, v1h, v2h;
        sm_v(qi, lo, v1l, v2l); sm_v(qj, hi, v1h, v2h);
        PnT[lane] += om1 * (v1l * v1h) + om2 * (v2l * v2h);
    }
    if (lane < 21) {
        const int r = lane / 7, c = lane - 7 * r;
        const double qc[3] = {w.Quy[c], w.Quy[7 + c], w.Quy[14 + c]};
        double v1, v2;
        sm_v(qc, c, v1, v2);
        const double kg = o.Kg[lane] + om1 * t1[r] * v1 + om2 * t2[r] * v2;
        o.Kg[lane] = kg;
    }
    if (lane < 9) {
        const int r = lane / 3, c = lane - 3 * r;
        o.Qi[lane] = Qi[lane] - om1 * t1[r] * t1[c] - om2 * t2[r] * t2[c];
    }
}
#endif


#ifdef MPCX_TWO_WAVE
// The same factorisation shared by the two waves of a small-batch workgroup (role 0 / role 1), one hardware barrier per node.
// Role 0 keeps what the next node waits for -- the critical chain P_{k+1} -> LDL^T -> X1 -> Pt -> T = Pt F -> S = F^T T ->
// Q_uu^-1 -> P_k -- and the operand prefetch; role 1 takes everything else off that chain: its own (redundant) LDL^T for
// X2, the blocks G and Minv the sweeps need, the fused backward sweep of the node before (k+1, whose matrices sit complete
// in another operand buffer) and all stores to the factor record.  Every element is computed by the same expressions as in
// the one-wave form (bit-identical results: the library is built with -ffp-contract=on).  ~7 750 -> ~4 800 cycles per node for a wave that is alone on its
// SIMD (64 satellites on a 1024-SIMD chip: the small-batch regime of BASELINE configs[1]).
__device__ __noinline__ bool riccati_factor2(const Sat &s_in, SatData &sd, Scratch &w, int lane, int role, bool keep_pt)
{
    const Sat s = uniform_view(s_in);
    const int K = s.K;
    bool good = true;
    const int sc = lane >> 3, sr = lane & 7;
    const bool sact = sr < 7;
    const int srr = (sr < 7) ? sr : 6, sr3 = (sr < 3) ? sr : 2;
    const int sink_e = s.o_sink + lane;
    const int mi = lane / 7, mj = lane - 7 * mi;
    const int xc = (lane < 7) ? lane : 6;
    // ---- role 0: operand prefetch (as in the one-wave form, three buffers) ----
    double pre[3] = {0.0, 0.0, 0.0};
    const int e1 = lane + 64, e2 = lane + 128;
    auto wx_src = [](int q) -> int {
        const int i = q / 7, j = q - 7 * i;
        return (i < 3 && j < 3) ? N_W3 + i * 3 + j : (i == j ? N_DIAG : N_ZERO);
    };
    const int wx1 = (e1 >= 91) ? wx_src(e1 - 91) : 0;
    const int src2 = (e2 < 140) ? wx_src(e2 - 91) : (e2 < OPS_IN ? N_WU + (e2 - 140) : 0);
    auto fetch = [&](int k) {
        cgf64 *stk = s.stage + (size_t)(k <= K - 2 ? k : K - 2) * MPCX_STAGE_DOUBLES;
        cgf64 *stm = s.stage + (size_t)(k >= 1 ? k - 1 : 0) * MPCX_STAGE_DOUBLES;
        cgf64 *nb = s.nb + (size_t)k * NB_N;
        cgf64 *p1 = (e1 < 70) ? stk + e1 : (e1 < 91) ? stm + e1 : nb + wx1;
        cgf64 *p2 = nb + src2;
        pre[0] = stk[lane]; pre[1] = *p1; pre[2] = *p2;
    };
    auto ops_slot = [](int e) -> int {
        if (e < 49) return (int)offsetof(StageOps, F) + 8 * ((e / 7) * FS + e % 7);
        if (e < 70) return (int)offsetof(StageOps, Bn) + 8 * (e - 49);
        if (e < 91) return (int)offsetof(StageOps, Bpm) + 8 * (e - 70);
        if (e < 140) return (int)offsetof(StageOps, G2) + 8 * (((e - 91) / 7) * FS + (e - 91) % 7);
        if (e < 149) return (int)offsetof(StageOps, Wu) + 8 * (e - 140);
        if (e < 156) return (int)offsetof(StageOps, D) + 8 * (e - 149);
        return (int)offsetof(StageOps, SX) + 8 * ((e < OPS_IN) ? e - 156 : 0);
    };
    const int slot0 = ops_slot(lane), slot1 = ops_slot(e1), slot2 = ops_slot(e2);
    auto stash = [&](StageOps &o, int k) {
        const bool dynk = (k <= K - 2);
        char *base = (char *)&o;
        *(double *)(base + slot0) = dynk ? pre[0] : 0.0;
        *(double *)(base + slot1) = ((e1 < 70) ? dynk : (e1 < 91) ? (k >= 1) : true) ? pre[1] : 0.0;
        if (e2 < OPS_IN) *(double *)(base + slot2) = (e2 < 149 || e2 >= 156 || dynk) ? pre[2] : 0.0;
    };
    if (role == 1) {                     // (the operand prefetch is the second wave's: it has the slack)
        fetch(K - 1);
        stash(w.ops[(K - 1) % 3], K - 1);
        if (lane < 49) w.ops[(K - 1) % 3].G2[(lane / 7) * FS + lane % 7] = sd.WxK[lane];
    } else {
        for (int e = lane; e < 49; e += 64) w.Pn2[K & 1][e] = 0.0;       // P_K = 0 (read as "P of node k+1" by node K-1)
        if (lane == 0) { w.zero = 0.0; w.good_flag = 1; }
    }
    WG_BARRIER();
    // ---- role 0 lane roles (P1, P5, P6: as in the one-wave form) ----
    const bool p1_bh = lane < 21, p1_wx = lane >= 32 && lane < 53;
    const int p1_e = p1_wx ? lane - 32 : (p1_bh ? lane : 0), p1_i = p1_e / 3, p1_j = p1_e - 3 * p1_i;
    const int p5_i = lane / FS, p5_j = lane - FS * p5_i;
    const bool p5b_t = lane < 6, p5b_g = lane >= 6 && lane < 36;
    const int p5b_q = p5b_g ? lane - 6 : 0;
    const int p5b_r = p5b_q / FS;
    const int p5b_j = p5b_t ? 4 + lane : p5b_q - FS * p5b_r;
    const int p5b_sa = p5b_t ? 1 : 3;
    const bool p5b_wu = p5b_g && p5b_j >= 7, p5b_qy = p5b_g && p5b_j < 7;
    int p6_i = 0, p6_j = 0;
    { int tt = lane; for (int i = 0; i < FS; ++i) { const int n = FS - i; if (tt < n) { p6_i = i; p6_j = i + tt; break; } tt -= n; } }
    const bool p6_on = lane < 55;
    const bool p6_qyy = p6_on && p6_j < 7, p6_quy = p6_on && p6_i < 7 && p6_j >= 7, p6_quu = p6_on && p6_i >= 7;
    // ---- role 1: the fused backward sweep, one node behind (same arithmetic as sweep_backward) ----
    ChanIn cur{0.0, 0.0, 0.0, 0.0};
    ChanRaw nraw{0.0, 0.0, 0.0, 0.0};
    double pnext = 0.0;
    auto sweep_node = [&](const StageOps &o, int j) {        // node j's p, qu from its complete operand buffer
        const bool dynj = (j <= K - 2);
        double sw_G[7], sw_Pt[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) { sw_G[q] = o.G[srr * 7 + q]; sw_Pt[q] = o.Pt[srr * 7 + q]; }
        const double sw_v = cur.rho + pnext;
        double tt = pnext;
#pragma unroll
        for (int q = 0; q < 7; ++q) tt += -sw_G[q] * gshfl8(sw_v, q) + sw_Pt[q] * gshfl8(cur.aff, q);
        double Acol[7], Bpmcol[7], Bhcol[7], Kgcol[3];
#pragma unroll
        for (int q = 0; q < 7; ++q) { Acol[q] = o.F[q * FS + srr]; Bpmcol[q] = o.Bpm[q * 3 + sr3]; Bhcol[q] = o.F[q * FS + 7 + sr3]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) Kgcol[q] = o.Kg[q * 7 + srr];
        if (!dynj || !sact) tt = 0.0;
        double qu = cur.gu;
#pragma unroll
        for (int q = 0; q < 7; ++q) qu += Bpmcol[q] * gshfl8(cur.gx, q) + Bhcol[q] * gshfl8(tt, q);
        if (sr >= 3 || !sact) qu = 0.0;
        double pp = cur.gx;
#pragma unroll
        for (int q = 0; q < 7; ++q) pp += Acol[q] * gshfl8(tt, q);
#pragma unroll
        for (int q = 0; q < 3; ++q) pp -= Kgcol[q] * gshfl8(qu, q);
        ustore(s.ws, sact ? s.o_ch + j * CH_N + C_P + sc * 7 + sr : sink_e, pp);
        ustore(s.ws, (sact && sr < 3) ? s.o_ch + j * CH_N + C_QU + sc * 3 + sr3 : sink_e, qu);
        pnext = sact ? pp : pnext;
        // ... and the part of node j's factor record that the first wave left in LDS: gain, Bh, Q_uu^-1
        const bool on21 = lane < 21;
        const int l21 = on21 ? lane : 0;
        const int fb = s.o_fac + j * FAC_N;
        ustore(s.ws, on21 ? fb + F_KG + lane : sink_e, o.Kg[l21]);
        ustore(s.ws, on21 ? fb + F_BH + lane : sink_e, o.F[(l21 / 3) * FS + 7 + l21 % 3]);
        ustore(s.ws, lane < 9 ? fb + F_QI + lane : sink_e, o.Qi[lane < 9 ? lane : 0]);
    };
    for (int k = K - 1; k >= 0; --k) {
        StageOps &o = w.ops[k % 3];
        const double *Pn = w.Pn2[(k + 1) & 1];
        const bool dyn = (k <= K - 2);
        if (role == 0) {
            // P1: Bh = A Bpm + Bn ; WxBp = Wx Bpm
            {
                const double dot = dotN<7>((p1_wx ? o.G2 : o.F) + p1_i * FS, 1, o.Bpm + p1_j, 3);
                const double val = p1_wx ? dot : (dyn ? o.Bn[p1_e] + dot : 0.0);
                double *dst = p1_wx ? &o.G2[p1_i * FS + 7 + p1_j] : (p1_bh ? &o.F[p1_i * FS + 7 + p1_j] : &w.sink[lane]);
                *dst = val;
            }
            double rd[7] = {0, 0, 0, 0, 0, 0, 0};
            if (dyn) {
                // P2: LDL^T of M = D + Pn in registers; P3 (this wave's half): X1 = Lt^-1 Pn, lane c < 7 owns column c
                double m[28];
#pragma unroll
                for (int i = 0, n = 0; i < 7; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j, ++n) m[n] = Pn[i * 7 + j] + (i == j ? o.D[i] : 0.0);
#pragma unroll
                for (int pp = 0; pp < 7; ++pp) {
                    const double d = m[pp * (pp + 1) / 2 + pp];
                    if (!(d > 0.0)) good = false;
                    rd[pp] = rcp_pos(d);
                    double col[7];
#pragma unroll
                    for (int i = pp + 1; i < 7; ++i) col[i] = m[i * (i + 1) / 2 + pp];
#pragma unroll
                    for (int i = pp + 1; i < 7; ++i) {
                        const double lip = col[i] * rd[pp];
#pragma unroll
                        for (int j = pp + 1; j <= i; ++j) m[i * (i + 1) / 2 + j] -= lip * col[j];
                        m[i * (i + 1) / 2 + pp] = lip;
                    }
                }
                double x[7];
#pragma unroll
                for (int pp = 0; pp < 7; ++pp) x[pp] = Pn[pp * 7 + xc];
#pragma unroll
                for (int pp = 1; pp < 7; ++pp)
#pragma unroll
                    for (int q = 0; q < pp; ++q) x[pp] -= m[pp * (pp + 1) / 2 + q] * x[q];
                {
                    double *dst = (lane < 7) ? &w.WlLi[lane] : &w.sink[lane];
                    const int st = (lane < 7) ? 14 : 0;
#pragma unroll
                    for (int pp = 0; pp < 7; ++pp) dst[pp * st] = x[pp];
                }
            }
            wsync();
            {
                // P4 (this wave's third): Pt = Pn - X1^T R X1, symmetric by construction
                const bool on = lane < 49;
                const int ci = on ? mi : 0, cj = on ? mj : 0;
                const int lo = (ci < cj) ? ci : cj, hi = (ci < cj) ? cj : ci;
                double a1 = 0.0;
#pragma unroll
                for (int l = 0; l < 7; ++l) a1 += w.WlLi[l * 14 + lo] * (rd[l] * w.WlLi[l * 14 + hi]);
                const double pt = dyn ? Pn[lo * 7 + hi] - a1 : 0.0;
                *(on ? &o.Pt[lane] : &w.sink[lane]) = pt;
                if (keep_pt) ustore(s.ws, on ? s.o_fac + k * FAC_N + F_PT + lane : sink_e, pt);
                wsync();
            }
            // P5
            w.T[p5_i * FS + p5_j] = dotN<7>(o.Pt + p5_i * 7, 1, o.F + p5_j, FS);
            {
                const double *a = p5b_t ? o.Pt + 42 : o.Bpm + p5b_r;
                const double *b = (p5b_t ? o.F : o.G2) + p5b_j;
                double acc = 0.0;
#pragma unroll
                for (int l = 0; l < 7; ++l) acc += a[l * p5b_sa] * b[l * FS];
                const double *add = p5b_wu ? &o.Wu[p5b_r * 3 + p5b_j - 7] : &w.zero;
                double *dst = p5b_t ? &w.T[6 * FS + p5b_j] : (p5b_wu ? &w.Quu[p5b_r * 3 + p5b_j - 7] : (p5b_qy ? &w.Quy[p5b_r * 7 + p5b_j] : &w.sink[lane]));
                *dst = *add + acc;
            }
            wsync();
            // P6
            {
                const double sdot = dotN<7>(o.F + p6_i, FS, w.T + p6_j, FS);
                double *dst = p6_qyy ? &w.Qyy[p6_i * 7 + p6_j] : (p6_quy ? &w.Quy[(p6_j - 7) * 7 + p6_i] : (p6_quu ? &w.Quu[(p6_i - 7) * 3 + p6_j - 7] : &w.sink[lane]));
                const double *add = p6_qyy ? &o.G2[p6_i * FS + p6_j] : dst;
                *dst = *add + sdot;
            }
            wsync();
            // P7-P9: Q_uu^-1, P_k (into the other P buffer), the gain and Q_uu^-1 into the node's operand buffer
            double Qi[9];
            if (!inv3_spd(w.Quu, Qi)) good = false;
            if (lane < 49) {
                const int lo = (mi < mj) ? mi : mj, hi = (mi < mj) ? mj : mi;
                double qi[3], qj[3];
#pragma unroll
                for (int l = 0; l < 3; ++l) { qi[l] = w.Quy[l * 7 + lo]; qj[l] = w.Quy[l * 7 + hi]; }
                double a1 = w.Qyy[lo * 7 + hi];
#pragma unroll
                for (int l = 0; l < 3; ++l) {
                    const double kj = Qi[l * 3] * qj[0] + Qi[l * 3 + 1] * qj[1] + Qi[l * 3 + 2] * qj[2];
                    a1 -= qi[l] * kj;
                }
                w.Pn2[k & 1][lane] = a1;
            }
            {
                const bool on21 = lane < 21;
                const int l21 = on21 ? lane : 0;
                const int r = l21 / 7, c = l21 - 7 * r;
                const double q0 = w.Quy[c], q1 = w.Quy[7 + c], q2 = w.Quy[14 + c];
                const double k0 = Qi[0] * q0 + Qi[1] * q1 + Qi[2] * q2, k1 = Qi[3] * q0 + Qi[4] * q1 + Qi[5] * q2,
                             k2 = Qi[6] * q0 + Qi[7] * q1 + Qi[8] * q2;
                const double kg = (r == 0) ? k0 : (r == 1 ? k1 : k2);
                *(on21 ? &o.Kg[lane] : &w.sink[lane]) = kg;
                double qv = Qi[0];
#pragma unroll
                for (int e = 1; e < 9; ++e) qv = (lane == e) ? Qi[e] : qv;
                *(lane < 9 ? &o.Qi[lane] : &w.sink[lane]) = qv;
            }
            // stiff stage terms (rare): rank-1 update of P_k, the gain and Q_uu^-1 -- on this node's LDS copies, which the
            // second wave writes to the record afterwards
            if (o.SX[SX_EX] > 0.0 || o.SX[SX_EU] > 0.0) {
                wsync();
                stiff_stage_update2(o, w, w.Pn2[k & 1], lane);
            }
            if (!__all(good) && lane == 0) w.good_flag = 0;
        } else {
            // ---- role 1 ----
            if (k >= 1) fetch(k - 1);
            nraw = chan_fetch(s, k, sc, srr, sr3);
            if (dyn) {
                double m[28], rd[7];
#pragma unroll
                for (int i = 0, n = 0; i < 7; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j, ++n) m[n] = Pn[i * 7 + j] + (i == j ? o.D[i] : 0.0);
#pragma unroll
                for (int pp = 0; pp < 7; ++pp) {
                    const double d = m[pp * (pp + 1) / 2 + pp];
                    rd[pp] = rcp_pos(d);
                    double col[7];
#pragma unroll
                    for (int i = pp + 1; i < 7; ++i) col[i] = m[i * (i + 1) / 2 + pp];
#pragma unroll
                    for (int i = pp + 1; i < 7; ++i) {
                        const double lip = col[i] * rd[pp];
#pragma unroll
                        for (int j = pp + 1; j <= i; ++j) m[i * (i + 1) / 2 + j] -= lip * col[j];
                        m[i * (i + 1) / 2 + pp] = lip;
                    }
                }
                double x[7];
#pragma unroll
                for (int pp = 0; pp < 7; ++pp) {
                    const double pv = Pn[pp * 7 + xc];
                    x[pp] = (lane < 7) ? pv : (lane - 7 == pp ? 1.0 : 0.0);
                }
#pragma unroll
                for (int pp = 1; pp < 7; ++pp)
#pragma unroll
                    for (int q = 0; q < pp; ++q) x[pp] -= m[pp * (pp + 1) / 2 + q] * x[q];
                {
                    double *dst = (lane < 14) ? &w.WlLi1[lane] : &w.sink[lane];
                    const int st = (lane < 14) ? 14 : 0;
#pragma unroll
                    for (int pp = 0; pp < 7; ++pp) dst[pp * st] = x[pp];
                }
                wsync();
                // G = X1^T R X2, Minv = X2^T R X2 into the node's operand buffer and the factor record
                const bool on = lane < 49;
                const int ci = on ? mi : 0, cj = on ? mj : 0;
                double a2 = 0.0, a3 = 0.0;
#pragma unroll
                for (int l = 0; l < 7; ++l) {
                    const double x1i = w.WlLi1[l * 14 + ci], x2i = w.WlLi1[l * 14 + 7 + ci], x2j = w.WlLi1[l * 14 + 7 + cj];
                    a2 += x1i * (rd[l] * x2j); a3 += x2i * (rd[l] * x2j);
                }
                *(on ? &o.G[lane] : &w.sink[lane]) = a2;
                *(on ? &o.Minv[lane] : &w.sink[lane]) = a3;
                const int fb = s.o_fac + k * FAC_N;
                ustore(s.ws, on ? fb + F_G + lane : sink_e, a2);
                ustore(s.ws, on ? fb + F_MINV + lane : sink_e, a3);
            } else {
                // the terminal node has no dynamics: zero blocks (as the one-wave form stores them)
                const bool on = lane < 49;
                *(on ? &o.G[lane] : &w.sink[lane]) = 0.0;
                *(on ? &o.Minv[lane] : &w.sink[lane]) = 0.0;
                const int fb = s.o_fac + k * FAC_N;
                ustore(s.ws, on ? fb + F_G + lane : sink_e, 0.0);
                ustore(s.ws, on ? fb + F_MINV + lane : sink_e, 0.0);
            }
            if (dyn) sweep_node(w.ops[(k + 1) % 3], k + 1);          // node k+1: complete since the last barrier
            // inputs of node k for its sweep in the next slot
            cur = chan_mask(nraw, sc, sr, sact);
            if (!dyn) {
                const double tg = (sc == 2) ? sd.avt[srr] : sd.ta[sc >= 3 ? sc - 3 : 0][srr];
                cur.gx = (sact && sc >= 2) ? tg : cur.gx;
                cur.rho = 0.0; cur.aff = 0.0;
            }
            if (k >= 1) stash(w.ops[(k - 1) % 3], k - 1);
        }
        WG_BARRIER();
        if (w.good_flag == 0) { good = false; break; }
    }
    if (role == 1 && good) sweep_node(w.ops[0], 0);
    WG_BARRIER();
    return good;
}

// What the first wave's driver calls: tell the second wave (parked in solve_kernel2w's command loop) to join, take role 0.
enum { CMD_FACTOR = 1, CMD_EXIT = 2 };
__device__ __forceinline__ bool riccati_factor(const Sat &s, SatData &sd, Scratch &w, int lane, bool fuse_sweep, bool keep_pt)
{
    (void)fuse_sweep;                              // (the backward sweep always rides along: it is the second wave's)
    if (lane == 0) { w.cmd = CMD_FACTOR; w.cmd_arg = keep_pt ? 1 : 0; }
    WG_BARRIER();
    return riccati_factor2(s, sd, w, lane, 0, keep_pt);
}
#else
// Backward Riccati sweep: factorisation (DESIGN.md "Solver algorithm").  Returns false on breakdown.
// With fuse_sweep the backward linear-term sweep of all 8 channels rides along: node k's p_k, qu_k are formed
// right after its matrices, while they are still in LDS (same arithmetic as sweep_backward).
__device__ __noinline__ bool riccati_factor(const Sat &s_in, SatData &sd, Scratch &w, int lane, bool fuse_sweep, bool keep_pt)
{
    const Sat s = uniform_view(s_in);   // private copy: scalar registers, not re-read after every LDS fence
    const int K = s.K;
    bool good = true;
    const int sc = lane >> 3, sr = lane & 7;
    const bool sact = fuse_sweep && sr < 7;
    const int srr = (sr < 7) ? sr : 6, sr3 = (sr < 3) ? sr : 2;
    // The fused backward sweep runs one node behind the factorisation: node k+1's sweep sits in the same straight-line
    // block as node k's LDL^T chain, so that the two dependent chains fill each other's latency gaps.
    ChanIn cur{0.0, 0.0, 0.0, 0.0};           // inputs of the node swept in this iteration (k+1)
    ChanRaw nraw{0.0, 0.0, 0.0, 0.0};         // raw inputs of node k, in flight during iteration k
    double pnext = 0.0;
    // one node of the sweep: t = p+ - G(rho + p+) + Pt aff ; qu = gu + Bpm^T gx + Bh^T t ; p = gx + A^T t - Kg^T qu
    // (written in three pieces so that the first matrix-vector product can be spread over the pivots of the LDL^T)
    double sw_G[7], sw_Pt[7], sw_v = 0.0, sw_t = 0.0;
    auto sweep_begin = [&](const StageOps &o) {
#pragma unroll
        for (int q = 0; q < 7; ++q) { sw_G[q] = o.G[srr * 7 + q]; sw_Pt[q] = o.Pt[srr * 7 + q]; }
        sw_v = cur.rho + pnext; sw_t = pnext;
    };
    auto sweep_col = [&](int q) { sw_t += -sw_G[q] * gshfl8(sw_v, q) + sw_Pt[q] * gshfl8(cur.aff, q); };
    auto sweep_finish = [&](const StageOps &o, int j, double &pp, double &qu) {
        const bool dynj = (j <= K - 2);
        double Acol[7], Bpmcol[7], Bhcol[7], Kgcol[3];
#pragma unroll
        for (int q = 0; q < 7; ++q) { Acol[q] = o.F[q * FS + srr]; Bpmcol[q] = o.Bpm[q * 3 + sr3]; Bhcol[q] = o.F[q * FS + 7 + sr3]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) Kgcol[q] = o.Kg[q * 7 + srr];
        double tt = sw_t;
        if (!dynj || !sact) tt = 0.0;
        qu = cur.gu;
#pragma unroll
        for (int q = 0; q < 7; ++q) qu += Bpmcol[q] * gshfl8(cur.gx, q) + Bhcol[q] * gshfl8(tt, q);
        if (sr >= 3 || !sact) qu = 0.0;
        pp = cur.gx;
#pragma unroll
        for (int q = 0; q < 7; ++q) pp += Acol[q] * gshfl8(tt, q);
#pragma unroll
        for (int q = 0; q < 3; ++q) pp -= Kgcol[q] * gshfl8(qu, q);
    };
    const int sink_e = s.o_sink + lane;                       // this lane's sink slot (element offset in the workspace)
    auto sweep_store = [&](int j, double pp, double qu) {
        ustore(s.ws, sact ? s.o_ch + j * CH_N + C_P + sc * 7 + sr : sink_e, pp);
        ustore(s.ws, (sact && sr < 3) ? s.o_ch + j * CH_N + C_QU + sc * 3 + sr3 : sink_e, qu);
        pnext = sact ? pp : pnext;
    };
    // operand prefetch: node k's (A, Bn | Bpm | Wx, Wu, D) -> registers -> LDS buffer.  Three branch-free loads per
    // lane: A, Bn are the head of stage record k, Bpm = B_kp of record k-1 (same offset), Wx (expanded from its compact
    // form) | Wu | D | SX from the Newton-block record; what node k does not have (no dynamics at K-1, no Bpm at 0) is
    // zeroed when stashed.
    double pre[3];
    const int e1 = lane + 64, e2 = lane + 128;
    // element q of the 7 x 7 stage Hessian in the compact Newton record: its 3x3 block entry, the common diagonal value, or
    // the record's zero
    auto wx_src = [](int q) -> int {
        const int i = q / 7, j = q - 7 * i;
        return (i < 3 && j < 3) ? N_W3 + i * 3 + j : (i == j ? N_DIAG : N_ZERO);
    };
    const int wx1 = (e1 >= 91) ? wx_src(e1 - 91) : 0;
    const int src2 = (e2 < 140) ? wx_src(e2 - 91) : (e2 < OPS_IN ? N_WU + (e2 - 140) : 0);
    auto fetch = [&](int k) {
        cgf64 *stk = s.stage + (size_t)(k <= K - 2 ? k : K - 2) * MPCX_STAGE_DOUBLES;
        cgf64 *stm = s.stage + (size_t)(k >= 1 ? k - 1 : 0) * MPCX_STAGE_DOUBLES;
        cgf64 *nb = s.nb + (size_t)k * NB_N;
        cgf64 *p1 = (e1 < 70) ? stk + e1 : (e1 < 91) ? stm + e1 : nb + wx1;
        cgf64 *p2 = nb + src2;
        pre[0] = stk[lane]; pre[1] = *p1; pre[2] = *p2;
    };
    // LDS slot (byte offset inside StageOps) of element e of the fetch order [A 49 | Bn 21 | Bpm 21 | Wx 49 | Wu 9 | D 7 | SX 8]
    auto ops_slot = [](int e) -> int {
        if (e < 49) return (int)offsetof(StageOps, F) + 8 * ((e / 7) * FS + e % 7);
        if (e < 70) return (int)offsetof(StageOps, Bn) + 8 * (e - 49);
        if (e < 91) return (int)offsetof(StageOps, Bpm) + 8 * (e - 70);
        if (e < 140) return (int)offsetof(StageOps, G2) + 8 * (((e - 91) / 7) * FS + (e - 91) % 7);
        if (e < 149) return (int)offsetof(StageOps, Wu) + 8 * (e - 140);
        if (e < 156) return (int)offsetof(StageOps, D) + 8 * (e - 149);
        return (int)offsetof(StageOps, SX) + 8 * ((e < OPS_IN) ? e - 156 : 0);
    };
    const int slot0 = ops_slot(lane), slot1 = ops_slot(e1), slot2 = ops_slot(e2);
    auto stash = [&](StageOps &o, int k) {
        const bool dynk = (k <= K - 2);
        char *base = (char *)&o;
        *(double *)(base + slot0) = dynk ? pre[0] : 0.0;
        *(double *)(base + slot1) = ((e1 < 70) ? dynk : (e1 < 91) ? (k >= 1) : true) ? pre[1] : 0.0;
        if (e2 < OPS_IN) *(double *)(base + slot2) = (e2 < 149 || e2 >= 156 || dynk) ? pre[2] : 0.0;
    };
    fetch(K - 1);
    stash(w.ops[(K - 1) & 1], K - 1);
    // (the terminal node's Hessian -- soft part, capped rank-1 terms, AL term -- is a full matrix: from SatData)
    if (lane < 49) w.ops[(K - 1) & 1].G2[(lane / 7) * FS + lane % 7] = sd.WxK[lane];
    for (int e = lane; e < 49; e += 64) w.Pn[e] = 0.0;
    if (lane == 0) w.zero = 0.0;
    WG_SYNC();
    const int mi = lane / 7, mj = lane - 7 * mi;
    const int xc = (lane < 7) ? lane : 6;                 // column of [Pn | I] this lane substitutes (lanes 0..13)
    // P1 roles: lanes 0..20 element e of Bh = A Bpm + Bn (into F), lanes 32..52 element e of Wx Bpm (into G2), one body
    const bool p1_bh = lane < 21, p1_wx = lane >= 32 && lane < 53;
    const int p1_e = p1_wx ? lane - 32 : (p1_bh ? lane : 0), p1_i = p1_e / 3, p1_j = p1_e - 3 * p1_i;
    // P5 roles: first round task lane of the 70 of T = Pt F; second round lanes 0..5 the other 6 (row 6, columns 4..9),
    // lanes 6..35 element (r, j) of Bpm^T G2 (j < 7: Quy0, j >= 7: Quu0)
    const int p5_i = lane / FS, p5_j = lane - FS * p5_i;
    const bool p5b_t = lane < 6, p5b_g = lane >= 6 && lane < 36;
    const int p5b_q = p5b_g ? lane - 6 : 0;
    const int p5b_r = p5b_q / FS;
    const int p5b_j = p5b_t ? 4 + lane : p5b_q - FS * p5b_r;
    const int p5b_sa = p5b_t ? 1 : 3;
    const bool p5b_wu = p5b_g && p5b_j >= 7, p5b_qy = p5b_g && p5b_j < 7;
    // P6 roles: lane t < 55 is entry (i, j), i <= j, of the 10 x 10 matrix S = F^T T
    int p6_i = 0, p6_j = 0;
    { int tt = lane; for (int i = 0; i < FS; ++i) { const int n = FS - i; if (tt < n) { p6_i = i; p6_j = i + tt; break; } tt -= n; } }
    const bool p6_on = lane < 55;
    const bool p6_qyy = p6_on && p6_j < 7, p6_quy = p6_on && p6_i < 7 && p6_j >= 7, p6_quu = p6_on && p6_i >= 7;
    for (int k = K - 1; k >= 0; --k) {
        StageOps &o = w.ops[k & 1];
        gf64 *fac = s.fac + (size_t)k * FAC_N;
        FT_DECL
        if (k >= 1) fetch(k - 1);
        const bool dyn = (k <= K - 2);
        if (fuse_sweep) nraw = chan_fetch(s, k, sc, srr, sr3);
        FT_MARK(0)
        // P1: Bh = A Bpm + Bn ; WxBp = Wx Bpm
        {
            const double dot = dotN<7>((p1_wx ? o.G2 : o.F) + p1_i * FS, 1, o.Bpm + p1_j, 3);
            const double val = p1_wx ? dot : (dyn ? o.Bn[p1_e] + dot : 0.0);
            double *dst = p1_wx ? &o.G2[p1_i * FS + 7 + p1_j] : (p1_bh ? &o.F[p1_i * FS + 7 + p1_j] : &w.sink[lane]);
            *dst = val;
        }
        double rd[7] = {0, 0, 0, 0, 0, 0, 0};
        double sw_p = 0.0, sw_qu = 0.0;
        if (dyn) {
            // P2: LDL^T of M = D + Pn, redundantly in the registers of every lane (broadcast LDS reads, no exchange):
            // m holds the lower triangle, the strict part ends up as Lt.  Same arithmetic as the oracle's ldl_solve7.
            double m[28];
#pragma unroll
            for (int i = 0, n = 0; i < 7; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j, ++n) m[n] = w.Pn[i * 7 + j] + (i == j ? o.D[i] : 0.0);
            // fused backward sweep of node k+1 (its matrices are still in the other operand buffer), interleaved
            // with the pivots: one column of its first matrix-vector product per pivot
            const StageOps &on = w.ops[(k + 1) & 1];
            if (fuse_sweep) sweep_begin(on);
#pragma unroll
            for (int pp = 0; pp < 7; ++pp) {
                if (fuse_sweep) sweep_col(pp);
                const double d = m[pp * (pp + 1) / 2 + pp];
                if (!(d > 0.0)) good = false;
                rd[pp] = rcp_pos(d);
                double col[7];
#pragma unroll
                for (int i = pp + 1; i < 7; ++i) col[i] = m[i * (i + 1) / 2 + pp];
#pragma unroll
                for (int i = pp + 1; i < 7; ++i) {
                    const double lip = col[i] * rd[pp];
#pragma unroll
                    for (int j = pp + 1; j <= i; ++j) m[i * (i + 1) / 2 + j] -= lip * col[j];
                    m[i * (i + 1) / 2 + pp] = lip;
                }
            }
            // P3: [X1 | X2] = Lt^-1 [Pn | I] (unit lower), lane c < 14 owns column c
            double x[7];
#pragma unroll
            for (int pp = 0; pp < 7; ++pp) {
                const double pv = w.Pn[pp * 7 + xc];
                x[pp] = (lane < 7) ? pv : (lane - 7 == pp ? 1.0 : 0.0);
            }
#pragma unroll
            for (int pp = 1; pp < 7; ++pp)
#pragma unroll
                for (int q = 0; q < pp; ++q) x[pp] -= m[pp * (pp + 1) / 2 + q] * x[q];
            {
                double *dst = (lane < 14) ? &w.WlLi[lane] : &w.sink[lane];
                const int st = (lane < 14) ? 14 : 0;
#pragma unroll
                for (int pp = 0; pp < 7; ++pp) dst[pp * st] = x[pp];
            }
            if (fuse_sweep) sweep_finish(on, k + 1, sw_p, sw_qu);
        }
        if (fuse_sweep && dyn) sweep_store(k + 1, sw_p, sw_qu);
        wsync();
        FT_MARK(1)
        {
            FT_MARK(2)
            // P4: Pt = Pn - X1^T R X1 ; G = X1^T R X2 ; Minv = X2^T R X2 with R = diag(1/d).  Pt is symmetric by
            // construction: lanes (i,j) and (j,i) evaluate the same expression in (min, max) order on a symmetric Pn.
            // Branch-free: every lane computes (idle lanes on element 0), LDS / global stores of idle lanes go to sinks;
            // the terminal node (no dynamics) stores zeros.
            const bool on = lane < 49;
            const int ci = on ? mi : 0, cj = on ? mj : 0;
            const int lo = (ci < cj) ? ci : cj, hi = (ci < cj) ? cj : ci;
            double a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int l = 0; l < 7; ++l) {
                const double x1i = w.WlLi[l * 14 + ci], x1lo = w.WlLi[l * 14 + lo], x1hi = w.WlLi[l * 14 + hi];
                const double x2i = w.WlLi[l * 14 + 7 + ci], x2j = w.WlLi[l * 14 + 7 + cj];
                a1 += x1lo * (rd[l] * x1hi);
                a2 += x1i * (rd[l] * x2j); a3 += x2i * (rd[l] * x2j);
            }
            const double pt = dyn ? w.Pn[lo * 7 + hi] - a1 : 0.0;
            a2 = dyn ? a2 : 0.0; a3 = dyn ? a3 : 0.0;
            *(on ? &o.Pt[lane] : &w.sink[lane]) = pt;
            *(on ? &o.G[lane] : &w.sink[lane]) = a2;
            *(on ? &o.Minv[lane] : &w.sink[lane]) = a3;
            const int fb = s.o_fac + k * FAC_N;
            ustore(s.ws, on ? fb + F_G + lane : sink_e, a2);
            ustore(s.ws, on ? fb + F_MINV + lane : sink_e, a3);
            if (keep_pt) ustore(s.ws, on ? fb + F_PT + lane : sink_e, pt);
            wsync();
        }
        FT_MARK(3)
        // P5: T = Pt F (70 dot products of one pattern: 64 in the first round, 6 in the second) and
        //     [Quy0 | Quu0 - Wu] = Bpm^T G2 (30, second round, lanes 6..35)
        w.T[p5_i * FS + p5_j] = dotN<7>(o.Pt + p5_i * 7, 1, o.F + p5_j, FS);
        {
            const double *a = p5b_t ? o.Pt + 42 : o.Bpm + p5b_r;
            const double *b = (p5b_t ? o.F : o.G2) + p5b_j;
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < 7; ++l) acc += a[l * p5b_sa] * b[l * FS];
            const double *add = p5b_wu ? &o.Wu[p5b_r * 3 + p5b_j - 7] : &w.zero;
            double *dst = p5b_t ? &w.T[6 * FS + p5b_j] : (p5b_wu ? &w.Quu[p5b_r * 3 + p5b_j - 7] : (p5b_qy ? &w.Quy[p5b_r * 7 + p5b_j] : &w.sink[lane]));
            *dst = *add + acc;
        }
        wsync();
        FT_MARK(4)
        // P6: the upper triangle of S = F^T T (55 dot products of one pattern): Qyy = Wx + A^T Pt A (upper part only, read
        //     back through (min, max)), Quy += Bh^T Pt A, Quu += Bh^T Pt Bh (upper part: all the 3x3 inverse reads)
        {
            const double sdot = dotN<7>(o.F + p6_i, FS, w.T + p6_j, FS);
            double *dst = p6_qyy ? &w.Qyy[p6_i * 7 + p6_j] : (p6_quy ? &w.Quy[(p6_j - 7) * 7 + p6_i] : (p6_quu ? &w.Quu[(p6_i - 7) * 3 + p6_j - 7] : &w.sink[lane]));
            const double *add = p6_qyy ? &o.G2[p6_i * FS + p6_j] : dst;
            *dst = *add + sdot;
        }
        wsync();
        FT_MARK(5)
        // P7-P9: every lane inverts the 3x3 itself; P_k = sym(Qyy - Quy^T Qi Quy) straight from its own two columns of
        // Quy (no exchange of the gain on the way); the gain Kg = Qi Quy goes to LDS / the factor record for the sweeps
        double Qi[9];
        if (!inv3_spd(w.Quu, Qi)) good = false;
        if (lane < 49) {
            const int lo = (mi < mj) ? mi : mj, hi = (mi < mj) ? mj : mi;
            double qi[3], qj[3];
#pragma unroll
            for (int l = 0; l < 3; ++l) { qi[l] = w.Quy[l * 7 + lo]; qj[l] = w.Quy[l * 7 + hi]; }
            // (qi, qj) = columns (min, max) of Quy: lanes (i,j) and (j,i) evaluate the same expression, P_k is
            // symmetric by construction; Qyy is symmetrised through the same (min, max) read (rounding-level asymmetry
            // of A^T (Pt A) otherwise)
            double a1 = w.Qyy[lo * 7 + hi];
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const double kj = Qi[l * 3] * qj[0] + Qi[l * 3 + 1] * qj[1] + Qi[l * 3 + 2] * qj[2];     // Kg(l, hi)
                a1 -= qi[l] * kj;
            }
            w.Pn[lane] = a1;
        }
        // (Qi is indexed with constants only and picked by selects: a register array indexed by a lane-dependent value
        //  is placed in scratch memory, and its store / load pair would sit behind an s_waitcnt vmcnt(0) in every node)
        {
            // gain Kg = Qi Quy (lanes 0..20) and the node's record entries Kg, Bh, Qi: branch-free (see ustore)
            const bool on21 = lane < 21;
            const int l21 = on21 ? lane : 0;
            const int r = l21 / 7, c = l21 - 7 * r;
            const double q0 = w.Quy[c], q1 = w.Quy[7 + c], q2 = w.Quy[14 + c];
            const double k0 = Qi[0] * q0 + Qi[1] * q1 + Qi[2] * q2, k1 = Qi[3] * q0 + Qi[4] * q1 + Qi[5] * q2,
                         k2 = Qi[6] * q0 + Qi[7] * q1 + Qi[8] * q2;
            const double kg = (r == 0) ? k0 : (r == 1 ? k1 : k2);
            *(on21 ? &o.Kg[lane] : &w.sink[lane]) = kg;
            const int fb = s.o_fac + k * FAC_N;
            ustore(s.ws, on21 ? fb + F_KG + lane : sink_e, kg);
            ustore(s.ws, on21 ? fb + F_BH + lane : sink_e, o.F[(l21 / 3) * FS + 7 + l21 % 3]);
            double qv = Qi[0];
#pragma unroll
            for (int e = 1; e < 9; ++e) qv = (lane == e) ? Qi[e] : qv;
            ustore(s.ws, lane < 9 ? fb + F_QI + lane : sink_e, qv);
        }
        // stiff stage terms (rare: an active r_min plane / radius or thrust ball late in the iteration): rank-1 update of
        // what was just written; out of line so that the common path keeps its register allocation
        if (o.SX[SX_EX] > 0.0 || o.SX[SX_EU] > 0.0) stiff_stage_update(o, w, fac, lane);
        FT_MARK(6)
        FT_MARK(7)
        // inputs of node k for its sweep in the next iteration (the terminal node's come from LDS)
        if (fuse_sweep) {
            // (the terminal node's inputs come through the same branch-free fetch: a conditional load here would make
            //  the first use of `cur` in the next node wait with vmcnt(0), i.e. for that node's whole prefetch)
            cur = chan_mask(nraw, sc, sr, sact);
            if (!dyn) {
                const double tg = (sc == 2) ? sd.avt[srr] : sd.ta[sc >= 3 ? sc - 3 : 0][srr];
                cur.gx = (sact && sc >= 2) ? tg : cur.gx;
                cur.rho = 0.0; cur.aff = 0.0;
            }
        }
        FT_MARK(8)
        // a breakdown (every lane sees the same pivots) ends the sweep here: the caller retries with a larger delta_w
        if (!__all(good)) break;
        if (k >= 1) stash(w.ops[(k - 1) & 1], k - 1);
        wsync();
        FT_MARK(9)
    }
    if (fuse_sweep && __all(good)) {             // the sweep of node 0
        double sw_p, sw_qu;
        sweep_begin(w.ops[0]);
#pragma unroll
        for (int q = 0; q < 7; ++q) sweep_col(q);
        sweep_finish(w.ops[0], 0, sw_p, sw_qu);
        sweep_store(0, sw_p, sw_qu);
    }
    WG_SYNC();
    return __all(good);
}

#endif

// ---- linear-term sweeps: lane group c = channel, lane r = component ------------------------------
// Stage matrices are staged through a double-buffered LDS copy (prefetched one node ahead); each lane reads
// its own rows/columns into registers and the channel vectors travel by ds_bpermute inside the 8-lane group,
// so a node costs one barrier (the buffer swap).
struct SweepPre { double v[6]; };

// (PT: the backward sweep of a refinement pass reads Pt; the forward sweep does not, and outside refinement the
//  factorisation does not even write it -- its 49 doubles, three of the record's 12.5 cache lines, are not fetched then)
template <bool PT>
__device__ __forceinline__ void sweep_fetch_mats(const Sat &s, int k, int lane, SweepPre &pre)
{
    const int K = s.K;
    cgf64 *fac = s.fac + (size_t)k * FAC_N;
#pragma unroll
    for (int q = 0; q < 2; ++q) pre.v[q] = fac[lane + 64 * q];
    pre.v[2] = fac[(PT || lane + 128 < F_PT) ? lane + 128 : F_PT - 1];
    pre.v[3] = PT ? fac[(lane + 192 < FAC_N) ? lane + 192 : FAC_N - 1] : 0.0;
    // A: head of stage record k; Bpm: B_kp of record k-1; D: Newton record k (what a node lacks is zeroed when stashed)
    cgf64 *stk = s.stage + (size_t)(k <= K - 2 ? k : K - 2) * MPCX_STAGE_DOUBLES;
    cgf64 *stm = s.stage + (size_t)(k >= 1 ? k - 1 : 0) * MPCX_STAGE_DOUBLES;
    cgf64 *nb = s.nb + (size_t)k * NB_N;
    pre.v[4] = stk[lane];
    cgf64 *p5 = (lane < 21) ? stm + 70 + lane : nb + N_D + ((lane < 28) ? lane - 21 : 0);
    pre.v[5] = *p5;
}

__device__ __forceinline__ void sweep_stash_mats(double *f, int K, int k, int lane, const SweepPre &pre)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) f[lane + 64 * q] = pre.v[q];
    f[F_A + lane] = (k <= K - 2) ? pre.v[4] : 0.0;
    f[F_BPM + lane] = ((lane < 21) ? (k >= 1) : (k <= K - 2)) ? pre.v[5] : 0.0;
}

// Backward sweep for channels [c0, c1): p_k and qu_k stored per channel.
__device__ __noinline__ void sweep_backward(const Sat &s_in, SatData &sd, Scratch &w, int c0, int c1, int lane)
{
    const Sat s = uniform_view(s_in);
    const int K = s.K;
    const int c = lane >> 3, r = lane & 7;
    const bool act = (c >= c0 && c < c1) && r < 7;
    const int rr = (r < 7) ? r : 6, r3 = (r < 3) ? r : 2;
    SweepPre pre;
    sweep_fetch_mats<true>(s, K - 1, lane, pre);
    sweep_stash_mats(w.flat[(K - 1) & 1], K, K - 1, lane, pre);
    ChanIn cur = chan_inputs(s, sd, K - 1, c, r, act), nxt = cur;
    double pnext = 0.0;
    WG_SYNC();
    for (int k = K - 1; k >= 0; --k) {
        const double *f = w.flat[k & 1];
        if (k >= 1) { sweep_fetch_mats<true>(s, k - 1, lane, pre); nxt = chan_inputs(s, sd, k - 1, c, r, act); }
        const bool dyn = (k <= K - 2);
        double Grow[7], Ptrow[7], Acol[7], Bpmcol[7], Bhcol[7], Kgcol[3];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            Grow[q] = f[F_G + rr * 7 + q]; Ptrow[q] = f[F_PT + rr * 7 + q]; Acol[q] = f[F_A + q * 7 + rr];
            Bpmcol[q] = f[F_BPM + q * 3 + r3]; Bhcol[q] = f[F_BH + q * 3 + r3];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) Kgcol[q] = f[F_KG + q * 7 + rr];
        const double v = cur.rho + pnext;
        double t = pnext;
#pragma unroll
        for (int q = 0; q < 7; ++q) t += -Grow[q] * gshfl8(v, q) + Ptrow[q] * gshfl8(cur.aff, q);
        if (!dyn || !act) t = 0.0;
        double qu = cur.gu;
#pragma unroll
        for (int q = 0; q < 7; ++q) qu += Bpmcol[q] * gshfl8(cur.gx, q) + Bhcol[q] * gshfl8(t, q);
        if (r >= 3 || !act) qu = 0.0;
        double p = cur.gx;
#pragma unroll
        for (int q = 0; q < 7; ++q) p += Acol[q] * gshfl8(t, q);
#pragma unroll
        for (int q = 0; q < 3; ++q) p -= Kgcol[q] * gshfl8(qu, q);
        if (act) {
            gf64 *ch = s.ch + (size_t)k * CH_N;
            ch[C_P + c * 7 + r] = p;
            if (r < 3) ch[C_QU + c * 3 + r] = qu;
            pnext = p;
        }
        if (k >= 1) sweep_stash_mats(w.flat[(k - 1) & 1], K, k - 1, lane, pre);
        cur = nxt;
        wsync();
    }
    WG_SYNC();
}

// Forward sweep for channels [c0, c1): stores each channel's trajectory (x, u, nu, lam) per node and
// accumulates the border coefficients (Sigma.lam, x_K).
__device__ __noinline__ void sweep_forward(const Sat &s_in, SatData &sd, Scratch &w, int c0, int c1, int lane)
{
    const Sat s = uniform_view(s_in);
    const int K = s.K;
    const int c = lane >> 3, r = lane & 7;
    const bool act = (c >= c0 && c < c1) && r < 7;
    const int rr = (r < 7) ? r : 6, r3 = (r < 3) ? r : 2;
    SweepPre pre;
    sweep_fetch_mats<false>(s, 0, lane, pre);
    sweep_stash_mats(w.flat[0], K, 0, lane, pre);
    ChanIn cur = chan_inputs(s, sd, 0, c, r, act);
    ChanRaw nraw{0.0, 0.0, 0.0, 0.0};
    // qu_k, p_{k+1} and Sigma_k of the lane's channel / component: branch-free loads, masked after arrival
    auto load_pq = [&](int k, double &qu, double &pn, double &sg) {
        cgf64 *ch = s.ch + (size_t)k * CH_N;
        qu = ch[C_QU + c * 3 + r3];
        pn = (ch + (k <= K - 2 ? CH_N : 0))[C_P + c * 7 + rr];
        sg = s.Sig(k <= K - 2 ? k : K - 2)[rr];
    };
    double quc, pnc, sgc, qun = 0.0, pnn = 0.0, sgn = 0.0;
    load_pq(0, quc, pnc, sgc);
    if (!(act && r < 3)) quc = 0.0;
    if (!(act && K >= 2)) pnc = 0.0;
    double y = 0.0, siglam = 0.0;
    WG_SYNC();
    for (int k = 0; k < K; ++k) {
        const double *f = w.flat[k & 1];
        FT_DECL
        if (k + 1 < K) { sweep_fetch_mats<false>(s, k + 1, lane, pre); nraw = chan_fetch(s, k + 1, c, rr, r3); load_pq(k + 1, qun, pnn, sgn); }
        const bool dyn = (k <= K - 2);
        FT_MARK(10)
        double Kgrow[7], Arow[7], Gcol[7], Mrow[7], Qirow[3], Bpmrow[3], Bhrow[3];
#pragma unroll
        for (int q = 0; q < 7; ++q) { Kgrow[q] = f[F_KG + r3 * 7 + q]; Arow[q] = f[F_A + rr * 7 + q]; Gcol[q] = f[F_G + q * 7 + rr]; Mrow[q] = f[F_MINV + rr * 7 + q]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) { Qirow[q] = f[F_QI + r3 * 3 + q]; Bpmrow[q] = f[F_BPM + rr * 3 + q]; Bhrow[q] = f[F_BH + rr * 3 + q]; }
        const double Dr = f[F_D + rr];
        double u = 0.0;
#pragma unroll
        for (int q = 0; q < 7; ++q) u -= Kgrow[q] * gshfl8(y, q);
#pragma unroll
        for (int q = 0; q < 3; ++q) u -= Qirow[q] * gshfl8(quc, q);
        if (r >= 3 || !act) u = 0.0;
        double x = y, yh = cur.aff;
#pragma unroll
        for (int q = 0; q < 3; ++q) { const double uq = gshfl8(u, q); x += Bpmrow[q] * uq; yh += Bhrow[q] * uq; }
#pragma unroll
        for (int q = 0; q < 7; ++q) yh += Arow[q] * gshfl8(y, q);
        if (!dyn || !act) yh = 0.0;
        FT_MARK(11)
        const double wv = cur.rho + pnc;
        double nu = 0.0;
#pragma unroll
        for (int q = 0; q < 7; ++q) nu -= Gcol[q] * gshfl8(yh, q) + Mrow[q] * gshfl8(wv, q);
        FT_MARK(12)
        {
            // the channel's trajectory at this node: branch-free stores (see ustore)
            const int tb = s.o_traj + (k * NCH + c) * TR_N, sink_t = s.o_sink + lane;
            const double lam = Dr * nu + cur.rho;
            const bool ad = act && dyn;
            ustore(s.ws, act ? tb + T_X + r : sink_t, x);
            ustore(s.ws, (act && r < 3) ? tb + T_U + r3 : sink_t, u);
            ustore(s.ws, ad ? tb + T_NU + r : sink_t, nu);
            if (act && k == K - 1) sd.xK[c][r] = x;
            siglam += ad ? sgc * lam : 0.0;
            y = ad ? yh + nu : y;
        }
        FT_MARK(13)
        if (k + 1 < K) {
            sweep_stash_mats(w.flat[(k + 1) & 1], K, k + 1, lane, pre);
            cur = chan_mask(nraw, c, r, act);
            quc = (act && r < 3) ? qun : 0.0; pnc = (act && k + 1 <= K - 2) ? pnn : 0.0; sgc = sgn;
        }
        wsync();
        FT_MARK(14)
    }
    siglam += __shfl_xor(siglam, 1, 8);
    siglam += __shfl_xor(siglam, 2, 8);
    siglam += __shfl_xor(siglam, 4, 8);
    if (act && r == 0) sd.siglam[c] = siglam;
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
    gf64 *dr = wave_uniform(s.dr);
    cgf64 *it = wave_uniform((cgf64 *)s.it), *traj = wave_uniform((cgf64 *)s.traj);
    // Rounds of 32 nodes.  The trajectories are read in their own order (node, channel, component: contiguous), the
    // combination goes through LDS (stg: the recursion's scratch, [component][node of the round]) and leaves in the
    // direction's field-major order, consecutive lanes on consecutive nodes: written straight from the reading lanes
    // the direction was 8-byte stores scattered over as many cache lines as lanes.
    for (int k0 = 0; k0 < K; k0 += 32) {
        const int nk = (K - k0 < 32) ? K - k0 : 32;
        const int n = nk * TR_N;
        for (int e0 = 0; e0 < n; e0 += 256) {
            double v[4];
            int slot[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
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
            for (int q = 0; q < 4; ++q) if (slot[q] >= 0) stg[slot[q]] = v[q];
        }
        WG_SYNC();
        for (int e = lane; e < DIR_N * 32; e += 64) {
            const int i = e >> 5, kl = e & 31, k = k0 + kl;
            const bool act = kl < nk && !(k == K - 1 && i >= T_NU);
            const int off = (i < T_U) ? I_X + i : (i < T_NU ? I_U + (i - T_U) : (i < T_LAM ? I_NU + (i - T_NU) : I_LAM + (i - T_LAM)));
            const int dst = off * KP + (kl < nk ? k : k0);
            // starting value: -lam for the multiplier part of the first solve, the current direction when refining
            const double cur = first ? it[dst] : dr[dst];
            const double base = first ? ((i >= T_LAM) ? -cur : 0.0) : cur;
            // multiplier part: D_k nu_k + rho_k from the combined nu
            const int kc = (kl < nk) ? k : k0, j = (i >= T_LAM) ? i - T_LAM : 0;
            const double Dj = s.nb[(size_t)kc * NB_N + N_D + j], rj = s.ch[(size_t)kc * CH_N + C_RHS + R_RHO + j];
            const double val = (i >= T_LAM) ? fma(Dj, stg[(T_NU + j) * CMB_LD + kl], rj) : stg[(i < T_LAM ? i : 0) * CMB_LD + kl];
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

// The bordered system ("Solver algorithm"): unknowns dtf, the multiplier of the vt row and one zeta per
// terminal barrier term with excess weight.  Kept in the order (vt, zeta_1..5, dtf) = channels (2, 3..7, 1): in that
// order the matrix is symmetric, its leading 6x6 block (constraint-type rows, zeta rows in their 1/wex form) is
// negative definite and the Schur complement of dtf is positive exactly when the reduced KKT matrix has the inertia
// of a convex problem.  An LDL^T without pivoting in that order (stable for such quasi-definite matrices) therefore
// serves three purposes: the solve, the inertia check ipopt gets from its linear solver -- every constraint pivot
// negative, dtf's positive (Sylvester); a wrong inertia is reported like a breakdown and regularised by delta_w,
// without it the iteration can alternate between a descent and an ascent direction in tf on short-arc references --
// and it runs redundantly in the registers of every lane.  A zeta without excess weight is decoupled (pivot -1).
__device__ __forceinline__ int border_channel(int q) { return q == NBD - 1 ? 1 : 2 + q; }
// Row p < NBD-1 of the border: 1/wex of a zeta row with excess weight, 0 for a zeta row without (decoupled), and `eq`
// set for the tangential equality (row 0 of the exact variant; in the convex variant row 0 is the zeta row of the pair)
__device__ __forceinline__ double border_iw(const SatData &sd, int p, bool &eq)
{
    eq = (p == 0) && !sd.linvt;
    const double wex = (p == 0) ? (sd.linvt ? sd.w_vt - sd.gam : 0.0) : sd.tw[p - 1] - sd.twin[p - 1];
    return (!eq && wex > 0.0) ? 1.0 / wex : 0.0;
}

__device__ __noinline__ bool border_factor(SatData &sd, int lane)
{
    // assembly, one lane per entry: rows 0..5 measure a . x_K of the unit channels, row 6 the tf stationarity
    if (lane < NBD * NBD) {
        const int p = lane / NBD, q = lane - NBD * p;
        const int c = border_channel(q);
        double v;
        if (p < NBD - 1) {
            const double *a = (p == 0) ? sd.avt : sd.ta[p - 1];
            v = 0.0;
#pragma unroll
            for (int l = 0; l < 7; ++l) v += a[l] * sd.xK[c][l];
        } else v = (q == NBD - 1 ? sd.Wtf : 0.0) - sd.siglam[c];
        if (sd.fixed_tf && (p == NBD - 1 || q == NBD - 1)) v = (p == q) ? 1.0 : 0.0;      // dtf = 0: out of the border
        sd.Mb[p][q] = v; sd.Sb[p][q] = v;    // Sb keeps the matrix for the refinement step of border_solve
    }
    WG_SYNC();
    double S[NBD][NBD];
#pragma unroll
    for (int p = 0; p < NBD; ++p)
#pragma unroll
        for (int q = 0; q < NBD; ++q) S[p][q] = sd.Mb[p][q];
#pragma unroll
    for (int p = 0; p < NBD - 1; ++p) {
        bool eq;
        const double iw = border_iw(sd, p, eq);
        const bool on = iw > 0.0;
        if (eq) continue;
#pragma unroll
        for (int q = 0; q < NBD; ++q) if (!on && q != p) { S[p][q] = 0.0; S[q][p] = 0.0; }
        S[p][p] = on ? S[p][p] - iw : -1.0;
    }
    bool ok = true;
    double rd[NBD];
#pragma unroll
    for (int p = 0; p < NBD; ++p) {
        const double d = S[p][p];
        if (p == NBD - 1) { if (!(d > 0.0)) ok = false; } else if (!(d < 0.0)) ok = false;
        rd[p] = 1.0 / d;
#pragma unroll
        for (int i = p + 1; i < NBD; ++i) {
            const double m = S[i][p] * rd[p];
#pragma unroll
            for (int j = p + 1; j < NBD; ++j) S[i][j] -= m * S[p][j];
            S[i][p] = m;                                  // unit lower factor
        }
    }
    WG_SYNC();
    if (lane == 0) {                                       // factors for border_solve (also of the refinement passes)
#pragma unroll
        for (int p = 0; p < NBD; ++p) {
            sd.Mb[p][p] = rd[p];
#pragma unroll
            for (int i = p + 1; i < NBD; ++i) sd.Mb[i][p] = S[i][p];
        }
    }
    WG_SYNC();
    return ok;
}

// Right-hand side of the border system from channel 0, then L D L^T solve with the stored factors and one step of
// iterative refinement against the matrix itself (every lane, in registers); sd.sol in channel order (dtf, vt
// multiplier, zeta_1..5).
__device__ __noinline__ void border_solve(SatData &sd, double gtf_rhs, double rvt_rhs, const double *gex, int lane)
{
    double rb[NBD], x[NBD], r[NBD];
#pragma unroll
    for (int p = 0; p < NBD - 1; ++p) {
        const double *a = (p == 0) ? sd.avt : sd.ta[p - 1];
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < 7; ++l) acc += a[l] * sd.xK[0][l];
        rb[p] = -acc;
    }
    rb[0] += rvt_rhs;
    double iw[NBD - 1];
    bool eqr[NBD - 1];
#pragma unroll
    for (int p = 0; p < NBD - 1; ++p) {
        iw[p] = border_iw(sd, p, eqr[p]);
        if (p >= 1) rb[p] = (iw[p] > 0.0) ? rb[p] - gex[p - 1] * iw[p] : 0.0;
        else if (!eqr[0] && !(iw[0] > 0.0)) rb[0] = 0.0;       // convex variant, pair without excess weight: decoupled
    }
    rb[NBD - 1] = sd.fixed_tf ? 0.0 : -gtf_rhs + sd.siglam[0];
    auto ldl_solve = [&](double (&v)[NBD]) {
#pragma unroll
        for (int p = 0; p < NBD; ++p)
#pragma unroll
            for (int i = p + 1; i < NBD; ++i) v[i] -= sd.Mb[i][p] * v[p];
#pragma unroll
        for (int p = 0; p < NBD; ++p) v[p] *= sd.Mb[p][p];
#pragma unroll
        for (int p = NBD - 1; p >= 0; --p)
#pragma unroll
            for (int i = p + 1; i < NBD; ++i) v[p] -= sd.Mb[i][p] * v[i];
    };
#pragma unroll
    for (int p = 0; p < NBD; ++p) x[p] = rb[p];
    ldl_solve(x);
    // r = rb - S x with S rebuilt from the kept matrix (zeta rows in their 1/wex form, decoupled ones as -1)
#pragma unroll
    for (int p = 0; p < NBD; ++p) {
        double acc = rb[p];
#pragma unroll
        for (int q = 0; q < NBD; ++q) {
            double s = sd.Sb[p][q];
            const int pc = p < NBD - 1 ? p : 0, qc = q < NBD - 1 ? q : 0;
            const bool zp = p < NBD - 1 && !eqr[pc], zq = q < NBD - 1 && !eqr[qc];      // zeta rows / columns
            const bool offp = zp && !(iw[pc] > 0.0), offq = zq && !(iw[qc] > 0.0);
            if (p == q && zp) s = offp ? -1.0 : s - iw[pc];
            else if (offp || offq) s = 0.0;
            acc -= s * x[q];
        }
        r[p] = acc;
    }
    ldl_solve(r);
#pragma unroll
    for (int p = 0; p < NBD; ++p) x[p] += r[p];
    WG_SYNC();
    if (lane == 0) {
        sd.sol[0] = x[NBD - 1];
#pragma unroll
        for (int p = 0; p < NBD - 1; ++p) sd.sol[1 + p] = x[p];
    }
    WG_SYNC();
}


// ---- launch-wide reductions of the shared-tf mode (solve_shared_kernel: every workgroup resident, cooperative launch) ----
// A reduction is also the barrier between two phases of the lock-step iteration: every workgroup publishes GR_N values,
// waits until all S have arrived, and folds the S contributions in a fixed order (same result on every workgroup, the
// same from run to run).  Slots alternate between two rings: a workgroup can be at most one phase ahead of the slowest.
constexpr int GR_SUM = 6, GR_MAX = 3, GR_MIN = 3, GR_N = GR_SUM + GR_MAX + GR_MIN;
constexpr long kSpinMax = 20000000;        // ~ seconds: a workgroup that never arrives aborts the launch instead of hanging it
struct GridSync {
    double *red;
    int32_t *arrive, *abort_flag;
    int S, blk, phase;
    bool aborted;
};

__device__ __noinline__ void grid_reduce(GridSync &g, double (&v)[GR_N], int lane)
{
    double *slot = g.red + ((size_t)(g.phase & 1) * g.S + g.blk) * GR_N;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < GR_N; ++j) __hip_atomic_store(slot + j, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    WG_SYNC();
    if (lane == 0) {
        __hip_atomic_fetch_add(g.arrive, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const int target = (g.phase + 1) * g.S;
        long spins = 0;
        while (__hip_atomic_load(g.arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(g.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (++spins > kSpinMax) { __hip_atomic_store(g.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    WG_SYNC();
    __threadfence();
    if (__hip_atomic_load(g.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) g.aborted = true;
    const double *base = g.red + (size_t)(g.phase & 1) * g.S * GR_N;
#pragma unroll
    for (int j = 0; j < GR_N; ++j) {
        double acc = (j < GR_SUM) ? 0.0 : (j < GR_SUM + GR_MAX ? -1e300 : 1e300);
        for (int b = lane; b < g.S; b += 64) {
            const double x = __hip_atomic_load(base + (size_t)b * GR_N + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc = (j < GR_SUM) ? acc + x : (j < GR_SUM + GR_MAX ? fmax(acc, x) : fmin(acc, x));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double x = __shfl_xor(acc, o, 64);
            acc = (j < GR_SUM) ? acc + x : (j < GR_SUM + GR_MAX ? fmax(acc, x) : fmin(acc, x));
        }
        v[j] = acc;
    }
    ++g.phase;
    WG_SYNC();
}
__device__ __forceinline__ void gr_clear(double (&v)[GR_N])
{
#pragma unroll
    for (int j = 0; j < GR_N; ++j) v[j] = (j < GR_SUM) ? 0.0 : (j < GR_SUM + GR_MAX ? -1e300 : 1e300);
}

// The border of a satellite whose tf is shared by the launch.  Same matrix as border_factor builds, the dtf row carrying
// only this satellite's share of the tf row (W_tf = 2 w_tr, its -Sigma.lambda terms): the six constraint-type pivots are
// eliminated here, the seventh -- the Schur complement of dtf -- is this satellite's ADDEND to the launch's tf pivot
// (sd.tS) and is neither tested nor inverted.  Returns false if a constraint pivot has the wrong sign.
__device__ __noinline__ bool border_factor_shared(SatData &sd, int lane)
{
    if (lane < NBD * NBD) {
        const int p = lane / NBD, q = lane - NBD * p;
        const int c = border_channel(q);
        double v;
        if (p < NBD - 1) {
            const double *a = (p == 0) ? sd.avt : sd.ta[p - 1];
            v = 0.0;
#pragma unroll
            for (int l = 0; l < 7; ++l) v += a[l] * sd.xK[c][l];
        } else v = (q == NBD - 1 ? sd.Wtf : 0.0) - sd.siglam[c];
        sd.Mb[p][q] = v; sd.Sb[p][q] = v;
    }
    WG_SYNC();
    double S[NBD][NBD];
#pragma unroll
    for (int p = 0; p < NBD; ++p)
#pragma unroll
        for (int q = 0; q < NBD; ++q) S[p][q] = sd.Mb[p][q];
#pragma unroll
    for (int p = 0; p < NBD - 1; ++p) {
        bool eq;
        const double iw = border_iw(sd, p, eq);
        const bool on = iw > 0.0;
        if (eq) continue;
#pragma unroll
        for (int q = 0; q < NBD; ++q) if (!on && q != p) { S[p][q] = 0.0; S[q][p] = 0.0; }
        S[p][p] = on ? S[p][p] - iw : -1.0;
    }
    bool ok = true;
    double rd[NBD];
#pragma unroll
    for (int p = 0; p < NBD - 1; ++p) {
        const double d = S[p][p];
        if (!(d < 0.0)) ok = false;
        rd[p] = 1.0 / d;
#pragma unroll
        for (int i = p + 1; i < NBD; ++i) {
            const double m = S[i][p] * rd[p];
#pragma unroll
            for (int j = p + 1; j < NBD; ++j) S[i][j] -= m * S[p][j];
            S[i][p] = m;
        }
    }
    WG_SYNC();
    if (lane == 0) {
#pragma unroll
        for (int p = 0; p < NBD - 1; ++p) {
            sd.Mb[p][p] = rd[p];
#pragma unroll
            for (int i = p + 1; i < NBD; ++i) sd.Mb[i][p] = S[i][p];
        }
        sd.tS = S[NBD - 1][NBD - 1];
    }
    WG_SYNC();
    return ok;
}

// Solve with the shared tf: forward substitution here, dtf = (sum of the satellites' right-hand-side shares + the launch's
// own part r_glob) / (sum of their pivot shares + W_glob) across the launch, back substitution here; then one step of
// iterative refinement of the whole bordered system, its tf row again summed across the launch.  `fail`: this satellite
// cannot contribute (breakdown upstream).  Returns false -- on every workgroup alike -- if any satellite failed or the
// launch's tf pivot is not positive (wrong inertia: regularise).
__device__ __noinline__ bool border_solve_shared(SatData &sd, GridSync &g, double gtf_share, double rvt_rhs, const double *gex,
                                                 double W_glob, double r_glob, bool fail, int lane)
{
    double rb[NBD], v[NBD], x[NBD];
#pragma unroll
    for (int p = 0; p < NBD - 1; ++p) {
        const double *a = (p == 0) ? sd.avt : sd.ta[p - 1];
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < 7; ++l) acc += a[l] * sd.xK[0][l];
        rb[p] = -acc;
    }
    rb[0] += rvt_rhs;
    double iw[NBD - 1];
    bool eqr[NBD - 1];
#pragma unroll
    for (int p = 0; p < NBD - 1; ++p) {
        iw[p] = border_iw(sd, p, eqr[p]);
        if (p >= 1) rb[p] = (iw[p] > 0.0) ? rb[p] - gex[p - 1] * iw[p] : 0.0;
        else if (!eqr[0] && !(iw[0] > 0.0)) rb[0] = 0.0;
    }
    rb[NBD - 1] = -gtf_share + sd.siglam[0];
    auto forward = [&](double (&w)[NBD]) {
#pragma unroll
        for (int p = 0; p < NBD - 1; ++p)
#pragma unroll
            for (int i = p + 1; i < NBD; ++i) w[i] -= sd.Mb[i][p] * w[p];
    };
    auto backward = [&](double (&w)[NBD], double dtf) {      // w: forward-substituted; on return the solution
        w[NBD - 1] = dtf;
#pragma unroll
        for (int p = NBD - 2; p >= 0; --p) {
            double acc = w[p] * sd.Mb[p][p];
#pragma unroll
            for (int i = p + 1; i < NBD; ++i) acc -= sd.Mb[i][p] * w[i];
            w[p] = acc;
        }
    };
#pragma unroll
    for (int p = 0; p < NBD; ++p) v[p] = rb[p];
    forward(v);
    double gr[GR_N];
    gr_clear(gr);
    gr[0] = fail ? 0.0 : sd.tS; gr[1] = fail ? 0.0 : v[NBD - 1]; gr[2] = fail ? 1.0 : 0.0;
    grid_reduce(g, gr, lane);
    const double D = gr[0] + W_glob;
    if (g.aborted || gr[2] > 0.0 || !(D > 0.0)) return false;
    const double dtf = (gr[1] + r_glob) / D;
#pragma unroll
    for (int p = 0; p < NBD; ++p) x[p] = v[p];
    backward(x, dtf);
    // refinement: r = rb - S x with S rebuilt from the kept matrix (zeta rows in their 1/wex form, decoupled ones as -1);
    // the tf row's residual is summed across the launch together with the launch's own part r_glob - W_glob dtf
    double r[NBD];
#pragma unroll
    for (int p = 0; p < NBD; ++p) {
        double acc = rb[p];
#pragma unroll
        for (int q = 0; q < NBD; ++q) {
            double sv = sd.Sb[p][q];
            const int pc = p < NBD - 1 ? p : 0, qc = q < NBD - 1 ? q : 0;
            const bool zp = p < NBD - 1 && !eqr[pc], zq = q < NBD - 1 && !eqr[qc];
            const bool offp = zp && !(iw[pc] > 0.0), offq = zq && !(iw[qc] > 0.0);
            if (p == q && zp) sv = offp ? -1.0 : sv - iw[pc];
            else if (offp || offq) sv = 0.0;
            acc -= sv * x[q];
        }
        r[p] = acc;
    }
    forward(r);
    gr_clear(gr);
    gr[1] = r[NBD - 1];
    grid_reduce(g, gr, lane);
    if (g.aborted) return false;
    const double ddtf = (gr[1] + (r_glob - W_glob * dtf)) / D;
    backward(r, ddtf);
#pragma unroll
    for (int p = 0; p < NBD; ++p) x[p] += r[p];
    WG_SYNC();
    if (lane == 0) {
        sd.sol[0] = x[NBD - 1];
#pragma unroll
        for (int p = 0; p < NBD - 1; ++p) sd.sol[1 + p] = x[p];
    }
    WG_SYNC();
    return true;
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
__device__ __noinline__ void reduced_residual(const Sat &s, SatData &sd, double *stg, int lane, double &gtf_rhs, double &rvt_rhs, double *gex)
{
    const int K = s.K;
    double gtf_part = 0.0;
    const double dtf = s.drg[G_TF];
    // terminal-node completion terms (the rank-1 terms and the AL shift): from the direction at node K-1, known up front
    double gin[NTERM], rvt_x;
    {
        const auto dK = s.drn(K - 1);
        double av = 0.0;
        for (int i = 0; i < 7; ++i) av += sd.avt[i] * dK[I_X + i];
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
            for (int i = 0; i < 7; ++i) adx += sd.ta[t][i] * dK[I_X + i];
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
        cgf64 *nb = s.nb + (size_t)k * NB_N;
        const auto ns = s.nsn(k);
        const auto p = s.itn(k), d = s.drn(k);
        double lt[7], ltm[7];     // total multipliers lam + dlam of rows k and k-1
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            lt[i] = (k <= K - 2) ? p[I_LAM + i] + d[I_LAM + i] : 0.0;
            ltm[i] = (k >= 1) ? p.node(-1)[I_LAM + i] + d.node(-1)[I_LAM + i] : 0.0;
        }
        double gx[7], gu[3];
        if (k >= 1) {
            if (k == K - 1) {
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    double acc = sd.gxKsoft[i] + ltm[i];
#pragma unroll
                    for (int j = 0; j < 7; ++j) acc += sd.WxKsoft[i * 7 + j] * d[I_X + j];
                    gx[i] = acc;
                }
            } else {
                // stage Hessian in its compact form: the 3x3 position block, the common diagonal value elsewhere (the
                // entries left out are exact zeros: same sums as with the full matrix)
                const double dg = nb[N_DIAG];
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    double acc = ns[NS_GX + i] + ltm[i];
                    if (i < 3) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) acc += nb[N_W3 + i * 3 + j] * d[I_X + j];
                    } else acc += dg * d[I_X + i];
                    gx[i] = acc;
                }
            }
            if (k == K - 1) {
                const double lvt = s.itg[G_LVT] + s.drg[G_LVT];
                for (int i = 0; i < 7; ++i) gx[i] += sd.avt[i] * lvt;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 7; ++i) gx[i] = 0.0;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double acc = ns[NS_GU + i];
#pragma unroll
            for (int j = 0; j < 3; ++j) acc += nb[N_WU + i * 3 + j] * d[I_U + j];
            gu[i] = acc;
        }
        {
            // the stiff stage terms' excess weight, which the blocks N_W3 / N_WU do not carry (newton_blocks)
            const double ex_x = nb[N_SX + SX_EX], ex_u = nb[N_SX + SX_EU];
            const double a0 = nb[N_SX + SX_A], a1 = nb[N_SX + SX_A + 1], a2 = nb[N_SX + SX_A + 2];
            const double c0 = nb[N_SX + SX_CU], c1 = nb[N_SX + SX_CU + 1], c2 = nb[N_SX + SX_CU + 2];
            const double px = ex_x * (a0 * d[I_X] + a1 * d[I_X + 1] + a2 * d[I_X + 2]);
            const double pu = ex_u * (c0 * d[I_U] + c1 * d[I_U + 1] + c2 * d[I_U + 2]);
            if (k >= 1 && k <= K - 2) { gx[0] += px * a0; gx[1] += px * a1; gx[2] += px * a2; }
            gu[0] += pu * c0; gu[1] += pu * c1; gu[2] += pu * c2;
        }
        if (k >= 1) {
            const auto Bp = s.Bpt(k - 1);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < 7; ++i) acc += Bp[i * 3 + j] * ltm[i];
                gu[j] -= acc;
            }
        }
        if (k <= K - 2) {
            const auto A = s.At(k), Bn = s.Bnt(k), Bp = s.Bpt(k), Sg = s.Sigt(k);
            const auto dn = d.node(1);
            if (k >= 1) {
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < 7; ++i) acc += A[i * 7 + j] * lt[i];
                    gx[j] -= acc;
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < 7; ++i) acc += Bn[i * 3 + j] * lt[i];
                gu[j] -= acc;
            }
            double sl = 0.0;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                rec[R_RHO + i] = ns[NS_RHO + i] + ns[NS_D + i] * d[I_NU + i] - lt[i];
                double acc = dn[I_X + i] - Sg[i] * dtf - d[I_NU + i];
#pragma unroll
                for (int j = 0; j < 7; ++j) acc -= A[i * 7 + j] * d[I_X + j];
#pragma unroll
                for (int j = 0; j < 3; ++j) acc -= Bn[i * 3 + j] * d[I_U + j] + Bp[i * 3 + j] * dn[I_U + j];
                rec[R_AFF + i] = -ns[NS_E + i] - acc;
                sl += Sg[i] * lt[i];
            }
            gtf_part -= sl;
        }
        if (k == K - 1) {
            // terminal-node completion: the rank-1 terms and the AL shift; its rho / aff slots are zero as in the first record
            for (int t = 0; t < NTERM; ++t)
                for (int i = 0; i < 7; ++i) gx[i] += gin[t] * sd.ta[t][i];
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
          gf64 *ch = s.ch + (size_t)k0 * CH_N + C_RHS;
          for (int e = lane; e < nr; e += 64) { const int kl = e / RHS_N, i = e - kl * RHS_N; ch[(size_t)kl * CH_N + i] = stg[kl * RHS_LD + i]; }
      }
      WG_SYNC();
    }
    gtf_rhs = sd.gtf + sd.Wtf * dtf + wave_sum(gtf_part);
    WG_SYNC();
}

// Right-hand side of the first solve of an iteration: direction 0, total multipliers 0, i.e. the Newton blocks
// themselves (what reduced_residual returns for d = (0, -lam, -lam_vt)).  Lane k writes node k's record.
__device__ __forceinline__ void first_rhs_scalars(const SatData &sd, double &gtf_rhs, double &rvt_rhs, double *gex)
{
    gtf_rhs = sd.gtf;
    rvt_rhs = sd.linvt ? -sd.gh_vt / sd.w_vt : -sd.cv;       // (convex variant: the pair's zeta row at the zero direction)
    for (int t = 0; t < NTERM; ++t) {
        const double share = (sd.tw[t] > 0.0) ? sd.twin[t] / sd.tw[t] : 1.0;
        gex[t] = sd.tgh[t] * (1.0 - share);         // = wex * gh / w: the zeta row's residual at the zero direction, times wex
    }
}

// The fraction-to-the-boundary step of the direction and the finite check on it.  The directions of the eliminated pairs
// (dt, ds, dz by back-substitution: pair_dir, l1_dir) are formed here only to be measured against their variables; they are
// not stored -- every trial evaluation forms them again (eval_residual).  The handful of terminal / tf pairs live in the
// global part of the direction record, as before.
__device__ __noinline__ double finish_direction(const Sat &s, SatData &sd, double mu, double tau, int lane, bool &finite)
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
        // chunk 0 (both halves compute; the step limit and the finite flag are idempotent): the ball pairs
        double x[7], dx[7];
        {
            double u[3], du[3], bs[6];
#pragma unroll
            for (int i = 0; i < 7; ++i) { x[i] = p[I_X + i]; dx[i] = d[I_X + i]; }
#pragma unroll
            for (int i = 0; i < 3; ++i) { u[i] = p[I_U + i]; du[i] = d[I_U + i]; }
#pragma unroll
            for (int i = 0; i < 6; ++i) bs[i] = p[I_SU + i];
#pragma unroll
            for (int i = 0; i < 7; ++i) CHK(dx[i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) CHK(du[i]);
            const double rb0 = rb[0], rb1 = rb[1], rb2 = rb[2];
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
        CHUNK_END
        // four rounds: component i = 4*half + r of the eliminated t and of the two L1 slack pairs
        {
            const bool dyn = (k <= K - 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int iv = 4 * half + r;
                const bool valid = (iv < 7) && dyn;
                const int i = (iv < 7) ? iv : 6;
                const double nu = p[I_NU + i], tt = p[I_T + i], stp = p[I_STP + i], ztp = p[I_ZTP + i];
                const double stn = p[I_STN + i], ztn = p[I_ZTN + i], dnu = d[I_NU + i], dlam = d[I_LAM + i];
                const L1Dir q = l1_dir(nu, tt, stp, ztp, stn, ztn, dnu, mu, w_nu);
                if (valid) {
                    CHK(dnu); CHK(dlam); CHK(q.dt);
                    LIM(stp, q.dstp); LIM(ztp, q.dztp);
                    LIM(stn, q.dstn); LIM(ztn, q.dztn);
                }
                CHUNK_END
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
    finite = (wave_max(bad) == 0.0);
    WG_SYNC();
    return amax;
}

#ifndef MPCX_TWO_WAVE
// Launch order: satellites sorted by the previous solve's iteration count, longest first (counting sort, one block).
// The order inside one count is whatever the atomics give; the solver's results do not depend on the order.
__global__ __launch_bounds__(1024) void launch_order_kernel(int S, const int32_t *prev_iters, int32_t *order)
{
    __shared__ int hist[256], base[256];
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < S; i += 1024) atomicAdd(&hist[min(max(prev_iters[i], 0), 255)], 1);
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int key = 255; key >= 0; --key) { base[key] = acc; acc += hist[key]; }
    }
    // identity first: every slot holds a valid satellite whatever the scatter below leaves unwritten
    for (int i = tid; i < S; i += 1024) order[i] = i;
    __syncthreads();
    // (prev_iters is written by the previous solve on the SAME stream, include/mpcx.h; the clamp keeps a scatter past
    // the table impossible even if a caller breaks that rule and the two passes see different counts)
    for (int i = tid; i < S; i += 1024) order[min(atomicAdd(&base[min(max(prev_iters[i], 0), 255)], 1), S - 1)] = i;
}

// The predictor the launch order sorts by: the largest iteration count of the satellite's last kPredHist solves.  A launch
// ends with its slowest workgroup, and with about two satellites per workgroup slot one long satellite that starts late
// costs its whole length: a satellite that needed many iterations in ANY of the last few solves is started early (an early
// start costs nothing if it turns out short); the last count alone forgets it as soon as the problem changes a little
// (successive MPC steps, the two SCP iterations of a step, the benchmark's rotating variants).
constexpr int kPredHist = 8;
__global__ void update_prediction_kernel(int S, const int32_t *iters, int32_t *hist, int32_t *pred, int slot, int n_valid)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    hist[(size_t)slot * S + i] = iters[i];
    int m = 0;
    for (int h = 0; h < n_valid; ++h) m = max(m, hist[(size_t)h * S + i]);
    pred[i] = m;
}

// a satellite whose discretisation failed reports that code instead of the solver's
__global__ void merge_status_kernel(int S, const int32_t *dstat, int32_t *status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S && dstat[i] != 0) status[i] = dstat[i];
}
static inline void merge_status_kernel_launch(int S, const int32_t *dstat, int32_t *status, hipStream_t st)
{
    hipLaunchKernelGGL(merge_status_kernel, dim3((S + 255) / 256), dim3(256), 0, st, S, dstat, status);
}

#endif  // !MPCX_TWO_WAVE

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
    s.itg = ws; ws += GL_N;
    s.drg = ws; ws += GL_N;
    s.itgB = ws; ws += GL_N;
    s.sink = ws;
    // (offsets as integers computed from the layout, not as pointer differences: the compiler would fold base + (sink -
    //  base) back into a second pointer and emit a branch with one store per path)
    s.o_fac = (int)(KP * (3 * IT_N + NS_N + MPCX_STAGE_DOUBLES + 3) + K * NB_N);
    s.o_ch = s.o_fac + K * FAC_N; s.o_traj = s.o_ch + K * CH_N; s.o_sink = s.o_traj + K * NCH * TR_N + 3 * GL_N;
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
    const int K = a.Ks ? a.Ks[sat] : Kmax;        // (wave-uniform: one satellite per workgroup)
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
    Sat s = sat_view(a, sat, slot, K, Kmax);
    const int KP = s.KP;
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
    bool pushed = false;
    for (int k = lane; k < K; k += 64) {
        double x[7], u[3];
        for (int i = 0; i < 7; ++i) x[i] = s.xbar[(size_t)i * s.ldk + k];
        for (int i = 0; i < 3; ++i) u[i] = s.ubar[(size_t)i * s.ldk + k];
        const double rn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        const auto rb = s.rbn(k);
        for (int i = 0; i < 3; ++i) rb[i] = x[i] / rn;           // optimizer.py:129-130
        const auto p = s.itn(k), d = s.drn(k);
        for (int i = 0; i < IT_N; ++i) { p[i] = 0.0; d[i] = 0.0; }
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
    bool clean = !SHARED && !sd.fixed_tf && !__any(pushed);      // (fixed-tf solves feed a host root search with their g_tf: left as they were)
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
    const double mu0 = clean ? kMuInitClean : kMuInit;
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

    double mu = mu0, dw_last = 0.0;            // mu: this iteration's complementarity target
    // (shared tf: the counts of the whole launch -- S satellites without their own tf rows plus tf's two range inequalities)
    const int nzc = SHARED ? a.S * n_ineq(K, sd.nT, 1) + 2 : n_ineq(K, sd.nT, sd.fixed_tf);
    const int nlc = (SHARED ? a.S : 1) * (7 * (K - 1) + (sd.nT == 6 ? 1 : 0));
    // shared tf: slacks / multipliers of 0 <= tf <= tf_max, their trial values, and the launch's part of the tf row
    double gs[2] = {0.0, 0.0}, gz[2] = {0.0, 0.0}, gst[2] = {0.0, 0.0}, gzt[2] = {0.0, 0.0}, gds[2] = {0.0, 0.0}, gdz[2] = {0.0, 0.0};
    if (SHARED) {
        const double tf = sd.tfbar;
        gs[0] = fmax(-(-tf - sd.b_tf[0]), kBoundPush * fmax(1.0, fabs(sd.b_tf[0]))); gz[0] = kMuInit / gs[0];
        gs[1] = fmax(-(tf - sd.b_tf[1]), kBoundPush * fmax(1.0, fabs(sd.b_tf[1]))); gz[1] = kMuInit / gs[1];
    }
    const double b_tf2[2] = {sd.b_tf[0], sd.b_tf[1]};
    int n_acc = 0, status = MPCX_ST_MAXITER, it_count = 0, n_reg = 0, first_reg = -1;
    // second safeguard of the adaptive barrier rule (the first is the kMuErr bound below): after kFbN consecutive accepted
    // steps shorter than kFbAlpha -- the iterate is jammed against its bounds -- mu is lifted to kFbBoost * mean(s z) and
    // follows ipopt's monotone Fiacco-McCormick rule from then on.  kFbN = 8: benchmark problems at K = 100 take up to seven
    // short regularised steps in a row and recover by themselves in 16 / 25 iterations (the monotone rule: 32 / 41).
    bool mono = false;
    int n_small = 0;
    double E0 = 0.0;
    // residual of the start point; afterwards the accepted trial of the line search is the next iteration's evaluation
    // (sq in its mu = 0 form: it serves E_0 and, for any mu, the line search's ||F_mu||)
    ResAcc r0;
    PT_END(6)
    PT_BEGIN
    eval_residual<false>(s, sd, 0.0, 0.0, 0.0, lane, r0);
    PT_END(0)
    if (SHARED) shared_fold(*gsync, r0, sd.tfbar, b_tf2, gs, gz, 0.0, lane);
    for (int iter = 0;; ++iter) {
        it_count = iter;
        E0 = scaled_error_n(r0, nzc, nlc, 0.0);
        if (SHARED && gsync->aborted) { status = MPCX_ST_NUMERIC; break; }
        if (!(E0 == E0) || !(E0 < 1e300)) { status = MPCX_ST_NUMERIC; break; }
        if (E0 <= o.tol) { status = MPCX_ST_OK; break; }
        n_acc = (E0 <= o.acc_tol) ? n_acc + 1 : 0;
        if (n_acc >= o.acc_iter) { status = MPCX_ST_ACCEPTABLE; break; }
        if (iter >= o.max_iter) { status = (E0 <= o.acc_tol) ? MPCX_ST_ACCEPTABLE : MPCX_ST_MAXITER; break; }
        // adaptive barrier parameter: a fixed fraction of the iterate's mean complementarity ("Solver algorithm")
        const double mu_cur = r0.prod_sum / (double)nzc;
        if (!mono && n_small >= kFbN) {
            mono = true;
            mu = fmax(o.tol / 10.0, fmin(kMuInit, kFbBoost * mu_cur));
        }
        // (never below kMuErr * E_0: the mean complementarity may collapse while the iterate is still infeasible)
        if (!mono) mu = fmax(fmax(clean ? fmin(kSigma * mu_cur, mu_cur * sqrt(mu_cur)) : kSigma * mu_cur, o.tol / 10.0), kMuErr * E0);
        else {
            // mu moves on only when the barrier problem is solved to E_mu <= 10 mu: mu <- max(tol/10, min(0.2 mu, mu^1.5))
            for (int lv = 0; lv < 64 && mu > o.tol / 10.0 && scaled_error_n(r0, nzc, nlc, mu) <= 10.0 * mu; ++lv)
                mu = fmax(o.tol / 10.0, fmin(0.2 * mu, mu * sqrt(mu)));
        }
        // Newton direction, with Hessian regularisation retries on breakdown
        bool have_dir = false;
        double delta_w = 0.0, alpha = 1.0;
#ifdef MPCX_ITER_LOG
        int fail_mask = 0;     // decimal digits: factor, border, finite-check failures of this iteration
#endif
        const double tau = fmax(0.99, 1.0 - mu);
        // Hessian regularisation on breakdown follows ipopt's inertia-correction schedule: 0 first, then a third of
        // the last value that worked (1e-4 the first time), growing by 8 (by 100 until some value has worked), up to 1e40
        while (!have_dir && delta_w <= kDwMax) {
            PT_BEGIN
            newton_blocks<false>(s, sd, (double *)&w, mu, delta_w, lane);
            PT_END(1)
            double gtf_rhs, rvt_rhs, gex[NTERM];
            first_rhs_scalars(sd, gtf_rhs, rvt_rhs, gex);    // (the node records of the first right-hand side: newton_blocks)
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
                if (passes > 1) newton_blocks<true>(s, sd, (double *)&w, mu, delta_w, lane);
                bool okl = riccati_factor(s, sd, w, lane, true, passes > 1);     // (a local breakdown is reported through the border's reduction)
                bool ok = true;
                for (int pass = 0; pass < passes && ok; ++pass) {
                    if (okl && pass > 0) { reduced_residual(s, sd, (double *)&w, lane, gtf_rhs, rvt_rhs, gex); sweep_backward(s, sd, w, 0, 1, lane); }
                    if (okl) {
                        sweep_forward(s, sd, w, 0, (pass == 0) ? NCH : 1, lane);
                        if (pass == 0) okl = border_factor_shared(sd, lane);
                    }
                    const double dtf_cur = (pass == 0) ? 0.0 : s.drg[G_TF];
                    ok = border_solve_shared(sd, g, gtf_rhs, rvt_rhs, gex, W_glob, -(g_glob + W_glob * dtf_cur), !okl, lane);
                    if (!ok) break;
                    combine_channels(s, sd, (double *)&w, lane, pass == 0);
                }
                if (ok) {
                    bool fin = true;
                    alpha = finish_direction(s, sd, mu, tau, lane, fin);
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
                if (ok) have_dir = true;
                else if (g.aborted) break;
                else if (delta_w == 0.0) delta_w = (dw_last == 0.0) ? kDwFirst : fmax(kDwMin, dw_last / 3.0);
                else delta_w *= (dw_last == 0.0) ? 100.0 : 8.0;
                continue;
            }
            // iterative refinement only once a barrier weight (terminal rank-1 terms, stage balls and planes, the tf
            // bounds) is stiff enough to cost digits
            double twmax = sd.sigmax;
            for (int t = 0; t < NTERM; ++t) twmax = fmax(twmax, sd.tw[t]);
            const int passes = 1 + ((delta_w == 0.0 && twmax > kRefineTw) ? o.n_refine : 0);
            if (passes > 1) newton_blocks<true>(s, sd, (double *)&w, mu, delta_w, lane);      // (the scalars reduced_residual reads)
            PT_BEGIN
            bool ok = riccati_factor(s, sd, w, lane, true, passes > 1);   // factorisation + backward sweep of all 8 channels
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
                        reduced_residual(s, sd, (double *)&w, lane, gtf_rhs, rvt_rhs, gex);
                        PT_END(5)
                    }
                    // pass 0: all 8 channels (right-hand side + the 7 border columns); refinement: channel 0 only
                    const int c1 = (pass == 0) ? NCH : 1;
                    if (pass > 0) {
                        PT_BEGIN
                        sweep_backward(s, sd, w, 0, c1, lane);
                        PT_END(3)
                    }
                    PT_BEGIN
                    sweep_forward(s, sd, w, 0, c1, lane);
                    if (pass == 0) ok = border_factor(sd, lane);
                    PT_END(4)
#ifdef MPCX_ITER_LOG
                    if (!ok) fail_mask += 100;
#endif
                    if (!ok) break;
                    PT_BEGIN
                    border_solve(sd, gtf_rhs, rvt_rhs, gex, lane);
#if defined(MPCX_ITER_LOG) && defined(MPCX_LOG_IT)
                    // diagnostic build only: the border system of one chosen iteration into this satellite's NU block
                    if (iter == MPCX_LOG_IT && pass == 0 && lane == 0) {
                        double *lg = a.NU + (size_t)sat * 7 * Kmax; int n = 0;
                        for (int j = 0; j < NBD; ++j) lg[n++] = sd.sol[j];
                        for (int j = 0; j < NTERM; ++j) lg[n++] = sd.tw[j];
                        for (int j = 0; j < NTERM; ++j) lg[n++] = sd.twin[j];
                        for (int j = 0; j < NCH; ++j) lg[n++] = sd.siglam[j];
                        for (int c = 0; c < NCH; ++c) for (int j = 0; j < 7; ++j) lg[n++] = sd.xK[c][j];
                        lg[n++] = gtf_rhs; lg[n++] = rvt_rhs;
                        for (int j = 0; j < NTERM; ++j) lg[n++] = gex[j];
                        lg[n++] = sd.Wtf; lg[n++] = sd.gam; lg[n++] = delta_w;
                    }
#endif
                    combine_channels(s, sd, (double *)&w, lane, pass == 0);
                    PT_END(7)
                }
            }
            if (ok) {
                // dt, ds, dz, the fraction-to-the-boundary step and the finite check on the direction
                PT_BEGIN
                alpha = finish_direction(s, sd, mu, tau, lane, ok);
                PT_END(8)
#ifdef MPCX_ITER_LOG
                if (!ok) fail_mask += 10000;
#endif
            }
            if (ok) have_dir = true;
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
        const double rn0 = sqrt(fmax(0.0, r0.sq - 2.0 * mu * r0.prod_sum + (double)nzc * mu * mu));
        // every trial is evaluated as the iterate it would become (slack reset and multiplier safeguard applied) and
        // left in the second iterate buffer
        const double mu_clip = fmax(mu, mu_cur);
        ResAcc rt;
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
        bool have_trial = false;
        for (int ls = 0; ls < 30; ++ls) {
            if (0.5 * alpha < kAlphaFloor) break;      // a rejection could not shorten the step any more: take it
            PT_BEGIN
            eval_residual<true>(s, sd, alpha, mu, mu_clip, lane, rt);
            PT_END(10)
            if (SHARED) shared_trial();
            const bool dec = sqrt(rt.sq) <= (1.0 - 1e-4 * alpha) * rn0;
            const bool cen = rt.prod_min >= kGammaNbhd * fmin(mu, rt.prod_sum / (double)nzc);
#ifdef MPCX_ITER_LOG
            // diagnostic build only: the first trial's margins into this satellite's U block
            if (ls == 0 && lane == 0 && 3 * iter + 2 < 3 * K) { double *lg = a.U + (size_t)sat * 3 * Kmax + 3 * iter; lg[0] = alpha; lg[1] = sqrt(rt.sq) / rn0; lg[2] = rt.prod_min / (kGammaNbhd * fmin(mu, rt.prod_sum / (double)nzc)); }
#endif
            if (dec && cen) { have_trial = true; break; }
            alpha *= 0.5;
        }
        if (!have_trial) {                              // the step taken untested
            PT_BEGIN
            eval_residual<true>(s, sd, alpha, mu, mu_clip, lane, rt);
            PT_END(9)
            if (SHARED) shared_trial();
        }
#ifdef MPCX_ITER_LOG
        // diagnostic build only: iteration log (mu, E0, accepted step, regularisation) into this satellite's X block
        if (lane == 0 && 5 * iter + 4 < 7 * K) { double *lg = a.X + (size_t)sat * 7 * Kmax + 5 * iter; lg[0] = mu; lg[1] = E0; lg[2] = alpha; lg[3] = delta_w; lg[4] = (double)fail_mask; }
#endif
        n_small = (alpha < kFbAlpha) ? n_small + 1 : 0;
        // accept: the candidate becomes the iterate, its residual (sq back in the mu = 0 form) the next iteration's
        { gf64 *q = s.it; s.it = s.itB; s.itB = q; q = s.itg; s.itg = s.itgB; s.itgB = q; }
        if (SHARED) { gs[0] = gst[0]; gs[1] = gst[1]; gz[0] = gzt[0]; gz[1] = gzt[1]; }
        r0 = rt;
        r0.sq = rt.sq + 2.0 * mu * rt.prod_sum - (double)nzc * mu * mu;
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
#endif
    }
}

// The two kernels' LDS working set: ONE pair of module-scope objects, so that it sits at the same LDS address in both and
// the out-of-line phase functions (which take it by reference) keep addressing it with compile-time offsets -- with a
// pair per kernel the addresses reach them as run-time pointers (measured: solve_kernel 6.85 -> 8.4 ms at S4096).
__shared__ SatData g_sd;
__shared__ Scratch g_w;

#ifndef MPCX_TWO_WAVE
// Persistent workgroups: the launch has as many single-wave workgroups as the device holds at once (or S, if fewer), each
// takes satellites off a counter until none is left, in launch order (longest first when the previous solve's iteration
// counts are known).  A workgroup keeps ONE workspace slot for all its satellites: the solver's working set is
// slots x 206 KB whatever the batch size (8192 satellites: 0.44 GB instead of 1.8 GB), and a slot's lines are rewritten
// by the next satellite while they are still cached instead of being written back as dead data.
__global__ __launch_bounds__(64, MPCX_SOLVE_WAVES) void solve_kernel(SolveArgs a)
{
    SatData &sd = g_sd;
    Scratch &w = g_w;
    __shared__ int next_item;
    const int lane = threadIdx.x;
    for (;;) {
        if (lane == 0) next_item = atomicAdd(a.counter, 1);
        __syncthreads();
        const int b = __builtin_amdgcn_readfirstlane(next_item);
        __syncthreads();
        if (b >= a.S) return;
        // (an entry outside [0, S) can only come from a caller that broke the same-stream rule of include/mpcx.h: never an
        //  out-of-bounds satellite)
        int sat = a.order ? a.order[b] : b;
        if ((unsigned)sat >= (unsigned)a.S) sat = b;
        solve_satellite<false>(a, sat, (int)blockIdx.x, sd, w, lane);
        __syncthreads();
    }
}

// Shared final time: one workgroup per satellite, all resident (cooperative launch), one lock-step iteration.
__global__ __launch_bounds__(64, MPCX_SOLVE_WAVES) void solve_shared_kernel(SolveArgs a)
{
    SatData &sd = g_sd;
    Scratch &w = g_w;
    const int lane = threadIdx.x;
    GridSync g{a.red, a.arrive, a.abort_flag, a.S, (int)blockIdx.x, 0, false};
    solve_satellite<true>(a, (int)blockIdx.x, (int)blockIdx.x, sd, w, lane, &g);
}

}  // namespace MPCX_NS

using namespace MPCX_NS;

int mpcx2w_launch(const void *args, size_t args_bytes, int blocks, hipStream_t stream);      // solve2w.hip
#ifndef MPCX_TWO_WAVE_MAX
#define MPCX_TWO_WAVE_MAX 1024      // two waves per satellite pay up to one satellite per SIMD (profiles/r03/batch_size_sweep.txt)
#endif
constexpr int kTwoWaveMax = MPCX_TWO_WAVE_MAX;
constexpr int kCounterRing = 64;    // work-queue counters per context: solves in flight at once on different streams

static SolveOpts to_dev_opts(const mpcx_solve_opts *o)
{
    SolveOpts d;
    d.min_mass = o->min_mass; d.u_max = o->u_max; d.r_min = o->r_min; d.r_max = o->r_max; d.eps_r = o->eps_r;
    d.eps_vr = o->eps_vr; d.eps_vn = o->eps_vn; d.eps_vt = o->eps_vt; d.tf_max = o->tf_max; d.w_nu = o->w_nu; d.w_tr = o->w_tr;
    d.tol = o->tol; d.acc_tol = o->acceptable_tol; d.max_iter = o->max_iter; d.acc_iter = o->acceptable_iter;
    d.n_refine = o->n_refine; d.linvt = (o->flags & MPCX_SOLVE_LINEAR_VT) ? 1 : 0;
    d.fixed_tf = (o->flags & MPCX_SOLVE_FIXED_TF) ? 1 : 0; d.shared_tf = (o->flags & MPCX_SOLVE_SHARED_TF) ? 1 : 0;
    return d;
}


namespace MPCX_NS {
// Diagnostic export of what solve_kernel builds before its first iteration: the terminal inequality rows a_j . x_K <= b_j
// (build_terminal: Optimizer.get_constraint_terms, optimizer.py:80-170, as consumed by the rules :398-403, 406-446,
// 471-489, 351-352) and the relaxed scalar bounds.  Same device function, same lane, same LDS struct as in the solve.
__global__ __launch_bounds__(64) void constraint_terms_kernel(int S, int K, const double *xbar, const double *consts, const double *r_des,
                                                              SolveOpts o, double *aT, double *bT, double *scal)
{
    __shared__ SatData sd;
    const int sat = blockIdx.x, lane = threadIdx.x;
    if (sat >= S) return;
    if (lane == 0) {
        double xK[7], x0[3];
        for (int i = 0; i < 7; ++i) xK[i] = xbar[(size_t)sat * 7 * K + (size_t)i * K + K - 1];
        for (int i = 0; i < 3; ++i) x0[i] = xbar[(size_t)sat * 7 * K + (size_t)i * K];
        build_terminal(xK, consts[(size_t)sat * MPCX_NCONST + MPCX_C_MU], r_des[sat], o, sd);
        sd.infeas = structural_violation(x0, K, sd);
    }
    __syncthreads();
    if (lane < 56) aT[(size_t)sat * 56 + lane] = sd.aT[lane / 7][lane % 7];
    if (lane < 8) bT[(size_t)sat * 8 + lane] = sd.bT[lane];
    if (lane == 0) {
        double *q = scal + (size_t)sat * MPCX_NTERM_SCALARS;
        q[0] = sd.b_u; q[1] = sd.b_rmax; q[2] = sd.b_rmin; q[3] = sd.b_rfmax; q[4] = sd.b_tf[0]; q[5] = sd.b_tf[1];
        q[6] = sd.vt_des; q[7] = sd.infeas;
    }
}
}  // namespace MPCX_NS

extern "C" int mpcx_constraint_terms_dev(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *consts,
                                         const double *r_des, const mpcx_solve_opts *opts, double *aT, double *bT,
                                         double *scalars, void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 2 || !opts || !xbar || !consts || !r_des || !aT || !bT || !scalars)
        return ctx_fail(ctx, MPCX_E_BADARG, "constraint_terms: need S>=1, K>=2, options and all arrays");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(mpcx::constraint_terms_kernel, dim3(S), dim3(64), 0, (hipStream_t)stream, S, K, xbar, consts, r_des,
                       to_dev_opts(opts), aT, bT, scalars);
    MPCX_HIP(ctx, hipGetLastError());
    return MPCX_OK;
}

extern "C" int mpcx_constraint_terms(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *consts,
                                     const double *r_des, const mpcx_solve_opts *opts, double *aT, double *bT, double *scalars)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 2 || !opts) return ctx_fail(ctx, MPCX_E_BADARG, "constraint_terms: need S>=1, K>=2 and options");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    DeviceArena ar(ctx);
    double *dx = ar.upload(xbar, (size_t)S * 7 * K), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST), *drd = ar.upload(r_des, S);
    double *da = ar.alloc<double>((size_t)S * 56), *db = ar.alloc<double>((size_t)S * 8), *ds = ar.alloc<double>((size_t)S * MPCX_NTERM_SCALARS);
    if (ar.failed()) return ar.code();
    int rc = mpcx_constraint_terms_dev(ctx, S, K, dx, dc, drd, opts, da, db, ds, ctx->stream);
    if (rc) return rc;
    ar.download(aT, da, (size_t)S * 56); ar.download(bT, db, (size_t)S * 8); ar.download(scalars, ds, (size_t)S * MPCX_NTERM_SCALARS);
    return ar.finish();
}

extern "C" int mpcx_solve_regularised_dev(mpcx_ctx *ctx, int S, int32_t *out, void *stream)
{
    if (!ctx || !out) return MPCX_E_BADARG;
    if (S < 1 || S != ctx->nreg_S || !ctx->nreg) return ctx_fail(ctx, MPCX_E_BADARG, "solve_regularised: S must be the batch size of the last solve on this context");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    MPCX_HIP(ctx, hipMemcpyAsync(out, ctx->nreg, (size_t)S * 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MPCX_OK;
}

extern "C" int mpcx_solve_regularised(mpcx_ctx *ctx, int S, int32_t *out)
{
    if (!ctx || !out) return MPCX_E_BADARG;
    if (S < 1 || S != ctx->nreg_S || !ctx->nreg) return ctx_fail(ctx, MPCX_E_BADARG, "solve_regularised: S must be the batch size of the last solve on this context");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    // (the host-pointer solves ran on the context's stream and have completed; a _dev solve is ordered by its stream)
    MPCX_HIP(ctx, hipMemcpy(out, ctx->nreg, (size_t)S * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MPCX_OK;
}

extern "C" void mpcx_default_solve_opts(mpcx_solve_opts *o)
{
    // reference defaults: optimizer.py:178-188; ipopt defaults: tol 1e-8, acceptable_tol 1e-6
    o->min_mass = 0.1; o->u_max = 5.0; o->r_min = 0.99; o->r_max = 5.0; o->eps_r = 0.01;
    o->eps_vr = 1e-5; o->eps_vn = 1e-5; o->eps_vt = 1e-5; o->tf_max = 5.0; o->w_nu = 1000.0; o->w_tr = 0.002;
    o->tol = 1e-8; o->acceptable_tol = 1e-6; o->max_iter = 200; o->acceptable_iter = 15; o->n_refine = 1; o->flags = 0;
}

extern "C" size_t mpcx_solve_workspace_bytes(int S, int K)
{
    return (size_t)S * ws_doubles(K) * sizeof(double);
}

// what a solve on THIS context's device touches: one slot per persistent workgroup, min(S, workgroups resident at once)
extern "C" size_t mpcx_solve_workspace_bytes_ctx(const mpcx_ctx *ctx, int S, int K)
{
    const int slots = (ctx && S > ctx->n_slots) ? ctx->n_slots : S;
    return (size_t)slots * ws_doubles(K) * sizeof(double);
}

extern "C" int mpcx_solve_batch_ragged_dev(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *stage, const double *xbar,
                                           const double *ubar, const double *tf, const double *consts,
                                           const double *r_des, const mpcx_solve_opts *opts, double *X, double *U,
                                           double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                           void *workspace, void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 3 || !opts) return ctx_fail(ctx, MPCX_E_BADARG, "solve: need S>=1, K>=3 and options");
    if (!workspace) return ctx_fail(ctx, MPCX_E_BADARG, "solve: workspace of mpcx_solve_workspace_bytes(S,K) required");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    SolveArgs a;
    a.S = S; a.K = K; a.Ks = Ks; a.stage = stage; a.xbar = xbar; a.ubar = ubar; a.tfbar = tf; a.consts = consts; a.r_des = r_des;
    a.o = to_dev_opts(opts);
    a.X = X; a.U = U; a.NU = NU; a.tf_out = tf_out; a.kkt = kkt; a.status = status; a.iters = iters;
    a.ws = (double *)workspace; a.ws_stride = ws_doubles(K);
    // per-satellite regularisation counts of this solve (library-owned, grow-only; read back by mpcx_solve_regularised)
    if (ctx->nreg_cap < S) {
        if (ctx->nreg) (void)hipFree(ctx->nreg);
        ctx->nreg = nullptr; ctx->nreg_cap = 0;
        MPCX_HIP(ctx, hipMalloc((void **)&ctx->nreg, (size_t)S * 2 * sizeof(int32_t)));
        ctx->nreg_cap = S;
    }
    a.nreg = ctx->nreg; ctx->nreg_S = S;
    // longest-first launch order from the previous solve's iteration counts (include/mpcx.h, MPCX_SOLVE_INDEX_ORDER)
    // (a batch the device holds at once has no order to choose: every satellite starts at time 0)
    const bool adaptive = !(opts->flags & MPCX_SOLVE_INDEX_ORDER) && S > ctx->n_slots;
    a.order = nullptr;
    if (adaptive) {
        // grow-only buffers (a smaller batch reuses them: no free / allocation, hence no implicit device synchronisation,
        // when ConstellationMPC alternates group sizes on one context); the stored counts are valid only for a following
        // solve of the same batch size
        if (ctx->order_cap < S) {
            if (ctx->prev_iters) (void)hipFree(ctx->prev_iters);
            if (ctx->pred_hist) (void)hipFree(ctx->pred_hist);
            ctx->pred_hist = nullptr;
            if (ctx->order) (void)hipFree(ctx->order);
            ctx->prev_iters = ctx->order = nullptr; ctx->order_cap = 0; ctx->order_S = 0; ctx->order_valid = 0;
            MPCX_HIP(ctx, hipMalloc((void **)&ctx->prev_iters, (size_t)S * sizeof(int32_t)));
            MPCX_HIP(ctx, hipMalloc((void **)&ctx->pred_hist, (size_t)kPredHist * S * sizeof(int32_t)));
            MPCX_HIP(ctx, hipMalloc((void **)&ctx->order, (size_t)S * sizeof(int32_t)));
            ctx->order_cap = S;
        }
        if (ctx->order_S != S) { ctx->order_S = S; ctx->order_valid = 0; }
        if (ctx->order_valid) {
            hipLaunchKernelGGL(launch_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, S, ctx->prev_iters, ctx->order);
            a.order = ctx->order;
        }
    }
    if (opts->flags & MPCX_SOLVE_SHARED_TF) {
        // one final time for the whole batch: a cooperative launch, one workgroup per satellite, all of them resident
        if (Ks) return ctx_fail(ctx, MPCX_E_BADARG, "solve: MPCX_SOLVE_SHARED_TF needs the same node count for every satellite (no ragged batch)");
        if (opts->flags & MPCX_SOLVE_FIXED_TF) return ctx_fail(ctx, MPCX_E_BADARG, "solve: MPCX_SOLVE_SHARED_TF and MPCX_SOLVE_FIXED_TF exclude each other");
        if (ctx->coop_max == 0) {
            int coop = 0, per_cu = 0;
            MPCX_HIP(ctx, hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, ctx->device));
            MPCX_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, solve_shared_kernel, 64, 0));
            ctx->coop_max = coop ? per_cu * (ctx->n_slots / 8) : -1;
        }
        if (ctx->coop_max < 0) return ctx_fail(ctx, MPCX_E_HIP, "solve: the device does not support cooperative launches (MPCX_SOLVE_SHARED_TF)");
        if (S > ctx->coop_max) return ctx_fail(ctx, MPCX_E_BADARG, "solve: MPCX_SOLVE_SHARED_TF takes at most as many satellites as the device holds workgroups at once");
        if (ctx->red_cap < S) {
            if (ctx->red) (void)hipFree(ctx->red);
            ctx->red = nullptr; ctx->red_cap = 0;
            MPCX_HIP(ctx, hipMalloc((void **)&ctx->red, ((size_t)2 * S * GR_N + 2) * sizeof(double)));
            ctx->red_cap = S;
        }
        a.red = ctx->red;
        a.arrive = (int32_t *)(ctx->red + (size_t)2 * ctx->red_cap * GR_N);
        a.abort_flag = a.arrive + 1;
        a.counter = nullptr; a.order = nullptr;
        MPCX_HIP(ctx, hipMemsetAsync(a.arrive, 0, 2 * sizeof(int32_t), (hipStream_t)stream));
        void *kargs[] = {(void *)&a};
        MPCX_HIP(ctx, hipLaunchCooperativeKernel((const void *)solve_shared_kernel, dim3(S), dim3(64), kargs, 0, (hipStream_t)stream));
        ctx->order_valid = 0;
        return MPCX_OK;
    }
    // the launch's own work-queue counter: one of a ring, so that two solves of one context enqueued on different streams
    // do not share (and reset) one queue -- each queue position must go to exactly one workgroup of ITS launch
    if (!ctx->counter) MPCX_HIP(ctx, hipMalloc((void **)&ctx->counter, kCounterRing * sizeof(int32_t)));
    a.counter = ctx->counter + (ctx->launch_seq++ % kCounterRing);
    MPCX_HIP(ctx, hipMemsetAsync(a.counter, 0, sizeof(int32_t), (hipStream_t)stream));
    // (the workspace is the caller's: slot b of THIS call's buffer)
    const int slots = S < ctx->n_slots ? S : ctx->n_slots;
    // small batches -- at most one satellite per SIMD -- go to the two-wave build (solve2w.hip): a second wave per
    // satellite shares the factorisation; results are bit for bit the one-wave kernel's (-ffp-contract=on, build.py;
    // tests/test_full_size_gpu.py::test_two_wave_small_batch_kernel).  MPCX_SOLVE_ONE_WAVE keeps the one-wave kernel.
    if (S <= kTwoWaveMax && !(opts->flags & MPCX_SOLVE_ONE_WAVE)) {
        if (mpcx2w_launch(&a, sizeof a, slots, (hipStream_t)stream) != 0) return ctx_fail(ctx, MPCX_E_HIP, "solve: two-wave launch failed");
    } else
    hipLaunchKernelGGL(solve_kernel, dim3(slots), dim3(64), 0, (hipStream_t)stream, a);
    MPCX_HIP(ctx, hipGetLastError());
    if (adaptive) {
        // (order_valid counts the solves of this batch size recorded so far)
        const int slot = ctx->order_valid % kPredHist, n_valid = ctx->order_valid + 1 < kPredHist ? ctx->order_valid + 1 : kPredHist;
        hipLaunchKernelGGL(update_prediction_kernel, dim3((S + 255) / 256), dim3(256), 0, (hipStream_t)stream, S, iters, ctx->pred_hist,
                           ctx->prev_iters, slot, n_valid);
        MPCX_HIP(ctx, hipGetLastError());
        ctx->order_valid += 1;
        if (ctx->order_valid >= 2 * kPredHist) ctx->order_valid -= kPredHist;     // (keeps slot and n_valid as they are)
    }
    return MPCX_OK;
}

extern "C" int mpcx_solve_batch_dev(mpcx_ctx *ctx, int S, int K, const double *stage, const double *xbar,
                                    const double *ubar, const double *tf, const double *consts,
                                    const double *r_des, const mpcx_solve_opts *opts, double *X, double *U,
                                    double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                    void *workspace, void *stream)
{
    return mpcx_solve_batch_ragged_dev(ctx, S, K, nullptr, stage, xbar, ubar, tf, consts, r_des, opts, X, U, NU, tf_out, status,
                                       iters, kkt, workspace, stream);
}

extern "C" int mpcx_mpc_step_batch_ragged_dev(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *xbar, const double *ubar,
                                              const double *tf, const double *consts, const double *r_des, int flags,
                                              double max_step, const mpcx_solve_opts *opts, double *X, double *U,
                                              double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                              void *workspace, void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    if (!workspace) return ctx_fail(ctx, MPCX_E_BADARG, "mpc_step: workspace of mpcx_mpc_step_workspace_bytes(S,K) required");
    // workspace = [stage records | int32 discretize status | solver workspace]
    double *stage = (double *)workspace;
    const size_t nstage = (size_t)S * (K - 1) * MPCX_STAGE_DOUBLES;
    int32_t *dstat = (int32_t *)(stage + nstage);
    double *sws = stage + nstage + ((size_t)S + 1) / 2 + 1;
    // (a ragged batch's thrust tables have as many columns as the satellite has nodes)
    int rc = mpcx_discretize_stages_ragged_dev(ctx, S, K, Ks, K, Ks, xbar, ubar, tf, consts, flags, max_step, stage, dstat, stream);
    if (rc) return rc;
    rc = mpcx_solve_batch_ragged_dev(ctx, S, K, Ks, stage, xbar, ubar, tf, consts, r_des, opts, X, U, NU, tf_out, status, iters,
                                     kkt, sws, stream);
    if (rc) return rc;
    merge_status_kernel_launch(S, dstat, status, (hipStream_t)stream);
    MPCX_HIP(ctx, hipGetLastError());
    return MPCX_OK;
}

extern "C" int mpcx_mpc_step_batch_dev(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *ubar,
                                       const double *tf, const double *consts, const double *r_des, int flags,
                                       double max_step, const mpcx_solve_opts *opts, double *X, double *U,
                                       double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                       void *workspace, void *stream)
{
    return mpcx_mpc_step_batch_ragged_dev(ctx, S, K, nullptr, xbar, ubar, tf, consts, r_des, flags, max_step, opts, X, U, NU,
                                          tf_out, status, iters, kkt, workspace, stream);
}

extern "C" size_t mpcx_mpc_step_workspace_bytes(int S, int K)
{
    return ((size_t)S * (K - 1) * MPCX_STAGE_DOUBLES + ((size_t)S + 1) / 2 + 1) * sizeof(double) +
           mpcx_solve_workspace_bytes(S, K);
}

extern "C" size_t mpcx_mpc_step_workspace_bytes_ctx(const mpcx_ctx *ctx, int S, int K)
{
    return ((size_t)S * (K - 1) * MPCX_STAGE_DOUBLES + ((size_t)S + 1) / 2 + 1) * sizeof(double) +
           mpcx_solve_workspace_bytes_ctx(ctx, S, K);
}

extern "C" int mpcx_mpc_step_batch_ragged(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *xbar, const double *ubar,
                                          const double *tf, const double *consts, const double *r_des, int flags,
                                          double max_step, const mpcx_solve_opts *opts, double *X, double *U, double *NU,
                                          double *tf_out, int32_t *status, int32_t *iters, double *kkt)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 3 || !opts) return ctx_fail(ctx, MPCX_E_BADARG, "mpc_step: need S>=1, K>=3 and options");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    void *ws = ctx_workspace(ctx, mpcx_mpc_step_workspace_bytes_ctx(ctx, S, K));
    if (!ws) return MPCX_E_NOMEM;
    DeviceArena ar(ctx);
    double *dx = ar.upload(xbar, (size_t)S * 7 * K), *du = ar.upload(ubar, (size_t)S * 3 * K);
    double *dtf = ar.upload(tf, S), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST), *drd = ar.upload(r_des, S);
    int32_t *dKs = Ks ? ar.upload(Ks, S) : nullptr;
    double *dX = ar.alloc<double>((size_t)S * 7 * K), *dU = ar.alloc<double>((size_t)S * 3 * K);
    const bool fixed_tf = (opts->flags & MPCX_SOLVE_FIXED_TF) != 0;          // tf_out is an input too (include/mpcx.h)
    double *dNU = ar.alloc<double>((size_t)S * 7 * K), *dtfo = fixed_tf ? ar.upload(tf_out, S) : ar.alloc<double>(S), *dk = ar.alloc<double>(S);
    int32_t *dst = ar.alloc<int32_t>(S), *dit = ar.alloc<int32_t>(S);
    if (ar.failed()) return ar.code();
    int rc = mpcx_mpc_step_batch_ragged_dev(ctx, S, K, dKs, dx, du, dtf, dc, drd, flags, max_step, opts, dX, dU, dNU, dtfo, dst,
                                            dit, dk, ws, ctx->stream);
    if (rc) return rc;
    ar.download(X, dX, (size_t)S * 7 * K); ar.download(U, dU, (size_t)S * 3 * K); ar.download(NU, dNU, (size_t)S * 7 * K);
    ar.download(tf_out, dtfo, S); ar.download(status, dst, S); ar.download(iters, dit, S); ar.download(kkt, dk, S);
    return ar.finish();
}

extern "C" int mpcx_mpc_step_batch(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *ubar,
                                   const double *tf, const double *consts, const double *r_des, int flags,
                                   double max_step, const mpcx_solve_opts *opts, double *X, double *U, double *NU,
                                   double *tf_out, int32_t *status, int32_t *iters, double *kkt)
{
    return mpcx_mpc_step_batch_ragged(ctx, S, K, nullptr, xbar, ubar, tf, consts, r_des, flags, max_step, opts, X, U, NU, tf_out,
                                      status, iters, kkt);
}

// One SCP iteration of OptimalController.update (control.py:183-227) for S satellites, host buffers in and out: the nonlinear
// rollout under the given thrust law sampled at the satellite's nodes (its thrust at those nodes = extract_uk), the
// linearisation / discretisation about it and the solve -- x_bar and u_bar never leave the device.
extern "C" int mpcx_scp_iteration_batch_ragged(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *y0, const double *tf,
                                               const double *consts, const double *r_des, int prop_flags, int ctrl_kind,
                                               const double *ctrl_vec, int Ku, const int32_t *Kus, const double *end_tau,
                                               double prop_max_step, int disc_flags, double disc_max_step,
                                               const mpcx_solve_opts *opts, double *xbar_out, double *ubar_out, double *X, double *U,
                                               double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                               int32_t *prop_status)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 3 || !opts || !prop_status) return ctx_fail(ctx, MPCX_E_BADARG, "scp_iteration: need S>=1, K>=3, options and prop_status");
    if (opts->flags & (MPCX_SOLVE_FIXED_TF | MPCX_SOLVE_SHARED_TF)) return ctx_fail(ctx, MPCX_E_BADARG, "scp_iteration: free per-satellite tf only");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    void *ws = ctx_workspace(ctx, mpcx_mpc_step_workspace_bytes_ctx(ctx, S, K));
    if (!ws) return MPCX_E_NOMEM;
    DeviceArena ar(ctx);
    double *dy0 = ar.upload(y0, (size_t)S * 7), *dtf = ar.upload(tf, S), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST), *drd = ar.upload(r_des, S);
    size_t nv = 0;
    if (ctrl_kind == MPCX_CTRL_CONSTANT) nv = (size_t)S * 3;
    else if (ctrl_kind == MPCX_CTRL_TANGENTIAL) nv = S;
    else if (ctrl_kind == MPCX_CTRL_SEQUENCE) nv = (size_t)S * 3 * Ku;
    double *dv = (nv && ctrl_vec) ? ar.upload(ctrl_vec, nv) : nullptr;
    double *de = (ctrl_kind == MPCX_CTRL_SEQUENCE && end_tau) ? ar.upload(end_tau, S) : nullptr;
    int32_t *dKs = Ks ? ar.upload(Ks, S) : nullptr, *dKus = Kus ? ar.upload(Kus, S) : nullptr;
    double *dx = ar.alloc<double>((size_t)S * 7 * K), *du = ar.alloc<double>((size_t)S * 3 * K);
    double *dX = ar.alloc<double>((size_t)S * 7 * K), *dU = ar.alloc<double>((size_t)S * 3 * K), *dNU = ar.alloc<double>((size_t)S * 7 * K);
    double *dtfo = ar.alloc<double>(S), *dk = ar.alloc<double>(S);
    int32_t *dst = ar.alloc<int32_t>(S), *dit = ar.alloc<int32_t>(S), *dps = ar.alloc<int32_t>(S), *dpn = ar.alloc<int32_t>(S);
    if (ar.failed()) return ar.code();
    if (Ks) {                                                                                        // the unused columns
        MPCX_HIP(ctx, hipMemsetAsync(dx, 0, (size_t)S * 7 * K * sizeof(double), ctx->stream));
        MPCX_HIP(ctx, hipMemsetAsync(du, 0, (size_t)S * 3 * K * sizeof(double), ctx->stream));
    }
    int rc = mpcx_propagate_thrust_batch_ragged_dev(ctx, S, K, dKs, dy0, dtf, dc, prop_flags, ctrl_kind, dv, Ku, dKus, de, prop_max_step,
                                                    dx, du, dps, dpn, ctx->stream);
    if (rc) return rc;
    rc = mpcx_mpc_step_batch_ragged_dev(ctx, S, K, dKs, dx, du, dtf, dc, drd, disc_flags, disc_max_step, opts, dX, dU, dNU, dtfo, dst,
                                        dit, dk, ws, ctx->stream);
    if (rc) return rc;
    if (xbar_out) ar.download(xbar_out, dx, (size_t)S * 7 * K);
    if (ubar_out) ar.download(ubar_out, du, (size_t)S * 3 * K);
    ar.download(X, dX, (size_t)S * 7 * K); ar.download(U, dU, (size_t)S * 3 * K); ar.download(NU, dNU, (size_t)S * 7 * K);
    ar.download(tf_out, dtfo, S); ar.download(status, dst, S); ar.download(iters, dit, S); ar.download(kkt, dk, S);
    ar.download(prop_status, dps, S);
    return ar.finish();
}

namespace MPCX_NS {
// Node counts of the next SCP iteration: Simulator.run samples a rollout over tf at int(base_res * tf) points (simulator.py:38;
// control.py:227 passes tf_u), computed where tf_u lives.  The counts are clamped to the row length only in the sense that a
// count outside 3..K makes that satellite's next solve report MPCX_ST_BADK.
__global__ void node_count_kernel(int S, double base_res, const double *tf, int32_t *Kn)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S) Kn[i] = (int32_t)(base_res * tf[i]);
}
__global__ void fill_f64_kernel(int n, double v, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}
__global__ void scale_f64_kernel(int n, const double *a, double d, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] / d;
}
}  // namespace MPCX_NS

// OptimalController.update (control.py:170-235) for S satellites as ONE call, everything between the first input and the
// last result resident in HBM (include/mpcx.h).
extern "C" int mpcx_mpc_update_batch(mpcx_ctx *ctx, int S, int K, int n_scp, double base_res, const double *y0, const double *tf0,
                                     const double *consts, const double *r_des, double ref_thrust, double prop_max_step,
                                     int disc_flags, double disc_max_step, const mpcx_solve_opts *opts, double *X, double *U,
                                     double *NU, double *tf_out, int32_t *Ks_out, int32_t *status, int32_t *iters, double *kkt,
                                     int32_t *prop_status, double sim_tf, double sim_interval, int sim_n_eval, int sim_flags,
                                     double sim_max_step, double *y_sim, int32_t *sim_status)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 3 || n_scp < 1 || !opts || !y0 || !tf0 || !consts || !r_des || !X || !U || !NU || !tf_out || !Ks_out || !status ||
        !iters || !kkt || !prop_status || !(base_res > 0.0))
        return ctx_fail(ctx, MPCX_E_BADARG, "mpc_update: need S>=1, K>=3, n_scp>=1, base_res>0, options and all arrays");
    if (opts->flags & (MPCX_SOLVE_FIXED_TF | MPCX_SOLVE_SHARED_TF)) return ctx_fail(ctx, MPCX_E_BADARG, "mpc_update: free per-satellite tf only");
    if (y_sim && (sim_n_eval < 1 || !(sim_tf > 0.0) || !(sim_interval > 0.0) || !sim_status))
        return ctx_fail(ctx, MPCX_E_BADARG, "mpc_update: segment flight needs sim_tf>0, sim_interval>0, sim_n_eval>=1, sim_status");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    void *ws = ctx_workspace(ctx, mpcx_mpc_step_workspace_bytes_ctx(ctx, S, K));
    if (!ws) return MPCX_E_NOMEM;
    DeviceArena ar(ctx);
    hipStream_t st = ctx->stream;
    double *dy0 = ar.upload(y0, (size_t)S * 7), *dtf0 = ar.upload(tf0, S), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST), *drd = ar.upload(r_des, S);
    const size_t n7 = (size_t)S * 7 * K, n3 = (size_t)S * 3 * K;
    double *dx = ar.alloc<double>(n7), *du = ar.alloc<double>(n3);                      // reference trajectory / thrust of the iteration
    double *dX = ar.alloc<double>(n7), *dNU = ar.alloc<double>(n7);
    double *dU[2] = {ar.alloc<double>(n3), ar.alloc<double>(n3)};                       // plan thrust: iteration i writes dU[i & 1], the next rollout plays it
    double *dtfu[2] = {ar.alloc<double>(S), ar.alloc<double>(S)};                       // tf_u of the iterations, alternating
    double *dmag = ar.alloc<double>(S), *done = ar.alloc<double>(S), *dk = ar.alloc<double>(S), *dend = ar.alloc<double>(S);
    int32_t *dKn[2] = {ar.alloc<int32_t>(S), ar.alloc<int32_t>(S)};                     // node counts, alternating
    int32_t *dst = ar.alloc<int32_t>((size_t)n_scp * S), *dit = ar.alloc<int32_t>((size_t)n_scp * S);
    int32_t *dps = ar.alloc<int32_t>(S), *dpn = ar.alloc<int32_t>(S), *dps2 = ar.alloc<int32_t>(S);
    double *dys = y_sim ? ar.alloc<double>((size_t)S * 7 * sim_n_eval) : nullptr;
    int32_t *dss = y_sim ? ar.alloc<int32_t>(S) : nullptr;
    if (ar.failed()) return ar.code();
    const dim3 gS((S + 255) / 256), b256(256);
    hipLaunchKernelGGL(fill_f64_kernel, gS, b256, 0, st, S, ref_thrust, dmag);
    hipLaunchKernelGGL(fill_f64_kernel, gS, b256, 0, st, S, 1.0, done);
    MPCX_HIP(ctx, hipMemsetAsync(dps, 0, sizeof(int32_t) * S, st));
    const double *tf_cur = dtf0;
    const int32_t *Ks = nullptr;            // node counts of the current iteration (nullptr: K for everybody)
    int rc = MPCX_OK;
    for (int it = 0; it < n_scp && rc == MPCX_OK; ++it) {
        double *Uw = dU[it & 1], *tfw = dtfu[it & 1];
        if (Ks) {                                                                         // ragged rows: the unused columns
            MPCX_HIP(ctx, hipMemsetAsync(dx, 0, n7 * sizeof(double), st));
            MPCX_HIP(ctx, hipMemsetAsync(du, 0, n3 * sizeof(double), st));
        }
        // control.py:178-180 / :217-227: rollout under the tangential reference law, then under the sequence just optimised,
        // played over its own horizon (end_tau = 1) and sampled at int(base_res * tf_u) nodes; u_bar = extract_uk (:188)
        if (it == 0)
            rc = mpcx_propagate_thrust_batch_ragged_dev(ctx, S, K, nullptr, dy0, tf_cur, dc, 0, MPCX_CTRL_TANGENTIAL, dmag, 0, nullptr, nullptr,
                                                        prop_max_step, dx, du, dps2, dpn, st);
        else
            rc = mpcx_propagate_thrust_batch_ragged_dev(ctx, S, K, Ks, dy0, tf_cur, dc, 0, MPCX_CTRL_SEQUENCE, dU[(it - 1) & 1], K,
                                                        it >= 2 ? dKn[(it - 1) & 1] : nullptr, done, prop_max_step, dx, du, dps2, dpn, st);
        if (rc) break;
        merge_status_kernel_launch(S, dps2, dps, st);                                    // (any rollout's failure is the update's)
        rc = mpcx_mpc_step_batch_ragged_dev(ctx, S, K, Ks, dx, du, tf_cur, dc, drd, disc_flags, disc_max_step, opts, dX, Uw, dNU, tfw,
                                            dst + (size_t)it * S, dit + (size_t)it * S, dk, ws, st);
        if (rc) break;
        tf_cur = tfw;
        if (it + 1 < n_scp) {
            int32_t *kn = dKn[(it + 1) & 1];
            hipLaunchKernelGGL(node_count_kernel, gS, b256, 0, st, S, base_res, tfw, kn);
            Ks = kn;
        }
    }
    if (rc) return rc;
    MPCX_HIP(ctx, hipGetLastError());
    const double *Uplan = dU[(n_scp - 1) & 1];
    if (y_sim) {
        // Simulator.run_segment (simulator.py:58-65): fly sim_tf under the truth model with SequenceController(u_opt, tf_u,
        // tf_sim = sim_interval): end_tau = tf_u / sim_interval (control.py:102), the plan's table with its own column count
        // (end_tau as the host computes it: a division, not a product with the reciprocal)
        hipLaunchKernelGGL(scale_f64_kernel, gS, b256, 0, st, S, tf_cur, sim_interval, dend);
        hipLaunchKernelGGL(fill_f64_kernel, gS, b256, 0, st, S, sim_tf, dmag);           // (dmag is free again: the flight time per satellite)
        rc = mpcx_propagate_thrust_batch_ragged_dev(ctx, S, sim_n_eval, nullptr, dy0, dmag, dc, sim_flags, MPCX_CTRL_SEQUENCE, Uplan, K, Ks, dend,
                                                    sim_max_step, dys, nullptr, dss, dpn, st);
        if (rc) return rc;
    }
    ar.download(X, dX, n7); ar.download(U, (const double *)Uplan, n3); ar.download(NU, dNU, n7);
    ar.download(tf_out, tf_cur, S);
    if (Ks) ar.download(Ks_out, Ks, S);
    else for (int i = 0; i < S; ++i) Ks_out[i] = K;                                      // (a single iteration: K nodes for everybody)
    ar.download(status, dst, (size_t)n_scp * S); ar.download(iters, dit, (size_t)n_scp * S); ar.download(kkt, dk, S);
    ar.download(prop_status, dps, S);
    if (y_sim) { ar.download(y_sim, dys, (size_t)S * 7 * sim_n_eval); ar.download(sim_status, dss, S); }
    return ar.finish();
}

extern "C" int mpcx_solve_batch(mpcx_ctx *ctx, int S, int K, const double *A, const double *Bp, const double *Bn,
                                const double *Sigma, const double *xi, const double *xbar, const double *ubar,
                                const double *tf, const double *consts, const double *r_des,
                                const mpcx_solve_opts *opts, double *X, double *U, double *NU, double *tf_out,
                                int32_t *status, int32_t *iters, double *kkt)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 3 || !opts) return ctx_fail(ctx, MPCX_E_BADARG, "solve: need S>=1, K>=3 and options");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    // pack the reference-shaped arrays into stage records on the host (tiny, O(S K) copies)
    const size_t n = (size_t)S * (K - 1);
    std::vector<double> st(n * MPCX_STAGE_DOUBLES);
    for (int s = 0; s < S; ++s)
        for (int k = 0; k < K - 1; ++k) {
            double *r = &st[((size_t)s * (K - 1) + k) * MPCX_STAGE_DOUBLES];
            const size_t b = (size_t)s * (K - 1) + k;
            for (int e = 0; e < 49; ++e) r[e] = A[b * 49 + e];
            for (int e = 0; e < 21; ++e) { r[49 + e] = Bn[b * 21 + e]; r[70 + e] = Bp[b * 21 + e]; }
            for (int i = 0; i < 7; ++i) {
                r[91 + i] = Sigma[(size_t)s * 7 * (K - 1) + (size_t)i * (K - 1) + k];
                r[98 + i] = xi[(size_t)s * 7 * (K - 1) + (size_t)i * (K - 1) + k];
            }
        }
    void *ws = ctx_workspace(ctx, mpcx_solve_workspace_bytes_ctx(ctx, S, K));
    if (!ws) return MPCX_E_NOMEM;
    DeviceArena ar(ctx);
    double *dst_ = ar.upload(st.data(), st.size());
    double *dx = ar.upload(xbar, (size_t)S * 7 * K), *du = ar.upload(ubar, (size_t)S * 3 * K);
    double *dtf = ar.upload(tf, S), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST), *drd = ar.upload(r_des, S);
    double *dX = ar.alloc<double>((size_t)S * 7 * K), *dU = ar.alloc<double>((size_t)S * 3 * K);
    const bool fixed_tf = (opts->flags & MPCX_SOLVE_FIXED_TF) != 0;          // tf_out is an input too (include/mpcx.h)
    double *dNU = ar.alloc<double>((size_t)S * 7 * K), *dtfo = fixed_tf ? ar.upload(tf_out, S) : ar.alloc<double>(S), *dk = ar.alloc<double>(S);
    int32_t *dstat = ar.alloc<int32_t>(S), *dit = ar.alloc<int32_t>(S);
    if (ar.failed()) return ar.code();
    int rc = mpcx_solve_batch_dev(ctx, S, K, dst_, dx, du, dtf, dc, drd, opts, dX, dU, dNU, dtfo, dstat, dit, dk, ws,
                                  ctx->stream);
    if (rc) return rc;
    ar.download(X, dX, (size_t)S * 7 * K); ar.download(U, dU, (size_t)S * 3 * K); ar.download(NU, dNU, (size_t)S * 7 * K);
    ar.download(tf_out, dtfo, S); ar.download(status, dstat, S); ar.download(iters, dit, S); ar.download(kkt, dk, S);
    return ar.finish();
}

#else  // MPCX_TWO_WAVE: the small-batch kernel and its launcher (called by mpcx_solve_batch_ragged_dev in the other build)

// Two waves per satellite.  The first runs solve_satellite exactly as the one-wave kernel's wave does; the second waits in a
// command loop and joins it for every factorisation (riccati_factor2).  Same work queue, same slot workspaces.
__global__ __launch_bounds__(128, MPCX_SOLVE_WAVES) void solve_kernel2w(SolveArgs a)
{
    SatData &sd = g_sd;
    Scratch &w = g_w;
    __shared__ int next_item;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (;;) {
        if (threadIdx.x == 0) next_item = atomicAdd(a.counter, 1);
        WG_BARRIER();
        const int b = __builtin_amdgcn_readfirstlane(next_item);
        WG_BARRIER();
        if (b >= a.S) return;
        int sat = a.order ? a.order[b] : b;
        if ((unsigned)sat >= (unsigned)a.S) sat = b;
        if (wave == 0) {
            solve_satellite<false>(a, sat, (int)blockIdx.x, sd, w, lane);
            if (lane == 0) w.cmd = CMD_EXIT;
            WG_BARRIER();
        } else {
            const int Kmax = a.K;
            int K = a.Ks ? a.Ks[sat] : Kmax;
            if (K < 3 || K > Kmax) K = Kmax;                       // (the first wave reports MPCX_ST_BADK and sends CMD_EXIT at once)
            const Sat s = sat_view(a, sat, (int)blockIdx.x, K, Kmax);
            for (;;) {
                WG_BARRIER();
                if (w.cmd == CMD_EXIT) break;
                (void)riccati_factor2(s, sd, w, lane, 1, w.cmd_arg != 0);
            }
        }
        WG_BARRIER();
    }
}

}  // namespace MPCX_NS

// (SolveArgs of the two builds are the same struct compiled twice: handed over as bytes)
int mpcx2w_launch(const void *args, size_t args_bytes, int blocks, hipStream_t stream)
{
    MPCX_NS::SolveArgs a;
    if (args_bytes != sizeof a) return -1;
    memcpy(&a, args, sizeof a);
    hipLaunchKernelGGL(MPCX_NS::solve_kernel2w, dim3(blocks), dim3(128), 0, stream, a);
    return 0;
}
#endif
