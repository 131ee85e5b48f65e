// solve_layout.hpp -- what the host side and the kernels of the batched solve share: the workspace layout of one satellite,
// the solver's constants, the launch arguments.  No device code.  (solve.hip / solve2w.hip: the kernels; solve_api.hip:
// the C entry points.)
#pragma once
#include <cstddef>
#include <stdint.h>
#include "../../include/mpcx.h"

namespace mpcx {
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
    int tp_selftest;                            // MPCX_SOLVE_TP_SELFTEST_DEAD: the first segment's workgroup leaves unasked (test hook)
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

// what the LDS-resident build (solve_lds.hip) keeps of it in LDS: everything but the stage copy, the Newton scalars and
// the channel trajectories
__host__ __device__ inline size_t lds_ws_doubles(int K);
// padded node count: leading dimension of the field-major arrays (rows start on 128-byte boundaries)
__host__ __device__ inline int padded_nodes(int K) { return (K + 15) & ~15; }

__host__ __device__ inline size_t ws_doubles(int K)
{
    const size_t KP = (size_t)padded_nodes(K);
    const size_t n = KP * (3 * IT_N + NS_N + MPCX_STAGE_DOUBLES + 3) + (size_t)K * (NB_N + FAC_N + CH_N + NCH * TR_N) + 3 * GL_N + 64;
    return (n + 15) & ~(size_t)15;
}

__host__ __device__ inline size_t lds_ws_doubles(int K)
{
    const size_t KP = (size_t)padded_nodes(K);
    return KP * (3 * IT_N + 3) + (size_t)K * (NB_N + FAC_N + CH_N) + 3 * GL_N + 64;
}

// ---- time-parallel build (solve_tp.hip, MPCX_SOLVE_TIME_PARALLEL): the horizon cut into segments, a pair of waves each ----
// DESIGN.md section 8 / tests/tools/partitioned_riccati.py: every segment but the last runs the recursion from a zero
// cost-to-go and carries, beside the right-hand side and the dtf channel, seven unit-price channels (terminal price e_i on the
// state behind its last node) and seven unit-state channels (start state e_i); a coarse 7 x 7 recursion over the cuts joins them.
constexpr int TP_MAXSEG = 4;
constexpr int CHX_N = 10;             // per node: backward vectors (p 7, qu 3) of the right-hand-side channel of a segment whose lane groups carry the price channels
__host__ __device__ inline int tp_segments(int K) { return K >= 24 ? 4 : (K >= 8 ? 2 : 1); }
// segment j = nodes tp_cut(j) .. tp_cut(j+1)-1.  The last segment is longer: its workgroup has no right-hand-side sweep of its own
// to run behind the factorisation (it is fused there) and starts before the others have seen the command -- 7 / 7 / 7 / 9 of 30.
__host__ __device__ inline int tp_cut(int K, int nseg, int j)
{
    if (j >= nseg) return K;
#ifndef MPCX_TP_LASTFRAC
#define MPCX_TP_LASTFRAC 18          // the last segment's share of the horizon in sixtieths (the others share the rest evenly)
#endif
    return nseg == 4 ? (j * (60 - MPCX_TP_LASTFRAC) * K) / 180 : (j * K) / nseg;
}
// what a segment's workgroup hands to the one that runs the coarse problem (global memory): W, N, Phi, the ends of its local
// trajectories and the Sigma . lam sums of its trajectories
constexpr int TP_XCH_N = 256, TP_MAIL_N = 16;
// slot workspace of the time-parallel kernel: the other kernels' slot for the call's row length K, then the extra backward
// record, a second bank of 8 trajectories per node, the mailbox of the satellite's workgroups and their exchange records
__host__ __device__ inline size_t tp_extras_offset(int K) { return ws_doubles(K); }
__host__ __device__ inline size_t tp_mail_offset(int K) { return ws_doubles(K) + (size_t)K * (CHX_N + NCH * TR_N); }
__host__ __device__ inline size_t ws_doubles_tp(int K)
{
    const size_t n = tp_mail_offset(K) + TP_MAIL_N + (size_t)TP_MAXSEG * TP_XCH_N;
    return (n + 15) & ~(size_t)15;
}

constexpr int GR_SUM = 6, GR_MAX = 3, GR_MIN = 3, GR_N = GR_SUM + GR_MAX + GR_MIN;      // launch-wide reductions of the shared-tf mode (solve_riccati.hpp: grid_reduce)
constexpr int kPredHist = 8;      // solves whose iteration counts the launch-order predictor remembers (solve.hip: update_prediction_kernel)
}  // namespace mpcx
