// solve_riccati.hpp -- the structure-exploiting linear solve of one interior-point iteration: Riccati factorisation in the
// shifted state (one lane per matrix element, operands in LDS), the 8-channel linear-term sweeps, the 7 x 7 border and -- for
// the shared-tf launch -- the launch-wide reductions.  Included after solve_phases.hpp by solve.hip and solve2w.hip.
#pragma once

namespace MPCX_NS {
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

// Wave priority of the factorisation (s_setprio; 0 = off).  Two waves share a SIMD; when one is inside the recursion -- a chain of
// dependent fp64 / LDS steps, where every lost issue slot lengthens the critical path -- and the other in a node-parallel phase --
// independent loads and arithmetic that fill any slot -- the arbiter should prefer the first.  Measured (profiles/r05/
// occupancy_sweep.txt, three alternating rounds on one box, bit-identical): S8192_K30 9.63 -> 9.43 ms, S4096_K100 12.06 -> 11.97,
// S4096_K30 4.85 -> 4.86 (its 2 x 2048 satellites start in lock step: the SIMD's two waves are in the same phase most of the time);
// the sweeps and the border solve at the same priority add nothing, a higher priority for the node-parallel phases or for one
// wave slot of every SIMD loses.
#ifndef MPCX_PRIO_RIC
#define MPCX_PRIO_RIC 3
#endif
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
#ifdef MPCX_TP
    double flatB[2][FLAT_N];       // the pair's second wave sweeps its own channels at the same time: its own staging buffers
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
        cwf64 *ch = s.ch + (size_t)k * CH_N + C_RHS;
        ci.gx = xld(ch + R_GX + r);
        if (r < 3) ci.gu = xld(ch + R_GU + r);
        if (dyn) { ci.rho = xld(ch + R_RHO + r); ci.aff = xld(ch + R_AFF + r); }
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
    cwf64 *ch = s.ch + (size_t)k * CH_N + C_RHS;
    ChanRaw cr;
#ifdef MPCX_WS_LDS
    // (the right-hand-side record is in LDS, Sigma in the global stage record: two loads and a select)
    const double sg = s.Sig(k < K - 2 ? k : K - 2)[rr], af = ch[R_AFF + rr];
    cr.gx = ch[R_GX + rr]; cr.gu = ch[R_GU + r3]; cr.rho = ch[R_RHO + rr]; cr.aff = (c == 1) ? sg : af;
#else
    cgf64 *pa = (c == 1) ? s.Sig(k < K - 2 ? k : K - 2) + rr : ch + R_AFF + rr;
    cr.gx = xld(ch + R_GX + rr); cr.gu = xld(ch + R_GU + r3); cr.rho = xld(ch + R_RHO + rr); cr.aff = xld(pa);
#endif
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
// Pdst: the buffer of P_k the update goes to in the two-wave factorisation (its double-buffered copy; the one-wave one updates w.Pn).  fac: the node's factor
// record -- the one-wave factorisation stores the updated gain and Q_uu^-1 there itself; nullptr in the two-wave factorisation,
// where they go to the node's LDS copies (o.Kg, o.Qi) and the second wave stores them.  One body for both (round 4 had it twice).
template <bool TO_RECORD>
__device__ __noinline__ void stiff_stage_update(StageOps &o, Scratch &w, double *Pdst, wf64 *fac, int lane)
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
        double *P = Pdst;
        if constexpr (TO_RECORD) P = w.Pn;          // (the one-wave factorisation's own buffer: an LDS address the compiler knows)
        P[lane] += om1 * (v1l * v1h) + om2 * (v2l * v2h);
    }
    if (lane < 21) {
        const int r = lane / 7, c = lane - 7 * r;
        const double qc[3] = {w.Quy[c], w.Quy[7 + c], w.Quy[14 + c]};
        double v1, v2;
        sm_v(qc, c, v1, v2);
        const double kg = o.Kg[lane] + om1 * t1[r] * v1 + om2 * t2[r] * v2;
        o.Kg[lane] = kg;
        if constexpr (TO_RECORD) fac[F_KG + lane] = kg;
    }
    if (lane < 9) {
        const int r = lane / 3, c = lane - 3 * r;
        const double qv = Qi[lane] - om1 * t1[r] * t1[c] - om2 * t2[r] * t2[c];
        if constexpr (TO_RECORD) fac[F_QI + lane] = qv;
#ifdef MPCX_TWO_WAVE
        else o.Qi[lane] = qv;
#endif
    }
}


#ifdef MPCX_TWO_WAVE
#ifdef MPCX_TP
// Time-parallel build: the workgroup (a pair of waves) factorises the nodes lo .. hi-1 of its segment from a zero cost-to-go
// behind node hi-1 (the last segment, hi = K, has the terminal node as before).  Lane groups of the fused backward sweep of a
// segment that is not the last: group 0 the dtf channel, groups 1..7 the unit prices e_0..e_6 on the state behind node hi-1; its
// ninth channel, the right-hand side, rides in the lanes of group 0 a second time (results in the extra record chx).
struct TpRange { int lo, hi; bool last; double *Wout; };
#define RF_ARGS , const TpRange rg          /* (by value: four scalars in registers; by reference they were flat loads from the caller's stack) */
#define RF_HI rg.hi
#define RF_LO rg.lo
#define RF_LAST rg.last
#define RF_GOOD w.good_flag
#else
#define RF_ARGS
#define RF_HI K
#define RF_LO 0
#define RF_LAST true
#define RF_GOOD w.good_flag
#endif
// The same factorisation shared by the two waves of a small-batch workgroup (role 0 / role 1), one hardware barrier per node.
// Role 0 keeps what the next node waits for -- the critical chain P_{k+1} -> LDL^T -> X1 -> Pt -> T = Pt F -> S = F^T T ->
// Q_uu^-1 -> P_k -- and the operand prefetch; role 1 takes everything else off that chain: its own (redundant) LDL^T for
// X2, the blocks G and Minv the sweeps need, the fused backward sweep of the node before (k+1, whose matrices sit complete
// in another operand buffer) and all stores to the factor record.  Every element is computed by the same expressions as in
// the one-wave form (bit-identical results: the library is built with -ffp-contract=on).  ~7 750 -> ~4 800 cycles per node for a wave that is alone on its
// SIMD (64 satellites on a 1024-SIMD chip: the small-batch regime of BASELINE configs[1]).
__device__ __noinline__ bool riccati_factor2(const Sat &s_in, SatData &sd, Scratch &w, int lane, int role, bool keep_pt RF_ARGS)
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
        cwf64 *nb = s.nb + (size_t)k * NB_N;
#ifdef MPCX_WS_LDS
        // (stage records in global memory, the Newton record in LDS: the lane's second element comes from one or the other)
        cgf64 *pg = (e1 < 70) ? stk + e1 : stm + (e1 < 91 ? e1 : 70);
        const double g1 = *pg, l1 = nb[wx1];
        pre[0] = stk[lane]; pre[1] = (e1 < 91) ? g1 : l1; pre[2] = nb[src2];
#else
        cgf64 *p1 = (e1 < 70) ? stk + e1 : (e1 < 91) ? stm + e1 : nb + wx1;
        cgf64 *p2 = nb + src2;
        pre[0] = stk[lane]; pre[1] = xld(p1); pre[2] = xld(p2);
#endif
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
        fetch(RF_HI - 1);
        stash(w.ops[(RF_HI - 1) % 3], RF_HI - 1);
        if (RF_LAST && lane < 49) w.ops[(RF_HI - 1) % 3].G2[(lane / 7) * FS + lane % 7] = sd.WxK[lane];
    } else {
        for (int e = lane; e < 49; e += 64) w.Pn2[RF_HI & 1][e] = 0.0;       // P_K = 0 (read as "P of node k+1" by node K-1)
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
    const int scl = RF_LAST ? sc : (sc == 0 ? 1 : 2);        // the group's channel as chan_fetch / chan_mask know it (2: no stage data)
    double pnext = RF_LAST ? 0.0 : ((sc >= 1 && sr == sc - 1) ? 1.0 : 0.0);
#ifdef MPCX_TP
    // the ninth channel of a segment that is not the last -- its right-hand side -- rides in the lanes of group 0 a second
    // time: the same matrix rows in registers, its own vectors; results into the extra record
    ChanIn cur0{0.0, 0.0, 0.0, 0.0};
    ChanRaw nraw0{0.0, 0.0, 0.0, 0.0};
    double pnext0 = 0.0;
    const bool act0 = sact && sc == 0 && !RF_LAST;
#endif
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
#ifdef MPCX_TP
        if (!RF_LAST) {
            const double v0 = cur0.rho + pnext0;
            double t0 = pnext0;
#pragma unroll
            for (int q = 0; q < 7; ++q) t0 += -sw_G[q] * gshfl8(v0, q) + sw_Pt[q] * gshfl8(cur0.aff, q);
            if (!dynj || !act0) t0 = 0.0;
            double qu0 = cur0.gu;
#pragma unroll
            for (int q = 0; q < 7; ++q) qu0 += Bpmcol[q] * gshfl8(cur0.gx, q) + Bhcol[q] * gshfl8(t0, q);
            if (sr >= 3 || !act0) qu0 = 0.0;
            double pp0 = cur0.gx;
#pragma unroll
            for (int q = 0; q < 7; ++q) pp0 += Acol[q] * gshfl8(t0, q);
#pragma unroll
            for (int q = 0; q < 3; ++q) pp0 -= Kgcol[q] * gshfl8(qu0, q);
            ustore(s.ws, act0 ? s.o_chx + j * CHX_N + sr : sink_e, pp0);
            ustore(s.ws, (act0 && sr < 3) ? s.o_chx + j * CHX_N + 7 + sr3 : sink_e, qu0);
            pnext0 = act0 ? pp0 : pnext0;
        }
#endif
        // ... and the part of node j's factor record that the first wave left in LDS: gain, Bh, Q_uu^-1
        const bool on21 = lane < 21;
        const int l21 = on21 ? lane : 0;
        const int fb = s.o_fac + j * FAC_N;
        ustore(s.ws, on21 ? fb + F_KG + lane : sink_e, o.Kg[l21]);
        ustore(s.ws, on21 ? fb + F_BH + lane : sink_e, o.F[(l21 / 3) * FS + 7 + l21 % 3]);
        ustore(s.ws, lane < 9 ? fb + F_QI + lane : sink_e, o.Qi[lane < 9 ? lane : 0]);
    };
    for (int k = RF_HI - 1; k >= RF_LO; --k) {
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
                stiff_stage_update<false>(o, w, w.Pn2[k & 1], nullptr, lane);
            }
            if (!__all(good) && lane == 0) RF_GOOD = 0;
        } else {
            // ---- role 1 ----
            if (k > RF_LO) fetch(k - 1);
            nraw = chan_fetch(s, k, scl, srr, sr3);
#ifdef MPCX_TP
            if (!RF_LAST) nraw0 = chan_fetch(s, k, 0, srr, sr3);
#endif
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
            if (k + 1 < RF_HI) sweep_node(w.ops[(k + 1) % 3], k + 1);          // node k+1: complete since the last barrier
            // inputs of node k for its sweep in the next slot
            cur = chan_mask(nraw, scl, sr, sact);
#ifdef MPCX_TP
            if (!RF_LAST) cur0 = chan_mask(nraw0, 0, sr, act0);
#endif
            if (!dyn) {
                const double tg = (sc == 2) ? sd.avt[srr] : sd.ta[sc >= 3 ? sc - 3 : 0][srr];
                cur.gx = (sact && sc >= 2) ? tg : cur.gx;
                cur.rho = 0.0; cur.aff = 0.0;
            }
            if (k > RF_LO) stash(w.ops[(k - 1) % 3], k - 1);
        }
        WG_BARRIER();
        if (RF_GOOD == 0) { good = false; break; }
    }
#ifdef MPCX_TP
    if (role == 0 && good && lane < 49) rg.Wout[lane] = w.Pn2[RF_LO & 1][lane];      // the segment's cost-to-go Hessian at its first node
#endif
    if (role == 1 && good) sweep_node(w.ops[RF_LO % 3], RF_LO);
    WG_BARRIER();
    return good;
}

// What the first wave's driver calls: tell the second wave (parked in solve_kernel2w's command loop) to join, take role 0.
enum { CMD_FACTOR = 1, CMD_EXIT = 2, CMD_SWEEP = 3, CMD_COMBINE = 4 };
#ifndef MPCX_TP
__device__ __forceinline__ bool riccati_factor(const Sat &s, SatData &sd, Scratch &w, int lane, bool fuse_sweep, bool keep_pt)
{
    (void)fuse_sweep;                              // (the backward sweep always rides along: it is the second wave's)
    if (lane == 0) { w.cmd = CMD_FACTOR; w.cmd_arg = keep_pt ? 1 : 0; }
    WG_BARRIER();
    return riccati_factor2(s, sd, w, lane, 0, keep_pt);
}
#endif
#else
// Backward Riccati sweep: factorisation (DESIGN.md "Solver algorithm").  Returns false on breakdown.
// With fuse_sweep the backward linear-term sweep of all 8 channels rides along: node k's p_k, qu_k are formed
// right after its matrices, while they are still in LDS (same arithmetic as sweep_backward).
__device__ __noinline__ bool riccati_factor(const Sat &s_in, SatData &sd, Scratch &w, int lane, bool fuse_sweep, bool keep_pt)
{
    if (MPCX_PRIO_RIC) __builtin_amdgcn_s_setprio(MPCX_PRIO_RIC);
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
        cwf64 *nb = s.nb + (size_t)k * NB_N;
#ifdef MPCX_WS_LDS
        // (stage records in global memory, the Newton record in LDS: the lane's second element comes from one or the other)
        cgf64 *pg = (e1 < 70) ? stk + e1 : stm + (e1 < 91 ? e1 : 70);
        const double g1 = *pg, l1 = nb[wx1];
        pre[0] = stk[lane]; pre[1] = (e1 < 91) ? g1 : l1; pre[2] = nb[src2];
#else
        cgf64 *p1 = (e1 < 70) ? stk + e1 : (e1 < 91) ? stm + e1 : nb + wx1;
        cgf64 *p2 = nb + src2;
        pre[0] = stk[lane]; pre[1] = xld(p1); pre[2] = xld(p2);
#endif
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
        wf64 *fac = s.fac + (size_t)k * FAC_N;
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
        if (o.SX[SX_EX] > 0.0 || o.SX[SX_EU] > 0.0) stiff_stage_update<true>(o, w, nullptr, fac, lane);
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
    if (MPCX_PRIO_RIC) __builtin_amdgcn_s_setprio(0);
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
#ifdef MPCX_WS_LDS
// LDS-resident build: the factor record and the Newton record ARE in LDS -- the sweeps read them where they lie (FAC_AT, D_AT
// below); only the node's global operands, A (head of stage record k) and Bpm (B_kp of record k-1), are fetched one node ahead
// and staged in the double-buffered copy.
template <bool PT>
__device__ __forceinline__ void sweep_fetch_mats(const Sat &s, int k, int lane, SweepPre &pre)
{
    const int K = s.K;
    cgf64 *stk = s.stage + (size_t)(k <= K - 2 ? k : K - 2) * MPCX_STAGE_DOUBLES;
    cgf64 *stm = s.stage + (size_t)(k >= 1 ? k - 1 : 0) * MPCX_STAGE_DOUBLES;
    pre.v[4] = stk[lane];
    pre.v[5] = stm[70 + (lane < 21 ? lane : 0)];
}

__device__ __forceinline__ void sweep_stash_mats(double *f, int K, int k, int lane, const SweepPre &pre)
{
    f[F_A + lane] = (k <= K - 2) ? pre.v[4] : 0.0;
    if (lane < 21) f[F_BPM + lane] = (k >= 1) ? pre.v[5] : 0.0;
}
#define FAC_AT(s, f, k) ((s).fac + (size_t)(k) * FAC_N)
#define D_AT(s, f, k, rr) (((k) <= (s).K - 2) ? (s).nb[(size_t)(k) * NB_N + N_D + (rr)] : 0.0)
#else
template <bool PT>
__device__ __forceinline__ void sweep_fetch_mats(const Sat &s, int k, int lane, SweepPre &pre)
{
    const int K = s.K;
    cwf64 *fac = s.fac + (size_t)k * FAC_N;
#pragma unroll
    for (int q = 0; q < 2; ++q) pre.v[q] = fac[lane + 64 * q];
    pre.v[2] = fac[(PT || lane + 128 < F_PT) ? lane + 128 : F_PT - 1];
    pre.v[3] = PT ? fac[(lane + 192 < FAC_N) ? lane + 192 : FAC_N - 1] : 0.0;
    // A: head of stage record k; Bpm: B_kp of record k-1; D: Newton record k (what a node lacks is zeroed when stashed)
    cgf64 *stk = s.stage + (size_t)(k <= K - 2 ? k : K - 2) * MPCX_STAGE_DOUBLES;
    cgf64 *stm = s.stage + (size_t)(k >= 1 ? k - 1 : 0) * MPCX_STAGE_DOUBLES;
    cwf64 *nb = s.nb + (size_t)k * NB_N;
    pre.v[4] = stk[lane];
#ifdef MPCX_WS_LDS
    const double g5 = stm[70 + (lane < 21 ? lane : 0)], l5 = nb[N_D + ((lane >= 21 && lane < 28) ? lane - 21 : 0)];
    pre.v[5] = (lane < 21) ? g5 : l5;
#else
    cgf64 *p5 = (lane < 21) ? stm + 70 + lane : nb + N_D + ((lane < 28) ? lane - 21 : 0);
    pre.v[5] = xld(p5);
#endif
}

__device__ __forceinline__ void sweep_stash_mats(double *f, int K, int k, int lane, const SweepPre &pre)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) f[lane + 64 * q] = pre.v[q];
    f[F_A + lane] = (k <= K - 2) ? pre.v[4] : 0.0;
    f[F_BPM + lane] = ((lane < 21) ? (k >= 1) : (k <= K - 2)) ? pre.v[5] : 0.0;
}

#define FAC_AT(s, f, k) (f)
#define D_AT(s, f, k, rr) ((f)[F_D + (rr)])
#endif

// Backward sweep for channels [c0, c1): p_k and qu_k stored per channel.
// (time-parallel build: over the nodes lo .. hi-1 of a segment from p = 0 behind node hi-1; chx: the right-hand-side channel
//  of a segment that is not the last -- its vectors go to the extra record, the channel slots there hold dtf and the prices)
#ifdef MPCX_TP
#define SB_ARGS , int sb_lo, int sb_hi, gf64 *chx
#define SB_LO sb_lo
#define SB_HI sb_hi
#else
#define SB_ARGS
#define SB_LO 0
#define SB_HI K
#endif
__device__ __noinline__ void sweep_backward(const Sat &s_in, SatData &sd, Scratch &w, int c0, int c1, int lane SB_ARGS)
{
    const Sat s = uniform_view(s_in);
    const int K = s.K;
    const int c = lane >> 3, r = lane & 7;
    const bool act = (c >= c0 && c < c1) && r < 7;
    const int rr = (r < 7) ? r : 6, r3 = (r < 3) ? r : 2;
    SweepPre pre;
    sweep_fetch_mats<true>(s, SB_HI - 1, lane, pre);
    sweep_stash_mats(w.flat[(SB_HI - 1) & 1], K, SB_HI - 1, lane, pre);
    ChanIn cur = chan_inputs(s, sd, SB_HI - 1, c, r, act), nxt = cur;
    double pnext = 0.0;
    WG_SYNC();
    for (int k = SB_HI - 1; k >= SB_LO; --k) {
        const double *f = w.flat[k & 1];
        const auto fr = FAC_AT(s, f, k);               // the node's factor record (its LDS home, or the staged copy)
        if (k > SB_LO) { sweep_fetch_mats<true>(s, k - 1, lane, pre); nxt = chan_inputs(s, sd, k - 1, c, r, act); }
        const bool dyn = (k <= K - 2);
        double Grow[7], Ptrow[7], Acol[7], Bpmcol[7], Bhcol[7], Kgcol[3];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            Grow[q] = fr[F_G + rr * 7 + q]; Ptrow[q] = fr[F_PT + rr * 7 + q]; Acol[q] = f[F_A + q * 7 + rr];
            Bpmcol[q] = f[F_BPM + q * 3 + r3]; Bhcol[q] = fr[F_BH + q * 3 + r3];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) Kgcol[q] = fr[F_KG + q * 7 + rr];
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
#ifdef MPCX_TP
            if (chx) {
                gf64 *cx = chx + (size_t)k * CHX_N;
                cx[r] = p;
                if (r < 3) cx[7 + r] = qu;
            } else
#endif
            {
            wf64 *ch = s.ch + (size_t)k * CH_N;
            ch[C_P + c * 7 + r] = p;
            if (r < 3) ch[C_QU + c * 3 + r] = qu;
            }
            pnext = p;
        }
        if (k > SB_LO) sweep_stash_mats(w.flat[(k - 1) & 1], K, k - 1, lane, pre);
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
        cwf64 *ch = s.ch + (size_t)k * CH_N;
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
        const auto fr = FAC_AT(s, f, k);
        FT_DECL
        if (k + 1 < K) { sweep_fetch_mats<false>(s, k + 1, lane, pre); nraw = chan_fetch(s, k + 1, c, rr, r3); load_pq(k + 1, qun, pnn, sgn); }
        const bool dyn = (k <= K - 2);
        FT_MARK(10)
        double Kgrow[7], Arow[7], Gcol[7], Mrow[7], Qirow[3], Bpmrow[3], Bhrow[3];
#pragma unroll
        for (int q = 0; q < 7; ++q) { Kgrow[q] = fr[F_KG + r3 * 7 + q]; Arow[q] = f[F_A + rr * 7 + q]; Gcol[q] = fr[F_G + q * 7 + rr]; Mrow[q] = fr[F_MINV + rr * 7 + q]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) { Qirow[q] = fr[F_QI + r3 * 3 + q]; Bpmrow[q] = f[F_BPM + rr * 3 + q]; Bhrow[q] = fr[F_BH + rr * 3 + q]; }
        const double Dr = D_AT(s, f, k, rr);
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
            const double lam = Dr * nu + cur.rho;
            const bool ad = act && dyn;
            // (the trajectories live in the global workspace in every build: WS_GLOBAL / SINK_GLOBAL are the plain base and sink
            //  offset except in the LDS-resident build, whose LDS part has a base and a sink of its own)
            const int tb = s.o_traj + (k * NCH + c) * TR_N, sink_t = SINK_GLOBAL(s) + lane;
            ustore(WS_GLOBAL(s), act ? tb + T_X + r : sink_t, x);
            ustore(WS_GLOBAL(s), (act && r < 3) ? tb + T_U + r3 : sink_t, u);
            ustore(WS_GLOBAL(s), ad ? tb + T_NU + r : sink_t, nu);
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
__device__ __noinline__ void border_solve(SatData &sd, int lane)
{
    const double gtf_rhs = sd.rs_gtf, rvt_rhs = sd.rs_rvt;
    const double *gex = sd.rs_gex;
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
__device__ __noinline__ bool border_solve_shared(SatData &sd, GridSync &g, double W_glob, double r_glob, bool fail, int lane)
{
    const double gtf_share = sd.rs_gtf, rvt_rhs = sd.rs_rvt;
    const double *gex = sd.rs_gex;
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

}  // namespace MPCX_NS
