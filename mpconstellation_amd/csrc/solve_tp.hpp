// solve_tp.hpp -- the TIME-PARALLEL form of the linear solve of one interior-point iteration (solve_tp.hip only; DESIGN.md
// section 8, tests/tools/partitioned_riccati.py is the same algebra on the CPU oracle).
//
// The K nodes are cut into nseg <= TP_MAXSEG segments; a WORKGROUP of two waves (role 0 / role 1 of riccati_factor2) owns a
// segment -- on its own compute unit: eight waves of one satellite on ONE compute unit share its LDS pipe and two a SIMD, and
// a node then costs more than the four-fold shorter chain saves (measured: 21 k cycles per node instead of 5 k).  The
// satellite's workgroups talk through a mailbox and exchange records in global memory (cooperative launch: all resident).
// Every segment but the last runs the recursion from a ZERO cost-to-go behind its last node; the last one is the tail of the
// sequential recursion (terminal node, border channels).  Channels of a segment:
//   local      its own trajectory from the start state 0 and the terminal price 0 for the channels that have data in it --
//              right-hand side and dtf everywhere, all eight in the last segment;
//   price i    zero data, linear terminal cost e_i on the state behind its last node (not in the last segment):
//              end state -N e_i, start co-state Psi e_i;
//   state i    zero data, start state e_i (not in the first segment): end state Phi e_i.
// With a_j the start state and l_j the terminal price of segment j for one of the eight channels of the reduced solve:
//       a_{j+1} = y0_j + Phi_j a_j - N_j l_j ,      l_j = W_{j+1} a_{j+1} + p0_{j+1} + Psi_{j+1} l_{j+1}       (a_0 = 0, no price on the last)
// -- a coarse problem over the cuts with 7 x 7 blocks: backward  What_j = W_j + Psi_j What_{j+1} (I + N_j What_{j+1})^-1 Phi_j  (once
// per factorisation) and qhat_j (per channel), forward a_j, l_j.  (I + N W)^-1 r goes through a similar symmetric positive
// definite matrix -- I + C'N C with W = C C', or I + C'W C with N = C C' where W is not positive definite (tp_iface_factor) --:
// two LDL^T without pivoting, in registers.  The segment's trajectory is then
// local + sum_i a_i (state i) + sum_i l_i (price i): the combination that forms the direction takes these coefficients.
#pragma once

namespace MPCX_NS {

struct TpSeg {
    double W[49], N[49], Psi[49], Phi[49], What[49];
    double rcd[7];                       // 1 / diag(Cw)
    double Cw[49], Ls[49], rds[7];       // interface behind this segment: What_{j+1} = Cw Cw' (Cw lower), I + Cw'N Cw = Ls diag(1/rds) Ls' (Ls unit lower)
    double E[49], EPhi[49], EN[49];      // E = (I + N What_{j+1})^-1 explicitly, E Phi, E N: the per-channel passes are matrix-vector products
    double M1[49], M2[49];               // qhat_j = p0_j + M1 y0_j + M2 qhat_{j+1}:  M1 = Psi What_{j+1} E,  M2 = Psi (I - What_{j+1} E N)
    double y0[2][7];                     // state behind the last node of the local trajectories (0: right-hand side, 1: dtf)
    double p0[NCH][7];                   // linear term of the local cost-to-go at the first node
    double qhat[NCH][7], a[NCH][7], ell[NCH][7];
    double sl_loc[NCH], sl_ua[7], sl_up[7];      // sum over the segment of Sigma_k . lam_k of each trajectory
    double coef[16];                     // coefficients of the segment's 16 trajectory slots in the direction
    int mode;                            // form of the interface factors (tp_iface_factor)
};
struct TpData {
    int nseg, cut[TP_MAXSEG + 1];
    int seq;                             // commands posted so far (first workgroup) / seen so far (the others)
    int dead;                            // a wait ran out: the satellite's solve ends with MPCX_ST_NUMERIC
    int light;                           // the satellite's workgroups share an XCD: light fences (tp_release / tp_acquire)
    wf64 *it_cur;                        // the iterate as the first wave sees it now (tp_combine on the second wave)
#ifdef MPCX_PHASE_TIMING
    unsigned long long t_wait0;
#endif
    double xK_loc[NCH][7], xK_ua[7][7];
    double T1[49], T2[49];
    TpSeg seg[TP_MAXSEG];
};
// what a segment's workgroup writes to global memory for the one that runs the coarse problem (TP_XCH_N doubles)
struct TpXch { double W[49], N[49], Phi[49], y0[2][7], sl_loc[2], sl_ua[7], sl_up[7], p0[2][7], Psi[49]; };
enum { XO_W = 0, XO_N = 49, XO_PHI = 98, XO_Y0 = 147, XO_SLLOC = 161, XO_SLUA = 163, XO_SLUP = 170, XO_P0 = 177, XO_PSI = 191, XO_END = 240 };
static_assert(sizeof(TpXch) <= TP_XCH_N * sizeof(double), "exchange record");

__shared__ SatData g_sd;
__shared__ Scratch g_w;
__shared__ Sat g_s;                      // (the view of the satellite's problem and workspace: solve_driver.hpp)
__shared__ TpData g_tp;

// mailbox of a satellite's workgroups (ints in global memory, zeroed by the host before the launch)
enum { TPM_SEQ = 0, TPM_CMD = 1, TPM_ARG = 2, TPM_DONE = 3, TPM_OK = 4, TPM_DEAD = 5, TPM_PROG = 8, TPM_XCC = 12 };
// No wait without an end: a workgroup that has polled this many times (about a tenth of a second) declares the satellite's
// solve dead -- the first workgroup reports MPCX_ST_NUMERIC, the others leave.
#ifdef MPCX_TP_DEBUG
constexpr int kTpSpinMax = 1 << 23;
#else
constexpr int kTpSpinMax = 1 << 17;
#endif
// Visibility between the satellite's workgroups.  Agent-scope release / acquire on this chip are an L2 write-back and an L2
// invalidate of the whole XCD (its L2 is not coherent with the other seven): every compute unit of the XCD pays for them, and
// four satellites per XCD already cost more than the segments gain (measured: 1.32 ms at 16 satellites, 1.60 at 32).  The
// workgroups of a satellite sit on ONE XCD (solve_tp.hip gives them block indices of one residue mod 8; each checks the XCC
// it really runs on and the first workgroup compares them before anything is exchanged): their common L2 is coherent for
// them, a writer only has to wait until its stores have reached it, a reader to drop its compute unit's L1 -- `light`.
// MPCX_TP_ACQ_MODE (solve_phases.hpp, xld): 4 -- no cache maintenance at all, every load of another workgroup's data goes past
// the L1 (xld); 2 -- buffer_inv sc1, what the agent fence issues (drops the XCD's L2); 1 -- buffer_inv sc0: leaves stale lines
// in the reader's L1, measured wrong
#ifndef MPCX_TP_REL_MODE
#define MPCX_TP_REL_MODE 1
#endif
__device__ __forceinline__ void tp_release(bool light)
{
    // light: the wave's stores must have reached the common L2 (the L1 is write-through) before the mailbox word that announces them
    // is written.  Outside tgsplit mode a workgroup-scope release fence waits for lgkmcnt only (LDS is what a workgroup shares), so
    // the wait for the global stores is spelled out: without it the mailbox update, which travels to another L2 channel than the
    // data, could become visible first (round-4 advice; the ISA of the round-4 build had no vmcnt wait between the last
    // exchange-record store and the DONE atomic).  s_waitcnt counts per wave: issued under any exec mask it covers all lanes' stores.
    if (light && MPCX_TP_REL_MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}
__device__ __forceinline__ void tp_acquire(bool light)
{
    if (light && MPCX_TP_ACQ_MODE == 1) { asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
    else if (light && MPCX_TP_ACQ_MODE == 2) { asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
    else if (light && MPCX_TP_ACQ_MODE == 3) { asm volatile("buffer_inv sc0 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
    else if (light && MPCX_TP_ACQ_MODE == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// pause between two polls of a mailbox word: short at first, then longer -- a hundred and ninety waves polling every hundred
// cycles are a request storm on the L2s that slows every other workgroup's memory accesses (measured between 16 and 32 satellites)
__device__ __forceinline__ void tp_pause(int spins)
{
    if (spins < 8) __builtin_amdgcn_s_sleep(2);
    else if (spins < 64) __builtin_amdgcn_s_sleep(8);
    else __builtin_amdgcn_s_sleep(24);
}
__device__ __forceinline__ int tp_xcc_id() { return (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15; }      // HW_REG_XCC_ID[3:0]

enum { TPK_OFF = 0, TPK_LOCAL = 1, TPK_PRICE = 2, TPK_STATE = 3 };
#ifdef MPCX_TP_DEBUG
#define TP_DBG(...) do { if (lane == 0) printf(__VA_ARGS__); } while (0)
#else
#define TP_DBG(...)
#endif

// Forward sweep of one wave's eight lane groups over the nodes of segment j (the arithmetic of sweep_forward).
// which 0 (the pair's first wave): last segment -- the eight channels' local trajectories -> slots 0..7; other segments --
//   right-hand side (backward vectors in the extra record), dtf (channel slot 0), prices 0..5 (channel slots 1..6) -> slots 0..7.
// which 1 (second wave): last segment -- states 0..6 -> slots 8..14; other segments -- price 6 (channel slot 7) -> slot 8,
//   states 0..6 -> slots 9..15 (zeros in the first segment, whose start state is the fixed x_0).
// ngroups: 8, or 1 in a refinement pass (the right-hand-side channel alone; the other slots keep the first pass's trajectories).
__device__ __noinline__ void tp_sweep_forward(const Sat &s_in, SatData &sd, double (*flat)[FLAT_N], TpData &tp, int j, int which, int ngroups, int lane)
{
    const Sat s = uniform_view(s_in);
    const int K = s.K;
    const int lo = __builtin_amdgcn_readfirstlane(tp.cut[j]), hi = __builtin_amdgcn_readfirstlane(tp.cut[j + 1]);
    const int nseg = __builtin_amdgcn_readfirstlane(tp.nseg);
    const bool last = (j == nseg - 1), first = (j == 0);
    const int g = lane >> 3, r = lane & 7;
    const int rr = (r < 7) ? r : 6, r3 = (r < 3) ? r : 2;
    int kind, cdata, pq, idx;            // pq: channel slot of the backward vectors, -1 none (zero), -2 the extra record
    if (last) {
        if (which == 0) { kind = TPK_LOCAL; cdata = g; pq = g; idx = g; }
        else { kind = (g < 7 && !first) ? TPK_STATE : TPK_OFF; cdata = 2; pq = -1; idx = g; }
    } else if (which == 0) {
        if (g == 0) { kind = TPK_LOCAL; cdata = 0; pq = -2; idx = 0; }
        else if (g == 1) { kind = TPK_LOCAL; cdata = 1; pq = 0; idx = 1; }
        else { kind = TPK_PRICE; cdata = 2; pq = g - 1; idx = g - 2; }
    } else {
        if (g == 0) { kind = TPK_PRICE; cdata = 2; pq = 7; idx = 6; }
        else { kind = first ? TPK_OFF : TPK_STATE; cdata = 2; pq = -1; idx = g - 1; }
    }
    const bool live = g < ngroups;                   // (a group that is off still writes its slot: zeros)
    const bool act = live && r < 7;
    const int tslot = which * 8 + g;
    SweepPre pre;
    sweep_fetch_mats<false>(s, lo, lane, pre);
    sweep_stash_mats(flat[lo & 1], K, lo, lane, pre);
    ChanIn cur = chan_inputs(s, sd, lo, cdata, r, act);
    ChanRaw nraw{0.0, 0.0, 0.0, 0.0};
    cgf64 *chx = wave_uniform((cgf64 *)s.chx);
    // qu_k, p_{k+1} and Sigma_k of the lane's channel / component (behind the segment's last node p is the terminal price)
    auto load_pq = [&](int k, double &qu, double &pn, double &sg) {
        cwf64 *ch = s.ch + (size_t)k * CH_N;
        const int ps = pq >= 0 ? pq : 0;
        const double q1 = ch[C_QU + ps * 3 + r3], p1 = (ch + (k <= K - 2 ? CH_N : 0))[C_P + ps * 7 + rr];
        const double q2 = chx[(size_t)k * CHX_N + 7 + r3], p2 = chx[(size_t)(k + 1 < K ? k + 1 : k) * CHX_N + rr];
        qu = (pq >= 0) ? q1 : (pq == -2 ? q2 : 0.0);
        pn = (pq >= 0) ? p1 : (pq == -2 ? p2 : 0.0);
        if (k == hi - 1 && !last) pn = (kind == TPK_PRICE && rr == idx) ? 1.0 : 0.0;
        sg = s.Sig(k <= K - 2 ? k : K - 2)[rr];
    };
    double quc, pnc, sgc, qun = 0.0, pnn = 0.0, sgn = 0.0;
    load_pq(lo, quc, pnc, sgc);
    if (!(act && r < 3)) quc = 0.0;
    if (!(act && lo <= K - 2)) pnc = 0.0;
    double y = (kind == TPK_STATE && act && r == idx) ? 1.0 : 0.0, siglam = 0.0, xK = 0.0;
    WG_SYNC();
    for (int k = lo; k < hi; ++k) {
        const double *f = flat[k & 1];
        const auto fr = FAC_AT(s, f, k);
        if (k + 1 < hi) { sweep_fetch_mats<false>(s, k + 1, lane, pre); nraw = chan_fetch(s, k + 1, cdata, rr, r3); load_pq(k + 1, qun, pnn, sgn); }
        const bool dyn = (k <= K - 2);
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
        const double wv = cur.rho + pnc;
        double nu = 0.0;
#pragma unroll
        for (int q = 0; q < 7; ++q) nu -= Gcol[q] * gshfl8(yh, q) + Mrow[q] * gshfl8(wv, q);
        {
            const double lam = Dr * nu + cur.rho;
            const bool ad = act && dyn;
            const int tb = (which ? s.o_trajx : s.o_traj) + (k * NCH + g) * TR_N, sink_t = SINK_GLOBAL(s) + lane;
            ustore(WS_GLOBAL(s), act ? tb + T_X + r : sink_t, x);
            ustore(WS_GLOBAL(s), (act && r < 3) ? tb + T_U + r3 : sink_t, u);
            ustore(WS_GLOBAL(s), ad ? tb + T_NU + r : sink_t, nu);
            if (k == K - 1) xK = x;
            siglam += ad ? sgc * lam : 0.0;
            y = ad ? yh + nu : y;
        }
        if (k + 1 < hi) {
            sweep_stash_mats(flat[(k + 1) & 1], K, k + 1, lane, pre);
            cur = chan_mask(nraw, cdata, r, act);
            quc = (act && r < 3) ? qun : 0.0; pnc = (act && k + 1 <= K - 2) ? pnn : 0.0; sgc = sgn;
        }
        wsync();
    }
    siglam += __shfl_xor(siglam, 1, 8);
    siglam += __shfl_xor(siglam, 2, 8);
    siglam += __shfl_xor(siglam, 4, 8);
    (void)tslot;
    if (act) {
        TpSeg &sj = tp.seg[j];
        if (kind == TPK_LOCAL) {
            if (last) tp.xK_loc[cdata][r] = xK; else sj.y0[cdata][r] = y;
            if (r == 0) sj.sl_loc[cdata] = siglam;
        } else if (kind == TPK_PRICE) {
            sj.N[r * 7 + idx] = -y;
            if (r == 0) sj.sl_up[idx] = siglam;
        } else if (kind == TPK_STATE) {
            if (last) tp.xK_ua[idx][r] = xK; else sj.Phi[r * 7 + idx] = y;
            if (r == 0) sj.sl_ua[idx] = siglam;
        }
    }
    WG_SYNC();
}

// L D L^T of the symmetric 7 x 7 matrix M (LDS, row-major, lower triangle read) in the registers of every lane: m = lower
// triangle with the strict part replaced by the unit lower factor, the pivots on its diagonal.  False if a pivot is not positive.
__device__ __forceinline__ bool tp_ldl7(const double *M, double (&m)[28])
{
    bool ok = true;
#pragma unroll
    for (int i = 0, n = 0; i < 7; ++i)
#pragma unroll
        for (int jj = 0; jj <= i; ++jj, ++n) m[n] = M[i * 7 + jj];
#pragma unroll
    for (int pp = 0; pp < 7; ++pp) {
        const double d = m[pp * (pp + 1) / 2 + pp];
        if (!(d > 0.0)) ok = false;
        const double rd = rcp_pos(d > 0.0 ? d : 1.0);
        double col[7];
#pragma unroll
        for (int i = pp + 1; i < 7; ++i) col[i] = m[i * (i + 1) / 2 + pp];
#pragma unroll
        for (int i = pp + 1; i < 7; ++i) {
            const double lip = col[i] * rd;
#pragma unroll
            for (int jj = pp + 1; jj <= i; ++jj) m[i * (i + 1) / 2 + jj] -= lip * col[jj];
            m[i * (i + 1) / 2 + pp] = lip;
        }
    }
    return ok;
}

// Factors of the interface behind segment j for the coarse problem, (I + N_j What_{j+1})^-1 through a similar symmetric matrix:
//   mode 0   What_{j+1} = C C' positive definite (the usual case):  I + N What  ~  S = I + C'N C ,   z = C'^-1 S^-1 C' r ;
//   mode 1   What_{j+1} is not (the tangential equality's curvature lam_vt H_v at the terminal node is indefinite and reaches
//            the cut in the early iterations -- the sequential recursion does not ask for it either: it asks for D + P > 0):
//            N = C C' (N contains D^-1 of the segment's last node: positive definite),  S = I + C'What C ,  z = C S^-1 C^-1 r .
// S positive definite is the cut's share of the recursion's pivot test (N^-1 + What > 0); false = breakdown, regularised by delta_w.
__device__ __noinline__ bool tp_iface_factor(TpData &tp, int j, int lane)
{
    // (the segment's record and What_{j+1} addressed from tp, an LDS object at a known address: as reference / pointer
    //  parameters they were generic pointers, and the 100 accesses of this function flat loads and stores)
    TpSeg &sj = tp.seg[j];
    const double *Wn = tp.seg[j + 1].What;
    double m[28];
    const int a = (lane < 49) ? lane / 7 : 0, b = (lane < 49) ? lane - 7 * (lane / 7) : 0;
    bool ok = true;
    int mode = tp_ldl7(Wn, m) ? 0 : 1;
    if (lane < 49) tp.T2[lane] = 0.5 * (sj.N[a * 7 + b] + sj.N[b * 7 + a]);       // symmetric part of N
    wsync();
    if (mode == 1 && !tp_ldl7(tp.T2, m)) ok = false;
    if (lane == 0) {
        double sq[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {      // sqrt(d) = d / sqrt(d): reciprocal square root seed + two Newton steps
            const double d = (m[i * (i + 1) / 2 + i] > 0.0) ? m[i * (i + 1) / 2 + i] : 1.0;
            double rs = __builtin_amdgcn_rsq(d);
#pragma unroll
            for (int n = 0; n < 2; ++n) { const double e = fma(-d * rs, rs, 1.0); rs = fma(0.5 * rs, e, rs); }
            sq[i] = d * rs; sj.rcd[i] = rs;
        }
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int jj = 0; jj < 7; ++jj) sj.Cw[i * 7 + jj] = (jj > i) ? 0.0 : (jj == i ? sq[i] : m[i * (i + 1) / 2 + jj] * sq[jj]);
        sj.mode = mode;
    }
    wsync();
    {
        double acc = 0.0;      // (N C) or (What C)[a][b]
#pragma unroll
        for (int t = 0; t < 7; ++t) acc += (mode == 0 ? tp.T2[a * 7 + t] : 0.5 * (Wn[a * 7 + t] + Wn[t * 7 + a])) * sj.Cw[t * 7 + b];
        if (lane < 49) tp.T1[lane] = acc;
    }
    wsync();
    {
        double acc = (a == b) ? 1.0 : 0.0;
#pragma unroll
        for (int t = 0; t < 7; ++t) acc += sj.Cw[t * 7 + a] * tp.T1[t * 7 + b];
        wsync();
        if (lane < 49) tp.T2[lane] = acc;
    }
    wsync();
    {
        const double v = 0.5 * (tp.T2[a * 7 + b] + tp.T2[b * 7 + a]);
        wsync();
        if (lane < 49) tp.T2[lane] = v;
    }
    wsync();
    if (!tp_ldl7(tp.T2, m)) ok = false;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            sj.rds[i] = rcp_pos(m[i * (i + 1) / 2 + i] > 0.0 ? m[i * (i + 1) / 2 + i] : 1.0);
#pragma unroll
            for (int jj = 0; jj < 7; ++jj) sj.Ls[i * 7 + jj] = (jj > i) ? 0.0 : (jj == i ? 1.0 : m[i * (i + 1) / 2 + jj]);
        }
    }
    wsync();
    return ok;
}

// z = (I + N_j What_{j+1})^-1 r for the vector r of this lane (every lane its own; factors from LDS)
__device__ __forceinline__ void tp_iface_solve(const TpSeg &sj, double (&r)[7])
{
    double t[7];
    const bool m0 = (sj.mode == 0);
    if (m0) {                 // t = C' r
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int k = i; k < 7; ++k) acc += sj.Cw[k * 7 + i] * r[k];
            t[i] = acc;
        }
    } else {                  // t = C^-1 r
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            double acc = r[i];
#pragma unroll
            for (int k = 0; k < i; ++k) acc -= sj.Cw[i * 7 + k] * t[k];
            t[i] = acc * sj.rcd[i];
        }
    }
#pragma unroll
    for (int p = 1; p < 7; ++p)
#pragma unroll
        for (int q = 0; q < p; ++q) t[p] -= sj.Ls[p * 7 + q] * t[q];
#pragma unroll
    for (int p = 0; p < 7; ++p) t[p] *= sj.rds[p];
#pragma unroll
    for (int p = 5; p >= 0; --p)
#pragma unroll
        for (int q = p + 1; q < 7; ++q) t[p] -= sj.Ls[q * 7 + p] * t[q];
    if (m0) {                 // z = C'^-1 t
#pragma unroll
        for (int i = 6; i >= 0; --i) {
            double acc = t[i];
#pragma unroll
            for (int k = i + 1; k < 7; ++k) acc -= sj.Cw[k * 7 + i] * r[k];
            r[i] = acc * sj.rcd[i];
        }
    } else {                  // z = C t
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k <= i; ++k) acc += sj.Cw[i * 7 + k] * t[k];
            r[i] = acc;
        }
    }
}

// The coarse problem over the cuts (first wave): interface factors and What_j (first pass of a linear solve), then per channel
// qhat_j backward and a_j, l_j forward -- all eight channels in the first pass, the right-hand-side channel alone when refining --
// and from them what the border takes from the forward sweeps: x_K and sum_k Sigma_k . lam_k of every channel.
__device__ __noinline__ bool tp_coarse(const Sat &s, SatData &sd, TpData &tp, int lane, bool pass0)
{
    const int nseg = tp.nseg, last = nseg - 1;
    bool ok = true;
    cgf64 *chx = (cgf64 *)s.chx;
    // what the other segments' workgroups left in their exchange records (their start co-states included), and this
    // workgroup's own: the backward vectors p of the last segment's channels at its first node
    for (int j = 0; j < last; ++j) {
        TpSeg &sj = tp.seg[j];
        cgf64 *x = (cgf64 *)s.xch + (size_t)j * TP_XCH_N;
        for (int e = lane; e < XO_END; e += 64) {
            const double v = xld(x + e);
            if (e < XO_N) { if (pass0) sj.W[e] = v; }
            else if (e < XO_PHI) { if (pass0) sj.N[e - XO_N] = v; }
            else if (e < XO_Y0) { if (pass0) sj.Phi[e - XO_PHI] = v; }
            else if (e < XO_SLLOC) { if (pass0 || e < XO_Y0 + 7) (&sj.y0[0][0])[e - XO_Y0] = v; }
            else if (e < XO_SLUA) { if (pass0 || e == XO_SLLOC) sj.sl_loc[e - XO_SLLOC] = v; }
            else if (e < XO_SLUP) { if (pass0) sj.sl_ua[e - XO_SLUA] = v; }
            else if (e < XO_P0) { if (pass0) sj.sl_up[e - XO_SLUP] = v; }
            else if (e < XO_PSI) { if (pass0 || e < XO_P0 + 7) (&sj.p0[0][0])[e - XO_P0] = v; }
            else if (pass0) sj.Psi[e - XO_PSI] = v;
        }
    }
    if (lane < 56) {
        const int c = lane / 7, rw = lane - 7 * c;
        cwf64 *ch = s.ch + (size_t)tp.cut[last] * CH_N;
        if (pass0 || c == 0) tp.seg[last].p0[c][rw] = ch[C_P + c * 7 + rw];
    }
    (void)chx;
    wsync();
    // product of two 7 x 7 matrices in LDS, one lane per entry (out must not alias the operands; wsync before and after by the caller)
    const int ma = (lane < 49) ? lane / 7 : 0, mb = (lane < 49) ? lane - 7 * (lane / 7) : 0;
    auto mm = [&](const double *A, const double *B) {
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < 7; ++t) acc += A[ma * 7 + t] * B[t * 7 + mb];
        return acc;
    };
    if (pass0 && last >= 1) {
        if (lane < 49) tp.seg[last].What[lane] = tp.seg[last].W[lane];
        wsync();
        for (int j = last - 1; j >= 0; --j) {
            TpSeg &sj = tp.seg[j];
            const double *Wn = tp.seg[j + 1].What;
            if (!tp_iface_factor(tp, j, lane)) ok = false;
            // E column by column (lane i < 7 its column of the identity)
            {
                double col[7];
#pragma unroll
                for (int t = 0; t < 7; ++t) col[t] = (t == lane) ? 1.0 : 0.0;
                tp_iface_solve(sj, col);
                if (lane < 7) {
#pragma unroll
                    for (int t = 0; t < 7; ++t) sj.E[t * 7 + lane] = col[t];
                }
            }
            wsync();
            { const double v1 = mm(sj.E, sj.Phi), v2 = mm(sj.E, sj.N), v3 = mm(Wn, sj.E); if (lane < 49) { sj.EPhi[lane] = v1; sj.EN[lane] = v2; tp.T1[lane] = v3; } }     // T1 = What E
            wsync();
            if (j >= 1) {
                { const double v1 = mm(sj.Psi, tp.T1), v2 = mm(tp.T1, sj.N), v3 = mm(Wn, sj.EPhi); if (lane < 49) { sj.M1[lane] = v1; tp.T2[lane] = v2; sj.M2[lane] = v3; } }   // T2 = What E N ; M2 (scratch) = What E Phi
                wsync();
                { const double v1 = mm(sj.Psi, tp.T2), v2 = mm(sj.Psi, sj.M2); wsync(); if (lane < 49) { tp.T1[lane] = v2; tp.T2[lane] = sj.Psi[lane] - v1; } }   // T1 = Psi What E Phi ; T2 = Psi (I - What E N)
                wsync();
                if (lane < 49) { sj.M2[lane] = tp.T2[lane]; sj.What[lane] = sj.W[lane] + 0.5 * (tp.T1[ma * 7 + mb] + tp.T1[mb * 7 + ma]); }
                wsync();
            }
        }
    }
    // per channel: lane group c = channel, lane i of the group = component; matrix-vector products with the group exchange
    const int c = lane >> 3, ci = lane & 7, cr = (ci < 7) ? ci : 6;
    const bool con = (ci < 7) && (c < (pass0 ? NCH : 1));
    const bool has = c < 2;                         // channels with stage data in the segments before the last
    auto mv = [&](const double *M, double v) {      // component cr of M v, v spread over the group
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < 7; ++t) acc += M[cr * 7 + t] * gshfl8(v, t);
        return acc;
    };
    if (last >= 1) {
        double q = tp.seg[last].p0[c][cr];
        if (con) tp.seg[last].qhat[c][cr] = q;
        for (int j = last - 1; j >= 1; --j) {
            TpSeg &sj = tp.seg[j];
            const double y0 = has ? sj.y0[c & 1][cr] : 0.0, p0 = has ? sj.p0[c & 1][cr] : 0.0;
            q = p0 + mv(sj.M1, y0) + mv(sj.M2, q);
            if (ci >= 7) q = 0.0;
            if (con) sj.qhat[c][cr] = q;
        }
        wsync();
        double av = 0.0;
        if (con) tp.seg[0].a[c][cr] = 0.0;
        for (int j = 0; j < last; ++j) {
            TpSeg &sj = tp.seg[j];
            const double *Wn = tp.seg[j + 1].What;
            const double qn = (ci < 7) ? tp.seg[j + 1].qhat[c][cr] : 0.0;
            const double y0 = has ? sj.y0[c & 1][cr] : 0.0;
            double an = mv(sj.E, y0) - mv(sj.EN, qn);
            if (j > 0) an += mv(sj.EPhi, av);
            if (ci >= 7) an = 0.0;
            const double el = mv(Wn, an) + qn;
            if (con) { sj.ell[c][cr] = el; tp.seg[j + 1].a[c][cr] = an; }
            av = an;
        }
        wsync();
    } else if (con) tp.seg[0].a[c][cr] = 0.0;
    // what the border reads: x_K and sum Sigma . lam of the channel's whole trajectory
    {
        double xk = tp.xK_loc[c][cr], sl = 0.0;
        if (last >= 1) {
            const double al = (ci < 7) ? tp.seg[last].a[c][cr] : 0.0;
#pragma unroll
            for (int t = 0; t < 7; ++t) xk += gshfl8(al, t) * tp.xK_ua[t][cr];
        }
        // (lane i of the group adds the terms of index i; the group sum follows)
        for (int j = 0; j <= last; ++j) {
            const TpSeg &sj = tp.seg[j];
            if (ci == 0 && (j == last || has)) sl += sj.sl_loc[(j == last) ? c : (c & 1)];
            if (ci < 7 && j > 0) sl += sj.a[c][cr] * sj.sl_ua[cr];
            if (ci < 7 && j < last) sl += sj.ell[c][cr] * sj.sl_up[cr];
        }
        sl += __shfl_xor(sl, 1, 8); sl += __shfl_xor(sl, 2, 8); sl += __shfl_xor(sl, 4, 8);
        if (con) { sd.xK[c][cr] = xk; if (ci == 0) sd.siglam[c] = sl; }
    }
    WG_SYNC();
    return __all(ok);
}

// combine_channels of the time-parallel build: the direction is, per segment, a combination of its 16 trajectory slots with
// the coefficients 1 / border solution for the local ones, sum_c sol_c a_j[c] for the states and sum_c sol_c l_j[c] for the prices.
// (both waves of the first workgroup: the second takes every other round of each half; wave 0 computes the coefficients first --
//  tp_cmd_combine -- and tells the second where the iterate is: accepting a trial swaps two pointers of the FIRST wave's view)
__device__ __forceinline__ void tp_combine_coefs(SatData &sd, TpData &tp, int lane)
{
    const int nseg = tp.nseg, last = nseg - 1;
    if (lane < 16 * nseg) {
        const int j = lane >> 4, slot = lane & 15;
        const TpSeg &sj = tp.seg[j];
        auto solp = [&](int c) { return c == 0 ? 1.0 : sd.sol[c - 1]; };
        auto Acoef = [&](int i) { double v = 0.0; for (int c = 0; c < NCH; ++c) v += solp(c) * sj.a[c][i]; return v; };
        auto Lcoef = [&](int i) { double v = 0.0; for (int c = 0; c < NCH; ++c) v += solp(c) * sj.ell[c][i]; return v; };
        double v;
        if (j == last) v = (slot < 8) ? solp(slot) : ((slot < 15 && j > 0) ? Acoef(slot - 8) : 0.0);
        else if (slot < 2) v = solp(slot);
        else if (slot < 8) v = Lcoef(slot - 2);
        else if (slot == 8) v = Lcoef(6);
        else v = (j > 0) ? Acoef(slot - 9) : 0.0;
        tp.seg[j].coef[slot] = v;
    }
    wsync();
}
__device__ __noinline__ void tp_combine(const Sat &s_in, SatData &sd, TpData &tp, double *stg, int lane, bool first, int wave)
{
    const Sat s = uniform_view(s_in);
    const int K = s.K, KP = s.KP;
    const int nseg = tp.nseg;
    wf64 *dr = wave_uniform(s.dr);
    cwf64 *it = wave_uniform((cwf64 *)tp.it_cur);
    cgf64 *traj = wave_uniform((cgf64 *)s.traj), *trajx = wave_uniform((cgf64 *)s.trajx);
    for (int k0 = 0; k0 < K; k0 += 32) {
        const int nk = (K - k0 < 32) ? K - k0 : 32;
        const int n = nk * TR_N;
        for (int e0 = 256 * wave; e0 < n; e0 += 512) {
            double v[4];
            int slot[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + 64 * q + lane;
                const int ec = (e < n) ? e : 0;
                const int kl = ec / TR_N, i = ec - kl * TR_N, k = k0 + kl;
                int j = 0;
                for (int t = 1; t < nseg; ++t) if (k >= tp.cut[t]) j = t;
                const double *cf = tp.seg[j].coef;
                cgf64 *tr = traj + (size_t)k * NCH * TR_N + i, *tx = trajx + (size_t)k * NCH * TR_N + i;
                double acc = 0.0;
#pragma unroll
                for (int t = 0; t < NCH; ++t) acc += cf[t] * xld(tr + t * TR_N) + cf[8 + t] * xld(tx + t * TR_N);
                v[q] = acc; slot[q] = (e < n) ? i * CMB_LD + kl : -1;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) if (slot[q] >= 0) stg[slot[q]] = v[q];
        }
        WG_BARRIER();
        // (this wave's 6 of the round's 12 steps of 64 entries: the three loads of all of them first, as in combine_channels)
        constexpr int NQ = DIR_N * 32 / 128;
        double curv[NQ], Dv[NQ], rv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = lane + 64 * wave + 128 * q;
            const int i = e >> 5, kl = e & 31, k = k0 + kl;
            const int off = (i < T_U) ? I_X + i : (i < T_NU ? I_U + (i - T_U) : (i < T_LAM ? I_NU + (i - T_NU) : I_LAM + (i - T_LAM)));
            const int dst = off * KP + (kl < nk ? k : k0);
            const int kc = (kl < nk) ? k : k0, jj = (i >= T_LAM) ? i - T_LAM : 0;
            curv[q] = first ? it[dst] : dr[dst];
            Dv[q] = s.nb[(size_t)kc * NB_N + N_D + jj]; rv[q] = s.ch[(size_t)kc * CH_N + C_RHS + R_RHO + jj];
        }
        CHUNK_END
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = lane + 64 * wave + 128 * q;
            const int i = e >> 5, kl = e & 31, k = k0 + kl;
            const bool act = kl < nk && !(k == K - 1 && i >= T_NU);
            const int off = (i < T_U) ? I_X + i : (i < T_NU ? I_U + (i - T_U) : (i < T_LAM ? I_NU + (i - T_NU) : I_LAM + (i - T_LAM)));
            const int dst = off * KP + (kl < nk ? k : k0);
            const double cur = curv[q];
            const double base = first ? ((i >= T_LAM) ? -cur : 0.0) : cur;
            const int jj = (i >= T_LAM) ? i - T_LAM : 0;
            const double val = (i >= T_LAM) ? fma(Dv[q], stg[(T_NU + jj) * CMB_LD + kl], rv[q]) : stg[(i < T_LAM ? i : 0) * CMB_LD + kl];
            if (act) dr[dst] = base + val;
        }
        WG_BARRIER();
    }
    if (wave == 0 && lane == 0) {
        if (first) { s.drg[G_TF] = sd.sol[0]; s.drg[G_LVT] = sd.linvt ? 0.0 : -s.itg[G_LVT] + sd.sol[1]; }
        else { s.drg[G_TF] += sd.sol[0]; if (!sd.linvt) s.drg[G_LVT] += sd.sol[1]; }
        if (sd.linvt) sd.zeta_vt = (first ? 0.0 : sd.zeta_vt) + sd.sol[1];
        for (int t = 0; t < NTERM; ++t) sd.zeta[t] = (first ? 0.0 : sd.zeta[t]) + sd.sol[2 + t];
    }
    WG_SYNC();
}

// ---- the satellite's workgroups: geometry, the commands of the first one, what the others do for them -------------------
__device__ __forceinline__ void tp_geometry(TpData &tp, int K)
{
    const int nseg = tp_segments(K);
    tp.nseg = nseg;
    for (int j = 0; j <= nseg; ++j) tp.cut[j] = tp_cut(K, nseg, j);
}

__device__ __forceinline__ TpRange tp_range(TpData &tp, int j)
{
    const int nseg = __builtin_amdgcn_readfirstlane(tp.nseg);
    return TpRange{__builtin_amdgcn_readfirstlane(tp.cut[j]), __builtin_amdgcn_readfirstlane(tp.cut[j + 1]), j == nseg - 1, tp.seg[j].W};
}

// the sweeps of one pass on the segment's two waves: role 0 -- in a refinement pass the right-hand-side channel's backward sweep
// (in the first pass all backward sweeps are fused into the factorisation), then the first eight trajectory slots; role 1 -- the others
__device__ __forceinline__ void tp_sweeps_pair(const Sat &s, SatData &sd, Scratch &w, TpData &tp, int j, int role, int lane, bool pass0)
{
    const bool last = (j == tp.nseg - 1);
    if (role == 0) {
        if (!pass0) sweep_backward(s, sd, w, 0, 1, lane, tp.cut[j], tp.cut[j + 1], last ? (gf64 *)nullptr : s.chx);
        tp_sweep_forward(s, sd, w.flat, tp, j, 0, pass0 ? NCH : 1, lane);
    } else if (pass0)
        tp_sweep_forward(s, sd, w.flatB, tp, j, 1, NCH, lane);      // (one segment only: zeros, so that every slot the combination reads is defined)
}

// first workgroup, first wave: post a command to the satellite's other workgroups / wait until all of them have answered it
__device__ __forceinline__ void tp_post(const Sat &s, TpData &tp, int cmd, int arg, int lane)
{
    if (lane == 0) {
        int *m = s.mail;
        if (tp.seq == 0) {
            // before the first command: where do the other workgroups run?  (they said so before their first poll)
            const int mine = 1 + tp_xcc_id();
            int same = 1;
            for (int j = 0; j < tp.nseg - 1; ++j) {
                int spins = 0, v = 0;
                while ((v = __hip_atomic_load(m + TPM_XCC + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++spins < kTpSpinMax) __builtin_amdgcn_s_sleep(1);
                if (v != mine) same = 0;
            }
#ifdef MPCX_TP_HEAVY
            same = 0;
#endif
            tp.light = same;
        }
        tp.seq += 1;
        if (cmd == CMD_FACTOR) __hip_atomic_store(m + TPM_OK, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(m + TPM_CMD, cmd | (tp.light ? 256 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(m + TPM_ARG, arg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tp_release(tp.light != 0);               // (everything this workgroup wrote before -- Newton records, right-hand sides -- is visible with the command)
        __hip_atomic_store(m + TPM_SEQ, tp.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ bool tp_wait(const Sat &s, TpData &tp, int lane)
{
    int ok = 1;
    if (lane == 0) {
        int *m = s.mail;
        const int want = (tp.nseg - 1) * tp.seq;
        int spins = 0;
        // (polling without cache maintenance: the acquire fence comes once, below)
        while (__hip_atomic_load(m + TPM_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < kTpSpinMax) tp_pause(spins);
        ok = __hip_atomic_load(m + TPM_OK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (spins >= kTpSpinMax) { ok = 0; tp.dead = 1; __hip_atomic_store(m + TPM_DEAD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    tp_acquire(tp.light != 0);                               // (all lanes: the others' records, trajectories and exchange records)
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}

// (the factorisation and the first pass's sweeps are ONE command: no segment's sweeps wait for another segment's factors)
__device__ __forceinline__ bool tp_cmd_factor(const Sat &s, SatData &sd, TpData &tp, int lane, bool keep_pt)
{
    if (tp.dead) return false;
    tp_post(s, tp, CMD_FACTOR, keep_pt ? 1 : 0, lane);
    TP_DBG("[drv b%d] posted FACTOR seq %d\n", (int)blockIdx.x, tp.seq);
    if (lane == 0) { g_w.cmd = CMD_FACTOR; g_w.cmd_arg = keep_pt ? 1 : 0; }
    WG_BARRIER();
#ifdef MPCX_PHASE_TIMING
    const unsigned long long dt0 = __builtin_amdgcn_s_memtime();
#endif
    const bool mine = riccati_factor2(s, sd, g_w, lane, 0, keep_pt, tp_range(tp, tp.nseg - 1));
#ifdef MPCX_PHASE_TIMING
    const unsigned long long dt1 = __builtin_amdgcn_s_memtime();
#endif
    if (mine) { tp_sweeps_pair(s, sd, g_w, tp, tp.nseg - 1, 0, lane, true); WG_BARRIER(); }
#ifdef MPCX_PHASE_TIMING
    if (lane == 0) { int *m = s.mail; const unsigned long long dt2 = __builtin_amdgcn_s_memtime(); m[28] = (int)(dt1 - dt0); m[29] = (int)(dt2 - dt1); tp.t_wait0 = dt2; }
#endif
    TP_DBG("[drv b%d] own factor done ok %d\n", (int)blockIdx.x, (int)mine);
    const bool theirs = tp_wait(s, tp, lane);
    TP_DBG("[drv b%d] workers done ok %d dead %d\n", (int)blockIdx.x, (int)theirs, tp.dead);
#ifdef MPCX_PHASE_TIMING
    if (lane == 0) s.mail[30] = (int)(__builtin_amdgcn_s_memtime() - tp.t_wait0);
#endif
    return mine && theirs;
}
__device__ __forceinline__ void tp_cmd_sweeps(const Sat &s, SatData &sd, TpData &tp, int lane, bool pass0)
{
    if (tp.dead) return;
    tp_post(s, tp, CMD_SWEEP, pass0 ? 1 : 0, lane);
    TP_DBG("[drv b%d] posted SWEEP seq %d\n", (int)blockIdx.x, tp.seq);
    if (lane == 0) { g_w.cmd = CMD_SWEEP; g_w.cmd_arg = pass0 ? 1 : 0; }
    WG_BARRIER();
    tp_sweeps_pair(s, sd, g_w, tp, tp.nseg - 1, 0, lane, pass0);
    WG_BARRIER();
    TP_DBG("[drv b%d] own sweeps done\n", (int)blockIdx.x);
    (void)tp_wait(s, tp, lane);
    TP_DBG("[drv b%d] workers' sweeps done dead %d\n", (int)blockIdx.x, tp.dead);
}

__device__ __forceinline__ void tp_cmd_combine(const Sat &s, SatData &sd, TpData &tp, double *stg, int lane, bool first)
{
    tp_combine_coefs(sd, tp, lane);
    if (lane == 0) { tp.it_cur = s.it; g_w.cmd = CMD_COMBINE; g_w.cmd_arg = first ? 1 : 0; }
    WG_BARRIER();
    tp_combine(s, sd, tp, stg, lane, first, 0);
}

// a workgroup that owns another segment: wait for the first workgroup's commands (its first wave polls the mailbox, the second
// follows through LDS), run them on the segment, leave the exchange record, report
__device__ __forceinline__ void tp_worker(const Sat &s, SatData &sd, TpData &tp, int j, int wave_in, int lane, bool selftest_dead)
{
    int *m = s.mail;
    // (every branch on the wave index or on the command is made scalar: with the loop's exit depending on values the compiler
    //  takes for per-lane ones, the structurised loop dropped all lanes but one of the polling wave after its first pass)
    const int wave = __builtin_amdgcn_readfirstlane(wave_in);
    if (wave == 0 && lane == 0) __hip_atomic_store(m + TPM_XCC + j, 1 + tp_xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (selftest_dead && j == 0) return;           // MPCX_SOLVE_TP_SELFTEST_DEAD: this workgroup never answers (the first one's wait must run out)
    for (;;) {
        if (wave == 0) {
            TP_DBG("[wrk b%d seg %d] polling for seq > %d\n", (int)blockIdx.x, j, tp.seq);
            const int seen = __builtin_amdgcn_readfirstlane(tp.seq);
            int spins = 0, cur = seen;
            do {
                cur = __builtin_amdgcn_readfirstlane(__hip_atomic_load(m + TPM_SEQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));     // (every lane the same word)
                if (cur != seen) break;
                tp_pause(spins);
            } while (++spins < 4 * kTpSpinMax);
            const int cmd_in = (cur == seen) ? (int)CMD_EXIT : __hip_atomic_load(m + TPM_CMD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int arg_in = __hip_atomic_load(m + TPM_ARG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) { tp.seq = seen + 1; g_w.cmd = cmd_in; g_w.cmd_arg = arg_in; }
        }
        WG_BARRIER();
        const int cmd = __builtin_amdgcn_readfirstlane(g_w.cmd) & 255, arg = __builtin_amdgcn_readfirstlane(g_w.cmd_arg);
        const bool light = (__builtin_amdgcn_readfirstlane(g_w.cmd) & 256) != 0;     // (the first workgroup says which with every command)
        tp_acquire(light);
        TP_DBG("[wrk b%d seg %d wave %d] command %d arg %d seq %d\n", (int)blockIdx.x, j, wave, cmd, arg, tp.seq);
        if (cmd == CMD_EXIT) break;
        bool ok = true;
        const bool pass0 = (cmd == CMD_FACTOR);
#ifdef MPCX_PHASE_TIMING
        const unsigned long long wt0 = __builtin_amdgcn_s_memtime();
        unsigned long long wt1 = wt0, wt2 = wt0;
#endif
        if (pass0) ok = riccati_factor2(s, sd, g_w, lane, wave, arg != 0, tp_range(tp, j));
#ifdef MPCX_PHASE_TIMING
        wt1 = __builtin_amdgcn_s_memtime();
#endif
        if (ok) {              // (the same in both waves: the breakdown flag is the workgroup's)
            tp_sweeps_pair(s, sd, g_w, tp, j, wave, lane, pass0);
            WG_BARRIER();
#ifdef MPCX_PHASE_TIMING
            wt2 = __builtin_amdgcn_s_memtime();
#endif
            // the exchange record of the segment: what the sweeps left in LDS and the start co-states -- the backward vectors p at
            // the segment's first node (dtf in channel slot 0, price i in slot 1 + i, the right-hand side in the extra record)
            if (wave == 0) {
                const TpSeg &sj = tp.seg[j];
                gf64 *x = s.xch + (size_t)j * TP_XCH_N;
                cwf64 *ch = s.ch + (size_t)tp.cut[j] * CH_N;
                cgf64 *cx = (cgf64 *)s.chx + (size_t)tp.cut[j] * CHX_N;
                for (int e = lane; e < XO_END; e += 64) {
                    double v;
                    if (e < XO_N) v = sj.W[e];
                    else if (e < XO_PHI) v = sj.N[e - XO_N];
                    else if (e < XO_Y0) v = sj.Phi[e - XO_PHI];
                    else if (e < XO_SLLOC) v = (&sj.y0[0][0])[e - XO_Y0];
                    else if (e < XO_SLUA) v = sj.sl_loc[e - XO_SLLOC];
                    else if (e < XO_SLUP) v = sj.sl_ua[e - XO_SLUA];
                    else if (e < XO_P0) v = sj.sl_up[e - XO_SLUP];
                    else if (e < XO_P0 + 7) v = cx[e - XO_P0];
                    else if (e < XO_PSI) v = ch[C_P + (e - XO_P0 - 7)];
                    else { const int q = e - XO_PSI, rw = q / 7, i = q - 7 * rw; v = ch[C_P + (i + 1) * 7 + rw]; }
                    x[e] = v;
                }
            }
        }
        tp_release(light);                                     // (both waves: factor records, backward vectors, trajectories, the exchange record)
        WG_BARRIER();
        TP_DBG("[wrk b%d seg %d wave %d] command %d finished ok %d\n", (int)blockIdx.x, j, wave, cmd, (int)ok);
        if (wave == 0) {
            const int okw = __builtin_amdgcn_readfirstlane((int)ok);
#ifdef MPCX_PHASE_TIMING
            if (lane == 0 && pass0) {      // diagnostic build only: this command's cycles -- factorisation, sweeps, exchange record + release
                m[16 + 4 * j] = (int)(wt1 - wt0); m[17 + 4 * j] = (int)(wt2 - wt1); m[18 + 4 * j] = (int)(__builtin_amdgcn_s_memtime() - wt2);
            }
#endif
            if (lane == 0) {
                if (!okw) __hip_atomic_store(m + TPM_OK, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(m + TPM_PROG + j, 10 * tp.seq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(m + TPM_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            TP_DBG("[wrk b%d seg %d] reported\n", (int)blockIdx.x, j);
        }
    }
}

}  // namespace MPCX_NS
