// packed_probe.hip -- measurement for DESIGN.md section 8, item 0: the Riccati factorisation step of solve_kernel with its
// fused backward sweep of 8 channels, for FOUR satellites per wave.  16 lanes per satellite: lane c < 7 holds column c of
// every 7 x 7 operand in registers, lanes 8..15 hold the 8 channel vectors, operands travel by v_mov_b64_dpp row_newbcast
// (one instruction per broadcast on gfx950), no LDS.  The channel lanes ride on the matrix lanes' broadcasts with their own
// operands (G v = X1^T R L^-1 v, Pt aff, B^T t, A^T t, Quy^T Qi qu): one instruction stream serves the factorisation of four
// satellites and the sweeps of their 32 channels.  Not part of libmpcx.so; built and run by packed_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#ifndef PROBE_WAVES
#define PROBE_WAVES 2
#endif

template <int N>
__device__ __forceinline__ double bc(double v)      // value of lane N of the caller's 16-lane row, in all 16 lanes
{
    const long long b = __double_as_longlong(v);
    return __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, b, 0x150 + N, 0xF, 0xF, true));
}
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
#define SFOR(i, n) sfor<0, n>([&](auto i##_c) __attribute__((always_inline)) { constexpr int i = decltype(i##_c)::value;
#define SEND });

__device__ __forceinline__ double rcp_pos(double d)
{
    double r = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int n = 0; n < 2; ++n) { const double e = fma(-d, r, 1.0); r = fma(r, e, r); }
    return r;
}

enum { R_A = 0, R_BH = 49, R_BPM = 70, R_WX = 91, R_WU = 140, R_D = 149, REC_N = 156, CH_IN = 24, NCH = 8 };

struct ProbeArgs {
    int S, K;
    const double *rec;     // [S][K][REC_N]: A | Bh | Bpm | Wx | Wu | D (row-major blocks)
    const double *chin;    // [S][K][NCH][24]: gx 7 | gu 3 | rho 7 | aff 7
    double *P, *Kg, *Qi;   // [S][K][49], [S][K][21], [S][K][9]
    double *Lrd;           // [S][K][28]: L (21, row-wise strict lower triangle) | 1/d (7)
    double *p, *qu;        // [S][K][NCH][7], [S][K][NCH][3]
    long long *cycles;     // [blocks]
};

__global__ __launch_bounds__(64, PROBE_WAVES) void packed_probe_kernel(ProbeArgs a)
{
    const int lane = threadIdx.x, l = lane & 15;
    int sat = blockIdx.x * 4 + (lane >> 4);
    const bool live = sat < a.S;
    if (!live) sat = a.S - 1;
    const int K = a.K;
    const bool mat = l < 7, chn = l >= 8;
    const int c = mat ? l : 0, c3 = (l < 3) ? l : 0, ch = chn ? l - 8 : 0;
    const double *rec = a.rec + (size_t)sat * K * REC_N;
    const double *chin = a.chin + ((size_t)sat * K * NCH + ch) * CH_IN;
    double pc[7] = {0, 0, 0, 0, 0, 0, 0};                 // P_{k+1}[:, c] in the matrix lanes, p_{k+1} in the channel lanes
    double nA[7], nbh[7], nbpm[7], nwx[7], nwu[3], nD[7], ngx[7], ngu[3], nrho[7], naff[7];
    auto fetch = [&](int k) __attribute__((always_inline)) {
        const double *rk = rec + (size_t)k * REC_N;
        const double *ck = chin + (size_t)k * NCH * CH_IN;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            nA[r] = rk[R_A + r * 7 + c]; nwx[r] = rk[R_WX + r * 7 + c];
            nbh[r] = rk[R_BH + r * 3 + c3]; nbpm[r] = rk[R_BPM + r * 3 + c3];
            nD[r] = rk[R_D + r];
            ngx[r] = ck[r]; nrho[r] = ck[10 + r]; naff[r] = ck[17 + r];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) { nwu[j] = rk[R_WU + j * 3 + c3]; ngu[j] = ck[7 + j]; }
    };
    fetch(K - 1);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = K - 1; k >= 0; --k) {
        double A[7], bh[7], bpm[7], wx[7], wu[3], D[7], gx[7], gu[3], rho[7], aff[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) { A[r] = nA[r]; wx[r] = nwx[r]; bh[r] = nbh[r]; bpm[r] = nbpm[r]; D[r] = nD[r]; gx[r] = ngx[r]; rho[r] = nrho[r]; aff[r] = naff[r]; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { wu[j] = nwu[j]; gu[j] = ngu[j]; }
        fetch(k >= 1 ? k - 1 : 0);       // the next node's operands travel while this node computes (branch-free)
        // ---- A: L D L^T of M = D + P_{k+1}; L and 1/d end up replicated in every lane ----
        double m[7], L[21], rd[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) m[r] = pc[r] + ((mat && r == c) ? D[r] : 0.0);
        SFOR(p, 7)
            rd[p] = rcp_pos(bc<p>(m[p]));
            const double mp = m[p];                       // M[p][c] = M[c][p]
            SFOR(q, 6 - p)
                constexpr int r = p + 1 + q;
                const double lrp = bc<p>(m[r]) * rd[p];   // L[r][p]
                L[r * (r - 1) / 2 + p] = lrp;
                m[r] = fma(-lrp, mp, m[r]);
            SEND
        SEND
        if (live && l == 7) {       // the factors of M for the forward sweep (the spare lane of the row stores them)
            double *lo = a.Lrd + ((size_t)sat * K + k) * 28;
#pragma unroll
            for (int e = 0; e < 21; ++e) lo[e] = L[e];
#pragma unroll
            for (int r = 0; r < 7; ++r) lo[21 + r] = rd[r];
        }
        // ---- B: X1 = L^-1 P_{k+1}[:, c] ; X2 = L^-1 e_c (matrix lanes) / L^-1 (rho + p_{k+1}) (channel lanes) ----
        double x1[7], x2[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) { x1[r] = pc[r]; x2[r] = mat ? ((r == c) ? 1.0 : 0.0) : rho[r] + pc[r]; }
#pragma unroll
        for (int r = 1; r < 7; ++r)
#pragma unroll
            for (int q = 0; q < r; ++q) { x1[r] = fma(-L[r * (r - 1) / 2 + q], x1[q], x1[r]); x2[r] = fma(-L[r * (r - 1) / 2 + q], x2[q], x2[r]); }
        // ---- C: Pt = P - X1^T R X1 ; G = X1^T R X2 (channel lanes: G (rho + p+)) on one set of broadcasts of X1 ----
        double w1[7], w2[7], pt[7], g[7];
#pragma unroll
        for (int p = 0; p < 7; ++p) { w1[p] = rd[p] * x1[p]; w2[p] = rd[p] * x2[p]; }
        SFOR(r, 7)
            double s1 = 0.0, s2 = 0.0;
            SFOR(p, 7)
                const double b = bc<r>(x1[p]);            // X1[p][r]
                s1 = fma(b, w1[p], s1); s2 = fma(b, w2[p], s2);
            SEND
            pt[r] = pc[r] - s1; g[r] = s2;
        SEND
        // ---- E: T = Pt [A | Bh] (matrix lanes) / Pt aff (channel lanes) on one set of broadcasts of Pt ----
        double e[7], ta[7], tb[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) e[r] = mat ? A[r] : aff[r];
        SFOR(r, 7)
            double s1 = 0.0, s2 = 0.0;
            SFOR(q, 7)
                const double b = bc<q>(pt[r]);            // Pt[r][q]
                s1 = fma(b, e[q], s1); s2 = fma(b, bh[q], s2);
            SEND
            ta[r] = s1; tb[r] = s2;
        SEND
        // channel lanes: t = p+ - G (rho + p+) + Pt aff ; from here on `ta` is TA[:, c] or t
#pragma unroll
        for (int r = 0; r < 7; ++r) ta[r] = mat ? ta[r] : pc[r] - g[r] + ta[r];
        // ---- F: Quy0 = Bpm^T Wx (matrix) / Bpm^T gx (channel); Quu0 = Quy0 Bpm; G: + Bh^T TA / Bh^T t; Quu += Bh^T TB ----
        double f[7], q[3] = {0, 0, 0}, quu[3];
#pragma unroll
        for (int r = 0; r < 7; ++r) f[r] = mat ? wx[r] : gx[r];
        SFOR(j, 3)
            SFOR(r, 7)
                q[j] = fma(bc<j>(bpm[r]), f[r], q[j]);
            SEND
        SEND
#pragma unroll
        for (int j = 0; j < 3; ++j) quu[j] = wu[j];
        SFOR(cc, 7)
            SFOR(j, 3)
                quu[j] = fma(bc<cc>(q[j]), bpm[cc], quu[j]);   // Quy0[j][cc] Bpm[cc][i]
            SEND
        SEND
        SFOR(j, 3)
            SFOR(r, 7)
                const double b = bc<j>(bh[r]);
                q[j] = fma(b, ta[r], q[j]); quu[j] = fma(b, tb[r], quu[j]);
            SEND
        SEND
#pragma unroll
        for (int j = 0; j < 3; ++j) q[j] += chn ? gu[j] : 0.0;
        // ---- H: Quu to every lane, inverse through its L D L^T ----
        double Q[6], Qi[6];
        Q[0] = bc<0>(quu[0]); Q[1] = bc<0>(quu[1]); Q[2] = bc<1>(quu[1]); Q[3] = bc<0>(quu[2]); Q[4] = bc<1>(quu[2]); Q[5] = bc<2>(quu[2]);
        {
            const double i0 = rcp_pos(Q[0]), l10 = Q[1] * i0, l20 = Q[3] * i0;
            const double d1 = fma(-l10, Q[1], Q[2]), i1 = rcp_pos(d1);
            const double l21 = fma(-l20, Q[1], Q[4]) * i1;
            const double d2 = fma(-l21 * l21, d1, fma(-l20, Q[3], Q[5])), i2 = rcp_pos(d2);
            // Quu^-1 = L^-T D^-1 L^-1, L^-1 = [[1,0,0],[-l10,1,0],[l10 l21 - l20, -l21, 1]]
            const double a20 = fma(l10, l21, -l20);
            Qi[5] = i2; Qi[4] = -l21 * i2; Qi[3] = a20 * i2;
            Qi[2] = fma(l21 * l21, i2, i1); Qi[1] = fma(-l10, i1, -l21 * a20 * i2);
            Qi[0] = fma(l10 * l10, i1, fma(a20 * a20, i2, i0));
        }
        // ---- I: Kg = Quu^-1 Quy (matrix) / w = Quu^-1 qu (channel) ----
        double kg[3];
        kg[0] = Qi[0] * q[0] + Qi[1] * q[1] + Qi[3] * q[2];
        kg[1] = Qi[1] * q[0] + Qi[2] * q[1] + Qi[4] * q[2];
        kg[2] = Qi[3] * q[0] + Qi[4] * q[1] + Qi[5] * q[2];
        // ---- J, K, L: new column / vector = f + A^T (TA | t) - Quy^T (Kg | w) ----
        SFOR(r, 7)
            double s = f[r];
            SFOR(qq, 7)
                s = fma(bc<r>(A[qq]), ta[qq], s);         // A[qq][r]
            SEND
            SFOR(j, 3)
                s = fma(-bc<r>(q[j]), kg[j], s);          // Quy[j][r]
            SEND
            pc[r] = s;
        SEND
        // ---- results of the node ----
        if (live) {
            if (mat) {
#pragma unroll
                for (int r = 0; r < 7; ++r) a.P[((size_t)sat * K + k) * 49 + r * 7 + c] = pc[r];
#pragma unroll
                for (int j = 0; j < 3; ++j) a.Kg[((size_t)sat * K + k) * 21 + j * 7 + c] = kg[j];
                if (l == 0) {
                    double *qo = a.Qi + ((size_t)sat * K + k) * 9;
                    qo[0] = Qi[0]; qo[1] = Qi[1]; qo[2] = Qi[3]; qo[3] = Qi[1]; qo[4] = Qi[2]; qo[5] = Qi[4]; qo[6] = Qi[3]; qo[7] = Qi[4]; qo[8] = Qi[5];
                }
            } else if (chn) {
#pragma unroll
                for (int r = 0; r < 7; ++r) a.p[(((size_t)sat * K + k) * NCH + ch) * 7 + r] = pc[r];
#pragma unroll
                for (int j = 0; j < 3; ++j) a.qu[(((size_t)sat * K + k) * NCH + ch) * 3 + j] = q[j];
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) a.cycles[blockIdx.x] = t1 - t0;
}

extern "C" int packed_probe_run(int S, int K, const double *rec, const double *chin, double *P, double *Kg, double *Qi, double *Lrd,
                                double *p, double *qu, long long *cycles, int reps, float *ms_out)
{
    ProbeArgs a{S, K, rec, chin, P, Kg, Qi, Lrd, p, qu, cycles};
    const int blocks = (S + 3) / 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(packed_probe_kernel, dim3(blocks), dim3(64), 0, 0, a);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(packed_probe_kernel, dim3(blocks), dim3(64), 0, 0, a);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return 1;
    hipEventElapsedTime(ms_out, e0, e1);
    *ms_out /= (float)reps;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// Forward sweep of the 8 channels of four satellites: the channel lanes carry y, u, x, nu, lam of their channel in registers,
// the matrix lanes only lend their columns (Kg, Bpm, A, Bh, P_{k+1}) to the broadcasts; nu = -L^-T R L^-1 (P_{k+1} yhat + rho
// + p_{k+1}) from the stored factors of M (no G, no M^-1 in the record: 104 doubles per node instead of 198).
struct FwdArgs {
    int S, K;
    const double *rec, *chin, *P, *Kg, *Qi, *Lrd, *p, *qu;
    double *X, *U, *NU, *LAM;      // [S][K][NCH][7|3|7|7]
    long long *cycles;
};

__global__ __launch_bounds__(64, PROBE_WAVES) void packed_forward_kernel(FwdArgs a)
{
    const int lane = threadIdx.x, l = lane & 15;
    int sat = blockIdx.x * 4 + (lane >> 4);
    const bool live = sat < a.S;
    if (!live) sat = a.S - 1;
    const int K = a.K;
    const bool mat = l < 7, chn = l >= 8;
    const int c = mat ? l : 0, c3 = (l < 3) ? l : 0, ch = chn ? l - 8 : 0;
    double y[7] = {0, 0, 0, 0, 0, 0, 0};
    double nA[7], nbh[7], nbpm[7], npn[7], nkg[3], nD[7], nL[21], nrd[7], nQi[9], nrho[7], naff[7], npp[7], nqu[3];
    auto fetch = [&](int k) __attribute__((always_inline)) {
        const size_t nk = (size_t)sat * K + k;
        const double *rk = a.rec + nk * REC_N;
        const double *ck = a.chin + (nk * NCH + ch) * CH_IN;
        const size_t nk1 = (k <= K - 2) ? nk + 1 : nk;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            nA[r] = rk[R_A + r * 7 + c]; nbh[r] = rk[R_BH + r * 3 + c3]; nbpm[r] = rk[R_BPM + r * 3 + c3];
            npn[r] = a.P[nk1 * 49 + r * 7 + c];
            nD[r] = rk[R_D + r]; nrd[r] = a.Lrd[nk * 28 + 21 + r];
            nrho[r] = ck[10 + r]; naff[r] = ck[17 + r];
            npp[r] = a.p[(nk1 * NCH + ch) * 7 + r];
        }
#pragma unroll
        for (int e = 0; e < 21; ++e) nL[e] = a.Lrd[nk * 28 + e];
#pragma unroll
        for (int e = 0; e < 9; ++e) nQi[e] = a.Qi[nk * 9 + e];
#pragma unroll
        for (int j = 0; j < 3; ++j) { nkg[j] = a.Kg[nk * 21 + j * 7 + c]; nqu[j] = a.qu[(nk * NCH + ch) * 3 + j]; }
    };
    fetch(0);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < K; ++k) {
        const size_t nk = (size_t)sat * K + k;
        const bool dyn = k <= K - 2;
        double A[7], bh[7], bpm[7], pn[7], kg[3], D[7], L[21], rd[7], Qi[9], rho[7], aff[7], pp[7], qu[3];
#pragma unroll
        for (int r = 0; r < 7; ++r) { A[r] = nA[r]; bh[r] = nbh[r]; bpm[r] = nbpm[r]; pn[r] = dyn ? npn[r] : 0.0; D[r] = nD[r]; rd[r] = nrd[r]; rho[r] = nrho[r]; aff[r] = naff[r]; pp[r] = dyn ? npp[r] : 0.0; }
#pragma unroll
        for (int e = 0; e < 21; ++e) L[e] = nL[e];
#pragma unroll
        for (int e = 0; e < 9; ++e) Qi[e] = nQi[e];
#pragma unroll
        for (int j = 0; j < 3; ++j) { kg[j] = nkg[j]; qu[j] = nqu[j]; }
        fetch(k + 1 < K ? k + 1 : K - 1);       // branch-free prefetch of the next node
        // u = -Kg y - Quu^-1 qu
        double u[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) u[j] = -(Qi[j * 3] * qu[0] + Qi[j * 3 + 1] * qu[1] + Qi[j * 3 + 2] * qu[2]);
        SFOR(cc, 7)
            SFOR(j, 3)
                u[j] = fma(-bc<cc>(kg[j]), y[cc], u[j]);
            SEND
        SEND
        // x = y + Bpm u ; yhat = A y + Bh u + aff
        double x[7], yh[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) { x[r] = y[r]; yh[r] = aff[r]; }
        SFOR(j, 3)
            SFOR(r, 7)
                x[r] = fma(bc<j>(bpm[r]), u[j], x[r]); yh[r] = fma(bc<j>(bh[r]), u[j], yh[r]);
            SEND
        SEND
        SFOR(cc, 7)
            SFOR(r, 7)
                yh[r] = fma(bc<cc>(A[r]), y[cc], yh[r]);
            SEND
        SEND
        // z = L^-1 (P_{k+1} yhat + rho + p_{k+1}) ; nu = -L^-T R z
        double z[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) z[r] = rho[r] + pp[r];
        SFOR(cc, 7)
            SFOR(r, 7)
                z[r] = fma(bc<cc>(pn[r]), yh[cc], z[r]);
            SEND
        SEND
#pragma unroll
        for (int r = 1; r < 7; ++r)
#pragma unroll
            for (int q = 0; q < r; ++q) z[r] = fma(-L[r * (r - 1) / 2 + q], z[q], z[r]);
#pragma unroll
        for (int r = 0; r < 7; ++r) z[r] *= -rd[r];
#pragma unroll
        for (int q = 6; q >= 1; --q)
#pragma unroll
            for (int r = 0; r < q; ++r) z[r] = fma(-L[q * (q - 1) / 2 + r], z[q], z[r]);
        if (live && chn) {
            double *xo = a.X + (nk * NCH + ch) * 7, *uo = a.U + (nk * NCH + ch) * 3, *no = a.NU + (nk * NCH + ch) * 7, *lo = a.LAM + (nk * NCH + ch) * 7;
#pragma unroll
            for (int r = 0; r < 7; ++r) { xo[r] = x[r]; no[r] = dyn ? z[r] : 0.0; lo[r] = dyn ? fma(D[r], z[r], rho[r]) : 0.0; }
#pragma unroll
            for (int j = 0; j < 3; ++j) uo[j] = u[j];
        }
#pragma unroll
        for (int r = 0; r < 7; ++r) y[r] = dyn ? yh[r] + z[r] : y[r];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) a.cycles[blockIdx.x] = t1 - t0;
}

extern "C" int packed_forward_run(int S, int K, const double *rec, const double *chin, const double *P, const double *Kg, const double *Qi,
                                  const double *Lrd, const double *p, const double *qu, double *X, double *U, double *NU, double *LAM,
                                  long long *cycles, int reps, float *ms_out)
{
    FwdArgs a{S, K, rec, chin, P, Kg, Qi, Lrd, p, qu, X, U, NU, LAM, cycles};
    const int blocks = (S + 3) / 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(packed_forward_kernel, dim3(blocks), dim3(64), 0, 0, a);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(packed_forward_kernel, dim3(blocks), dim3(64), 0, 0, a);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return 1;
    hipEventElapsedTime(ms_out, e0, e1);
    *ms_out /= (float)reps;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
