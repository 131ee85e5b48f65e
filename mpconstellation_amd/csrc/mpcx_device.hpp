// mpcx_device.hpp -- device-side building blocks shared by the gfx950 kernels.
// fp64 throughout.  Written for CDNA4 (wave64); no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mpcx.h"

namespace mpcx {

constexpr double kCd = 2.5;             // constants.py:7
constexpr double kRho500 = 9.983E-13;   // simulator.py:112
constexpr double kEps = 2.220446049250313e-16;

// Dormand-Prince 5(4) tableau as used by scipy's RK45 (scipy/integrate/_ivp/rk.py:377-404)
__device__ constexpr double RK_C[6] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0};
__device__ constexpr double RK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
__device__ constexpr double RK_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784,
                                       11.0 / 84};
__device__ constexpr double RK_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920,
                                       17253.0 / 339200, -22.0 / 525, 1.0 / 40};
__device__ constexpr double RK_P[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408,
     701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};
constexpr double RK_SAFETY = 0.9, RK_MIN_FACTOR = 0.2, RK_MAX_FACTOR = 10.0;
// scipy's RK23 (Bogacki-Shampine 3(2), rk.py:183-278; Discretizer.ivp_solver = 'RK23'): C = (0, 1/2, 3/4), A, B, E, P (4 x 3)
__device__ constexpr double RK23_C[3] = {0.0, 1.0 / 2, 3.0 / 4};
__device__ constexpr double RK23_A10 = 1.0 / 2, RK23_A21 = 3.0 / 4;      // (A[2][0] = 0)
__device__ constexpr double RK23_B[3] = {2.0 / 9, 1.0 / 3, 4.0 / 9};
__device__ constexpr double RK23_E[4] = {5.0 / 72, -1.0 / 12, -1.0 / 9, 1.0 / 8};
__device__ constexpr double RK23_P[4][3] = {{1, -4.0 / 3, 5.0 / 9}, {0, 1, -2.0 / 3}, {0, 4.0 / 3, -8.0 / 9}, {0, -1, 1}};

// Python / numpy float floor division (the `tau // dtau` of linearize_discretize.py:310)
__device__ __forceinline__ double py_floordiv(double a, double b)
{
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0 && ((b < 0) != (mod < 0))) div -= 1.0;
    if (div != 0.0) {
        double fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
        return fl;
    }
    return copysign(0.0, a / b);
}

// First-order hold of a (3,Ku) row-major table: linearize_discretize.py:294-315, control.py:104-126
// (ld: row length of the table in memory, Ku <= ld of its columns in use -- they differ only in ragged batches)
__device__ __forceinline__ void foh3(double tau, const double *__restrict__ u, int Ku, int ld, double (&out)[3],
                                     int &err)
{
    if (tau == 1.0) {
        out[0] = u[Ku - 1]; out[1] = u[ld + Ku - 1]; out[2] = u[2 * ld + Ku - 1];
        return;
    }
    const double km1 = (double)(Ku - 1);
    const double dtau = 1.0 / km1;
    int k = (int)py_floordiv(tau, dtau);
    if (k < 0 || k + 1 >= Ku) {          // the reference raises IndexError here
        err = MPCX_ST_FOH;
        k = k < 0 ? 0 : Ku - 2;
        if (Ku < 2) { out[0] = out[1] = out[2] = 0.0; return; }
    }
    const double tau_k = (double)k / km1, tau_kp1 = (double)(k + 1) / km1;
    const double lam_n = (tau_kp1 - tau) / (tau_kp1 - tau_k);
    const double lam_p = (tau - tau_k) / (tau_kp1 - tau_k);
    out[0] = lam_n * u[k] + lam_p * u[k + 1];
    out[1] = lam_n * u[ld + k] + lam_p * u[ld + k + 1];
    out[2] = lam_n * u[2 * ld + k] + lam_p * u[2 * ld + k + 1];
}

// The same first-order hold with the interval in use kept in registers: the two table columns, the node index and the
// interval's end points are reloaded / recomputed only when tau leaves the cached interval (an RK45 step is short
// against 1/(Ku-1): the stages of a step and many steps in a row stay inside one interval).  Inside -- away from both
// ends by more than any rounding of the index computation -- k is certain and the weights are foh3's up to one rounding
// (multiplication by the interval's cached reciprocal length instead of two divisions); otherwise -- node times included --
// foh3's own index computation and expressions decide.  Saves the fmod of py_floordiv, two divisions and six dependent
// global loads per right-hand side.
struct FohCache {
    int k;
    double tau_k, tau_kp1, inv, uk[3], uk1[3];
    __device__ __forceinline__ void reset() { k = -1; tau_k = 2.0; tau_kp1 = -1.0; inv = 0.0; }
};

__device__ __forceinline__ void foh3_cached(double tau, const double *__restrict__ u, int Ku, int ld, FohCache &c, double (&out)[3],
                                            int &err)
{
    if (!(tau > c.tau_k + 1e-12 && tau < c.tau_kp1 - 1e-12)) {      // rare: everything but "same interval" is behind this branch
        if (tau == 1.0) {
            out[0] = u[Ku - 1]; out[1] = u[ld + Ku - 1]; out[2] = u[2 * ld + Ku - 1];
            return;
        }
        const double km1 = (double)(Ku - 1);
        const double dtau = 1.0 / km1;
        int k = (int)py_floordiv(tau, dtau);
        if (k < 0 || k + 1 >= Ku) {          // the reference raises IndexError here
            err = MPCX_ST_FOH;
            k = k < 0 ? 0 : Ku - 2;
            if (Ku < 2) { out[0] = out[1] = out[2] = 0.0; return; }
        }
        if (k != c.k) {
            c.k = k;
            c.tau_k = (double)k / km1; c.tau_kp1 = (double)(k + 1) / km1;
#pragma unroll
            for (int i = 0; i < 3; ++i) { c.uk[i] = u[i * ld + k]; c.uk1[i] = u[i * ld + k + 1]; }
            c.inv = 1.0 / (c.tau_kp1 - c.tau_k);
        }
        const double lam_n = (c.tau_kp1 - tau) / (c.tau_kp1 - c.tau_k);
        const double lam_p = (tau - c.tau_k) / (c.tau_kp1 - c.tau_k);
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i] = lam_n * c.uk[i] + lam_p * c.uk1[i];
        return;
    }
    const double lam_n = (c.tau_kp1 - tau) * c.inv, lam_p = (tau - c.tau_k) * c.inv;
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = lam_n * c.uk[i] + lam_p * c.uk1[i];
}

// bcast8 for a 32-bit integer
template <int Q>
__device__ __forceinline__ int bcast8i(int v)
{
    const int t = __builtin_amdgcn_update_dpp(0, v, 0x150 + Q, 0xF, 0xF, true);
    return __builtin_amdgcn_update_dpp(t, v, 0x150 + 8 + Q, 0xF, 0xC, false);
}

// Value of lane q of the caller's 8-lane group in all 8 lanes (q a compile-time constant): two v_mov_b64_dpp -- gfx950 has
// the 64-bit DPP move for row_newbcast -- lane q of every 16-lane row into the whole row, then lane 8 + q over the row's
// upper half under a bank mask.  VALU only: no ds_bpermute, no LDS round trip.  Checked against __shfl for every q by
// profiles/tools/dpp_bcast_check.hip.
template <int Q>
__device__ __forceinline__ double bcast8(double v)
{
    const long long b = __double_as_longlong(v);
    const long long t = __builtin_amdgcn_update_dpp((long long)0, b, 0x150 + Q, 0xF, 0xF, true);
    return __longlong_as_double(__builtin_amdgcn_update_dpp(t, b, 0x150 + 8 + Q, 0xF, 0xC, false));
}

// Sum over the caller's 8-lane group, bitwise identical on its 8 lanes: neighbours inside the quad by quad_perm, then the
// other quad by row_half_mirror (lane i <-> 7 - i; both quads hold their quad's sum by then and a + b = b + a bit for bit).
template <int CTRL>
__device__ __forceinline__ double dpp64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double group_sum8(double v)
{
    v += dpp64<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp64<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp64<0x141>(v);     // row_half_mirror
    return v;
}

struct SatConst {
    double mu, re, j2, g0, isp, s, r0, rho;
    __device__ __forceinline__ void load(const double *__restrict__ c)
    {
        mu = c[MPCX_C_MU]; re = c[MPCX_C_R_E]; j2 = c[MPCX_C_J2]; g0 = c[MPCX_C_G0];
        isp = c[MPCX_C_ISP]; s = c[MPCX_C_S]; r0 = c[MPCX_C_R0]; rho = c[MPCX_C_RHO];
    }
};

// Simulator.satellite_dynamics (simulator.py:116-161) for a given thrust u; result NOT yet
// multiplied by tf (the caller scales, so Sigma_func's tf=1 evaluation reuses it).
__device__ __forceinline__ void dynamics_unscaled(const double (&y)[7], const double (&u)[3],
                                                  const SatConst &c, int flags, double (&yd)[7])
{
    const double r2 = y[0] * y[0] + y[1] * y[1] + y[2] * y[2];
    const double rn = sqrt(r2);
    const double r3 = rn * rn * rn;
    const double kg = -c.mu / r3;
    const double m = y[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        yd[i] = y[3 + i];
        yd[3 + i] = kg * y[i] + u[i] / m;
    }
    if (flags & MPCX_FLAG_DRAG) {
        const double vn = sqrt(y[3] * y[3] + y[4] * y[4] + y[5] * y[5]);
        const double coef = -0.5 * kCd * c.s * (1.0 / m) * (kRho500 / c.rho) * vn;
#pragma unroll
        for (int i = 0; i < 3; ++i) yd[3 + i] += coef * y[3 + i];
    }
    if (flags & MPCX_FLAG_J2) {
        const double q = y[2] / rn, q2 = q * q;
        const double r5 = r3 * rn * rn;
        const double coef = 1.5 * c.j2 * c.mu * (c.re * c.re) / r5;
        yd[3] += coef * ((5.0 * q2 - 1.0) * y[0]);
        yd[4] += coef * ((5.0 * q2 - 1.0) * y[1]);
        yd[5] += coef * ((5.0 * q2 - 3.0) * y[2]);
    }
    yd[6] = -sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) / (c.g0 * c.isp);
}

// The dense part of Dxf (linearize_discretize.py:144-179): G = d a / d r (3x3) and
// gm = d a / d m (3); rows 0-2 of Dxf are [0 I 0], row 6 is zero.  Not yet multiplied by tf.
__device__ __forceinline__ void jacobian_blocks(double rx, double ry, double rz, double m,
                                                const double (&u)[3], const SatConst &c, int flags,
                                                double (&G)[3][3], double (&gm)[3])
{
    const double r[3] = {rx, ry, rz};
    const double r2 = rx * rx + ry * ry + rz * rz;
    const double rn = sqrt(r2);
    const double r3 = rn * rn * rn;
    const double r5 = r3 * rn * rn;
    const double c1 = -c.mu / r3;
    const double c2 = 3.0 * c.mu / r5;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) G[i][j] = (i == j ? c1 : 0.0) + c2 * (r[i] * r[j]);
    if (flags & MPCX_FLAG_J2) {
        const double kJ2 = 1.5 * c.j2 * c.mu * (c.re * c.re);
        const double q = rz / rn, q2 = q * q;
        const double g[3] = {5.0 * q2 - 1.0, 5.0 * q2 - 1.0, 5.0 * q2 - 3.0};
        const double r4 = r2 * r2, r7 = r5 * r2;
        double ddr[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) ddr[j] = 5.0 * (rz * rz) * (-2.0 * (r[j] / r4));
        ddr[2] += (5.0 / r2) * (2.0 * rz);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double t = ((kJ2 * g[i]) * r[i]) * (-5.0 * r[j] / r7) + kJ2 / r5 * (r[i] * ddr[j]);
                if (i == j) t += kJ2 / r5 * g[i];
                G[i][j] += t;
            }
    }
    const double m2 = m * m;
#pragma unroll
    for (int i = 0; i < 3; ++i) gm[i] = -u[i] / m2;
}

}  // namespace mpcx
