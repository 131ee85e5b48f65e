/*
 * oracle/dynamics.c -- TEST INFRASTRUCTURE (see mpc_oracle.h).
 * CPU restatement of the orbital dynamics, its Jacobians and the thrust laws.
 * Operation order follows the reference expressions so results agree to rounding.
 */
#include <math.h>
#include <float.h>
#include <string.h>
#include "mpc_oracle.h"

#define C_D_CONST 2.5          /* constants.py:7 */
#define RHO_500KM 9.983E-13    /* simulator.py:112 (fixed density) */

static double norm3(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

static void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* simulator.py:116-161.  The thrust u = u_func(y, tau) is evaluated by the caller. */
int oracle_dynamics(const double y[7], const double u[3], double tf, const double *cst,
                    int flags, double ydot[7])
{
    const double *r = y, *v = y + 3;
    double m = y[6];
    int bad = (m <= 0.0);                                  /* :135-136 raises */
    double r_norm = norm3(r);                              /* :139 */
    double k_g = -cst[OC_MU] / pow(r_norm, 3.0);           /* :145 */
    double yd[7];
    for (int i = 0; i < 3; ++i) {
        yd[i] = v[i];                                      /* :142 */
        yd[3 + i] = k_g * r[i] + u[i] / m;                 /* :145-149 */
    }
    if (flags & ORACLE_FLAG_DRAG) {                        /* :150-153 */
        double coef = -1.0 / 2.0 * C_D_CONST * cst[OC_S] * (1.0 / m) * (RHO_500KM / cst[OC_RHO])
                      * norm3(v);
        for (int i = 0; i < 3; ++i) yd[3 + i] += coef * v[i];
    }
    if (flags & ORACLE_FLAG_J2) {                          /* :154-158 */
        double q = r[2] / r_norm;
        double q2 = q * q;
        double d[3] = {5.0 * q2 - 1.0, 5.0 * q2 - 1.0, 5.0 * q2 - 3.0};
        double coef = 1.5 * cst[OC_J2] * cst[OC_MU] * (cst[OC_R_E] * cst[OC_R_E]) / pow(r_norm, 5.0);
        for (int i = 0; i < 3; ++i) yd[3 + i] += coef * (d[i] * r[i]);
    }
    yd[6] = -norm3(u) / (cst[OC_G0] * cst[OC_ISP]);        /* :160 */
    for (int i = 0; i < 7; ++i) ydot[i] = tf * yd[i];      /* :161 */
    return bad;
}

/* linearize_discretize.py:119-183 (include_drag branch :162-169 is dead in the reference:
 * Constants has no CD and rho_func defaults to None, so it is out of scope). */
void oracle_A_func(const double x[7], const double u[3], double tf, const double *cst,
                   int flags, double A[49])
{
    double D[49];
    memset(D, 0, sizeof D);
    const double *r = x;
    double m = x[6];
    double r_norm = norm3(r);
    double c1 = -cst[OC_MU] / pow(r_norm, 3.0);            /* :146 */
    double c2 = 3.0 * cst[OC_MU] / pow(r_norm, 5.0);       /* :147 */
    for (int i = 0; i < 3; ++i) {
        D[i * 7 + 3 + i] = 1.0;                            /* :177 */
        for (int j = 0; j < 3; ++j)
            D[(3 + i) * 7 + j] = (i == j ? c1 : 0.0) + c2 * (r[i] * r[j]);
    }
    if (flags & ORACLE_FLAG_J2) {                          /* :149-158 */
        double kJ2 = 1.5 * cst[OC_J2] * cst[OC_MU] * (cst[OC_R_E] * cst[OC_R_E]);
        double q = r[2] / r_norm;
        double q2 = q * q;
        double g[3] = {5.0 * q2 - 1.0, 5.0 * q2 - 1.0, 5.0 * q2 - 3.0};
        double r4 = pow(r_norm, 4.0), r2 = r_norm * r_norm, r5 = pow(r_norm, 5.0),
               r7 = pow(r_norm, 7.0);
        double ddr[3];
        for (int j = 0; j < 3; ++j) ddr[j] = 5.0 * (r[2] * r[2]) * (-2.0 * (r[j] / r4));
        ddr[2] += (5.0 / r2) * (2.0 * r[2]);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double t1 = ((kJ2 * g[i]) * r[i]) * (-5.0 * r[j] / r7);
                double t2 = kJ2 / r5 * (r[i] * ddr[j]);
                double t3 = (i == j) ? kJ2 / r5 * g[i] : 0.0;
                D[(3 + i) * 7 + j] += t1 + t2 + t3;
            }
    }
    for (int i = 0; i < 3; ++i) D[(3 + i) * 7 + 6] = -u[i] / (m * m);   /* :175 */
    for (int i = 0; i < 49; ++i) A[i] = tf * D[i];                       /* :182 */
}

/* linearize_discretize.py:186-215 */
void oracle_B_func(const double x[7], const double u[3], double tf, const double *cst, double B[21])
{
    double m = x[6];
    double D[21];
    memset(D, 0, sizeof D);
    for (int i = 0; i < 3; ++i) D[(3 + i) * 3 + i] = 1.0 / m;            /* :203 */
    double nT = norm3(u);
    if (!(nT <= DBL_EPSILON)) {                                          /* :208-211 */
        double den = cst[OC_G0] * cst[OC_ISP] * nT;
        for (int j = 0; j < 3; ++j) D[6 * 3 + j] = -u[j] / den;
    }
    for (int i = 0; i < 21; ++i) B[i] = tf * D[i];                       /* :214 */
}

/* linearize_discretize.py:218-236 */
void oracle_xi_func(const double x[7], const double u[3], double tf, const double *cst, int flags,
                    double xi[7])
{
    double A[49], B[21];
    oracle_A_func(x, u, tf, cst, flags, A);
    oracle_B_func(x, u, tf, cst, B);
    for (int i = 0; i < 7; ++i) {
        double ax = 0.0, bu = 0.0;
        for (int j = 0; j < 7; ++j) ax += A[i * 7 + j] * x[j];
        for (int j = 0; j < 3; ++j) bu += B[i * 3 + j] * u[j];
        xi[i] = -(ax + bu);                                              /* :235 */
    }
}

/* Python/numpy float floor division (CPython float_floor_div, numpy npy_divmod). */
static double py_floordiv(double a, double b)
{
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0) {
        if ((b < 0) != (mod < 0)) div -= 1.0;
    }
    if (div != 0.0) {
        double fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
        return fl;
    }
    return copysign(0.0, a / b);
}

/* linearize_discretize.py:294-315 (and control.py:104-126, same code) */
int oracle_u_foh(double tau, const double *u, int Ku, double out[3])
{
    if (tau == 1.0) {                                                    /* :305-306 */
        for (int i = 0; i < 3; ++i) out[i] = u[i * Ku + Ku - 1];
        return 0;
    }
    double dtau = 1.0 / (double)(Ku - 1);                                /* :309 */
    int k = (int)py_floordiv(tau, dtau);                                 /* :310 */
    if (k < 0 || k + 1 >= Ku) return -1;                                 /* IndexError in the reference */
    double tau_k = (double)k / (double)(Ku - 1);
    double tau_kp1 = (double)(k + 1) / (double)(Ku - 1);
    double lam_n = (tau_kp1 - tau) / (tau_kp1 - tau_k);
    double lam_p = (tau - tau_k) / (tau_kp1 - tau_k);
    for (int i = 0; i < 3; ++i) out[i] = lam_n * u[i * Ku + k] + lam_p * u[i * Ku + k + 1];
    return 0;
}

/* control.py thrust laws u(x, tau) */
int oracle_ctrl_eval(const oracle_ctrl *c, const double y[7], double tau, double u[3])
{
    switch (c->kind) {
    case ORACLE_CTRL_CONSTANT:
        u[0] = c->thrust[0]; u[1] = c->thrust[1]; u[2] = c->thrust[2];
        return 0;
    case ORACLE_CTRL_TANGENTIAL: {                                       /* control.py:66-84 */
        const double *r = y, *v = y + 3;
        double rn = norm3(r), h[3], t[3], rh[3], hh[3];
        cross3(r, v, h);
        double hn = norm3(h);
        for (int i = 0; i < 3; ++i) { rh[i] = r[i] / rn; hh[i] = h[i] / hn; }
        cross3(hh, rh, t);
        /* R @ [0, T, 0] with R = [r_hat t_hat h_hat] */
        for (int i = 0; i < 3; ++i) u[i] = rh[i] * 0.0 + t[i] * c->thrust[0] + hh[i] * 0.0;
        return 0;
    }
    case ORACLE_CTRL_SEQUENCE:                                           /* control.py:132-142 */
        if (tau <= c->end_tau) return oracle_u_foh(tau / c->end_tau, c->useq, c->Ku, u);
        u[0] = u[1] = u[2] = 0.0;
        return 0;
    default:
        u[0] = u[1] = u[2] = 0.0;
        return 0;
    }
}

/* linearize_discretize.py:393-411 */
void oracle_extract_uk(int K, const double *x, const double *t, const oracle_ctrl *ctrl, double *u)
{
    for (int k = 0; k < K; ++k) {
        double y[7], uk[3];
        for (int i = 0; i < 7; ++i) y[i] = x[i * K + k];
        oracle_ctrl_eval(ctrl, y, t[k], uk);
        for (int i = 0; i < 3; ++i) u[i * K + k] = uk[i];
    }
}

/* satellite_scale.py:28-44 + constants.py:1-8 */
void oracle_scale(const double state[7], double sc[7], double cst[OC_NCONST])
{
    const double MU_EARTH = 3.986004418E14, R_EARTH = 6.371E6, J2c = 1.08262668E-3, G0c = 9.80665,
                 ISPc = 500.0, Sc = 55.44;
    double r0 = norm3(state);
    double s0 = 2.0 * M_PI * sqrt(pow(r0, 3.0) / MU_EARTH);
    double v0 = r0 / s0;
    double a0 = r0 / (s0 * s0);
    double m0 = state[6];
    double T0 = m0 * r0 / (s0 * s0);
    double mu0 = pow(r0, 3.0) / (s0 * s0);
    sc[0] = r0; sc[1] = s0; sc[2] = v0; sc[3] = a0; sc[4] = m0; sc[5] = T0; sc[6] = mu0;
    cst[OC_MU] = MU_EARTH / mu0;
    cst[OC_R_E] = R_EARTH / r0;
    cst[OC_J2] = J2c;
    cst[OC_G0] = G0c / a0;
    cst[OC_ISP] = ISPc / s0;
    cst[OC_S] = Sc / (r0 * r0);
    cst[OC_R0] = r0;
    cst[OC_RHO] = m0 / pow(r0, 3.0);
}
