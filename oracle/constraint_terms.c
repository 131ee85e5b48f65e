/*
 * oracle/constraint_terms.c -- TEST INFRASTRUCTURE (see mpc_oracle.h).
 * CPU restatement of Optimizer.get_constraint_terms (reference optimizer.py:80-170) for one
 * satellite, including the reference's operator-precedence quirk in Dv_h_hat (:122) and the
 * inverted ubar_hat mask (:136-138).
 */
#include <math.h>
#include <float.h>
#include <string.h>
#include "mpc_oracle.h"

static double norm3(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static void skew(const double x[3], double S[9]) /* optimizer.py:41-45 */
{
    S[0] = 0; S[1] = -x[2]; S[2] = x[1];
    S[3] = x[2]; S[4] = 0; S[5] = -x[0];
    S[6] = -x[1]; S[7] = x[0]; S[8] = 0;
}
static void mm3(const double *A, const double *B, double *C)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int l = 0; l < 3; ++l) s += A[i * 3 + l] * B[l * 3 + j];
            C[i * 3 + j] = s;
        }
}
static void vm3(const double v[3], const double *M, double out[3]) /* v @ M */
{
    for (int j = 0; j < 3; ++j) out[j] = v[0] * M[j] + v[1] * M[3 + j] + v[2] * M[6 + j];
}
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

void oracle_constraint_terms(int K, const double *x, const double *u, double mu, double *rbar_hat,
                             double *ubar_hat, double T[CT_NTERMS])
{
    double r[3], v[3], h[3], r_hat[3], h_hat[3], t_hat[3];
    for (int i = 0; i < 3; ++i) { r[i] = x[i * K + K - 1]; v[i] = x[(3 + i) * K + K - 1]; }   /* :112-114 */
    double rn = norm3(r);
    cross3(r, v, h);
    double hn = norm3(h);
    for (int i = 0; i < 3; ++i) { r_hat[i] = r[i] / rn; h_hat[i] = h[i] / hn; }
    cross3(h_hat, r_hat, t_hat);                                                              /* :119 */

    double Ph[9], hh3[9], Sv[9], Sr[9], Srh[9], Shh[9], nSv[9];
    double ihn = pow(hn, -1.0), ihn3 = pow(hn, -3.0), irn = pow(rn, -1.0), irn3 = pow(rn, -3.0);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            hh3[i * 3 + j] = ihn3 * (h[i] * h[j]);
            Ph[i * 3 + j] = (i == j ? ihn : 0.0) - hh3[i * 3 + j];
        }
    skew(v, Sv); skew(r, Sr); skew(r_hat, Srh); skew(h_hat, Shh);
    for (int i = 0; i < 9; ++i) nSv[i] = -Sv[i];
    double Dr_h_hat[9], Dv_h_hat[9], tmp[9], Dr_r_hat[9], Dr_t_hat[9], Dv_t_hat[9], nSrh[9], a[9], b[9];
    mm3(Ph, nSv, Dr_h_hat);                                                                   /* :121 */
    mm3(hh3, Sr, tmp);                                                                        /* :122 (precedence as written) */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Dv_h_hat[i * 3 + j] = (i == j ? ihn : 0.0) - tmp[i * 3 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Dr_r_hat[i * 3 + j] = (i == j ? irn : 0.0) - irn3 * (r[i] * r[j]);   /* :123 */
    for (int i = 0; i < 9; ++i) nSrh[i] = -Srh[i];
    mm3(nSrh, Dr_h_hat, a); mm3(Shh, Dr_r_hat, b);
    for (int i = 0; i < 9; ++i) Dr_t_hat[i] = a[i] + b[i];                                    /* :124 */
    mm3(nSrh, Dv_h_hat, Dv_t_hat);                                                            /* :125 */

    if (rbar_hat)                                                                             /* :129-130 */
        for (int k = 0; k < K - 1; ++k) {
            double rk[3] = {x[k], x[K + k], x[2 * K + k]};
            double n = norm3(rk);
            for (int i = 0; i < 3; ++i) rbar_hat[i * (K - 1) + k] = rk[i] / n;
        }
    if (ubar_hat)                                                                             /* :133-139 */
        for (int k = 0; k < K; ++k) {
            double uk[3] = {u[k], u[K + k], u[2 * K + k]};
            double n = norm3(uk);
            for (int i = 0; i < 3; ++i) ubar_hat[i * K + k] = (n <= DBL_EPSILON) ? uk[i] / n : 0.0;
        }

    double rv[6] = {r[0], r[1], r[2], v[0], v[1], v[2]};
    for (int i = 0; i < 3; ++i) T[CT_RF_HAT + i] = r_hat[i];                                  /* :144 */
    T[CT_VC] = sqrt(mu / rn);                                                                 /* :146 */
    double cvc = (-1.0 / 2.0) * pow(mu, 0.5) * pow(rn, -5.0 / 2.0);
    for (int i = 0; i < 3; ++i) T[CT_DRVC + i] = cvc * r[i];                                  /* :147 */
    T[CT_DRVC_RBAR] = dot3(&T[CT_DRVC], r);
    T[CT_VT] = dot3(v, t_hat);                                                                /* :150 */
    double DrVt[3], DvVt[3], w[3];
    vm3(v, Dr_t_hat, DrVt);
    vm3(v, Dv_t_hat, w);
    for (int i = 0; i < 3; ++i) DvVt[i] = t_hat[i] + w[i];                                    /* :152 */
    for (int i = 0; i < 3; ++i) { T[CT_DVT + i] = DrVt[i]; T[CT_DVT + 3 + i] = DvVt[i]; }
    double s = 0.0; for (int i = 0; i < 6; ++i) s += T[CT_DVT + i] * rv[i];
    T[CT_DVT_BAR] = s;
    T[CT_VR] = dot3(v, r_hat);                                                                /* :157 */
    double DrVr[3]; vm3(v, Dr_r_hat, DrVr);
    for (int i = 0; i < 3; ++i) { T[CT_DVR + i] = DrVr[i]; T[CT_DVR + 3 + i] = r_hat[i]; }
    s = 0.0; for (int i = 0; i < 6; ++i) s += T[CT_DVR + i] * rv[i];
    T[CT_DVR_BAR] = s;
    T[CT_VN] = dot3(v, h_hat);                                                                /* :164 */
    double DrVn[3]; vm3(v, Dr_h_hat, DrVn); vm3(v, Dv_h_hat, w);
    for (int i = 0; i < 3; ++i) { T[CT_DVN + i] = DrVn[i]; T[CT_DVN + 3 + i] = h_hat[i] + w[i]; }
    s = 0.0; for (int i = 0; i < 6; ++i) s += T[CT_DVN + i] * rv[i];
    T[CT_DVN_BAR] = s;
}
