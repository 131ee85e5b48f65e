/*
 * oracle/discretize.c -- TEST INFRASTRUCTURE (see mpc_oracle.h).
 * CPU restatement of Discretizer.discretize / get_matrices
 * (reference linearize_discretize.py:8-82, 257-291, 334-390).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mpc_oracle.h"
#include "rk45.h"

typedef struct {
    const double *u; /* (3,Ku) row-major */
    int Ku;
    double tf;
    const double *cst;
    int flags;
    int foh_err;
} dphi_ctx;

/* dPhi: linearize_discretize.py:262-290 */
static int dphi(double tau, const double *y, double *ydot, void *vctx)
{
    dphi_ctx *c = (dphi_ctx *)vctx;
    double u[3], A[49];
    if (oracle_u_foh(tau, c->u, c->Ku, u)) { c->foh_err = 1; u[0] = u[1] = u[2] = 0.0; }
    const double *Phi = y, *x = y + 49;
    oracle_A_func(x, u, c->tf, c->cst, c->flags, A);                    /* :282 */
    for (int i = 0; i < 7; ++i)                                          /* :284 */
        for (int j = 0; j < 7; ++j) {
            double acc = 0.0;
            for (int l = 0; l < 7; ++l) acc += A[i * 7 + l] * Phi[l * 7 + j];
            ydot[i * 7 + j] = acc;
        }
    return oracle_dynamics(x, u, c->tf, c->cst, c->flags, ydot + 49);    /* :287 */
}

/* 7x7 inverse by LU with partial pivoting (np.linalg.inv -> LAPACK gesv), :69 */
static int inv7(const double *M, double *Minv)
{
    double a[7][14];
    for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 7; ++j) { a[i][j] = M[i * 7 + j]; a[i][7 + j] = (i == j); }
    for (int p = 0; p < 7; ++p) {
        int piv = p;
        for (int i = p + 1; i < 7; ++i) if (fabs(a[i][p]) > fabs(a[piv][p])) piv = i;
        if (a[piv][p] == 0.0) return 1;
        if (piv != p) for (int j = 0; j < 14; ++j) { double t = a[p][j]; a[p][j] = a[piv][j]; a[piv][j] = t; }
        for (int i = p + 1; i < 7; ++i) {
            double m = a[i][p] / a[p][p];
            if (m != 0.0) for (int j = p; j < 14; ++j) a[i][j] -= m * a[p][j];
        }
    }
    for (int c = 0; c < 7; ++c)
        for (int i = 6; i >= 0; --i) {
            double s = a[i][7 + c];
            for (int j = i + 1; j < 7; ++j) s -= a[i][j] * Minv[j * 7 + c];
            Minv[i * 7 + c] = s / a[i][i];
        }
    return 0;
}

static void linspace01(int K, int j, double *out) /* np.linspace(0, 1, K)[j] */
{
    double step = 1.0 / (double)(K - 1);
    *out = (j == K - 1) ? 1.0 : (double)j * step + 0.0;
}

/* get_matrices for one interval k: linearize_discretize.py:8-82 */
static int get_matrices(int K, int Ku, const double *x, const double *u, double tf,
                        const double *cst, int flags, double max_step, int n_uniform, int k, double *A_k,
                        double *B_kp, double *B_kn, double *Sigma_k, double *xi_k, int32_t *n_nodes,
                        int32_t *n_fev, double *dump_t, double *dump_y, int dump_cap)
{
    double tau_k, tau_kp1;
    linspace01(K, k, &tau_k);
    linspace01(K, k + 1, &tau_kp1);
    double y0[56];
    memset(y0, 0, sizeof y0);
    for (int i = 0; i < 7; ++i) { y0[i * 7 + i] = 1.0; y0[49 + i] = x[i * K + k]; }   /* :31-34 */
    dphi_ctx ctx = {u, Ku, tf, cst, flags, 0};
    rk45 s;
    /* :37-41; method = options['ivp_solver'] (:40): flag bit 8 = 'RK23', otherwise the default 'RK45' (:105) */
    rk_init_method(&s, (flags & 8) ? 23 : 45, 56, dphi, &ctx, tau_k, y0, tau_kp1, max_step, 1e-3, 1e-6);

    int cap = n_uniform > 64 ? n_uniform + 1 : 64, n = 0;
    double *ts = malloc(cap * sizeof(double)), *ys = malloc(cap * 56 * sizeof(double));
    int status = 0;
    if (n_uniform >= 2) {
        /* use_uniform_steps (:27-30, :50-53): t_eval = np.linspace(tau_k, tau_kp1, integrator_steps); solve_ivp returns the
         * dense-output interpolant at those points (ivp.py main loop: searchsorted(t_eval, t, side='right')), the quadrature
         * nodes are the uniform points and A_k the interpolant at the last one */
        const double ustep = (tau_kp1 - tau_k) / (double)(n_uniform - 1);
        int ei = 0;
        while (!(s.t == s.t_bound)) {
            int r = rk45_step(&s);
            if (r) { status = r; break; }
            while (ei < n_uniform) {
                double te = (ei == n_uniform - 1) ? tau_kp1 : (double)ei * ustep + tau_k;
                if (te > s.t) break;
                ts[n] = te; rk45_dense_eval(&s, te, ys + n * 56); ++n; ++ei;
            }
        }
        if (n == 0) { ts[0] = tau_k; memcpy(ys, y0, sizeof y0); n = 1; }
    } else {
        ts[0] = tau_k; memcpy(ys, y0, sizeof y0); n = 1;
        while (!(s.t == s.t_bound)) {
            int r = rk45_step(&s);
            if (r) { status = r; break; }
            if (n == cap) { cap *= 2; ts = realloc(ts, cap * sizeof(double)); ys = realloc(ys, cap * 56 * sizeof(double)); }
            ts[n] = s.t; memcpy(ys + n * 56, s.y, 56 * sizeof(double)); ++n;
        }
    }
    if (ctx.foh_err) status = 3;
    if (s.fun_err && !status) status = 1;
    if (n_nodes) *n_nodes = n;
    if (n_fev) *n_fev = s.nfev;
    if (dump_t) for (int i = 0; i < n && i < dump_cap; ++i) { dump_t[i] = ts[i]; memcpy(dump_y + i * 56, ys + i * 56, 56 * sizeof(double)); }

    const double *Phi_kp1 = ys + (n - 1) * 56;                                          /* :43-44 */
    memcpy(A_k, Phi_kp1, 49 * sizeof(double));

    /* integrands at the accepted nodes :52-75 */
    double *Bn_int = malloc(n * 21 * sizeof(double)), *Bp_int = malloc(n * 21 * sizeof(double));
    double *S_int = malloc(n * 7 * sizeof(double)), *xi_int = malloc(n * 7 * sizeof(double));
    for (int i = 0; i < n; ++i) {
        double t = ts[i];
        const double *Phi = ys + i * 56, *xs = Phi + 49;
        double lam_n = (tau_kp1 - t) / (tau_kp1 - tau_k);                               /* :60 */
        double lam_p = (t - tau_k) / (tau_kp1 - tau_k);                                 /* :61 */
        double ut[3], B[21], Sig[7], xiv[7], Pinv[49];
        if (oracle_u_foh(t, u, Ku, ut)) status = 3;
        oracle_B_func(xs, ut, tf, cst, B);                                              /* :65 */
        oracle_dynamics(xs, ut, 1.0, cst, flags, Sig);                                  /* :66, Sigma_func :252-253 */
        oracle_xi_func(xs, ut, tf, cst, flags, xiv);                                    /* :67 */
        if (inv7(Phi, Pinv)) status = 4;                                                /* :69 */
        for (int r = 0; r < 7; ++r) {
            for (int c = 0; c < 3; ++c) {
                double an = 0.0, ap = 0.0;
                for (int l = 0; l < 7; ++l) {
                    an += Pinv[r * 7 + l] * (B[l * 3 + c] * lam_n);                     /* :71 */
                    ap += Pinv[r * 7 + l] * (B[l * 3 + c] * lam_p);                     /* :72 */
                }
                Bn_int[i * 21 + r * 3 + c] = an;
                Bp_int[i * 21 + r * 3 + c] = ap;
            }
            double as = 0.0, ax = 0.0;
            for (int l = 0; l < 7; ++l) { as += Pinv[r * 7 + l] * Sig[l]; ax += Pinv[r * 7 + l] * xiv[l]; }
            S_int[i * 7 + r] = as;                                                      /* :74 */
            xi_int[i * 7 + r] = ax;                                                     /* :75 */
        }
    }
    /* np.trapz :77-80 : sum(d * (y[1:] + y[:-1]) / 2.0) */
    double tBp[21] = {0}, tBn[21] = {0}, tS[7] = {0}, tX[7] = {0};
    for (int i = 0; i + 1 < n; ++i) {
        double d = ts[i + 1] - ts[i];
        for (int e = 0; e < 21; ++e) {
            tBp[e] += d * (Bp_int[(i + 1) * 21 + e] + Bp_int[i * 21 + e]) / 2.0;
            tBn[e] += d * (Bn_int[(i + 1) * 21 + e] + Bn_int[i * 21 + e]) / 2.0;
        }
        for (int e = 0; e < 7; ++e) {
            tS[e] += d * (S_int[(i + 1) * 7 + e] + S_int[i * 7 + e]) / 2.0;
            tX[e] += d * (xi_int[(i + 1) * 7 + e] + xi_int[i * 7 + e]) / 2.0;
        }
    }
    for (int r = 0; r < 7; ++r) {
        for (int c = 0; c < 3; ++c) {
            double ap = 0.0, an = 0.0;
            for (int l = 0; l < 7; ++l) { ap += Phi_kp1[r * 7 + l] * tBp[l * 3 + c]; an += Phi_kp1[r * 7 + l] * tBn[l * 3 + c]; }
            B_kp[r * 3 + c] = ap; B_kn[r * 3 + c] = an;
        }
        double as = 0.0, ax = 0.0;
        for (int l = 0; l < 7; ++l) { as += Phi_kp1[r * 7 + l] * tS[l]; ax += Phi_kp1[r * 7 + l] * tX[l]; }
        Sigma_k[r] = as; xi_k[r] = ax;
    }
    free(ts); free(ys); free(Bn_int); free(Bp_int); free(S_int); free(xi_int);
    return status;
}

/* Discretizer.discretize: linearize_discretize.py:334-390 (the mp.Pool fan-out is a serial loop here) */
int oracle_discretize(int K, int Ku, const double *x, const double *u, double tf, const double *cst,
                      int flags, double max_step, double *A, double *Bp, double *Bn, double *Sigma,
                      double *xi, int32_t *node_counts, int32_t *node_nfev, double *node_t,
                      double *node_y, int node_cap)
{
    return oracle_discretize_mode(K, Ku, x, u, tf, cst, flags, max_step, 0, A, Bp, Bn, Sigma, xi, node_counts, node_nfev,
                                  node_t, node_y, node_cap);
}

/* n_uniform >= 2: Discretizer.use_uniform_steps with integrator_steps = n_uniform; 0: the default adaptive nodes */
int oracle_discretize_mode(int K, int Ku, const double *x, const double *u, double tf, const double *cst,
                           int flags, double max_step, int n_uniform, double *A, double *Bp, double *Bn, double *Sigma,
                           double *xi, int32_t *node_counts, int32_t *node_nfev, double *node_t,
                           double *node_y, int node_cap)
{
    int status = 0, used = 0;
    for (int k = 0; k < K - 1; ++k) {
        double S7[7], X7[7];
        int32_t nn = 0, nf = 0;
        int r = get_matrices(K, Ku, x, u, tf, cst, flags, max_step, n_uniform, k, A + k * 49, Bp + k * 21,
                             Bn + k * 21, S7, X7, &nn, &nf, node_t ? node_t + used : 0,
                             node_y ? node_y + used * 56 : 0, node_cap - used);
        if (r && !status) status = r;
        for (int i = 0; i < 7; ++i) { Sigma[i * (K - 1) + k] = S7[i]; xi[i * (K - 1) + k] = X7[i]; }
        if (node_counts) node_counts[k] = nn;
        if (node_nfev) node_nfev[k] = nf;
        used += nn;
        if (used > node_cap) used = node_cap;
    }
    return status;
}
