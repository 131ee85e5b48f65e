/*
 * oracle/mpc_oracle.h -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.  The product (mpconstellation_amd/,
 * libmpcx.so) never links, loads or calls anything in oracle/.
 *
 * Each function cites the file:line of rgovindjee/mpconstellation it restates.
 * Discretize half (D1-D8, S2, U1 of SURVEY.md §8a): pinned against golden
 * vectors generated from the reference itself (tests/golden/make_golden.py).
 * Solve half (S3-S7): the reference hands the NLP to pyomo + ipopt (both
 * unpinned, neither installed here) => PARITY UNPINNED at that boundary; see
 * oracle/nlp_ipm.py (the numpy restatement of that half) and DESIGN.md.
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Normalised constants, same field order as reference constants.py:11-20. */
enum { OC_MU = 0, OC_R_E, OC_J2, OC_G0, OC_ISP, OC_S, OC_R0, OC_RHO, OC_NCONST };

enum { ORACLE_FLAG_DRAG = 1, ORACLE_FLAG_J2 = 2 };

/* Thrust laws u(y, tau): reference control.py */
enum {
    ORACLE_CTRL_ZERO = 0,       /* control.py:20-29   */
    ORACLE_CTRL_CONSTANT = 1,   /* control.py:37-53   */
    ORACLE_CTRL_TANGENTIAL = 2, /* control.py:55-84   */
    ORACLE_CTRL_SEQUENCE = 3    /* control.py:86-143  */
};

typedef struct {
    int kind;
    double thrust[3];   /* CONSTANT: ECI thrust; TANGENTIAL: thrust[0] = magnitude */
    const double *useq; /* SEQUENCE: (3, Ku) row-major */
    int Ku;
    double end_tau;     /* SEQUENCE: tf_u / tf_sim (control.py:102) */
} oracle_ctrl;

/* D1  simulator.py:116-161 ; returns 0, or 1 if mass <= 0 (reference raises) */
int oracle_dynamics(const double y[7], const double u[3], double tf, const double *cst,
                    int flags, double ydot[7]);
/* D2  linearize_discretize.py:119-183 (drag branch out of scope: cannot run in the reference) */
void oracle_A_func(const double x[7], const double u[3], double tf, const double *cst,
                   int flags, double A[49]);
/* D3  linearize_discretize.py:186-215 */
void oracle_B_func(const double x[7], const double u[3], double tf, const double *cst, double B[21]);
/* D4  linearize_discretize.py:218-236 */
void oracle_xi_func(const double x[7], const double u[3], double tf, const double *cst, int flags,
                    double xi[7]);
/* D6  linearize_discretize.py:294-315 ; returns 0 or -1 on the reference's IndexError */
int oracle_u_foh(double tau, const double *u, int Ku, double out[3]);

/* D5+D7  linearize_discretize.py:8-82, 334-390.  x (7,K) row-major, u (3,Ku) row-major.
 * Outputs in the reference's shapes: A (K-1,7,7) Bp (K-1,7,3) Bn (K-1,7,3) Sigma (7,K-1) xi (7,K-1).
 * node_counts/nfev (K-1) optional.  node_t/node_y optional dumps (capacity node_cap rows). */
int oracle_discretize(int K, int Ku, const double *x, const double *u, double tf, const double *cst,
                      int flags, double max_step, double *A, double *Bp, double *Bn, double *Sigma,
                      double *xi, int32_t *node_counts, int32_t *node_nfev, double *node_t,
                      double *node_y, int node_cap);
/* the same with Discretizer.use_uniform_steps = True and integrator_steps = n_uniform (>= 2); n_uniform = 0: default */
int oracle_discretize_mode(int K, int Ku, const double *x, const double *u, double tf, const double *cst,
                           int flags, double max_step, int n_uniform, double *A, double *Bp, double *Bn, double *Sigma,
                           double *xi, int32_t *node_counts, int32_t *node_nfev, double *node_t,
                           double *node_y, int node_cap);

/* simulator.py:164-189: solve_ivp(RK45, max_step, t_eval=linspace(0,1,n_eval)) with dense output.
 * y_out (7, n_eval) row-major as sol.y.  Returns 0, 1 = mass<=0, 2 = step too small. */
int oracle_propagate(const double y0[7], double tf, const double *cst, int flags,
                     const oracle_ctrl *ctrl, int n_eval, double max_step, double *y_out,
                     int32_t *nsteps);
/* linearize_discretize.py:393-411 */
void oracle_extract_uk(int K, const double *x, const double *t, const oracle_ctrl *ctrl, double *u);

/* S2  optimizer.py:80-170.  Output packed, see constraint_terms.c */
enum {
    CT_RF_HAT = 0,        /* 3 */
    CT_VC = 3,            /* 1 */
    CT_DRVC = 4,          /* 3 */
    CT_DRVC_RBAR = 7,     /* 1 */
    CT_VT = 8,            /* 1 */
    CT_DVT = 9,           /* 6 */
    CT_DVT_BAR = 15,      /* 1 */
    CT_VR = 16,           /* 1 */
    CT_DVR = 17,          /* 6 */
    CT_DVR_BAR = 23,      /* 1 */
    CT_VN = 24,           /* 1 */
    CT_DVN = 25,          /* 6 */
    CT_DVN_BAR = 31,      /* 1 */
    CT_NTERMS = 32
};
void oracle_constraint_terms(int K, const double *x, const double *u, double mu,
                             double *rbar_hat /* (3,K-1) */, double *ubar_hat /* (3,K) */,
                             double terms[CT_NTERMS]);

/* U1  satellite_scale.py:28-44 */
void oracle_scale(const double state[7], double scale7[7], double cst[OC_NCONST]);

#ifdef __cplusplus
}
#endif
#endif
