/* oracle/rk45.h -- TEST INFRASTRUCTURE: scipy RK45 restatement (see rk45.c). */
#ifndef ORACLE_RK45_H
#define ORACLE_RK45_H

#define RK45_MAXN 56

typedef int (*rk45_fun)(double t, const double *y, double *ydot, void *ctx);

typedef struct {
    int n;
    rk45_fun fun;
    void *ctx;
    double t, t_old, t_bound, h_abs, h_previous, max_step, rtol, atol;
    double y[RK45_MAXN], y_old[RK45_MAXN], f[RK45_MAXN];
    double K[7][RK45_MAXN];
    int nfev, nsteps, fun_err;
    int method;        /* 45: Dormand-Prince 5(4) (scipy 'RK45'); 23: Bogacki-Shampine 3(2) (scipy 'RK23') */
} rk45;

void rk45_init(rk45 *s, int n, rk45_fun fun, void *ctx, double t0, const double *y0, double t_bound,
               double max_step, double rtol, double atol);
/* the same stepper with scipy's RK23 tableau (rk.py:183-278): Discretizer.ivp_solver = 'RK23' (linearize_discretize.py:40,105) */
void rk_init_method(rk45 *s, int method, int n, rk45_fun fun, void *ctx, double t0, const double *y0, double t_bound,
                    double max_step, double rtol, double atol);
int rk45_step(rk45 *s);
void rk45_dense_eval(const rk45 *s, double t, double *y);

#endif
