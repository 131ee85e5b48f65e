/*
 * oracle/propagate.c -- TEST INFRASTRUCTURE (see mpc_oracle.h).
 * CPU restatement of Simulator.get_trajectory_ODE (reference simulator.py:164-189):
 * solve_ivp(RK45, rtol 1e-3, atol 1e-6, max_step, t_eval = linspace(0,1,n_eval)); samples come
 * from the RK45 dense-output interpolant (scipy ivp.py main loop, t_eval branch).
 */
#include <math.h>
#include <string.h>
#include "mpc_oracle.h"
#include "rk45.h"

int oracle_ctrl_eval(const oracle_ctrl *c, const double y[7], double tau, double u[3]);

typedef struct {
    const oracle_ctrl *ctrl;
    double tf;
    const double *cst;
    int flags;
    int err;
} prop_ctx;

static int prop_rhs(double tau, const double *y, double *ydot, void *vctx)
{
    prop_ctx *c = (prop_ctx *)vctx;
    double u[3];
    if (oracle_ctrl_eval(c->ctrl, y, tau, u)) c->err = 3;                /* simulator.py:147 */
    return oracle_dynamics(y, u, c->tf, c->cst, c->flags, ydot);
}

int oracle_propagate(const double y0[7], double tf, const double *cst, int flags,
                     const oracle_ctrl *ctrl, int n_eval, double max_step, double *y_out,
                     int32_t *nsteps)
{
    prop_ctx ctx = {ctrl, tf, cst, flags, 0};
    rk45 s;
    rk45_init(&s, 7, prop_rhs, &ctx, 0.0, y0, 1.0, max_step, 1e-3, 1e-6);
    int ei = 0, status = 0;
    double step = (n_eval > 1) ? 1.0 / (double)(n_eval - 1) : 0.0;
    while (!(s.t == s.t_bound)) {
        int r = rk45_step(&s);
        if (r) { status = r; break; }
        /* np.searchsorted(t_eval, t, side='right') */
        while (ei < n_eval) {
            double te = (ei == n_eval - 1 && n_eval > 1) ? 1.0 : (double)ei * step + 0.0;
            if (te > s.t) break;
            double yy[7];
            rk45_dense_eval(&s, te, yy);
            for (int i = 0; i < 7; ++i) y_out[i * n_eval + ei] = yy[i];
            ++ei;
        }
    }
    if (nsteps) *nsteps = s.nsteps;
    if (ctx.err && !status) status = ctx.err;
    if (s.fun_err && !status) status = 1;
    return status;
}
