/*
 * oracle/rk45.c -- TEST INFRASTRUCTURE (see mpc_oracle.h).
 * Restatement of the explicit Runge-Kutta 5(4) (Dormand-Prince) stepper the reference calls
 * through scipy.integrate.solve_ivp(method='RK45') (linearize_discretize.py:37-41,
 * simulator.py:187).  scipy 1.15.3 is the third-party dependency here; its published
 * algorithm is followed step by step:
 *   scipy/integrate/_ivp/rk.py   : SAFETY/MIN_FACTOR/MAX_FACTOR :8-11, rk_step :14-70,
 *                                  RungeKutta._step_impl :110-168, RK45 tableau :377-404
 *   scipy/integrate/_ivp/common.py: norm :63-65, select_initial_step :68-134
 * and, for Discretizer.ivp_solver = 'RK23', the Bogacki-Shampine 3(2) tableau of the same file (RK23, rk.py:183-278:
 * C, A, B, E, P, order 3, error_estimator_order 2, n_stages 3) driven by the same stepper.
 */
#include <math.h>
#include <float.h>
#include <string.h>
#include "rk45.h"

static const double RK_C[6] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0};
static const double RK_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
static const double RK_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
static const double RK_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200,
                               -22.0 / 525, 1.0 / 40};
const double RK45_P[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408,
     701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};

/* RK23 (rk.py:183-278) in the same array shapes */
static const double RK23_C[6] = {0.0, 1.0 / 2, 3.0 / 4, 0, 0, 0};
static const double RK23_A[6][5] = {{0, 0, 0, 0, 0}, {1.0 / 2, 0, 0, 0, 0}, {0, 3.0 / 4, 0, 0, 0}, {0}, {0}, {0}};
static const double RK23_B[6] = {2.0 / 9, 1.0 / 3, 4.0 / 9, 0, 0, 0};
static const double RK23_E[7] = {5.0 / 72, -1.0 / 12, -1.0 / 9, 1.0 / 8, 0, 0, 0};
static const double RK23_P[7][4] = {{1, -4.0 / 3, 5.0 / 9, 0}, {0, 1, -2.0 / 3, 0}, {0, 4.0 / 3, -8.0 / 9, 0}, {0, -1, 1, 0},
                                    {0}, {0}, {0}};

typedef struct { int n_stages, err_order, n_p; const double *C; const double (*A)[5]; const double *B, *E; const double (*P)[4]; } tableau;
static tableau tab_of(int method)
{
    tableau t;
    if (method == 23) { t.n_stages = 3; t.err_order = 2; t.n_p = 3; t.C = RK23_C; t.A = RK23_A; t.B = RK23_B; t.E = RK23_E; t.P = RK23_P; }
    else { t.n_stages = 6; t.err_order = 4; t.n_p = 4; t.C = RK_C; t.A = RK_A; t.B = RK_B; t.E = RK_E; t.P = RK45_P; }
    return t;
}

#define SAFETY 0.9
#define MIN_FACTOR 0.2
#define MAX_FACTOR 10.0

static double rms_norm(const double *x, int n) /* common.py:63-65 */
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += x[i] * x[i];
    return sqrt(s) / sqrt((double)n);
}

/* common.py:68-134 with direction = +1, order = the method's error_estimator_order (4 / 2) */
static double select_initial_step(rk45 *s, double t0, const double *y0, double t_bound,
                                  const double *f0)
{
    int n = s->n;
    double tmp[RK45_MAXN], y1[RK45_MAXN], f1[RK45_MAXN], scale[RK45_MAXN];
    double interval_length = fabs(t_bound - t0);
    if (interval_length == 0.0) return 0.0;
    for (int i = 0; i < n; ++i) scale[i] = s->atol + fabs(y0[i]) * s->rtol;
    for (int i = 0; i < n; ++i) tmp[i] = y0[i] / scale[i];
    double d0 = rms_norm(tmp, n);
    for (int i = 0; i < n; ++i) tmp[i] = f0[i] / scale[i];
    double d1 = rms_norm(tmp, n);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    if (interval_length < h0) h0 = interval_length;
    for (int i = 0; i < n; ++i) y1[i] = y0[i] + h0 * f0[i];
    if (s->fun(t0 + h0, y1, f1, s->ctx)) s->fun_err = 1;
    s->nfev++;
    for (int i = 0; i < n; ++i) tmp[i] = (f1[i] - f0[i]) / scale[i];
    double d2 = rms_norm(tmp, n) / h0;
    double h1;
    if (d1 <= 1e-15 && d2 <= 1e-15)
        h1 = fmax(1e-6, h0 * 1e-3);
    else
        h1 = pow(0.01 / fmax(d1, d2), 1.0 / (tab_of(s->method).err_order + 1.0));
    double h = 100.0 * h0;
    if (h1 < h) h = h1;
    if (interval_length < h) h = interval_length;
    if (s->max_step < h) h = s->max_step;
    return h;
}

void rk45_init(rk45 *s, int n, rk45_fun fun, void *ctx, double t0, const double *y0, double t_bound,
               double max_step, double rtol, double atol)
{
    rk_init_method(s, 45, n, fun, ctx, t0, y0, t_bound, max_step, rtol, atol);
}

void rk_init_method(rk45 *s, int method, int n, rk45_fun fun, void *ctx, double t0, const double *y0, double t_bound,
                    double max_step, double rtol, double atol)
{
    memset(s, 0, sizeof *s);
    s->method = method;
    s->n = n; s->fun = fun; s->ctx = ctx; s->t = t0; s->t_bound = t_bound;
    s->max_step = max_step; s->rtol = rtol; s->atol = atol;
    memcpy(s->y, y0, n * sizeof(double));
    if (fun(t0, s->y, s->f, ctx)) s->fun_err = 1;      /* rk.py:97 */
    s->nfev = 1;
    s->h_abs = select_initial_step(s, t0, s->y, t_bound, s->f);   /* rk.py:98-101 */
    s->t_old = t0;
}

/* One accepted step: rk.py:110-168 (+ rk_step :14-70).  Returns 0 ok, 2 step too small. */
int rk45_step(rk45 *s)
{
    int n = s->n;
    const tableau tb = tab_of(s->method);
    const int ns = tb.n_stages;
    const double expo = -1.0 / (tb.err_order + 1.0);           /* rk.py:93: error_exponent */
    double t = s->t;
    const double *y = s->y;
    double min_step = 10.0 * fabs(nextafter(t, INFINITY) - t);
    double h_abs;
    if (s->h_abs > s->max_step) h_abs = s->max_step;
    else if (s->h_abs < min_step) h_abs = min_step;
    else h_abs = s->h_abs;

    int step_rejected = 0;
    double y_new[RK45_MAXN], ytmp[RK45_MAXN], err[RK45_MAXN];
    double t_new, h;
    for (;;) {
        if (h_abs < min_step) return 2;
        h = h_abs;
        t_new = t + h;
        if (t_new - s->t_bound > 0) t_new = s->t_bound;
        h = t_new - t;
        h_abs = fabs(h);
        /* rk_step */
        memcpy(s->K[0], s->f, n * sizeof(double));
        for (int st = 1; st < ns; ++st) {
            for (int i = 0; i < n; ++i) {
                double dy = 0.0;
                for (int j = 0; j < st; ++j) dy += s->K[j][i] * tb.A[st][j];
                ytmp[i] = y[i] + dy * h;
            }
            if (s->fun(t + tb.C[st] * h, ytmp, s->K[st], s->ctx)) s->fun_err = 1;
        }
        for (int i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int j = 0; j < ns; ++j) acc += s->K[j][i] * tb.B[j];
            y_new[i] = y[i] + h * acc;
        }
        if (s->fun(t + h, y_new, s->K[ns], s->ctx)) s->fun_err = 1;
        s->nfev += ns;
        /* error estimate rk.py:104-108, :137-138 */
        for (int i = 0; i < n; ++i) {
            double e = 0.0;
            for (int j = 0; j <= ns; ++j) e += s->K[j][i] * tb.E[j];
            double scale = s->atol + fmax(fabs(y[i]), fabs(y_new[i])) * s->rtol;
            err[i] = e * h / scale;
        }
        double error_norm = rms_norm(err, n);
        if (error_norm < 1.0) {
            double factor;
            if (error_norm == 0.0) factor = MAX_FACTOR;
            else factor = fmin(MAX_FACTOR, SAFETY * pow(error_norm, expo));
            if (step_rejected) factor = fmin(1.0, factor);
            h_abs *= factor;
            break;
        }
        h_abs *= fmax(MIN_FACTOR, SAFETY * pow(error_norm, expo));
        step_rejected = 1;
    }
    s->h_previous = h;
    memcpy(s->y_old, s->y, n * sizeof(double));
    s->t_old = t;
    s->t = t_new;
    memcpy(s->y, y_new, n * sizeof(double));
    s->h_abs = h_abs;
    memcpy(s->f, s->K[ns], n * sizeof(double));
    s->nsteps++;
    return 0;
}

/* rk.py:170-172 + RkDenseOutput._call_impl :560-574 : y(t) on the last accepted step */
void rk45_dense_eval(const rk45 *s, double t, double *y)
{
    int n = s->n;
    const tableau tb = tab_of(s->method);
    double h = s->h_previous;
    double x = (t - s->t_old) / h;
    double p[4];
    p[0] = x; p[1] = p[0] * x; p[2] = p[1] * x; p[3] = p[2] * x;
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int c = 0; c < tb.n_p; ++c) {
            double q = 0.0;
            for (int j = 0; j <= tb.n_stages; ++j) q += s->K[j][i] * tb.P[j][c];
            acc += q * p[c];
        }
        y[i] = h * acc + s->y_old[i];
    }
}
