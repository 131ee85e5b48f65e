/*
 * mpcx.h -- C ABI of libmpcx.so, the MI355X (gfx950) batched constellation-MPC engine.
 *
 * The reference (rgovindjee/mpconstellation) has no FFI layer: its hot path is two Python
 * classes.  Each entry point below names the reference interface it replaces (file:line into
 * the reference repo); mpconstellation_amd/_ffi.py is the ctypes binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - plain C, fp64 everywhere, row-major (numpy C-order) arrays, satellite index outermost;
 *  - every function returns 0 on success or a negative MPCX_E_* code; mpcx_last_error() gives
 *    the text; per-satellite outcomes come back in int32 status arrays (MPCX_ST_*);
 *  - "_dev" variants take DEVICE pointers (HBM resident) and enqueue on `stream` (a hipStream_t
 *    passed as void*, NULL = default stream) without synchronising; the others take HOST
 *    pointers, stage through HBM and return when the results are in the caller's buffers;
 *  - the library never keeps a caller pointer after a call returns (host variants) or after
 *    the enqueued work has completed (device variants);
 *  - there is no CPU fallback: without a HIP device mpcx_create fails with MPCX_E_NODEVICE.
 */
#ifndef MPCX_H
#define MPCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCX_VERSION 500

/* return codes */
#define MPCX_OK 0
#define MPCX_E_NODEVICE (-1)
#define MPCX_E_BADARG (-2)
#define MPCX_E_HIP (-3)
#define MPCX_E_NOMEM (-4)

/* per-satellite status */
#define MPCX_ST_OK 0
#define MPCX_ST_MASS 1        /* mass <= 0 in the dynamics (reference raises, simulator.py:135-136) */
#define MPCX_ST_STEP 2        /* RK45 step size underflow (scipy: "Required step size is less than spacing") */
#define MPCX_ST_FOH 3         /* FOH index outside the input table (reference: IndexError) */
#define MPCX_ST_SINGULAR 4    /* state-transition matrix not invertible (np.linalg.inv raises) */
#define MPCX_ST_MAXITER 5     /* solver hit max_iter without meeting tol */
#define MPCX_ST_NUMERIC 6     /* solver: non-finite value or factorisation breakdown */
#define MPCX_ST_ACCEPTABLE 7  /* solver stopped at the 'acceptable' level (ipopt acceptable_tol) */
#define MPCX_ST_BADK 9        /* ragged batch: this satellite's node / table-column / output-point count is outside the range
                               * the call accepts (solve: 3..K, discretize: 2..K, propagate: 1..n_eval, tables: 2..Ku) */
#define MPCX_ST_INFEASIBLE 8  /* solver: the constraint set is empty whatever the dynamics (start node outside its own radius
                               * bounds, terminal window outside r_max, r_min > r_max, empty window or tf range); seen before
                               * the first iteration: x_bar, u_bar, tf_bar come back, kkt = the violation (ipopt: restoration
                               * failure / "converged to a point of local infeasibility", ignored by optimizer.py:603) */
#define MPCX_ST_TIMEOUT 10    /* time-parallel solve only (MPCX_SOLVE_TIME_PARALLEL): one of the satellite's workgroups did not answer
                               * within the wait limit (~0.1 s of polling) -- e.g. another long kernel kept it from becoming
                               * resident; not a numerical failure: X, U, NU hold the last iterate, kkt = -1; solve again
                               * without the flag or with the device to itself */

/* dynamics flags (reference include_drag / include_J2 keyword arguments) */
#define MPCX_FLAG_DRAG 1
#define MPCX_FLAG_J2 2
/* discretize entry points only: Discretizer.use_uniform_steps (linearize_discretize.py:27-30, 50-53) with
 * integrator_steps = n: flags |= MPCX_FLAG_UNIFORM_STEPS | MPCX_UNIFORM_STEPS(n).  The quadrature then runs over n uniform
 * points per interval of the RK45 dense output instead of the accepted step nodes. */
#define MPCX_FLAG_UNIFORM_STEPS 4
#define MPCX_UNIFORM_STEPS(n) ((n) << 8)
/* discretize entry points only: Discretizer.ivp_solver = 'RK23' (linearize_discretize.py:40,105: the attribute is solve_ivp's
 * `method`) -- scipy's Bogacki-Shampine 3(2) pair instead of the default 'RK45', same step-size controller, same tolerances.
 * The implicit methods scipy also offers (Radau, BDF, LSODA) and DOP853 are not implemented. */
#define MPCX_FLAG_RK23 8

/* normalised constants per satellite: reference constants.py:11-20 field order */
enum { MPCX_C_MU = 0, MPCX_C_R_E, MPCX_C_J2, MPCX_C_G0, MPCX_C_ISP, MPCX_C_S, MPCX_C_R0,
       MPCX_C_RHO, MPCX_NCONST };

/* packed per-interval stage record written by the discretizer and read by the solver:
 * [A 7x7 | B_kn 7x3 | B_kp 7x3 | Sigma 7 | xi 7], row-major blocks */
#define MPCX_STAGE_DOUBLES 105

typedef struct mpcx_ctx mpcx_ctx;

int mpcx_version(void);
/* one context per (device, host thread); calls on different contexts are thread-safe */
int mpcx_create(int device, mpcx_ctx **out);
void mpcx_destroy(mpcx_ctx *ctx);
const char *mpcx_last_error(const mpcx_ctx *ctx); /* ctx may be NULL: last create error */
int mpcx_synchronize(mpcx_ctx *ctx, void *stream);
/* The stream the HOST-POINTER entry points of this context work on (a context handle per (device, stream)).  A
 * context starts with a private non-blocking stream, so that contexts of different host threads overlap (one per device, or
 * several per device).  A process that drives the device through a stream of its own -- PyTorch's, say -- may hand that
 * stream in instead: the context's calls are then ordered with the rest of that stream's work.  stream: a hipStream_t, NULL
 * for the device's default stream, MPCX_STREAM_PRIVATE for a new private stream.  The old stream is drained first. */
#define MPCX_STREAM_PRIVATE ((void *)(intptr_t)-1)
int mpcx_set_stream(mpcx_ctx *ctx, void *stream);
/* Where a host-pointer call's time went, as data (the reference has no counterpart: its solve is a subprocess whose time
 * optimizer.py:603 does not look at).  mpcx_trace_enable(ctx, 1): every following host-pointer call on the context records, at
 * the price of four events and one polled marker per call,
 *   MPCX_TR_WALL          entry to return on the host, ms
 *   MPCX_TR_FIRST_MARKER  from entry until the stream has executed the call's FIRST packet (a bare marker, polled): time the
 *                         queue took to pick the call up -- before any work of this library ran
 *   MPCX_TR_HOST_STAGE    host copies into the staging pool + enqueueing of transfers and kernels
 *   MPCX_TR_HOST_WAIT     host blocked on the stream (events of the downloads, final synchronisation)
 *   MPCX_TR_HOST_COPYOUT  staging -> the caller's result arrays
 *   MPCX_TR_DEV_SPAN      the device's own time stamps: first marker -> last download done (includes any time the stream sat
 *                         waiting for the host to enqueue the next transfer)
 *   MPCX_TR_DEV_KERNELS   ... last upload done -> kernels done: the call's kernels alone, as the device ran them
 *   MPCX_TR_VALID         1 when the record belongs to a traced call
 * mpcx_last_call_trace copies the record of the context's last traced call (n <= MPCX_TRACE_N doubles).  The same marks are
 * printed to stderr for calls slower than MPCX_HOST_TRACE=<ms> (environment), with or without mpcx_trace_enable. */
enum { MPCX_TR_WALL = 0, MPCX_TR_FIRST_MARKER = 1, MPCX_TR_HOST_STAGE = 2, MPCX_TR_HOST_WAIT = 3, MPCX_TR_HOST_COPYOUT = 4,
       MPCX_TR_DEV_SPAN = 5, MPCX_TR_DEV_KERNELS = 6, MPCX_TR_VALID = 7, MPCX_TRACE_N = 8 };
int mpcx_trace_enable(mpcx_ctx *ctx, int on);
int mpcx_last_call_trace(const mpcx_ctx *ctx, double *out, int n);
/* Page-locked host memory for the arrays a caller hands to the host-pointer entry points again and again (the reference
 * keeps x_bar / u_bar / results in numpy arrays, optimizer.py:13-39, 192-217; a numpy array can live in such a buffer).
 * Arrays in page-locked memory are transferred by DMA straight from / to the caller's buffer; pageable ones go through the
 * context's own page-locked staging (an extra host copy each way). */
void *mpcx_host_alloc(mpcx_ctx *ctx, size_t bytes);
void mpcx_host_free(mpcx_ctx *ctx, void *p);

/*
 * Replaces Discretizer.discretize (linearize_discretize.py:334-390), i.e. get_matrices (:8-82)
 * for every interval of every satellite, including dPhi (:257-291), A_func (:119-183),
 * B_func (:186-215), xi_func (:218-236), Sigma_func (:239-254), u_FOH (:294-315) and the
 * RK45 integration the reference delegates to scipy (rtol 1e-3, atol 1e-6, max_step, automatic
 * first step), with f = Simulator.satellite_dynamics (simulator.py:116-161).
 *   xbar   [S][7][K]     reference trajectories         ubar [S][3][Ku]  reference thrust
 *   tf     [S]           reference final times          consts [S][MPCX_NCONST]
 *   A      [S][K-1][7][7]   Bp, Bn [S][K-1][7][3]   Sigma, xi [S][7][K-1]   (reference shapes,
 *   reference return order A, B_kp, B_kn, Sigma, xi)     status [S]
 */
int mpcx_discretize_batch(mpcx_ctx *ctx, int S, int K, int Ku, const double *xbar,
                          const double *ubar, const double *tf, const double *consts, int flags,
                          double max_step, double *A, double *Bp, double *Bn, double *Sigma,
                          double *xi, int32_t *status);
int mpcx_discretize_batch_dev(mpcx_ctx *ctx, int S, int K, int Ku, const double *xbar,
                              const double *ubar, const double *tf, const double *consts,
                              int flags, double max_step, double *A, double *Bp, double *Bn,
                              double *Sigma, double *xi, int32_t *status, void *stream);
/* same computation, output as packed stage records stage[S][K-1][MPCX_STAGE_DOUBLES] (the
 * layout the solver consumes; A/B never take the reference's five-array form) */
int mpcx_discretize_stages_dev(mpcx_ctx *ctx, int S, int K, int Ku, const double *xbar,
                               const double *ubar, const double *tf, const double *consts,
                               int flags, double max_step, double *stage, int32_t *status,
                               void *stream);

/* Options of the per-satellite solve: the keys of Optimizer.init_options (optimizer.py:178-188;
 * u_lim[1] -> u_max, r_lim -> r_min/r_max; r_des is per satellite; eps_vt is read only with
 * MPCX_SOLVE_LINEAR_VT: the reference as shipped enables the exact tangential constraint :577, which has no
 * tolerance) and the ipopt-level controls. */
typedef struct {
    double min_mass, u_max, r_min, r_max, eps_r, eps_vr, eps_vn, eps_vt, tf_max, w_nu, w_tr;
    double tol, acceptable_tol;
    int32_t max_iter, acceptable_iter, n_refine, flags;   /* flags: MPCX_SOLVE_* */
} mpcx_solve_opts;

/* By default the solver launches its per-satellite workgroups longest-first, ordered by the iteration counts of
 * the previous solve of the same batch size on this context (consecutive MPC steps pose similar problems; with a
 * few satellites per wave slot the launch ends when the slowest slot does).  Results never depend on the launch
 * order.  The counts are the library's own copy, written and read on the stream of the calls: consecutive solves on
 * one context must be enqueued on the same stream (or be ordered by the caller).  This flag keeps the plain index
 * order.  With it, solves of one context may be in flight on different streams at once: every launch has its own work
 * queue (up to 64 launches of a context in flight) and works in the caller's workspace; the regularisation record
 * (mpcx_solve_regularised) is then that of whichever solve wrote last. */
#define MPCX_SOLVE_INDEX_ORDER 1
/* Tangential velocity as the linearised pair max_tan_vel_rule / min_tan_vel_rule (optimizer.py:471-489,
 * |Vt_lin(x_K) - Vc_lin(r_K)| <= eps_vt), which the reference keeps commented out at :575-576, instead of the quartic
 * equality :492-517 it enables at :577.  Every constraint is then linear or convex quadratic and the objective strictly
 * convex in (x, u, tf): the minimiser is unique -- the variant the parity tests use to compare solvers exactly. */
#define MPCX_SOLVE_LINEAR_VT 2
/* The final time is not a variable: satellite s is solved with tf held at the value found in tf_out[s] ON ENTRY (its
 * range constraint optimizer.py:588 and its stationarity row drop out; the trust-region term w_tr (tf - tf_bar)^2 stays in
 * the objective as a constant).  ON EXIT tf_out[s] holds the satellite's term of the tf stationarity row,
 * g_s = 2 w_tr (tf - tf_bar) - sum_k Sigma_k . lambda_k = dV_s/dtf of its optimal value.  This is the inner problem of the
 * shared-tf mode: several satellites in one reference Optimizer share ONE tf (optimizer.py:287,311,322,336), and that NLP
 * separates given tf -- its KKT conditions are every inner problem's plus 1 + sum_s g_s(tf) = 0 (or tf on its bound),
 * a scalar equation the host solves (mpconstellation_amd/optimizer.py: Optimizer with shared tf).  Host-pointer entry
 * points read tf_out as an input too in this mode. */
#define MPCX_SOLVE_FIXED_TF 4

/* ONE final time for all S satellites of the call: the NLP of a reference Optimizer that holds several satellites
 * (optimizer.py:287: a single tf Var; its trust-region term :311,322 and the dynamics' Sigma_k tf :336 for every satellite,
 * the range constraint :588 once), solved as one problem -- one barrier parameter, one step length, one convergence test,
 * the tf row of every Newton system assembled across the satellites -- in a single cooperative launch with one workgroup per
 * satellite.  tf_out[s] is the same for every s (get_solved_tf ignores s, :199-203); status, iters, kkt are the launch's.
 * S is limited to the workgroups the device holds at once (2048 on MI355X); no ragged batches. */
#define MPCX_SOLVE_SHARED_TF 8

/* Batches of at most 1024 satellites (no more than one per SIMD of an MI355X) are solved by a kernel with TWO waves per
 * satellite that share the factorisation of every interior-point iteration.  Its results are bit for bit those of the
 * one-wave kernel larger batches use (both are compiled with -ffp-contract=on: the same expressions round alike): a
 * satellite's result does not depend on the size of the batch it is solved in.  This flag keeps the one-wave kernel for
 * small batches too (measurements, tests). */
#define MPCX_SOLVE_ONE_WAVE 16
/* Batches of at most one satellite per compute unit (256 on an MI355X) whose horizon's working set fits (K <= 30) are solved
 * with that working set -- iterate, direction, Newton and factor records, channel vectors: 134 KB at K = 30 -- held in the
 * compute unit's LDS instead of the global workspace (same kernel otherwise, same bits).  This flag keeps them on the
 * global-workspace kernel (measurements, tests). */
#define MPCX_SOLVE_NO_LDS 32
/* Time-parallel linear solve for small batches: the horizon is cut into four segments whose Riccati recursions and sweeps run
 * side by side, a workgroup of two waves per segment on its own compute unit, joined by a coarse 7 x 7 recursion over the cuts
 * (csrc/solve_tp.hip).  Honoured for batches of at most 128 satellites and row lengths K >= 24 (four
 * workgroups per satellite, all resident); other calls take the kernels they would take without the flag.  Same Newton
 * directions to ~1e-10 relative, the same iteration counts on 98-100 % of the problems, NOT the same bits as the other
 * kernels -- which is why it is a flag and not the default.  64 satellites: solve kernel 1.09 against 1.33 ms at 30 nodes, call 1.76
 * against 2.36 ms at 60.  The satellite's workgroups wait for each other (a cooperative launch, every wait with a
 * time limit that ends that satellite's solve with MPCX_ST_TIMEOUT): one time-parallel solve per device at a time -- a long
 * kernel of another stream or context that keeps some of a satellite's workgroups from becoming resident runs the limit out.
 * A batch the device cannot hold at once (more satellites than a quarter of its resident workgroups) takes the default kernels,
 * like a batch above 128. */
#define MPCX_SOLVE_TIME_PARALLEL 64
/* Test hook of the time-parallel kernel (tests/test_time_parallel_oracle_gpu.py): the workgroup of every satellite's first
 * segment leaves before its first command, so that the wait limit runs out -- every satellite must come back MPCX_ST_TIMEOUT with
 * defined results and the launch must end.  No effect without MPCX_SOLVE_TIME_PARALLEL. */
#define MPCX_SOLVE_TP_SELFTEST_DEAD (1 << 30)

void mpcx_default_solve_opts(mpcx_solve_opts *o);
/* Workspace of the _dev solves / fused steps.  The plain queries are device-independent upper bounds (one slot per
 * satellite); the _ctx queries return what a launch on the context's device touches -- one slot per persistent workgroup,
 * min(S, workgroups resident at once: 2048 on MI355X), 0.44 GB instead of 1.8 GB at S = 8192, K = 30 -- and are enough. */
size_t mpcx_solve_workspace_bytes(int S, int K);
size_t mpcx_mpc_step_workspace_bytes(int S, int K);
size_t mpcx_solve_workspace_bytes_ctx(const mpcx_ctx *ctx, int S, int K);
size_t mpcx_mpc_step_workspace_bytes_ctx(const mpcx_ctx *ctx, int S, int K);

/*
 * Replaces Optimizer.get_constraint_terms (optimizer.py:80-170) + the NLP transcription and
 * ipopt solve of Optimizer.solve_OPT (optimizer.py:254-613), one independent problem (own tf)
 * per satellite, given already discretised dynamics.
 *   host variant: A [S][K-1][7][7], Bp, Bn [S][K-1][7][3], Sigma, xi [S][7][K-1] in the reference's
 *   shapes; device variant: packed stage records stage[S][K-1][MPCX_STAGE_DOUBLES].
 *   xbar [S][7][K], ubar [S][3][K], tf [S], consts [S][MPCX_NCONST], r_des [S]
 * Results replace get_solved_trajectory / get_solved_u / get_solved_nu / get_solved_tf
 * (optimizer.py:192-217): X [S][7][K], U [S][3][K], NU [S][7][K], tf_out [S];
 * status [S] (MPCX_ST_*), iters [S], kkt [S] = final scaled optimality error (ipopt's E_0).
 */
int mpcx_solve_batch(mpcx_ctx *ctx, int S, int K, const double *A, const double *Bp, const double *Bn,
                     const double *Sigma, const double *xi, const double *xbar, const double *ubar,
                     const double *tf, const double *consts, const double *r_des,
                     const mpcx_solve_opts *opts, double *X, double *U, double *NU, double *tf_out,
                     int32_t *status, int32_t *iters, double *kkt);
int mpcx_solve_batch_dev(mpcx_ctx *ctx, int S, int K, const double *stage, const double *xbar,
                         const double *ubar, const double *tf, const double *consts,
                         const double *r_des, const mpcx_solve_opts *opts, double *X, double *U,
                         double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                         void *workspace, void *stream);

/* Per-satellite regularisation record of the LAST solve (or fused step) on this context, out [S][2] int32: the number of
 * interior-point iterations whose Newton system needed a Hessian regularisation delta_w > 0 (ipopt's inertia correction:
 * its iteration log prints the same quantity, lg(rg), which optimizer.py:603 shows with tee=verbose), and the index of the
 * first such iteration (-1: none).  S must be that solve's batch size.  The _dev variant copies on `stream` (the stream of
 * the solve) into device memory; the host variant returns when `out` is filled. */
int mpcx_solve_regularised(mpcx_ctx *ctx, int S, int32_t *out);
int mpcx_solve_regularised_dev(mpcx_ctx *ctx, int S, int32_t *out, void *stream);

/*
 * Replaces Optimizer.get_constraint_terms (optimizer.py:80-170) as the solver consumes it: what the device builds for
 * every satellite before its first iteration, returned instead of used.  Terminal inequality rows a_j . x_K <= b_j,
 *   aT [S][8][7], bT [S][8]:  0: -r_hat.r_K <= -(r_des - eps_r)          (optimizer.py:398-402)
 *                             1, 2: +-(Vr linearised) <= eps_vr           (:406-416, 432-433)
 *                             3, 4: +-(Vn linearised) <= eps_vn           (:436-446, 466-467; Dv_h_hat in its precedence form :122)
 *                             5: -m_K <= -min_mass                        (:351-352, 363)
 *                             6, 7: +-(Vt - Vc linearised) <= eps_vt      (:471-489; zero rows without MPCX_SOLVE_LINEAR_VT)
 * every b relaxed by 1e-8 max(1, |b|) (ipopt's bound_relax_factor), and
 *   scalars [S][MPCX_NTERM_SCALARS]: relaxed u_max^2 (:379-381), r_max^2 (:393-395), -r_min (:384-391), (r_des+eps_r)^2
 *   (:403), the tf range 0 / tf_max (:588), vt_des = sqrt(mu / r_des) (:492-517), and the structural violation (> 0: the
 *   constraint set is empty, MPCX_ST_INFEASIBLE).
 * xbar [S][7][K], consts [S][MPCX_NCONST], r_des [S].
 */
#define MPCX_NTERM_SCALARS 8
int mpcx_constraint_terms(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *consts,
                          const double *r_des, const mpcx_solve_opts *opts, double *aT, double *bT, double *scalars);
int mpcx_constraint_terms_dev(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *consts,
                              const double *r_des, const mpcx_solve_opts *opts, double *aT, double *bT,
                              double *scalars, void *stream);

/*
 * One satellite-MPC-step = Optimizer.solve_OPT as the reference runs it (optimizer.py:243-251 calls
 * discretize, then :254-603): discretize -> constraint terms -> solve, fused on the device; the
 * stage records never take the reference's five-array form and never leave HBM.
 */
int mpcx_mpc_step_batch(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *ubar,
                        const double *tf, const double *consts, const double *r_des, int flags,
                        double max_step, const mpcx_solve_opts *opts, double *X, double *U, double *NU,
                        double *tf_out, int32_t *status, int32_t *iters, double *kkt);
int mpcx_mpc_step_batch_dev(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *ubar,
                            const double *tf, const double *consts, const double *r_des, int flags,
                            double max_step, const mpcx_solve_opts *opts, double *X, double *U,
                            double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                            void *workspace, void *stream);

/* thrust laws of the reference's controllers (control.py) */
#define MPCX_CTRL_ZERO 0        /* Controller.get_u_func            control.py:20-29  */
#define MPCX_CTRL_CONSTANT 1    /* ConstantThrustController          control.py:37-53  ctrl_vec [S][3]      */
#define MPCX_CTRL_TANGENTIAL 2  /* ConstantTangentialThrustController control.py:55-84 ctrl_vec [S] magnitude */
#define MPCX_CTRL_SEQUENCE 3    /* SequenceController (FOH playback) control.py:86-143 ctrl_vec [S][3][Ku], end_tau [S] */

/*
 * Replaces Simulator.get_trajectory_ODE (simulator.py:164-189) for S satellites: scipy
 * solve_ivp(RK45, rtol 1e-3, atol 1e-6, max_step, t_eval=linspace(0,1,n_eval)) of
 * Simulator.satellite_dynamics (simulator.py:116-161, flags = MPCX_FLAG_DRAG|MPCX_FLAG_J2) under a
 * thrust law u(y, tau).  y0 [S][7] normalised states, y_out [S][7][n_eval] (= sol.y per satellite).
 */
int mpcx_propagate_batch(mpcx_ctx *ctx, int S, int n_eval, const double *y0, const double *tf,
                         const double *consts, int flags, int ctrl_kind, const double *ctrl_vec, int Ku,
                         const double *end_tau, double max_step, double *y_out, int32_t *status,
                         int32_t *nsteps);
int mpcx_propagate_batch_dev(mpcx_ctx *ctx, int S, int n_eval, const double *y0, const double *tf,
                             const double *consts, int flags, int ctrl_kind, const double *ctrl_vec,
                             int Ku, const double *end_tau, double max_step, double *y_out,
                             int32_t *status, int32_t *nsteps, void *stream);

/*
 * Ragged batches.  The reference re-samples every SCP re-rollout at int(base_res * tf_u) nodes (control.py:227 ->
 * simulator.py:38), a different count for every satellite of a constellation: the next discretize / solve then has K_s
 * nodes for satellite s.  The *_ragged entry points take one launch of satellites with different counts: arrays keep the
 * rectangular shapes of the plain entry points with K (n_eval, Ku) the ROW LENGTH, and satellite s uses the first Ks[s]
 * (n_evals[s], Kus[s]) columns of its rows; np.linspace(0, 1, Ks[s]) is its node grid.  Result columns past a satellite's
 * count come back zero (stage records past its last interval are unspecified).  A count outside the accepted range gives
 * that satellite MPCX_ST_BADK and leaves the others alone.  A NULL count array means "all K": the plain entry points are
 * these with NULL.  Count arrays are int32; device pointers in the _dev variants.
 */
int mpcx_discretize_stages_ragged_dev(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, int Ku, const int32_t *Kus,
                                      const double *xbar, const double *ubar, const double *tf,
                                      const double *consts, int flags, double max_step, double *stage,
                                      int32_t *status, void *stream);
int mpcx_solve_batch_ragged_dev(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *stage, const double *xbar,
                                const double *ubar, const double *tf, const double *consts,
                                const double *r_des, const mpcx_solve_opts *opts, double *X, double *U,
                                double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                void *workspace, void *stream);
/* the fused step: thrust tables have as many columns as the satellite has nodes (Kus = Ks) */
int mpcx_mpc_step_batch_ragged(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *xbar, const double *ubar,
                               const double *tf, const double *consts, const double *r_des, int flags,
                               double max_step, const mpcx_solve_opts *opts, double *X, double *U, double *NU,
                               double *tf_out, int32_t *status, int32_t *iters, double *kkt);
int mpcx_mpc_step_batch_ragged_dev(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *xbar, const double *ubar,
                                   const double *tf, const double *consts, const double *r_des, int flags,
                                   double max_step, const mpcx_solve_opts *opts, double *X, double *U,
                                   double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                   void *workspace, void *stream);
/* Simulator.get_trajectory_ODE with t_eval = linspace(0, 1, n_evals[s]) per satellite (simulator.py:38,185-187) and, for
 * MPCX_CTRL_SEQUENCE, thrust tables of Kus[s] columns */
int mpcx_propagate_batch_ragged(mpcx_ctx *ctx, int S, int n_eval, const int32_t *n_evals, const double *y0,
                                const double *tf, const double *consts, int flags, int ctrl_kind,
                                const double *ctrl_vec, int Ku, const int32_t *Kus, const double *end_tau,
                                double max_step, double *y_out, int32_t *status, int32_t *nsteps);
int mpcx_propagate_batch_ragged_dev(mpcx_ctx *ctx, int S, int n_eval, const int32_t *n_evals, const double *y0,
                                    const double *tf, const double *consts, int flags, int ctrl_kind,
                                    const double *ctrl_vec, int Ku, const int32_t *Kus, const double *end_tau,
                                    double max_step, double *y_out, int32_t *status, int32_t *nsteps, void *stream);
/*
 * The same rollout that also returns Discretizer.extract_uk (linearize_discretize.py:393-411) of its controller: u_out
 * [S][3][n_eval] = u_func(x_k, t_k) at every output point (control.py:66-84 for the tangential law, :104-142 for a
 * sequence, the constant vector, zeros) -- the reference thrust u_bar that OptimalController.update (control.py:187,222)
 * derives from the trajectory it has just computed.  u_out may be NULL (then it is mpcx_propagate_batch_ragged).
 */
int mpcx_propagate_thrust_batch_ragged(mpcx_ctx *ctx, int S, int n_eval, const int32_t *n_evals, const double *y0,
                                       const double *tf, const double *consts, int flags, int ctrl_kind,
                                       const double *ctrl_vec, int Ku, const int32_t *Kus, const double *end_tau,
                                       double max_step, double *y_out, double *u_out, int32_t *status, int32_t *nsteps);
int mpcx_propagate_thrust_batch_ragged_dev(mpcx_ctx *ctx, int S, int n_eval, const int32_t *n_evals, const double *y0,
                                           const double *tf, const double *consts, int flags, int ctrl_kind,
                                           const double *ctrl_vec, int Ku, const int32_t *Kus, const double *end_tau,
                                           double max_step, double *y_out, double *u_out, int32_t *status, int32_t *nsteps,
                                           void *stream);
/*
 * One SCP iteration of OptimalController.update (control.py:183-227) for S satellites in one call: the nonlinear rollout from
 * y0 [S][7] over tf [S] under the given thrust law (arguments as mpcx_propagate_batch_ragged), sampled at K nodes (Ks[s] of
 * them in a ragged batch) -- x_bar; the law at those nodes -- u_bar (extract_uk); the discretisation about (x_bar, u_bar,
 * tf) and the solve (arguments and results as mpcx_mpc_step_batch_ragged).  x_bar and u_bar stay on the device; they are
 * returned only when xbar_out [S][7][K] / ubar_out [S][3][K] are not NULL.  prop_status [S]: the rollout's MPCX_ST_* codes.
 */
int mpcx_scp_iteration_batch_ragged(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *y0, const double *tf,
                                    const double *consts, const double *r_des, int prop_flags, int ctrl_kind,
                                    const double *ctrl_vec, int Ku, const int32_t *Kus, const double *end_tau,
                                    double prop_max_step, int disc_flags, double disc_max_step, const mpcx_solve_opts *opts,
                                    double *xbar_out, double *ubar_out, double *X, double *U, double *NU, double *tf_out,
                                    int32_t *status, int32_t *iters, double *kkt, int32_t *prop_status);
/*
 * OptimalController.update (control.py:170-235) for S satellites in ONE call -- and, optionally, the segment flight of
 * Simulator.run_segment (simulator.py:58-65) that follows it -- with everything between the first input and the last result
 * resident in HBM: the reference rollout from y0 [S][7] over tf0 [S] (the horizon) under ConstantTangentialThrustController
 * (ref_thrust; control.py:178-180) sampled at K = int(base_res * horizon) nodes, then n_scp x [ extract_uk, discretise, solve ]
 * (:183-213) with, between two iterations, the nonlinear re-rollout under SequenceController(u_opt, tf_u, tf_u) sampled at
 * int(base_res * tf_u) nodes per satellite (:217-227, simulator.py:38: the node counts are computed on the device and the
 * next iteration is a ragged launch; the plan's thrust is consumed in place as the rollout's table).
 * Results: the last iteration's plan X [S][7][K], U [S][3][K], NU [S][7][K] (rows of length K, Ks_out[s] columns in use, zeros
 * behind them), tf_out [S] = tf_u, Ks_out [S]; status, iters [n_scp][S]: every iteration's solver outcome; kkt [S]: the last
 * iteration's; prop_status [S]: the first failure among the rollouts (MPCX_ST_*).
 * Segment flight (y_sim != NULL): from y0 over sim_tf under the truth model sim_flags (MPCX_FLAG_DRAG | MPCX_FLAG_J2) with
 * SequenceController(u_opt, tf_u, tf_sim = sim_interval) (end_tau = tf_u / sim_interval, control.py:102,217),
 * y_sim [S][7][sim_n_eval] = sol.y at linspace(0, 1, sim_n_eval), sim_status [S].
 * Environment (measurement switch, read per call): MPCX_UPDATE_SPLIT=1 runs a batch of 2048 or more satellites as two chains --
 * its halves, each on its own stream, so that one half's rollouts run under the other half's solve -- and =2 also delays the
 * second chain to the first's first solve; same bits either way; measured no faster than one chain, hence
 * not the default.
 */
int mpcx_mpc_update_batch(mpcx_ctx *ctx, int S, int K, int n_scp, double base_res, const double *y0, const double *tf0,
                          const double *consts, const double *r_des, double ref_thrust, double prop_max_step, int disc_flags,
                          double disc_max_step, const mpcx_solve_opts *opts, double *X, double *U, double *NU, double *tf_out,
                          int32_t *Ks_out, int32_t *status, int32_t *iters, double *kkt, int32_t *prop_status, double sim_tf,
                          double sim_interval, int sim_n_eval, int sim_flags, double sim_max_step, double *y_sim,
                          int32_t *sim_status);
/*
 * Replaces Discretizer.extract_uk (linearize_discretize.py:393-411) for a SequenceController played over its own horizon
 * (control.py:217-221, tf_sim = tf_u: end_tau = 1): the first-order hold (control.py:104-126) of table u [S][3][Ku]
 * (Kus[s] columns in use) at the nodes linspace(0, 1, ns[s]) -> u_out [S][3][n], the reference thrust of the next SCP
 * iteration.  status [S]: MPCX_ST_FOH / MPCX_ST_BADK.
 */
int mpcx_resample_sequence_dev(mpcx_ctx *ctx, int S, int Ku, const int32_t *Kus, const double *u, int n,
                               const int32_t *ns, double *u_out, int32_t *status, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPCX_H */
