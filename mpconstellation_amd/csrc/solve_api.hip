// solve_api.hip -- the C entry points of the batched solve, the fused MPC step, one SCP iteration and the whole
// OptimalController.update (include/mpcx.h): argument checks, staging of host buffers, launch order bookkeeping.  The kernels
// they launch are in solve.hip / solve2w.hip, reached through the launchers of solve_launch.hpp.
#include <vector>
#include "mpcx_host.hpp"
#include "solve_launch.hpp"


using namespace mpcx;

#ifndef MPCX_TWO_WAVE_MAX
#define MPCX_TWO_WAVE_MAX 1024      // two waves per satellite pay up to one satellite per SIMD (profiles/r03/batch_size_sweep.txt)
#endif
constexpr int kTwoWaveMax = MPCX_TWO_WAVE_MAX;
// MPCX_SOLVE_TIME_PARALLEL is honoured up to this many satellites (four workgroups each: 512 of them are two per compute unit);
// larger batches take the kernels they would take without the flag -- the chip is then busy with whole satellites
constexpr int kTimeParallelMax = 128;
// ... and from this row length on: four segments (below, two segments or one do not pay for the exchange between the workgroups:
// 1.53 against 1.38 ms at 16 nodes; satellites of a ragged batch with fewer nodes than that are still solved, in two segments or one)
constexpr int kTimeParallelMinK = 24;
constexpr int kCounterRing = 64;    // work-queue counters per context: solves in flight at once on different streams

static SolveOpts to_dev_opts(const mpcx_solve_opts *o)
{
    SolveOpts d;
    d.min_mass = o->min_mass; d.u_max = o->u_max; d.r_min = o->r_min; d.r_max = o->r_max; d.eps_r = o->eps_r;
    d.eps_vr = o->eps_vr; d.eps_vn = o->eps_vn; d.eps_vt = o->eps_vt; d.tf_max = o->tf_max; d.w_nu = o->w_nu; d.w_tr = o->w_tr;
    d.tol = o->tol; d.acc_tol = o->acceptable_tol; d.max_iter = o->max_iter; d.acc_iter = o->acceptable_iter;
    d.n_refine = o->n_refine; d.linvt = (o->flags & MPCX_SOLVE_LINEAR_VT) ? 1 : 0;
    d.fixed_tf = (o->flags & MPCX_SOLVE_FIXED_TF) ? 1 : 0; d.shared_tf = (o->flags & MPCX_SOLVE_SHARED_TF) ? 1 : 0;
    d.tp_selftest = (o->flags & MPCX_SOLVE_TP_SELFTEST_DEAD) ? 1 : 0;
    return d;
}



extern "C" int mpcx_constraint_terms_dev(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *consts,
                                         const double *r_des, const mpcx_solve_opts *opts, double *aT, double *bT,
                                         double *scalars, void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 2 || !opts || !xbar || !consts || !r_des || !aT || !bT || !scalars)
        return ctx_fail(ctx, MPCX_E_BADARG, "constraint_terms: need S>=1, K>=2, options and all arrays");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    mpcx_launch::constraint_terms(S, K, xbar, consts, r_des, to_dev_opts(opts), aT, bT, scalars, (hipStream_t)stream);
    MPCX_HIP(ctx, hipGetLastError());
    return MPCX_OK;
}

extern "C" int mpcx_constraint_terms(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *consts,
                                     const double *r_des, const mpcx_solve_opts *opts, double *aT, double *bT, double *scalars)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 2 || !opts) return ctx_fail(ctx, MPCX_E_BADARG, "constraint_terms: need S>=1, K>=2 and options");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    DeviceArena ar(ctx);
    double *dx = ar.upload(xbar, (size_t)S * 7 * K), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST), *drd = ar.upload(r_des, S);
    double *da = ar.alloc<double>((size_t)S * 56), *db = ar.alloc<double>((size_t)S * 8), *ds = ar.alloc<double>((size_t)S * MPCX_NTERM_SCALARS);
    if (ar.failed()) return ar.code();
    int rc = mpcx_constraint_terms_dev(ctx, S, K, dx, dc, drd, opts, da, db, ds, ctx->stream);
    if (rc) return rc;
    ar.download(aT, da, (size_t)S * 56); ar.download(bT, db, (size_t)S * 8); ar.download(scalars, ds, (size_t)S * MPCX_NTERM_SCALARS);
    return ar.finish();
}

extern "C" int mpcx_solve_regularised_dev(mpcx_ctx *ctx, int S, int32_t *out, void *stream)
{
    if (!ctx || !out) return MPCX_E_BADARG;
    if (S < 1 || S != ctx->nreg_S || !ctx->nreg) return ctx_fail(ctx, MPCX_E_BADARG, "solve_regularised: S must be the batch size of the last solve on this context");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    MPCX_HIP(ctx, hipMemcpyAsync(out, ctx->nreg, (size_t)S * 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MPCX_OK;
}

extern "C" int mpcx_solve_regularised(mpcx_ctx *ctx, int S, int32_t *out)
{
    if (!ctx || !out) return MPCX_E_BADARG;
    if (S < 1 || S != ctx->nreg_S || !ctx->nreg) return ctx_fail(ctx, MPCX_E_BADARG, "solve_regularised: S must be the batch size of the last solve on this context");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    // (the host-pointer solves ran on the context's stream and have completed; a _dev solve is ordered by its stream)
    MPCX_HIP(ctx, hipMemcpy(out, ctx->nreg, (size_t)S * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MPCX_OK;
}

extern "C" void mpcx_default_solve_opts(mpcx_solve_opts *o)
{
    // reference defaults: optimizer.py:178-188; ipopt defaults: tol 1e-8, acceptable_tol 1e-6
    o->min_mass = 0.1; o->u_max = 5.0; o->r_min = 0.99; o->r_max = 5.0; o->eps_r = 0.01;
    o->eps_vr = 1e-5; o->eps_vn = 1e-5; o->eps_vt = 1e-5; o->tf_max = 5.0; o->w_nu = 1000.0; o->w_tr = 0.002;
    o->tol = 1e-8; o->acceptable_tol = 1e-6; o->max_iter = 200; o->acceptable_iter = 15; o->n_refine = 1; o->flags = 0;
}

// (ws_doubles_tp >= ws_doubles: the slot of the time-parallel kernel, MPCX_SOLVE_TIME_PARALLEL, so that one buffer serves any flags)
extern "C" size_t mpcx_solve_workspace_bytes(int S, int K)
{
    return (size_t)S * ws_doubles_tp(K) * sizeof(double);
}

// what a solve on THIS context's device touches: one slot per persistent workgroup, min(S, workgroups resident at once)
// (the time-parallel kernel: larger slots, at most one workgroup per compute unit)
extern "C" size_t mpcx_solve_workspace_bytes_ctx(const mpcx_ctx *ctx, int S, int K)
{
    const int slots = (ctx && S > ctx->n_slots) ? ctx->n_slots : S;
    const int slots_tp = (S <= kTimeParallelMax && K >= kTimeParallelMinK) ? S : 0;
    const size_t a = (size_t)slots * ws_doubles(K), b = (size_t)slots_tp * ws_doubles_tp(K);
    return (a > b ? a : b) * sizeof(double);
}

extern "C" int mpcx_solve_batch_ragged_dev(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *stage, const double *xbar,
                                           const double *ubar, const double *tf, const double *consts,
                                           const double *r_des, const mpcx_solve_opts *opts, double *X, double *U,
                                           double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                           void *workspace, void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 3 || !opts) return ctx_fail(ctx, MPCX_E_BADARG, "solve: need S>=1, K>=3 and options");
    if (!workspace) return ctx_fail(ctx, MPCX_E_BADARG, "solve: workspace of mpcx_solve_workspace_bytes(S,K) required");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    SolveArgs a;
    a.S = S; a.K = K; a.Ks = Ks; a.stage = stage; a.xbar = xbar; a.ubar = ubar; a.tfbar = tf; a.consts = consts; a.r_des = r_des;
    a.o = to_dev_opts(opts);
    a.X = X; a.U = U; a.NU = NU; a.tf_out = tf_out; a.kkt = kkt; a.status = status; a.iters = iters;
    a.ws = (double *)workspace; a.ws_stride = ws_doubles(K);
    // per-satellite regularisation counts of this solve (library-owned, grow-only; read back by mpcx_solve_regularised)
    // (a split update -- two halves of one batch solved on two streams -- has sized the record for the whole batch beforehand and
    //  names this half's place in it: nreg_first / nreg_total)
    const int nreg_need = ctx->nreg_total > 0 ? ctx->nreg_total : S;
    if (ctx->nreg_cap < nreg_need) {
        if (ctx->nreg) (void)hipFree(ctx->nreg);
        ctx->nreg = nullptr; ctx->nreg_cap = 0;
        MPCX_HIP(ctx, hipMalloc((void **)&ctx->nreg, (size_t)nreg_need * 2 * sizeof(int32_t)));
        ctx->nreg_cap = nreg_need;
    }
    a.nreg = ctx->nreg + (ctx->nreg_total > 0 ? 2 * (size_t)ctx->nreg_first : 0); ctx->nreg_S = nreg_need;
    // longest-first launch order from the previous solve's iteration counts (include/mpcx.h, MPCX_SOLVE_INDEX_ORDER)
    // (a batch the device holds at once has no order to choose: every satellite starts at time 0)
    const bool adaptive = !(opts->flags & MPCX_SOLVE_INDEX_ORDER) && S > ctx->n_slots;
    a.order = nullptr;
    // the launch-order state of this sequence of solves: the context's, or -- second half of a split update -- its twin
    int32_t *&o_prev = ctx->cur_lane ? ctx->ord2.prev_iters : ctx->prev_iters, *&o_order = ctx->cur_lane ? ctx->ord2.order : ctx->order;
    int32_t *&o_hist = ctx->cur_lane ? ctx->ord2.pred_hist : ctx->pred_hist;
    int &o_S = ctx->cur_lane ? ctx->ord2.order_S : ctx->order_S, &o_valid = ctx->cur_lane ? ctx->ord2.order_valid : ctx->order_valid;
    int &o_cap = ctx->cur_lane ? ctx->ord2.order_cap : ctx->order_cap;
    if (adaptive) {
        // grow-only buffers (a smaller batch reuses them: no free / allocation, hence no implicit device synchronisation,
        // when ConstellationMPC alternates group sizes on one context); the stored counts are valid only for a following
        // solve of the same batch size
        if (o_cap < S) {
            if (o_prev) (void)hipFree(o_prev);
            if (o_hist) (void)hipFree(o_hist);
            o_hist = nullptr;
            if (o_order) (void)hipFree(o_order);
            o_prev = o_order = nullptr; o_cap = 0; o_S = 0; o_valid = 0;
            MPCX_HIP(ctx, hipMalloc((void **)&o_prev, (size_t)S * sizeof(int32_t)));
            MPCX_HIP(ctx, hipMalloc((void **)&o_hist, (size_t)kPredHist * S * sizeof(int32_t)));
            MPCX_HIP(ctx, hipMalloc((void **)&o_order, (size_t)S * sizeof(int32_t)));
            o_cap = S;
        }
        if (o_S != S) { o_S = S; o_valid = 0; }
        if (o_valid) {
            mpcx_launch::launch_order(S, o_prev, o_order, (hipStream_t)stream);
            a.order = o_order;
        }
    }
    if (opts->flags & MPCX_SOLVE_SHARED_TF) {
        // one final time for the whole batch: a cooperative launch, one workgroup per satellite, all of them resident
        if (Ks) return ctx_fail(ctx, MPCX_E_BADARG, "solve: MPCX_SOLVE_SHARED_TF needs the same node count for every satellite (no ragged batch)");
        if (opts->flags & MPCX_SOLVE_FIXED_TF) return ctx_fail(ctx, MPCX_E_BADARG, "solve: MPCX_SOLVE_SHARED_TF and MPCX_SOLVE_FIXED_TF exclude each other");
        if (ctx->coop_max == 0) {
            int coop = 0, per_cu = 0;
            MPCX_HIP(ctx, hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, ctx->device));
            MPCX_HIP(ctx, mpcx_launch::solve_shared_blocks_per_cu(&per_cu));
            ctx->coop_max = coop ? per_cu * (ctx->n_slots / 8) : -1;
        }
        if (ctx->coop_max < 0) return ctx_fail(ctx, MPCX_E_HIP, "solve: the device does not support cooperative launches (MPCX_SOLVE_SHARED_TF)");
        if (S > ctx->coop_max) return ctx_fail(ctx, MPCX_E_BADARG, "solve: MPCX_SOLVE_SHARED_TF takes at most as many satellites as the device holds workgroups at once");
        if (ctx->red_cap < S) {
            if (ctx->red) (void)hipFree(ctx->red);
            ctx->red = nullptr; ctx->red_cap = 0;
            MPCX_HIP(ctx, hipMalloc((void **)&ctx->red, ((size_t)2 * S * GR_N + 2) * sizeof(double)));
            ctx->red_cap = S;
        }
        a.red = ctx->red;
        a.arrive = (int32_t *)(ctx->red + (size_t)2 * ctx->red_cap * GR_N);
        a.abort_flag = a.arrive + 1;
        a.counter = nullptr; a.order = nullptr;
        MPCX_HIP(ctx, hipMemsetAsync(a.arrive, 0, 2 * sizeof(int32_t), (hipStream_t)stream));
        MPCX_HIP(ctx, mpcx_launch::solve_shared(a, (hipStream_t)stream));
        o_valid = 0;
        return MPCX_OK;
    }
    // the launch's own work-queue counter: one of a ring, so that two solves of one context enqueued on different streams
    // do not share (and reset) one queue -- each queue position must go to exactly one workgroup of ITS launch
    if (!ctx->counter) MPCX_HIP(ctx, hipMalloc((void **)&ctx->counter, kCounterRing * sizeof(int32_t)));
    a.counter = ctx->counter + (ctx->launch_seq++ % kCounterRing);
    MPCX_HIP(ctx, hipMemsetAsync(a.counter, 0, sizeof(int32_t), (hipStream_t)stream));
    // (the workspace is the caller's: slot b of THIS call's buffer)
    const int slots = S < ctx->n_slots ? S : ctx->n_slots;
    // small batches -- at most one satellite per SIMD -- go to the two-wave build (solve2w.hip): a second wave per
    // satellite shares the factorisation; results are bit for bit the one-wave kernel's (-ffp-contract=on, build.py;
    // tests/test_full_size_gpu.py::test_two_wave_small_batch_kernel).  MPCX_SOLVE_ONE_WAVE keeps the one-wave kernel.
    // at most one satellite per compute unit: the LDS-resident build (solve_lds.hip), if the horizon's working set fits
    int lds = 1;
    bool tp = (opts->flags & MPCX_SOLVE_TIME_PARALLEL) && S <= kTimeParallelMax && K >= kTimeParallelMinK;
    if (tp) {
        if (ctx->tp_max == 0) { const int per_cu = mpcxtp_blocks_per_cu(); ctx->tp_max = per_cu > 0 ? per_cu * (ctx->n_slots / 8) / TP_MAXSEG : -1; }
        if (ctx->tp_max < 0) return ctx_fail(ctx, MPCX_E_HIP, "solve: occupancy query of the time-parallel kernel failed");
        // (a device that cannot hold the batch's workgroups at once -- fewer compute units, another partition mode -- solves it with
        //  the default kernels, as it does a batch above kTimeParallelMax: the flag asks for speed, never for an error)
        if (((S + 7) / 8) * 8 > ctx->tp_max) tp = false;
    }
    if (tp) {
        // the time-parallel kernel: four workgroups per satellite, each on its own compute unit while the batch is that small,
        // all resident (they wait for each other); its own slot size, one slot per satellite; the satellites' mailboxes zeroed
        a.ws_stride = ws_doubles_tp(K);
        MPCX_HIP(ctx, hipMemset2DAsync((double *)workspace + tp_mail_offset(K), a.ws_stride * sizeof(double), 0, TP_MAIL_N * sizeof(double), (size_t)S, (hipStream_t)stream));
        if (mpcxtp_launch(&a, sizeof a, S, (hipStream_t)stream) != 0) return ctx_fail(ctx, MPCX_E_HIP, "solve: time-parallel launch failed");
        lds = 0;
    } else
    if (S <= ctx->n_slots / 8 && !(opts->flags & (MPCX_SOLVE_ONE_WAVE | MPCX_SOLVE_NO_LDS))) {
        lds = mpcxl_launch(&a, sizeof a, slots, (hipStream_t)stream);
        if (lds < 0) return ctx_fail(ctx, MPCX_E_HIP, "solve: LDS-resident launch failed");
    }
    if (lds == 0) {
    } else if (S <= kTwoWaveMax && !(opts->flags & MPCX_SOLVE_ONE_WAVE)) {
        if (mpcx2w_launch(&a, sizeof a, slots, (hipStream_t)stream) != 0) return ctx_fail(ctx, MPCX_E_HIP, "solve: two-wave launch failed");
    } else
        mpcx_launch::solve(a, slots, (hipStream_t)stream);
    MPCX_HIP(ctx, hipGetLastError());
    if (adaptive) {
        // (order_valid counts the solves of this batch size recorded so far)
        const int slot = o_valid % kPredHist, n_valid = o_valid + 1 < kPredHist ? o_valid + 1 : kPredHist;
        mpcx_launch::update_prediction(S, iters, o_hist, o_prev, slot, n_valid, (hipStream_t)stream);
        MPCX_HIP(ctx, hipGetLastError());
        o_valid += 1;
        if (o_valid >= 2 * kPredHist) o_valid -= kPredHist;     // (keeps slot and n_valid as they are)
    }
    return MPCX_OK;
}

extern "C" int mpcx_solve_batch_dev(mpcx_ctx *ctx, int S, int K, const double *stage, const double *xbar,
                                    const double *ubar, const double *tf, const double *consts,
                                    const double *r_des, const mpcx_solve_opts *opts, double *X, double *U,
                                    double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                    void *workspace, void *stream)
{
    return mpcx_solve_batch_ragged_dev(ctx, S, K, nullptr, stage, xbar, ubar, tf, consts, r_des, opts, X, U, NU, tf_out, status,
                                       iters, kkt, workspace, stream);
}

extern "C" int mpcx_mpc_step_batch_ragged_dev(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *xbar, const double *ubar,
                                              const double *tf, const double *consts, const double *r_des, int flags,
                                              double max_step, const mpcx_solve_opts *opts, double *X, double *U,
                                              double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                              void *workspace, void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    if (!workspace) return ctx_fail(ctx, MPCX_E_BADARG, "mpc_step: workspace of mpcx_mpc_step_workspace_bytes(S,K) required");
    // workspace = [stage records | int32 discretize status | solver workspace]
    double *stage = (double *)workspace;
    const size_t nstage = (size_t)S * (K - 1) * MPCX_STAGE_DOUBLES;
    int32_t *dstat = (int32_t *)(stage + nstage);
    double *sws = stage + nstage + ((size_t)S + 1) / 2 + 1;
    // (a ragged batch's thrust tables have as many columns as the satellite has nodes)
    int rc = mpcx_discretize_stages_ragged_dev(ctx, S, K, Ks, K, Ks, xbar, ubar, tf, consts, flags, max_step, stage, dstat, stream);
    if (rc) return rc;
    rc = mpcx_solve_batch_ragged_dev(ctx, S, K, Ks, stage, xbar, ubar, tf, consts, r_des, opts, X, U, NU, tf_out, status, iters,
                                     kkt, sws, stream);
    if (rc) return rc;
    mpcx_launch::merge_status(S, dstat, status, (hipStream_t)stream);
    MPCX_HIP(ctx, hipGetLastError());
    return MPCX_OK;
}

extern "C" int mpcx_mpc_step_batch_dev(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *ubar,
                                       const double *tf, const double *consts, const double *r_des, int flags,
                                       double max_step, const mpcx_solve_opts *opts, double *X, double *U,
                                       double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                       void *workspace, void *stream)
{
    return mpcx_mpc_step_batch_ragged_dev(ctx, S, K, nullptr, xbar, ubar, tf, consts, r_des, flags, max_step, opts, X, U, NU,
                                          tf_out, status, iters, kkt, workspace, stream);
}

extern "C" size_t mpcx_mpc_step_workspace_bytes(int S, int K)
{
    return ((size_t)S * (K - 1) * MPCX_STAGE_DOUBLES + ((size_t)S + 1) / 2 + 1) * sizeof(double) +
           mpcx_solve_workspace_bytes(S, K);
}

extern "C" size_t mpcx_mpc_step_workspace_bytes_ctx(const mpcx_ctx *ctx, int S, int K)
{
    return ((size_t)S * (K - 1) * MPCX_STAGE_DOUBLES + ((size_t)S + 1) / 2 + 1) * sizeof(double) +
           mpcx_solve_workspace_bytes_ctx(ctx, S, K);
}

extern "C" int mpcx_mpc_step_batch_ragged(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *xbar, const double *ubar,
                                          const double *tf, const double *consts, const double *r_des, int flags,
                                          double max_step, const mpcx_solve_opts *opts, double *X, double *U, double *NU,
                                          double *tf_out, int32_t *status, int32_t *iters, double *kkt)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 3 || !opts) return ctx_fail(ctx, MPCX_E_BADARG, "mpc_step: need S>=1, K>=3 and options");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    void *ws = ctx_workspace(ctx, mpcx_mpc_step_workspace_bytes_ctx(ctx, S, K));
    if (!ws) return MPCX_E_NOMEM;
    DeviceArena ar(ctx);
    double *dx = ar.upload(xbar, (size_t)S * 7 * K), *du = ar.upload(ubar, (size_t)S * 3 * K);
    double *dtf = ar.upload(tf, S), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST), *drd = ar.upload(r_des, S);
    int32_t *dKs = Ks ? ar.upload(Ks, S) : nullptr;
    double *dX = ar.alloc<double>((size_t)S * 7 * K), *dU = ar.alloc<double>((size_t)S * 3 * K);
    const bool fixed_tf = (opts->flags & MPCX_SOLVE_FIXED_TF) != 0;          // tf_out is an input too (include/mpcx.h)
    double *dNU = ar.alloc<double>((size_t)S * 7 * K), *dtfo = fixed_tf ? ar.upload(tf_out, S) : ar.alloc<double>(S), *dk = ar.alloc<double>(S);
    int32_t *dst = ar.alloc<int32_t>(S), *dit = ar.alloc<int32_t>(S);
    if (ar.failed()) return ar.code();
    int rc = mpcx_mpc_step_batch_ragged_dev(ctx, S, K, dKs, dx, du, dtf, dc, drd, flags, max_step, opts, dX, dU, dNU, dtfo, dst,
                                            dit, dk, ws, ctx->stream);
    if (rc) return rc;
    ar.download(X, dX, (size_t)S * 7 * K); ar.download(U, dU, (size_t)S * 3 * K); ar.download(NU, dNU, (size_t)S * 7 * K);
    ar.download(tf_out, dtfo, S); ar.download(status, dst, S); ar.download(iters, dit, S); ar.download(kkt, dk, S);
    return ar.finish();
}

extern "C" int mpcx_mpc_step_batch(mpcx_ctx *ctx, int S, int K, const double *xbar, const double *ubar,
                                   const double *tf, const double *consts, const double *r_des, int flags,
                                   double max_step, const mpcx_solve_opts *opts, double *X, double *U, double *NU,
                                   double *tf_out, int32_t *status, int32_t *iters, double *kkt)
{
    return mpcx_mpc_step_batch_ragged(ctx, S, K, nullptr, xbar, ubar, tf, consts, r_des, flags, max_step, opts, X, U, NU, tf_out,
                                      status, iters, kkt);
}

// One SCP iteration of OptimalController.update (control.py:183-227) for S satellites, host buffers in and out: the nonlinear
// rollout under the given thrust law sampled at the satellite's nodes (its thrust at those nodes = extract_uk), the
// linearisation / discretisation about it and the solve -- x_bar and u_bar never leave the device.
extern "C" int mpcx_scp_iteration_batch_ragged(mpcx_ctx *ctx, int S, int K, const int32_t *Ks, const double *y0, const double *tf,
                                               const double *consts, const double *r_des, int prop_flags, int ctrl_kind,
                                               const double *ctrl_vec, int Ku, const int32_t *Kus, const double *end_tau,
                                               double prop_max_step, int disc_flags, double disc_max_step,
                                               const mpcx_solve_opts *opts, double *xbar_out, double *ubar_out, double *X, double *U,
                                               double *NU, double *tf_out, int32_t *status, int32_t *iters, double *kkt,
                                               int32_t *prop_status)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 3 || !opts || !prop_status) return ctx_fail(ctx, MPCX_E_BADARG, "scp_iteration: need S>=1, K>=3, options and prop_status");
    if (opts->flags & (MPCX_SOLVE_FIXED_TF | MPCX_SOLVE_SHARED_TF)) return ctx_fail(ctx, MPCX_E_BADARG, "scp_iteration: free per-satellite tf only");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    void *ws = ctx_workspace(ctx, mpcx_mpc_step_workspace_bytes_ctx(ctx, S, K));
    if (!ws) return MPCX_E_NOMEM;
    DeviceArena ar(ctx);
    double *dy0 = ar.upload(y0, (size_t)S * 7), *dtf = ar.upload(tf, S), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST), *drd = ar.upload(r_des, S);
    size_t nv = 0;
    if (ctrl_kind == MPCX_CTRL_CONSTANT) nv = (size_t)S * 3;
    else if (ctrl_kind == MPCX_CTRL_TANGENTIAL) nv = S;
    else if (ctrl_kind == MPCX_CTRL_SEQUENCE) nv = (size_t)S * 3 * Ku;
    double *dv = (nv && ctrl_vec) ? ar.upload(ctrl_vec, nv) : nullptr;
    double *de = (ctrl_kind == MPCX_CTRL_SEQUENCE && end_tau) ? ar.upload(end_tau, S) : nullptr;
    int32_t *dKs = Ks ? ar.upload(Ks, S) : nullptr, *dKus = Kus ? ar.upload(Kus, S) : nullptr;
    double *dx = ar.alloc<double>((size_t)S * 7 * K), *du = ar.alloc<double>((size_t)S * 3 * K);
    double *dX = ar.alloc<double>((size_t)S * 7 * K), *dU = ar.alloc<double>((size_t)S * 3 * K), *dNU = ar.alloc<double>((size_t)S * 7 * K);
    double *dtfo = ar.alloc<double>(S), *dk = ar.alloc<double>(S);
    int32_t *dst = ar.alloc<int32_t>(S), *dit = ar.alloc<int32_t>(S), *dps = ar.alloc<int32_t>(S), *dpn = ar.alloc<int32_t>(S);
    if (ar.failed()) return ar.code();
    if (Ks) {                                                                                        // the unused columns
        MPCX_HIP(ctx, hipMemsetAsync(dx, 0, (size_t)S * 7 * K * sizeof(double), ctx->stream));
        MPCX_HIP(ctx, hipMemsetAsync(du, 0, (size_t)S * 3 * K * sizeof(double), ctx->stream));
    }
    int rc = mpcx_propagate_thrust_batch_ragged_dev(ctx, S, K, dKs, dy0, dtf, dc, prop_flags, ctrl_kind, dv, Ku, dKus, de, prop_max_step,
                                                    dx, du, dps, dpn, ctx->stream);
    if (rc) return rc;
    rc = mpcx_mpc_step_batch_ragged_dev(ctx, S, K, dKs, dx, du, dtf, dc, drd, disc_flags, disc_max_step, opts, dX, dU, dNU, dtfo, dst,
                                        dit, dk, ws, ctx->stream);
    if (rc) return rc;
    if (xbar_out) ar.download(xbar_out, dx, (size_t)S * 7 * K);
    if (ubar_out) ar.download(ubar_out, du, (size_t)S * 3 * K);
    ar.download(X, dX, (size_t)S * 7 * K); ar.download(U, dU, (size_t)S * 3 * K); ar.download(NU, dNU, (size_t)S * 7 * K);
    ar.download(tf_out, dtfo, S); ar.download(status, dst, S); ar.download(iters, dit, S); ar.download(kkt, dk, S);
    ar.download(prop_status, dps, S);
    return ar.finish();
}


// OptimalController.update (control.py:170-235) for S satellites as ONE call, everything between the first input and the
// last result resident in HBM (include/mpcx.h).
extern "C" int mpcx_mpc_update_batch(mpcx_ctx *ctx, int S, int K, int n_scp, double base_res, const double *y0, const double *tf0,
                                     const double *consts, const double *r_des, double ref_thrust, double prop_max_step,
                                     int disc_flags, double disc_max_step, const mpcx_solve_opts *opts, double *X, double *U,
                                     double *NU, double *tf_out, int32_t *Ks_out, int32_t *status, int32_t *iters, double *kkt,
                                     int32_t *prop_status, double sim_tf, double sim_interval, int sim_n_eval, int sim_flags,
                                     double sim_max_step, double *y_sim, int32_t *sim_status)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 3 || n_scp < 1 || !opts || !y0 || !tf0 || !consts || !r_des || !X || !U || !NU || !tf_out || !Ks_out || !status ||
        !iters || !kkt || !prop_status || !(base_res > 0.0))
        return ctx_fail(ctx, MPCX_E_BADARG, "mpc_update: need S>=1, K>=3, n_scp>=1, base_res>0, options and all arrays");
    if (opts->flags & (MPCX_SOLVE_FIXED_TF | MPCX_SOLVE_SHARED_TF)) return ctx_fail(ctx, MPCX_E_BADARG, "mpc_update: free per-satellite tf only");
    if (y_sim && (sim_n_eval < 1 || !(sim_tf > 0.0) || !(sim_interval > 0.0) || !sim_status))
        return ctx_fail(ctx, MPCX_E_BADARG, "mpc_update: segment flight needs sim_tf>0, sim_interval>0, sim_n_eval>=1, sim_status");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    // MPCX_UPDATE_SPLIT=1 / 2 (default 0): the batch as TWO chains -- the two halves of the satellites, each rollout -> discretise ->
    // solve -> ... -> flight on its own stream (satellites are independent: every satellite gets the bits the one-chain call
    // gives it, tests/test_mpc_loop_gpu.py::test_split_update_two_chains_equal_one), so that one half's rollouts (a sequential
    // chain of ~1000 RK steps on a quarter of the SIMDs whatever S: 13.5 of 79.5 ms of kernels per two segments at 4096
    // satellites) run under the other half's solve; =2 also delays the second chain's start until the first has reached its
    // first solve.  Round-4 verdict item 5; built, measured (profiles/r05/update_split.txt) and NOT the default: the kernels do
    // overlap, and the closed loop is exactly as fast -- 45.0 against 45.1 ms per segment at 4096 satellites, 80.9 / 79.7 at
    // 8192, 28.1 / 33.1 at 2048 -- because two launches of 2048 satellites fill the 2048 wave slots in index order, which gives
    // back what the longest-first order of ONE launch of 4096 had gained (9.5 against 10.3 ms per solve).
    const char *split_env = getenv("MPCX_UPDATE_SPLIT");            // (read per call: a test switches it inside one process)
    const int split_mode = split_env ? atoi(split_env) : 0;
    const bool split = split_mode > 0 && S >= 2 * kTwoWaveMax && !(opts->flags & MPCX_SOLVE_TIME_PARALLEL);
    const int cnt[2] = {split ? (S + 1) / 2 : S, split ? S / 2 : 0}, fst[2] = {0, cnt[0]};
    size_t ws_bytes[2] = {mpcx_mpc_step_workspace_bytes_ctx(ctx, cnt[0], K), split ? mpcx_mpc_step_workspace_bytes_ctx(ctx, cnt[1], K) : 0};
    ws_bytes[0] = (ws_bytes[0] + 255) & ~(size_t)255;
    char *wsb = (char *)ctx_workspace(ctx, ws_bytes[0] + ws_bytes[1]);
    if (!wsb) return MPCX_E_NOMEM;
    if (split && !ctx->stream2) {
        // (the stream last: a call that failed half-way through these leaves stream2 null and the next call completes the set)
        if (!ctx->ev_fork) MPCX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        if (!ctx->ev_join) MPCX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        if (!ctx->ev_stagger) MPCX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_stagger, hipEventDisableTiming));
        MPCX_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    }
    DeviceArena ar(ctx);
    double *dy0 = ar.upload(y0, (size_t)S * 7), *dtf0 = ar.upload(tf0, S), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST), *drd = ar.upload(r_des, S);
    const size_t n7 = (size_t)S * 7 * K, n3 = (size_t)S * 3 * K;
    double *dx = ar.alloc<double>(n7), *du = ar.alloc<double>(n3);                      // reference trajectory / thrust of the iteration
    double *dX = ar.alloc<double>(n7), *dNU = ar.alloc<double>(n7);
    double *dU[2] = {ar.alloc<double>(n3), ar.alloc<double>(n3)};                       // plan thrust: iteration i writes dU[i & 1], the next rollout plays it
    double *dtfu[2] = {ar.alloc<double>(S), ar.alloc<double>(S)};                       // tf_u of the iterations, alternating
    double *dmag = ar.alloc<double>(S), *done = ar.alloc<double>(S), *dk = ar.alloc<double>(S), *dend = ar.alloc<double>(S);
    int32_t *dKn[2] = {ar.alloc<int32_t>(S), ar.alloc<int32_t>(S)};                     // node counts, alternating
    int32_t *dst = ar.alloc<int32_t>((size_t)n_scp * S), *dit = ar.alloc<int32_t>((size_t)n_scp * S);
    int32_t *dps = ar.alloc<int32_t>(S), *dpn = ar.alloc<int32_t>(S), *dps2 = ar.alloc<int32_t>(S);
    double *dys = y_sim ? ar.alloc<double>((size_t)S * 7 * sim_n_eval) : nullptr;
    int32_t *dss = y_sim ? ar.alloc<int32_t>(S) : nullptr;
    if (ar.failed()) return ar.code();
    if (split) {
        // (the regularisation record is sized for the whole batch BEFORE anything is enqueued: no allocation under a running half)
        if (ctx->nreg_cap < S) {
            if (ctx->nreg) (void)hipFree(ctx->nreg);
            ctx->nreg = nullptr; ctx->nreg_cap = 0;
            MPCX_HIP(ctx, hipMalloc((void **)&ctx->nreg, (size_t)S * 2 * sizeof(int32_t)));
            ctx->nreg_cap = S;
        }
        MPCX_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));                   // the uploads are behind this point of the first stream
        MPCX_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    }
    const double *tf_fin = dtf0;
    const int32_t *Ks_fin = nullptr;
    const double *Uplan = dU[(n_scp - 1) & 1];
    // one half's chain: satellites f .. f + n - 1 on stream st, with its own workspace and (second half) launch-order state
    auto chain = [&](int f, int n, hipStream_t st, void *ws, int lane) -> int {
        const size_t o1 = (size_t)f, o7 = (size_t)f * 7 * K, o3 = (size_t)f * 3 * K;
        mpcx_launch::fill_f64(n, ref_thrust, dmag + o1, st);
        mpcx_launch::fill_f64(n, 1.0, done + o1, st);
        MPCX_HIP(ctx, hipMemsetAsync(dps + o1, 0, sizeof(int32_t) * n, st));
        const double *tf_cur = dtf0 + o1;
        const int32_t *Ks = nullptr;            // node counts of the current iteration (nullptr: K for everybody)
        int rc = MPCX_OK;
        ctx->cur_lane = lane; ctx->nreg_first = f; ctx->nreg_total = split ? S : 0;
        for (int it = 0; it < n_scp && rc == MPCX_OK; ++it) {
            double *Uw = dU[it & 1] + o3, *tfw = dtfu[it & 1] + o1;
            if (Ks) {                                                                         // ragged rows: the unused columns
                MPCX_HIP(ctx, hipMemsetAsync(dx + o7, 0, (size_t)n * 7 * K * sizeof(double), st));
                MPCX_HIP(ctx, hipMemsetAsync(du + o3, 0, (size_t)n * 3 * K * sizeof(double), st));
            }
            // control.py:178-180 / :217-227: rollout under the tangential reference law, then under the sequence just optimised,
            // played over its own horizon (end_tau = 1) and sampled at int(base_res * tf_u) nodes; u_bar = extract_uk (:188)
            if (it == 0)
                rc = mpcx_propagate_thrust_batch_ragged_dev(ctx, n, K, nullptr, dy0 + o1 * 7, tf_cur, dc + o1 * MPCX_NCONST, 0, MPCX_CTRL_TANGENTIAL,
                                                            dmag + o1, 0, nullptr, nullptr, prop_max_step, dx + o7, du + o3, dps2 + o1, dpn + o1, st);
            else
                rc = mpcx_propagate_thrust_batch_ragged_dev(ctx, n, K, Ks, dy0 + o1 * 7, tf_cur, dc + o1 * MPCX_NCONST, 0, MPCX_CTRL_SEQUENCE,
                                                            dU[(it - 1) & 1] + o3, K, it >= 2 ? dKn[(it - 1) & 1] + o1 : nullptr, done + o1,
                                                            prop_max_step, dx + o7, du + o3, dps2 + o1, dpn + o1, st);
            if (rc) break;
            mpcx_launch::merge_status(n, dps2 + o1, dps + o1, st);                          // (any rollout's failure is the update's)
            if (split && lane == 0 && it == 0 && split_mode == 2) MPCX_HIP(ctx, hipEventRecord(ctx->ev_stagger, st));
            rc = mpcx_mpc_step_batch_ragged_dev(ctx, n, K, Ks, dx + o7, du + o3, tf_cur, dc + o1 * MPCX_NCONST, drd + o1, disc_flags, disc_max_step,
                                                opts, dX + o7, Uw, dNU + o7, tfw, dst + (size_t)it * S + o1, dit + (size_t)it * S + o1, dk + o1, ws, st);
            if (rc) break;
            tf_cur = tfw;
            if (it + 1 < n_scp) {
                int32_t *kn = dKn[(it + 1) & 1] + o1;
                mpcx_launch::node_count(n, base_res, tfw, kn, st);
                Ks = kn;
            }
        }
        ctx->cur_lane = 0; ctx->nreg_first = 0; ctx->nreg_total = 0;
        if (rc) return rc;
        MPCX_HIP(ctx, hipGetLastError());
        if (f == 0) { tf_fin = tf_cur; Ks_fin = Ks; }                                       // (the whole-batch arrays the downloads read)
        if (y_sim) {
            // Simulator.run_segment (simulator.py:58-65): fly sim_tf under the truth model with SequenceController(u_opt, tf_u,
            // tf_sim = sim_interval): end_tau = tf_u / sim_interval (control.py:102), the plan's table with its own column count
            // (end_tau as the host computes it: a division, not a product with the reciprocal)
            mpcx_launch::divide_f64(n, tf_cur, sim_interval, dend + o1, st);
            mpcx_launch::fill_f64(n, sim_tf, dmag + o1, st);           // (dmag is free again: the flight time per satellite)
            rc = mpcx_propagate_thrust_batch_ragged_dev(ctx, n, sim_n_eval, nullptr, dy0 + o1 * 7, dmag + o1, dc + o1 * MPCX_NCONST, sim_flags,
                                                        MPCX_CTRL_SEQUENCE, Uplan + o3, K, Ks, dend + o1, sim_max_step, dys + o1 * 7 * sim_n_eval, nullptr,
                                                        dss + o1, dpn + o1, st);
            if (rc) return rc;
        }
        return MPCX_OK;
    };
    int rc = MPCX_OK;
    rc = chain(fst[0], cnt[0], ctx->stream, wsb, 0);
    ctx->cur_lane = 0; ctx->nreg_first = 0; ctx->nreg_total = 0;          // (also when the chain left early on an error)
    if (rc == MPCX_OK && split) {
        // (mode 2: the second chain starts when the first has reached its first solve -- its rollout and discretisation then run UNDER it)
        if (split_mode == 2) MPCX_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_stagger, 0));
        rc = chain(fst[1], cnt[1], ctx->stream2, wsb + ws_bytes[0], 1);
        ctx->cur_lane = 0; ctx->nreg_first = 0; ctx->nreg_total = 0;
    }
    if (split) {
        // join: the downloads on the first stream follow everything of the second (also on an error path: nothing of this call
        // is left running on the second stream when the arena's buffers are handed to the next call)
        (void)hipEventRecord(ctx->ev_join, ctx->stream2);
        (void)hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
    }
    if (rc) return rc;
    const double *tf_cur = tf_fin;
    const int32_t *Ks = Ks_fin;
    ar.download(X, dX, n7); ar.download(U, (const double *)Uplan, n3); ar.download(NU, dNU, n7);
    ar.download(tf_out, tf_cur, S);
    if (Ks) ar.download(Ks_out, Ks, S);
    else for (int i = 0; i < S; ++i) Ks_out[i] = K;                                      // (a single iteration: K nodes for everybody)
    ar.download(status, dst, (size_t)n_scp * S); ar.download(iters, dit, (size_t)n_scp * S); ar.download(kkt, dk, S);
    ar.download(prop_status, dps, S);
    if (y_sim) { ar.download(y_sim, dys, (size_t)S * 7 * sim_n_eval); ar.download(sim_status, dss, S); }
    return ar.finish();
}

extern "C" int mpcx_solve_batch(mpcx_ctx *ctx, int S, int K, const double *A, const double *Bp, const double *Bn,
                                const double *Sigma, const double *xi, const double *xbar, const double *ubar,
                                const double *tf, const double *consts, const double *r_des,
                                const mpcx_solve_opts *opts, double *X, double *U, double *NU, double *tf_out,
                                int32_t *status, int32_t *iters, double *kkt)
{
    if (!ctx) return MPCX_E_BADARG;
    if (S < 1 || K < 3 || !opts) return ctx_fail(ctx, MPCX_E_BADARG, "solve: need S>=1, K>=3 and options");
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    // pack the reference-shaped arrays into stage records on the host (tiny, O(S K) copies)
    const size_t n = (size_t)S * (K - 1);
    std::vector<double> st(n * MPCX_STAGE_DOUBLES);
    for (int s = 0; s < S; ++s)
        for (int k = 0; k < K - 1; ++k) {
            double *r = &st[((size_t)s * (K - 1) + k) * MPCX_STAGE_DOUBLES];
            const size_t b = (size_t)s * (K - 1) + k;
            for (int e = 0; e < 49; ++e) r[e] = A[b * 49 + e];
            for (int e = 0; e < 21; ++e) { r[49 + e] = Bn[b * 21 + e]; r[70 + e] = Bp[b * 21 + e]; }
            for (int i = 0; i < 7; ++i) {
                r[91 + i] = Sigma[(size_t)s * 7 * (K - 1) + (size_t)i * (K - 1) + k];
                r[98 + i] = xi[(size_t)s * 7 * (K - 1) + (size_t)i * (K - 1) + k];
            }
        }
    void *ws = ctx_workspace(ctx, mpcx_solve_workspace_bytes_ctx(ctx, S, K));
    if (!ws) return MPCX_E_NOMEM;
    DeviceArena ar(ctx);
    double *dst_ = ar.upload(st.data(), st.size());
    double *dx = ar.upload(xbar, (size_t)S * 7 * K), *du = ar.upload(ubar, (size_t)S * 3 * K);
    double *dtf = ar.upload(tf, S), *dc = ar.upload(consts, (size_t)S * MPCX_NCONST), *drd = ar.upload(r_des, S);
    double *dX = ar.alloc<double>((size_t)S * 7 * K), *dU = ar.alloc<double>((size_t)S * 3 * K);
    const bool fixed_tf = (opts->flags & MPCX_SOLVE_FIXED_TF) != 0;          // tf_out is an input too (include/mpcx.h)
    double *dNU = ar.alloc<double>((size_t)S * 7 * K), *dtfo = fixed_tf ? ar.upload(tf_out, S) : ar.alloc<double>(S), *dk = ar.alloc<double>(S);
    int32_t *dstat = ar.alloc<int32_t>(S), *dit = ar.alloc<int32_t>(S);
    if (ar.failed()) return ar.code();
    int rc = mpcx_solve_batch_dev(ctx, S, K, dst_, dx, du, dtf, dc, drd, opts, dX, dU, dNU, dtfo, dstat, dit, dk, ws,
                                  ctx->stream);
    if (rc) return rc;
    ar.download(X, dX, (size_t)S * 7 * K); ar.download(U, dU, (size_t)S * 3 * K); ar.download(NU, dNU, (size_t)S * 7 * K);
    ar.download(tf_out, dtfo, S); ar.download(status, dstat, S); ar.download(iters, dit, S); ar.download(kkt, dk, S);
    return ar.finish();
}

