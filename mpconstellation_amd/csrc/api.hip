// api.hip -- context management of libmpcx.so (see include/mpcx.h).
#include "mpcx_host.hpp"

static char g_create_err[512] = "";

extern "C" int mpcx_version(void) { return MPCX_VERSION; }

extern "C" int mpcx_create(int device, mpcx_ctx **out)
{
    if (!out) return MPCX_E_BADARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        snprintf(g_create_err, sizeof g_create_err,
                 "no HIP device available (%s); libmpcx has no CPU fallback",
                 e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return MPCX_E_NODEVICE;
    }
    if (device < 0 || device >= n) {
        snprintf(g_create_err, sizeof g_create_err, "device %d out of range [0,%d)", device, n);
        return MPCX_E_BADARG;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        snprintf(g_create_err, sizeof g_create_err, "hipGetDeviceProperties failed");
        return MPCX_E_HIP;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_create_err, sizeof g_create_err,
                 "device %d is %s; libmpcx is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return MPCX_E_NODEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) return MPCX_E_HIP;
    mpcx_ctx *c = new mpcx_ctx();
    c->device = device; c->err[0] = 0; c->ws = nullptr; c->ws_bytes = 0;
    c->prev_iters = nullptr; c->order = nullptr; c->pred_hist = nullptr; c->order_S = 0; c->order_valid = 0; c->order_cap = 0;
    c->nreg = nullptr; c->nreg_cap = 0; c->nreg_S = 0; c->nreg_first = 0; c->nreg_total = 0;
    c->ord2 = {nullptr, nullptr, nullptr, 0, 0, 0}; c->cur_lane = 0;
    c->stream2 = nullptr; c->ev_fork = c->ev_join = c->ev_stagger = nullptr;
    c->copier = nullptr;
    c->counter = nullptr; c->launch_seq = 0; c->n_slots = prop.multiProcessorCount * 8;
    c->red = nullptr; c->red_cap = 0; c->coop_max = 0; c->tp_max = 0;
    c->trace_on = 0; for (double &v : c->last_trace) v = 0.0;
    c->pool_dev.cur = c->pool_dev.off = 0; c->pool_dev.pinned = false;
    c->pool_host.cur = c->pool_host.off = 0; c->pool_host.pinned = true;
    c->own_stream = true;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        snprintf(g_create_err, sizeof g_create_err, "hipStreamCreate failed");
        return MPCX_E_HIP;
    }
    *out = c;
    return MPCX_OK;
}

extern "C" void mpcx_destroy(mpcx_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->prev_iters) (void)hipFree(ctx->prev_iters);
    if (ctx->order) (void)hipFree(ctx->order);
    if (ctx->pred_hist) (void)hipFree(ctx->pred_hist);
    if (ctx->nreg) (void)hipFree(ctx->nreg);
    if (ctx->ord2.prev_iters) (void)hipFree(ctx->ord2.prev_iters);
    if (ctx->ord2.order) (void)hipFree(ctx->ord2.order);
    if (ctx->ord2.pred_hist) (void)hipFree(ctx->ord2.pred_hist);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    for (hipEvent_t e : {ctx->ev_fork, ctx->ev_join, ctx->ev_stagger}) if (e) (void)hipEventDestroy(e);
    if (ctx->counter) (void)hipFree(ctx->counter);
    if (ctx->red) (void)hipFree(ctx->red);
    pool_free(ctx->pool_dev); pool_free(ctx->pool_host);
    for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
    delete ctx->copier;
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int mpcx_set_stream(mpcx_ctx *ctx, void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    MPCX_HIP(ctx, hipStreamSynchronize(ctx->stream));                  // nothing of this context is left on the old stream
    // (the new stream exists before the old one goes: a failed creation leaves the context on its old, valid stream)
    hipStream_t fresh = (hipStream_t)stream;                           // (NULL: the device's default stream)
    if (stream == MPCX_STREAM_PRIVATE) MPCX_HIP(ctx, hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = fresh;
    ctx->own_stream = (stream == MPCX_STREAM_PRIVATE);
    return MPCX_OK;
}

extern "C" int mpcx_trace_enable(mpcx_ctx *ctx, int on)
{
    if (!ctx) return MPCX_E_BADARG;
    ctx->trace_on = on ? 1 : 0;
    ctx->last_trace[MPCX_TR_VALID] = 0.0;
    return MPCX_OK;
}

extern "C" int mpcx_last_call_trace(const mpcx_ctx *ctx, double *out, int n)
{
    if (!ctx || !out || n < 1) return MPCX_E_BADARG;
    for (int i = 0; i < n; ++i) out[i] = i < MPCX_TRACE_N ? ctx->last_trace[i] : 0.0;
    return MPCX_OK;
}

extern "C" const char *mpcx_last_error(const mpcx_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

extern "C" int mpcx_synchronize(mpcx_ctx *ctx, void *stream)
{
    if (!ctx) return MPCX_E_BADARG;
    MPCX_HIP(ctx, hipSetDevice(ctx->device));
    MPCX_HIP(ctx, hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream));
    return MPCX_OK;
}

extern "C" void *mpcx_host_alloc(mpcx_ctx *ctx, size_t bytes)
{
    if (!ctx || !bytes) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    void *p = nullptr;
    // (portable: a multi-device call's result set is the DMA target of every device)
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { ctx_fail(ctx, MPCX_E_NOMEM, "page-locked allocation failed"); return nullptr; }
    return p;
}

extern "C" void mpcx_host_free(mpcx_ctx *ctx, void *p)
{
    if (ctx) (void)hipSetDevice(ctx->device);
    if (p) (void)hipHostFree(p);
}
