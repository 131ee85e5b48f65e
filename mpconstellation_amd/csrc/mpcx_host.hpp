// mpcx_host.hpp -- host-side plumbing of libmpcx.so: context, error reporting, staging arena.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../include/mpcx.h"

struct mpcx_ctx {
    int device;
    hipStream_t stream;       // stream used by the host-pointer entry points
    char err[512];
    // grow-only device workspace reused by the solver / fused step (never freed between calls)
    void *ws;
    size_t ws_bytes;
    // launch order of the solver's workgroups: the previous solve's iteration counts (library-owned copy) sorted
    // longest first; valid only for a following solve of the same batch size
    int32_t *prev_iters, *order;
    int order_S, order_valid, order_cap;
};

inline int ctx_fail(mpcx_ctx *ctx, int code, const char *msg)
{
    if (ctx) snprintf(ctx->err, sizeof ctx->err, "%s", msg);
    return code;
}

#define MPCX_HIP(ctx, call)                                                                   \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            if (ctx) snprintf((ctx)->err, sizeof (ctx)->err, "%s:%d %s -> %s", __FILE__,      \
                              __LINE__, #call, hipGetErrorString(e_));                        \
            return MPCX_E_HIP;                                                                \
        }                                                                                     \
    } while (0)

// Device workspace of at least `bytes` (grow-only).  Returns nullptr on failure.
inline void *ctx_workspace(mpcx_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->ws_bytes) return ctx->ws;
    if (ctx->ws) { (void)hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_bytes = 0; }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { ctx_fail(ctx, MPCX_E_NOMEM, "workspace allocation failed"); return nullptr; }
    ctx->ws = p; ctx->ws_bytes = bytes;
    return p;
}

// RAII staging buffers for the host-pointer entry points.
class DeviceArena {
  public:
    explicit DeviceArena(mpcx_ctx *c) : ctx_(c), code_(0) {}
    ~DeviceArena() { for (void *p : ptrs_) (void)hipFree(p); }
    template <typename T> T *alloc(size_t n)
    {
        void *p = nullptr;
        if (code_) return nullptr;
        hipError_t e = hipMalloc(&p, n * sizeof(T) + 16);
        if (e != hipSuccess) { code_ = ctx_fail(ctx_, MPCX_E_NOMEM, hipGetErrorString(e)); return nullptr; }
        ptrs_.push_back(p);
        return (T *)p;
    }
    template <typename T> T *upload(const T *h, size_t n)
    {
        T *d = alloc<T>(n);
        if (!d) return nullptr;
        hipError_t e = hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, ctx_->stream);
        if (e != hipSuccess) code_ = ctx_fail(ctx_, MPCX_E_HIP, hipGetErrorString(e));
        return d;
    }
    template <typename T> void download(T *h, const T *d, size_t n)
    {
        if (code_ || !h) return;
        hipError_t e = hipMemcpyAsync(h, d, n * sizeof(T), hipMemcpyDeviceToHost, ctx_->stream);
        if (e != hipSuccess) code_ = ctx_fail(ctx_, MPCX_E_HIP, hipGetErrorString(e));
    }
    bool failed() const { return code_ != 0; }
    int code() const { return code_; }

  private:
    mpcx_ctx *ctx_;
    int code_;
    std::vector<void *> ptrs_;
};
