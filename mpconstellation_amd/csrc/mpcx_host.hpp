// mpcx_host.hpp -- host-side plumbing of libmpcx.so: context, error reporting, staging arena.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/mpcx.h"

// Host-side copies between the caller's pageable arrays and the context's page-locked staging, spread over a few
// persistent worker threads.  A caller's result arrays are usually fresh allocations (numpy hands back newly mapped pages
// for every large array): the first write to each 4 KB page is a page fault, and one thread filling 17 MB of them was
// the largest single item of a 4096-satellite host-pointer call (round 2: 20.9 ms against 9.5 ms with page-locked
// arrays).  Faults of different threads are served concurrently.
class HostCopier {
  public:
    explicit HostCopier(int n) : stop_(false), pending_(0)
    {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
    }
    ~HostCopier()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    // copy `bytes` from src to dst in page-aligned slices, returns when all of it is done
    void copy(void *dst, const void *src, size_t bytes)
    {
        const size_t kMin = (size_t)1 << 20;                 // below 1 MB per slice a thread hand-off costs more than it saves
        size_t n = workers_.empty() ? 1 : (bytes + kMin - 1) / kMin;
        if (n > workers_.size() + 1) n = workers_.size() + 1;
        if (n <= 1) { memcpy(dst, src, bytes); return; }
        size_t slice = ((bytes + n - 1) / n + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> g(m_);
            for (size_t off = slice; off < bytes; off += slice) {
                jobs_.push_back({(char *)dst + off, (const char *)src + off, bytes - off < slice ? bytes - off : slice});
                ++pending_;
            }
        }
        cv_.notify_all();
        memcpy(dst, src, slice < bytes ? slice : bytes);     // the calling thread takes the first slice
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
    }

  private:
    struct Job { char *dst; const char *src; size_t bytes; };
    void run()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return stop_ || !jobs_.empty(); });
                if (stop_ && jobs_.empty()) return;
                j = jobs_.back(); jobs_.pop_back();
            }
            memcpy(j.dst, j.src, j.bytes);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::vector<Job> jobs_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    bool stop_;
    size_t pending_;
};

// Staging for the host-pointer entry points.  The buffers belong to the context and only grow: a device pool and a
// pinned host pool, each a list of chunks that a call bump-allocates from and the next call reuses from the start --
// in steady state a call allocates nothing (the first version paid a hipMalloc / hipFree pair per array and moved
// pageable memory: +12 ms on a 4096-satellite step).  Uploads go caller -> pinned -> HBM, downloads HBM -> pinned and,
// once the stream has drained (finish()), pinned -> caller.
struct StagePool {
    struct Chunk { char *p; size_t cap; };
    std::vector<Chunk> chunks;
    size_t cur, off;          // chunk in use and bump offset inside it
    bool pinned;
};

inline void *pool_take(StagePool &pl, size_t bytes)
{
    bytes = (bytes + 255) & ~(size_t)255;
    while (pl.cur < pl.chunks.size()) {
        if (pl.off + bytes <= pl.chunks[pl.cur].cap) { void *r = pl.chunks[pl.cur].p + pl.off; pl.off += bytes; return r; }
        ++pl.cur; pl.off = 0;
    }
    size_t cap = bytes;
    for (const auto &c : pl.chunks) cap = cap > c.cap ? cap : c.cap;
    cap = cap > bytes ? cap * 2 : bytes;                    // new chunks at least double the largest so far
    void *p = nullptr;
    hipError_t e = pl.pinned ? hipHostMalloc(&p, cap, hipHostMallocDefault) : hipMalloc(&p, cap);
    if (e != hipSuccess) return nullptr;
    pl.chunks.push_back({(char *)p, cap});
    pl.cur = pl.chunks.size() - 1; pl.off = bytes;
    return p;
}

inline void pool_reset(StagePool &pl) { pl.cur = 0; pl.off = 0; }

inline void pool_free(StagePool &pl)
{
    for (auto &c : pl.chunks) { if (pl.pinned) (void)hipHostFree(c.p); else (void)hipFree(c.p); }
    pl.chunks.clear(); pl.cur = pl.off = 0;
}

struct mpcx_ctx {
    int device;
    hipStream_t stream;       // stream used by the host-pointer entry points
    bool own_stream;          // created by the library (mpcx_create / MPCX_STREAM_PRIVATE), not handed in by mpcx_set_stream
    char err[512];
    // grow-only device workspace reused by the solver / fused step (never freed between calls)
    void *ws;
    size_t ws_bytes;
    // launch order of the solver's workgroups: the previous solve's iteration counts (library-owned copy) sorted
    // longest first; valid only for a following solve of the same batch size
    int32_t *prev_iters, *order, *pred_hist;     // prev_iters: the prediction the order is sorted by; pred_hist: the last solves' counts
    int order_S, order_valid, order_cap;
    // the same state for the SECOND half of a split update (mpcx_mpc_update_batch runs the two halves of a large batch as two
    // chains on two streams: each half is its own sequence of solves of its own batch size)
    struct OrderState { int32_t *prev_iters, *order, *pred_hist; int order_S, order_valid, order_cap; } ord2;
    int cur_lane;             // 0: the state above; 1: ord2 (set around the second half's solves only)
    // regularisation counts of the last solve ([S][2] int32, include/mpcx.h: mpcx_solve_regularised)
    int32_t *nreg;
    int nreg_cap, nreg_S;
    int nreg_first, nreg_total;        // split update: this solve's satellites start at nreg_first of a record of nreg_total
    hipStream_t stream2;               // the second half's stream and the events that fork / join it (created on first use)
    hipEvent_t ev_fork, ev_join, ev_stagger;
    int32_t *counter;         // ring of work-queue counters of the solver's persistent workgroups (one per launch in flight)
    unsigned launch_seq;      // solves launched so far: selects the counter
    int n_slots;              // single-wave workgroups of solve_kernel the device holds at once (compute units x 8)
    double *red;              // shared-tf launches: reduction slots + arrival counter + abort flag
    int red_cap, coop_max;    // coop_max: workgroups of solve_shared_kernel resident at once (0: not asked yet, -1: unsupported)
    int tp_max;               // satellites the time-parallel kernel holds at once (0: not asked yet, -1: query failed)
    int trace_on;             // mpcx_trace_enable: every host-pointer call records where its time went (last_trace), without the env switch
    double last_trace[MPCX_TRACE_N];   // the last traced call's record (include/mpcx.h: MPCX_TR_*)
    StagePool pool_dev, pool_host;     // staging of the host-pointer entry points
    HostCopier *copier;                // worker threads of the pageable <-> page-locked copies (created on first use)
    std::vector<hipEvent_t> events;    // one per download of a host-pointer call: its copy-out starts when ITS transfer is done
};

inline HostCopier *ctx_copier(mpcx_ctx *ctx)
{
    if (!ctx->copier) {
        unsigned hw = std::thread::hardware_concurrency();
        int n = hw >= 16 ? 7 : (hw >= 8 ? 3 : (hw >= 4 ? 1 : 0));      // + the calling thread
        ctx->copier = new HostCopier(n);
    }
    return ctx->copier;
}

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

// Diagnostic of the host-pointer path (MPCX_HOST_TRACE=<ms> in the environment): a call that takes longer than <ms>
// milliseconds reports on stderr where its time went -- host copies into the staging pool, enqueueing, waiting for each
// transfer's event, copy-out, the final stream synchronisation.  Off (one getenv per process) it costs a branch per mark.
inline double host_trace_threshold_ms()
{
    static const double t = [] { const char *e = getenv("MPCX_HOST_TRACE"); return e ? atof(e) : -1.0; }();
    return t;
}
struct HostTrace {
    enum { CAP = 96 };
    const char *name[CAP];
    double t[CAP];
    int n;
    bool on;
    explicit HostTrace(bool force = false) : n(0), on(force || host_trace_threshold_ms() >= 0.0) { mark("enter"); }
    static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void mark(const char *what) { if (on && n < CAP) { name[n] = what; t[n] = now(); ++n; } }
    // sum of the segments whose name starts with `prefix` (a segment is named by the mark that ENDS it)
    double sum(const char *prefix) const
    {
        double acc = 0.0;
        const size_t len = strlen(prefix);
        for (int i = 1; i < n; ++i) if (strncmp(name[i], prefix, len) == 0) acc += t[i] - t[i - 1];
        return acc;
    }
    void report(const char *call)
    {
        if (!on || n < 2 || host_trace_threshold_ms() < 0.0 || t[n - 1] - t[0] < host_trace_threshold_ms()) return;
        fprintf(stderr, "[mpcx host trace] %s: %.3f ms:", call, t[n - 1] - t[0]);
        for (int i = 1; i < n; ++i) fprintf(stderr, " %s %.3f", name[i], t[i] - t[i - 1]);
        fprintf(stderr, "\n");
    }
};

// Bump allocator over the two pools for the duration of one host-pointer call.
class DeviceArena {
  public:
    explicit DeviceArena(mpcx_ctx *c) : trace(c->trace_on != 0), ctx_(c), code_(0), dirty_(false)
    {
        pool_reset(c->pool_dev); pool_reset(c->pool_host);
        for (auto &e : dev_ev_) e = nullptr;
        if (trace.on) for (auto &e : dev_ev_) if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        dev_mark(0);
        // tracing only: how long the stream takes to execute the call's first packet -- a marker alone, polled: bench.py's
        // slow warm-up calls (40 ms against 8.8 ms) spend their extra time HERE, before any work of the call runs; the
        // device work (device view below) and the waits for it are as fast as in the steady state (profiles/r04/host_wait.txt)
        if (trace.on && dev_ev_[0]) { while (hipEventQuery(dev_ev_[0]) == hipErrorNotReady) {} trace.mark("q:first-marker"); }
    }
    HostTrace trace;
    // tracing only: device-side time stamps on the context's stream -- 0 call start, 1 last upload done (kernels follow),
    // 2 first download queued (kernels done), 3 last download queued -- so that a slow call can be split into transfer in,
    // kernels and transfer out AS THE DEVICE SAW THEM, beside the host's view of the same call
    hipEvent_t dev_ev_[4];
    int dev_stage_ = 0;
    bool have_up_ = false;
    void dev_mark(int i) { if (trace.on && dev_ev_[i]) (void)hipEventRecord(dev_ev_[i], ctx_->stream); }
    // A call that returns early (an error after its first transfer was queued) leaves copies from / to the staging pools in
    // flight; the next call resets the pools and would overwrite them.  The stream is drained before that can happen.
    ~DeviceArena()
    {
        if (dirty_) (void)hipStreamSynchronize(ctx_->stream);
        if (trace.on) for (auto &e : dev_ev_) if (e) (void)hipEventDestroy(e);
    }
    DeviceArena(const DeviceArena &) = delete;
    DeviceArena &operator=(const DeviceArena &) = delete;
    template <typename T> T *alloc(size_t n)
    {
        if (code_) return nullptr;
        void *p = pool_take(ctx_->pool_dev, n * sizeof(T) + 16);
        if (!p) { code_ = ctx_fail(ctx_, MPCX_E_NOMEM, "device staging allocation failed"); return nullptr; }
        return (T *)p;
    }
    // true if the caller's buffer is page-locked (mpcx_host_alloc, hipHostMalloc, hipHostRegister): the DMA engine reads /
    // writes it directly, no staging copy
    static bool is_pinned(const void *h)
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, h) != hipSuccess) { (void)hipGetLastError(); return false; }
        return at.type == hipMemoryTypeHost;
    }
    template <typename T> T *upload(const T *h, size_t n)
    {
        T *d = alloc<T>(n);
        if (!d) return nullptr;
        dirty_ = true;
        trace.mark("up:alloc");
        if (is_pinned(h)) {
            trace.mark("up:attr");
            hipError_t e = hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, ctx_->stream);
            trace.mark("up:enq");
            dev_mark(1); have_up_ = true;
            if (e != hipSuccess) code_ = ctx_fail(ctx_, MPCX_E_HIP, hipGetErrorString(e));
            return d;
        }
        trace.mark("up:attr");
        void *pin = pool_take(ctx_->pool_host, n * sizeof(T));
        if (!pin) { code_ = ctx_fail(ctx_, MPCX_E_NOMEM, "pinned staging allocation failed"); return nullptr; }
        trace.mark("up:take");
        ctx_copier(ctx_)->copy(pin, h, n * sizeof(T));
        trace.mark("up:copy");
        hipError_t e = hipMemcpyAsync(d, pin, n * sizeof(T), hipMemcpyHostToDevice, ctx_->stream);
        trace.mark("up:enq");
        dev_mark(1); have_up_ = true;          // (re-recorded by every upload: it ends up behind the LAST one, in front of the kernels)
        if (e != hipSuccess) code_ = ctx_fail(ctx_, MPCX_E_HIP, hipGetErrorString(e));
        return d;
    }
    template <typename T> void download(T *h, const T *d, size_t n)
    {
        if (code_ || !h) return;
        dirty_ = true;
        if (dev_stage_ == 0) { dev_mark(2); dev_stage_ = 1; }
        trace.mark("kernels");
        if (is_pinned(h)) {
            hipError_t e = hipMemcpyAsync(h, d, n * sizeof(T), hipMemcpyDeviceToHost, ctx_->stream);
            trace.mark("down:enq");
            if (e != hipSuccess) code_ = ctx_fail(ctx_, MPCX_E_HIP, hipGetErrorString(e));
            return;
        }
        void *pin = pool_take(ctx_->pool_host, n * sizeof(T));
        if (!pin) { code_ = ctx_fail(ctx_, MPCX_E_NOMEM, "pinned staging allocation failed"); return; }
        hipError_t e = hipMemcpyAsync(pin, d, n * sizeof(T), hipMemcpyDeviceToHost, ctx_->stream);
        if (e != hipSuccess) { code_ = ctx_fail(ctx_, MPCX_E_HIP, hipGetErrorString(e)); return; }
        // large transfers get their own event: the copy into the caller's array starts as soon as this transfer is done,
        // while the following ones are still on the bus
        hipEvent_t ev = nullptr;
        if (n * sizeof(T) >= ((size_t)1 << 20)) {
            if (out_ev_ >= ctx_->events.size()) {
                hipEvent_t ne;
                if (hipEventCreateWithFlags(&ne, hipEventDisableTiming) == hipSuccess) ctx_->events.push_back(ne);
            }
            if (out_ev_ < ctx_->events.size()) {
                ev = ctx_->events[out_ev_++];
                if (hipEventRecord(ev, ctx_->stream) != hipSuccess) ev = nullptr;
            }
        }
        out_.push_back({h, pin, n * sizeof(T), ev});
        trace.mark("down:enq");
    }
    // wait for the stream, then hand the downloads to the caller's buffers
    int finish()
    {
        if (code_) return code_;
        dev_mark(3);
        HostCopier *cp = ctx_copier(ctx_);
        for (const auto &o : out_) {                         // (stream order: an event's transfer done = all earlier ones done)
            if (!o.ev) continue;
            hipError_t e = hipEventSynchronize(o.ev);
            trace.mark("fin:event");
            if (e != hipSuccess) return ctx_fail(ctx_, MPCX_E_HIP, hipGetErrorString(e));
            cp->copy(o.dst, o.src, o.bytes);
            trace.mark("fin:copy");
        }
        hipError_t e = hipStreamSynchronize(ctx_->stream);
        trace.mark("fin:sync");
        if (e != hipSuccess) return ctx_fail(ctx_, MPCX_E_HIP, hipGetErrorString(e));
        dirty_ = false;
        for (const auto &o : out_) if (!o.ev) memcpy(o.dst, o.src, o.bytes);
        trace.mark("fin:small");
        if (trace.on && dev_ev_[0] && dev_ev_[2] && dev_ev_[3]) {
            float a = 0.f, b = 0.f;
            (void)hipEventElapsedTime(&a, dev_ev_[0], dev_ev_[2]); (void)hipEventElapsedTime(&b, dev_ev_[2], dev_ev_[3]);
            if (host_trace_threshold_ms() >= 0.0 && trace.t[trace.n - 1] - trace.t[0] >= host_trace_threshold_ms())
                fprintf(stderr, "[mpcx host trace] device view: uploads + kernels %.3f ms, downloads %.3f ms\n", a, b);
            // the call's record for mpcx_last_call_trace: the host's segments by kind, the device's own time stamps
            double *r = ctx_->last_trace;
            r[MPCX_TR_WALL] = trace.t[trace.n - 1] - trace.t[0];
            r[MPCX_TR_FIRST_MARKER] = trace.sum("q:");
            r[MPCX_TR_HOST_STAGE] = trace.sum("up:") + trace.sum("kernels") + trace.sum("down:");
            r[MPCX_TR_HOST_WAIT] = trace.sum("fin:event") + trace.sum("fin:sync");
            r[MPCX_TR_HOST_COPYOUT] = trace.sum("fin:copy") + trace.sum("fin:small");
            float kk = a;                                   // kernels alone: last upload done -> first download queued
            if (have_up_ && dev_ev_[1]) (void)hipEventElapsedTime(&kk, dev_ev_[1], dev_ev_[2]);
            r[MPCX_TR_DEV_SPAN] = (double)a + (double)b;
            r[MPCX_TR_DEV_KERNELS] = (double)kk;
            r[MPCX_TR_VALID] = 1.0;
        }
        trace.report("host-pointer call");
        return MPCX_OK;
    }
    bool failed() const { return code_ != 0; }
    int code() const { return code_; }

  private:
    struct Out { void *dst; const void *src; size_t bytes; hipEvent_t ev; };
    size_t out_ev_ = 0;
    mpcx_ctx *ctx_;
    int code_;
    bool dirty_;              // transfers queued and not yet waited for
    std::vector<Out> out_;
};
