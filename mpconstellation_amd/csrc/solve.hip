// solve.hip -- batched per-satellite solve of the SCP subproblem on gfx950: the KERNELS (one wave per satellite) and their
// launchers.  The C entry points are in solve_api.hip; the device code proper in solve_phases.hpp (node-parallel phases),
// solve_riccati.hpp (the structured linear solve) and solve_driver.hpp (the interior-point iteration); solve2w.hip compiles
// the same headers a second time for the two-wave small-batch kernel.
//
// Replaces Optimizer.get_constraint_terms + Optimizer.solve_OPT (reference optimizer.py:80-170,
// 219-613: the NLP that pyomo transcribes and ipopt solves) for S satellites at once.
//
// One wavefront (64 lanes) per satellite runs the whole primal-dual interior-point iteration:
//  * node-parallel phases (KKT residuals, barrier Hessian blocks, slack/multiplier updates, step
//    length, line-search trials) map one lane to one temporal node and work on field-major arrays
//    (coalesced), written as chunks "loads -> arithmetic -> stores" so that a chunk's loads are in flight together;
//  * the structure-exploiting linear solve is a Riccati recursion in the shifted state
//    y_k = x_k - Bp_{k-1} u_k (absorbs the first-order hold), 7x7 / 7x3 / 3x3 blocks staged in LDS
//    with one lane per matrix element and node k-1's operands prefetched while node k is worked on; the virtual
//    control nu_k is eliminated per stage by a 7x7 LDL^T (redundantly in the registers of every lane); the free
//    final time, the tangential-velocity equality and the five stiff rank-1 terminal barrier terms form a 7x7 border
//    solved by a symmetric quasi-definite LDL^T whose pivot signs also give the inertia; eight linear-term sweeps (1 right-hand side + 7 border columns) run side
//    by side in the 8 lane groups of the wave, the backward one fused into the factorisation loop; iterative
//    refinement on the reduced KKT system only once a terminal weight is stiff enough to cost digits.
// Per-satellite state lives in a global-memory workspace (ws_doubles: 199 KB at K = 30); no MFMA.
#include "solve_common.hpp"
#include "solve_launch.hpp"

namespace mpcx {
// Launch order: satellites sorted by the previous solve's iteration count, longest first (counting sort, one block).
// The order inside one count is whatever the atomics give; the solver's results do not depend on the order.
__global__ __launch_bounds__(1024) void launch_order_kernel(int S, const int32_t *prev_iters, int32_t *order)
{
    __shared__ int hist[256], base[256];
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < S; i += 1024) atomicAdd(&hist[min(max(prev_iters[i], 0), 255)], 1);
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int key = 255; key >= 0; --key) { base[key] = acc; acc += hist[key]; }
    }
    // identity first: every slot holds a valid satellite whatever the scatter below leaves unwritten
    for (int i = tid; i < S; i += 1024) order[i] = i;
    __syncthreads();
    // (prev_iters is written by the previous solve on the SAME stream, include/mpcx.h; the clamp keeps a scatter past
    // the table impossible even if a caller breaks that rule and the two passes see different counts)
    for (int i = tid; i < S; i += 1024) order[min(atomicAdd(&base[min(max(prev_iters[i], 0), 255)], 1), S - 1)] = i;
}

// The predictor the launch order sorts by: the largest iteration count of the satellite's last kPredHist solves.  A launch
// ends with its slowest workgroup, and with about two satellites per workgroup slot one long satellite that starts late
// costs its whole length: a satellite that needed many iterations in ANY of the last few solves is started early (an early
// start costs nothing if it turns out short); the last count alone forgets it as soon as the problem changes a little
// (successive MPC steps, the two SCP iterations of a step, the benchmark's rotating variants).
__global__ void update_prediction_kernel(int S, const int32_t *iters, int32_t *hist, int32_t *pred, int slot, int n_valid)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    hist[(size_t)slot * S + i] = iters[i];
    int m = 0;
    for (int h = 0; h < n_valid; ++h) m = max(m, hist[(size_t)h * S + i]);
    pred[i] = m;
}

// a satellite whose discretisation failed reports that code instead of the solver's
__global__ void merge_status_kernel(int S, const int32_t *dstat, int32_t *status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S && dstat[i] != 0) status[i] = dstat[i];
}
static inline void merge_status_kernel_launch(int S, const int32_t *dstat, int32_t *status, hipStream_t st)
{
    hipLaunchKernelGGL(merge_status_kernel, dim3((S + 255) / 256), dim3(256), 0, st, S, dstat, status);
}


// Persistent workgroups: the launch has as many single-wave workgroups as the device holds at once (or S, if fewer), each
// takes satellites off a counter until none is left, in launch order (longest first when the previous solve's iteration
// counts are known).  A workgroup keeps ONE workspace slot for all its satellites: the solver's working set is
// slots x 206 KB whatever the batch size (8192 satellites: 0.44 GB instead of 1.8 GB), and a slot's lines are rewritten
// by the next satellite while they are still cached instead of being written back as dead data.
__global__ __launch_bounds__(64, MPCX_SOLVE_WAVES) MPCX_NO_TAIL void solve_kernel(SolveArgs a)
{
    SatData &sd = g_sd;
    Scratch &w = g_w;
    __shared__ int next_item;
    const int lane = threadIdx.x;
#ifdef MPCX_LDS_PAD       // occupancy experiments only (profiles/r05/occupancy_sweep.txt): LDS the kernel does not use, to hold fewer workgroups per compute unit
    __shared__ double lds_pad[MPCX_LDS_PAD / 8];
    if (a.S < 0) { lds_pad[lane] = 1.0; __syncthreads(); a.kkt[0] = lds_pad[63 - lane]; }
#endif
    for (;;) {
        if (lane == 0) next_item = atomicAdd(a.counter, 1);
        __syncthreads();
        const int b = __builtin_amdgcn_readfirstlane(next_item);
        __syncthreads();
        if (b >= a.S) return;
        // (an entry outside [0, S) can only come from a caller that broke the same-stream rule of include/mpcx.h: never an
        //  out-of-bounds satellite)
        int sat = a.order ? a.order[b] : b;
        if ((unsigned)sat >= (unsigned)a.S) sat = b;
        solve_satellite<false>(a, sat, (int)blockIdx.x, sd, w, lane);
        __syncthreads();
    }
}

// Shared final time: one workgroup per satellite, all resident (cooperative launch), one lock-step iteration.
__global__ __launch_bounds__(64, MPCX_SOLVE_WAVES) MPCX_NO_TAIL void solve_shared_kernel(SolveArgs a)
{
    SatData &sd = g_sd;
    Scratch &w = g_w;
    const int lane = threadIdx.x;
    GridSync g{a.red, a.arrive, a.abort_flag, a.S, (int)blockIdx.x, 0, false};
    solve_satellite<true>(a, (int)blockIdx.x, (int)blockIdx.x, sd, w, lane, &g);
}


// Diagnostic export of what solve_kernel builds before its first iteration: the terminal inequality rows a_j . x_K <= b_j
// (build_terminal: Optimizer.get_constraint_terms, optimizer.py:80-170, as consumed by the rules :398-403, 406-446,
// 471-489, 351-352) and the relaxed scalar bounds.  Same device function, same lane, same LDS struct as in the solve.
__global__ __launch_bounds__(64) void constraint_terms_kernel(int S, int K, const double *xbar, const double *consts, const double *r_des,
                                                              SolveOpts o, double *aT, double *bT, double *scal)
{
    __shared__ SatData sd;
    const int sat = blockIdx.x, lane = threadIdx.x;
    if (sat >= S) return;
    if (lane == 0) {
        double xK[7], x0[3];
        for (int i = 0; i < 7; ++i) xK[i] = xbar[(size_t)sat * 7 * K + (size_t)i * K + K - 1];
        for (int i = 0; i < 3; ++i) x0[i] = xbar[(size_t)sat * 7 * K + (size_t)i * K];
        build_terminal(xK, consts[(size_t)sat * MPCX_NCONST + MPCX_C_MU], r_des[sat], o, sd);
        sd.infeas = structural_violation(x0, K, sd);
    }
    __syncthreads();
    if (lane < 56) aT[(size_t)sat * 56 + lane] = sd.aT[lane / 7][lane % 7];
    if (lane < 8) bT[(size_t)sat * 8 + lane] = sd.bT[lane];
    if (lane == 0) {
        double *q = scal + (size_t)sat * MPCX_NTERM_SCALARS;
        q[0] = sd.b_u; q[1] = sd.b_rmax; q[2] = sd.b_rmin; q[3] = sd.b_rfmax; q[4] = sd.b_tf[0]; q[5] = sd.b_tf[1];
        q[6] = sd.vt_des; q[7] = sd.infeas;
    }
}

// Node counts of the next SCP iteration: Simulator.run samples a rollout over tf at int(base_res * tf) points (simulator.py:38;
// control.py:227 passes tf_u), computed where tf_u lives.  The counts are clamped to the row length only in the sense that a
// count outside 3..K makes that satellite's next solve report MPCX_ST_BADK.
__global__ void node_count_kernel(int S, double base_res, const double *tf, int32_t *Kn)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S) Kn[i] = (int32_t)(base_res * tf[i]);
}
__global__ void fill_f64_kernel(int n, double v, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}
__global__ void scale_f64_kernel(int n, const double *a, double d, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] / d;
}
}  // namespace mpcx

// ---- launchers: what solve_api.hip sees of this translation unit (solve_launch.hpp) -------------------------------------
namespace mpcx_launch {
using namespace mpcx;

void solve(const SolveArgs &a, int slots, hipStream_t st) { hipLaunchKernelGGL(solve_kernel, dim3(slots), dim3(64), 0, st, a); }

hipError_t solve_shared(const SolveArgs &a, hipStream_t st)
{
    SolveArgs copy = a;
    void *kargs[] = {(void *)&copy};
    return hipLaunchCooperativeKernel((const void *)solve_shared_kernel, dim3(a.S), dim3(64), kargs, 0, st);
}

hipError_t solve_shared_blocks_per_cu(int *per_cu) { return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, solve_shared_kernel, 64, 0); }

void launch_order(int S, const int32_t *prev_iters, int32_t *order, hipStream_t st)
{
    hipLaunchKernelGGL(launch_order_kernel, dim3(1), dim3(1024), 0, st, S, prev_iters, order);
}

void update_prediction(int S, const int32_t *iters, int32_t *hist, int32_t *pred, int slot, int n_valid, hipStream_t st)
{
    hipLaunchKernelGGL(update_prediction_kernel, dim3((S + 255) / 256), dim3(256), 0, st, S, iters, hist, pred, slot, n_valid);
}

void merge_status(int S, const int32_t *dstat, int32_t *status, hipStream_t st) { merge_status_kernel_launch(S, dstat, status, st); }

void constraint_terms(int S, int K, const double *xbar, const double *consts, const double *r_des, const SolveOpts &o, double *aT,
                      double *bT, double *scal, hipStream_t st)
{
    hipLaunchKernelGGL(constraint_terms_kernel, dim3(S), dim3(64), 0, st, S, K, xbar, consts, r_des, o, aT, bT, scal);
}

void node_count(int S, double base_res, const double *tf, int32_t *Kn, hipStream_t st)
{
    hipLaunchKernelGGL(node_count_kernel, dim3((S + 255) / 256), dim3(256), 0, st, S, base_res, tf, Kn);
}

void fill_f64(int n, double v, double *out, hipStream_t st) { hipLaunchKernelGGL(fill_f64_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, v, out); }

void divide_f64(int n, const double *a, double d, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(scale_f64_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, a, d, out);
}
}  // namespace mpcx_launch
