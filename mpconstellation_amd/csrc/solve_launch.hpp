// solve_launch.hpp -- what the C entry points (solve_api.hip) see of the kernel translation units solve.hip / solve2w.hip:
// plain host functions that launch the kernels.  A change of the entry points leaves the kernel files untouched (and the
// source hashes the committed profiles record for them valid).
#pragma once
#include <hip/hip_runtime.h>
#include "solve_layout.hpp"

namespace mpcx_launch {
void solve(const mpcx::SolveArgs &a, int slots, hipStream_t st);                  // solve_kernel, `slots` persistent workgroups
hipError_t solve_shared(const mpcx::SolveArgs &a, hipStream_t st);                // solve_shared_kernel, cooperative, a.S workgroups
hipError_t solve_shared_blocks_per_cu(int *per_cu);
void launch_order(int S, const int32_t *prev_iters, int32_t *order, hipStream_t st);
void update_prediction(int S, const int32_t *iters, int32_t *hist, int32_t *pred, int slot, int n_valid, hipStream_t st);
void merge_status(int S, const int32_t *dstat, int32_t *status, hipStream_t st);
void constraint_terms(int S, int K, const double *xbar, const double *consts, const double *r_des, const mpcx::SolveOpts &o, double *aT,
                      double *bT, double *scal, hipStream_t st);
void node_count(int S, double base_res, const double *tf, int32_t *Kn, hipStream_t st);
void fill_f64(int n, double v, double *out, hipStream_t st);
void divide_f64(int n, const double *a, double d, double *out, hipStream_t st);
}  // namespace mpcx_launch

// solve2w.hip: the two-wave small-batch kernel (SolveArgs handed over as bytes: the struct is compiled into both units)
int mpcx2w_launch(const void *args, size_t args_bytes, int blocks, hipStream_t stream);
// solve_lds.hip: the same kernel with the satellite's working set in LDS (at most one satellite per compute unit at a time);
// returns 1 without launching when the working set of the call's node count does not fit
int mpcxl_launch(const void *args, size_t args_bytes, int blocks, hipStream_t stream);
// solve_tp.hip: the time-parallel kernel (a workgroup of two waves per segment of the horizon, four per satellite, cooperative
// launch; MPCX_SOLVE_TIME_PARALLEL); blocks = satellites
int mpcxtp_launch(const void *args, size_t args_bytes, int blocks, hipStream_t stream);
int mpcxtp_blocks_per_cu();
