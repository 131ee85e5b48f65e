// solve_common.hpp -- the includes and the two-build switch shared by solve.hip and solve2w.hip
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include "mpcx_device.hpp"
#include "solve_layout.hpp"

// Compiled twice: by solve.hip as it stands (namespace mpcx, solve_kernel: one wave per satellite, every barrier of a phase
// function is that wave's) and by solve2w.hip with MPCX_TWO_WAVE defined (namespace mpcx2w: the small-batch kernel whose
// workgroups have a second wave that shares the factorisation, riccati_factor2).  In the two-wave build the phase functions
// still run on the first wave alone, so their "workgroup barriers" must not be hardware barriers (the second wave never
// executes them): WG_SYNC() is then the memory fence only, and the real two-wave barriers are spelled WG_BARRIER().
#ifdef MPCX_TWO_WAVE
#ifdef MPCX_WS_LDS
#define MPCX_NS mpcxl          // (solve_lds.hip: the two-wave kernel with its working set in LDS)
#elif defined(MPCX_TP)
#define MPCX_NS mpcxtp         // (solve_tp.hip: the time-parallel kernel, a pair of waves per segment of the horizon)
#else
#define MPCX_NS mpcx2w
#endif
#define WG_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define MPCX_NS mpcx
// (measured in round 5, profiles/r05/streaming_phases_ab.txt: a fence + wave barrier here instead of __syncthreads() -- which for
//  this one-wave workgroup is "s_waitcnt vmcnt(0) lgkmcnt(0)", a drain of every outstanding store at each phase boundary --
//  gives the same bits and the same time, 5.32 against 5.32 ms at S4096_K30: the second wave of the SIMD hides the drains)
#define WG_SYNC() __syncthreads()
#endif
#define WG_BARRIER() __syncthreads()
// On the solve kernels: no `tail` markers on their calls.  The phase functions are local, non-recursive and never address-taken,
// so the compiler gives them no callee-saved registers (the caller keeps what it needs: interprocedural register allocation) --
// unless a call to them is MARKED as a tail-call candidate, which happens as soon as none of its arguments points into the
// caller's stack.  With the satellite's view in LDS (g_s) that was the case for newton_blocks, riccati_factor and
// combine_channels, each of which then saved and restored 112 registers to scratch per call: solve_kernel 5.34 -> 5.64 ms.
#define MPCX_NO_TAIL __attribute__((disable_tail_calls))

#include "solve_phases.hpp"
#include "solve_riccati.hpp"
#ifdef MPCX_TP
#include "solve_tp.hpp"
#endif
#include "solve_driver.hpp"
