// solve2w.hip -- the small-batch build of the solver: the headers of solve.hip compiled a second time, in their own namespace,
// with two waves per satellite (solve_common.hpp, riccati_factor2 in solve_riccati.hpp).  Up to 1024 satellites -- one per
// SIMD of the chip or fewer -- a satellite's time is the dependent chain of one wave; the second wave takes half of the
// factorisation's chain.  Results are bit for bit the one-wave kernel's (the library is built with -ffp-contract=on).
#define MPCX_TWO_WAVE 1
#include <cstring>
#include "solve_common.hpp"
#include "solve_launch.hpp"
#define MPCX_KERNEL2W_NAME solve_kernel2w
#include "solve_kernel2w.hpp"

// (SolveArgs of the two builds are the same struct compiled twice: handed over as bytes)
int mpcx2w_launch(const void *args, size_t args_bytes, int blocks, hipStream_t stream)
{
    MPCX_NS::SolveArgs a;
    if (args_bytes != sizeof a) return -1;
    memcpy(&a, args, sizeof a);
    hipLaunchKernelGGL(MPCX_NS::solve_kernel2w, dim3(blocks), dim3(128), 0, stream, a);
    return 0;
}
