// solve2w.hip -- the small-batch build of the solver: the headers of solve.hip compiled a second time, in their own namespace,
// with two waves per satellite (solve_common.hpp, riccati_factor2 in solve_riccati.hpp).  Up to 1024 satellites -- one per
// SIMD of the chip or fewer -- a satellite's time is the dependent chain of one wave; the second wave takes half of the
// factorisation's chain.  Results are bit for bit the one-wave kernel's (the library is built with -ffp-contract=on).
#define MPCX_TWO_WAVE 1
#include <cstring>
#include "solve_common.hpp"
#include "solve_launch.hpp"

namespace MPCX_NS {

// Two waves per satellite.  The first runs solve_satellite exactly as the one-wave kernel's wave does; the second waits in a
// command loop and joins it for every factorisation (riccati_factor2).  Same work queue, same slot workspaces.
__global__ __launch_bounds__(128, MPCX_SOLVE_WAVES) void solve_kernel2w(SolveArgs a)
{
    SatData &sd = g_sd;
    Scratch &w = g_w;
    __shared__ int next_item;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (;;) {
        if (threadIdx.x == 0) next_item = atomicAdd(a.counter, 1);
        WG_BARRIER();
        const int b = __builtin_amdgcn_readfirstlane(next_item);
        WG_BARRIER();
        if (b >= a.S) return;
        int sat = a.order ? a.order[b] : b;
        if ((unsigned)sat >= (unsigned)a.S) sat = b;
        if (wave == 0) {
            solve_satellite<false>(a, sat, (int)blockIdx.x, sd, w, lane);
            if (lane == 0) w.cmd = CMD_EXIT;
            WG_BARRIER();
        } else {
            const int Kmax = a.K;
            int K = a.Ks ? a.Ks[sat] : Kmax;
            if (K < 3 || K > Kmax) K = Kmax;                       // (the first wave reports MPCX_ST_BADK and sends CMD_EXIT at once)
            const Sat s = sat_view(a, sat, (int)blockIdx.x, K, Kmax);
            for (;;) {
                WG_BARRIER();
                if (w.cmd == CMD_EXIT) break;
                (void)riccati_factor2(s, sd, w, lane, 1, w.cmd_arg != 0);
            }
        }
        WG_BARRIER();
    }
}

}  // namespace MPCX_NS

// (SolveArgs of the two builds are the same struct compiled twice: handed over as bytes)
int mpcx2w_launch(const void *args, size_t args_bytes, int blocks, hipStream_t stream)
{
    MPCX_NS::SolveArgs a;
    if (args_bytes != sizeof a) return -1;
    memcpy(&a, args, sizeof a);
    hipLaunchKernelGGL(MPCX_NS::solve_kernel2w, dim3(blocks), dim3(128), 0, stream, a);
    return 0;
}
