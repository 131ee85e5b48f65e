// solve_kernel2w.hpp -- the kernel of the two builds with two waves per satellite (solve2w.hip: workspace in global memory;
// solve_lds.hip: most of it in LDS), named by MPCX_KERNEL2W_NAME.
#pragma once

namespace MPCX_NS {

// Two waves per satellite.  The first runs solve_satellite exactly as the one-wave kernel's wave does; the second waits in a
// command loop and joins it for every factorisation (riccati_factor2).  Same work queue, same slot workspaces.
__global__ __launch_bounds__(128, MPCX_SOLVE_WAVES) MPCX_NO_TAIL void MPCX_KERNEL2W_NAME(SolveArgs a)
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
            const Sat &s = g_s;                                    // (written by the first wave before its first command; a satellite
            for (;;) {                                             //  refused as MPCX_ST_BADK sends CMD_EXIT at once)
                WG_BARRIER();
                if (w.cmd == CMD_EXIT) break;
                (void)riccati_factor2(s, sd, w, lane, 1, w.cmd_arg != 0);
            }
        }
        WG_BARRIER();
    }
}

}  // namespace MPCX_NS

