// solve_lds.hip -- the LDS-RESIDENT build of the solver: the two-wave kernel of solve2w.hip compiled a third time with the
// satellite's working set -- iterate, candidate iterate, direction, Newton records, factor records, channel vectors: 134 KB
// at K = 30 -- carved from the workgroup's LDS instead of the global workspace (MPCX_WS_LDS: solve_phases.hpp, wf64;
// solve_driver.hpp, sat_view).  One workgroup then fills a compute unit (157 of its 160 KB), so the build serves batches of
// at most one satellite per CU (256 on an MI355X) and horizons whose working set fits; everything else runs solve2w.hip /
// solve.hip.  Same arithmetic, same bits.  What stays in global memory: the stage records and their field-major copy, the
// Newton scalars of the refinement, the channels' forward trajectories, inputs and results.
#define MPCX_TWO_WAVE 1
#define MPCX_WS_LDS 1
#include <cstring>
#include <mutex>
#include "solve_common.hpp"
#include "solve_launch.hpp"
#define MPCX_KERNEL2W_NAME solve_kernel_lds
#include "solve_kernel2w.hpp"

// 0: launched; 1: the working set of K nodes does not fit the LDS this kernel may still ask for (the caller takes the
// global-workspace kernel); -1: error
int mpcxl_launch(const void *args, size_t args_bytes, int blocks, hipStream_t stream)
{
    MPCX_NS::SolveArgs a;
    if (args_bytes != sizeof a) return -1;
    memcpy(&a, args, sizeof a);
    // dynamic LDS the kernel may have beside its static part: asked for once PER DEVICE (the attribute belongs to the device's
    // copy of the function; contexts of several devices call from their own threads: sharding.py)
    constexpr int kMaxDev = 64;
    static int limits[kMaxDev];
    static bool asked[kMaxDev];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return -1;
    if (dev >= kMaxDev) return 1;                          // (beyond the table: the global-workspace kernel, not an error)
    int lds_limit;
    {
        std::lock_guard<std::mutex> g(mu);
        if (!asked[dev]) {
            // Any failure here means "this device does not give the kernel its LDS": remembered as a limit of 0, the caller takes
            // the global-workspace kernel (return 1) -- never a failed solve.  The runtime reports 64 KB per block for gfx950 where
            // the hardware and the compiler allow 160 KB (SURVEY: 160 KB compiles, 161 KB does not), so 160 KB is TRIED first and
            // the reported value second; what hipFuncSetAttribute accepts is the limit.
            int room = 0;
            hipFuncAttributes at;
            int max_lds = 0;
            if (hipFuncGetAttributes(&at, (const void *)MPCX_NS::solve_kernel_lds) == hipSuccess &&
                hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess) {
                const int tries[2] = {160 * 1024 - (int)at.sharedSizeBytes, max_lds - (int)at.sharedSizeBytes};
                for (int t = 0; t < 2 && room == 0; ++t)
                    if (tries[t] > 0 && (t == 0 || tries[t] < tries[0]) &&
                        hipFuncSetAttribute((const void *)MPCX_NS::solve_kernel_lds, hipFuncAttributeMaxDynamicSharedMemorySize, tries[t]) == hipSuccess)
                        room = tries[t];
            }
            (void)hipGetLastError();                       // (a refused attribute must not surface as the next launch's error)
            limits[dev] = room;
            asked[dev] = true;
        }
        lds_limit = limits[dev];
    }
    const size_t need = mpcx::lds_ws_doubles(a.K) * sizeof(double);
    if (need > (size_t)lds_limit) return 1;
    hipLaunchKernelGGL(MPCX_NS::solve_kernel_lds, dim3(blocks), dim3(128), need, stream, a);
    return 0;
}
