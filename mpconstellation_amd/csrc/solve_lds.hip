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
    static int lds_limit = -1;                // dynamic LDS the kernel may have beside its static part (per device the same)
    if (lds_limit < 0) {
        hipFuncAttributes at;
        if (hipFuncGetAttributes(&at, (const void *)MPCX_NS::solve_kernel_lds) != hipSuccess) return -1;
        int dev = 0, max_lds = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return -1;
        if (max_lds < 160 * 1024) max_lds = 160 * 1024;                 // (gfx950: 160 KB per workgroup)
        const int room = max_lds - (int)at.sharedSizeBytes;
        if (room > 0 && hipFuncSetAttribute((const void *)MPCX_NS::solve_kernel_lds, hipFuncAttributeMaxDynamicSharedMemorySize, room) != hipSuccess) return -1;
        lds_limit = room > 0 ? room : 0;
    }
    const size_t need = mpcx::lds_ws_doubles(a.K) * sizeof(double);
    if (need > (size_t)lds_limit) return 1;
    hipLaunchKernelGGL(MPCX_NS::solve_kernel_lds, dim3(blocks), dim3(128), need, stream, a);
    return 0;
}
