// solve_tp.hip -- the TIME-PARALLEL build of the solver (MPCX_SOLVE_TIME_PARALLEL): the headers of solve.hip compiled a fourth
// time with the linear solve of an interior-point iteration cut into up to four segments of the horizon, a pair of waves each
// (solve_tp.hpp).  For small batches -- one workgroup of eight waves per satellite, at most one per
// compute unit at a time -- whose time is the dependent chain over the nodes: the chain is a segment long instead of the
// horizon.  One workgroup of two waves per segment, each on its own compute unit.  Same directions to ~1e-12, same iteration counts, not the other kernels' bits.
#define MPCX_TWO_WAVE 1
#define MPCX_TP 1
#include <cstring>
#include "solve_common.hpp"
#include "solve_launch.hpp"

namespace MPCX_NS {

// One workgroup of two waves per segment, TP_MAXSEG consecutive workgroups per satellite (a cooperative launch: they wait for
// each other).  The LAST segment's workgroup is the satellite's first: its first wave runs solve_satellite as in every build
// and sends the others their commands through the mailbox, its second wave follows through LDS as in the two-wave kernel.
__global__ __launch_bounds__(128, MPCX_SOLVE_WAVES) MPCX_NO_TAIL void solve_kernel_tp(SolveArgs a)
{
    SatData &sd = g_sd;
    TpData &tp = g_tp;
    Scratch &w = g_w;
    // (workgroups go round-robin to the 8 XCDs: the four of a satellite are given block indices of one residue mod 8, so that they
    //  share an L2 and their mailbox / exchange traffic stays in it)
    const int xcd = (int)blockIdx.x & 7, q8 = (int)blockIdx.x >> 3;
    const int pair = q8 % TP_MAXSEG, sat = (q8 / TP_MAXSEG) * 8 + xcd;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sat >= a.S) return;
    const int Kmax = a.K;
    int K = a.Ks ? a.Ks[sat] : Kmax;
    const bool badk = (K < 3 || K > Kmax);
    if (badk) K = Kmax;                                    // (the first wave reports MPCX_ST_BADK; the others only see the exit command)
    const int nseg = tp_segments(K);
    if (pair >= nseg) return;
    const int j = nseg - 1 - pair;                          // this workgroup's segment
    if (threadIdx.x == 0) { g_s = sat_view(a, sat, sat, K, Kmax); tp_geometry(tp, K); tp.seq = 0; tp.dead = 0; tp.light = 0; }
    WG_BARRIER();
    const Sat &s = g_s;
#ifdef MPCX_TP_DEBUG
    if (threadIdx.x == 0) printf("[b%d] sat %d pair %d seg %d of %d K %d mail %d %d %d %d\n", (int)blockIdx.x, sat, pair, j, nseg, K, s.mail[0], s.mail[1], s.mail[2], s.mail[3]);
#endif
    if (pair == 0) {
        if (wave == 0) {
            solve_satellite<false>(a, sat, sat, sd, w, lane);
            TP_DBG("[drv b%d] solve_satellite returned, dead %d\n", (int)blockIdx.x, tp.dead);
            if (tp.dead) {            // a wait ran out: not a numerical failure -- its own status; X, U, NU are the last iterate
                if (lane == 0) { a.status[sat] = MPCX_ST_TIMEOUT; a.kkt[sat] = -1.0; }
#ifdef MPCX_TP_DEBUG                 // (diagnostic build only: the mailbox as it stands in the first entries of the NU block)
                if (lane < TP_MAIL_N * 2) a.NU[(size_t)sat * 7 * Kmax + lane] = (double)s.mail[lane];
#endif
            }
            tp_post(s, tp, CMD_EXIT, 0, lane);
            if (lane == 0) w.cmd = CMD_EXIT;
            WG_BARRIER();
        } else {
            for (;;) {
                WG_BARRIER();
                const int cmd = __builtin_amdgcn_readfirstlane(w.cmd), arg = __builtin_amdgcn_readfirstlane(w.cmd_arg);
                TP_DBG("[drv b%d wave 1] command %d arg %d\n", (int)blockIdx.x, cmd, arg);
                if (cmd == CMD_EXIT) break;
                if (cmd == CMD_COMBINE) { tp_combine(s, sd, tp, (double *)&w, lane, arg != 0, 1); continue; }
                bool ok = true;
                if (cmd == CMD_FACTOR) ok = riccati_factor2(s, sd, w, lane, 1, arg != 0, tp_range(tp, j));
                if (ok) { tp_sweeps_pair(s, sd, w, tp, j, 1, lane, cmd == CMD_FACTOR); WG_BARRIER(); }
            }
        }
    } else
        tp_worker(s, sd, tp, j, wave, lane, a.o.tp_selftest != 0);
}

}  // namespace MPCX_NS

// (SolveArgs of the builds are the same struct compiled several times: handed over as bytes)
// workgroups of the kernel that can be resident per compute unit (for the host's limit on the batch size); < 0 on error
int mpcxtp_blocks_per_cu()
{
    int n = 0;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, MPCX_NS::solve_kernel_tp, 128, 0) == hipSuccess ? n : -1;
}

int mpcxtp_launch(const void *args, size_t args_bytes, int blocks, hipStream_t stream)
{
    MPCX_NS::SolveArgs a;
    if (args_bytes != sizeof a) return -1;
    memcpy(&a, args, sizeof a);
    // (cooperative: the workgroups of a satellite wait for each other, all of them must be resident)
    void *kargs[] = {(void *)&a};
    const int grid = ((blocks + 7) / 8) * 8 * mpcx::TP_MAXSEG;
    return hipLaunchCooperativeKernel((const void *)MPCX_NS::solve_kernel_tp, dim3(grid), dim3(128), kargs, 0, stream) == hipSuccess ? 0 : -1;
}
