// profiling helper: checks the DPP broadcasts inside 8-lane groups (two v_mov_b32_dpp per 32-bit half: quad_perm broadcast,
// then row_half_mirror into the other quad under a bank mask; and the two-instruction 64-bit row_newbcast form) against
// __shfl(v, q, 8) for every source lane q.
// build: hipcc --offload-arch=gfx950 -O2 -o dpp_bcast_check dpp_bcast_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int Q>
__device__ __forceinline__ int bcast8_i(int v)
{
    constexpr int qp = (Q & 3) * 0x55;                         // quad_perm: [q%4, q%4, q%4, q%4]
    const int t = __builtin_amdgcn_update_dpp(0, v, qp, 0xF, 0xF, false);
    // quads of the parity of q's quad hold lane q's value; mirror them into the others (lane i <-> 7 - i of the group)
    constexpr int other = (Q & 4) ? 0x5 : 0xA;                  // banks (quads) that do NOT hold q
    return __builtin_amdgcn_update_dpp(t, t, 0x141, 0xF, other, false);
}
template <int Q>
__device__ __forceinline__ double bcast8(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = bcast8_i<Q>((int)b), hi = bcast8_i<Q>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// The same broadcast as two 64-bit moves (gfx950: v_mov_b64_dpp exists for row_newbcast): lane q of every 16-lane row into
// the whole row, then lane 8+q over the row's upper half under a bank mask.
template <int Q>
__device__ __forceinline__ double bcast8_nb(double v)
{
    const long long b = __double_as_longlong(v);
    const long long t = __builtin_amdgcn_update_dpp((long long)0, b, 0x150 + Q, 0xF, 0xF, true);
    return __longlong_as_double(__builtin_amdgcn_update_dpp(t, b, 0x150 + 8 + Q, 0xF, 0xC, false));
}

__global__ void check(int *bad)
{
    const int lane = threadIdx.x;
    const double v = 1000.0 * lane + 0.25;
    int nb = 0;
#define CHK(Q) { const double a = bcast8<Q>(v), b = __shfl(v, Q, 8), c = bcast8_nb<Q>(v); if (a != b) ++nb; if (c != b) ++nb; }
    CHK(0) CHK(1) CHK(2) CHK(3) CHK(4) CHK(5) CHK(6) CHK(7)
    if (nb) atomicAdd(bad, nb);
}

int main()
{
    int *d, h = -1;
    hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("dpp bcast8 mismatches: %d\n", h);
    return h != 0;
}
