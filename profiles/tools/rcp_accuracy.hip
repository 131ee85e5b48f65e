// diagnostic: relative error of v_rcp_f64 and of 1, 2, 3 Newton steps on it, over log-uniform random positive doubles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double *x, double *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    out[i * 4] = r;
    for (int s = 0; s < 3; ++s) { double e = fma(-d, r, 1.0); r = fma(r, e, r); out[i * 4 + 1 + s] = r; }
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> h(n), o(4 * n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-40.0, 40.0);
    for (auto &v : h) v = std::pow(10.0, u(g));
    double *dx, *d_o;
    hipMalloc(&dx, n * 8); hipMalloc(&d_o, 4 * n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d_o, n);
    hipMemcpy(o.data(), d_o, 4 * n * 8, hipMemcpyDeviceToHost);
    double worst[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int s = 0; s < 4; ++s) {
            long double ex = 1.0L / (long double)h[i];
            double rel = (double)fabsl(((long double)o[i * 4 + s] - ex) / ex);
            if (rel > worst[s]) worst[s] = rel;
        }
    printf("max relative error: seed %.3e, 1 step %.3e, 2 steps %.3e, 3 steps %.3e (eps = %.3e)\n", worst[0], worst[1], worst[2], worst[3], ldexp(1.0, -53));
    return 0;
}
