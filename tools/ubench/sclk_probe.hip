// Shader clock a kernel actually runs at, as a function of how many waves it launches: clock64() (shader cycles) against
// wall_clock64() (constant 100 MHz) around a fixed dependent f64 chain.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_probe(double *out, long long *res, int n)
{
    double x = out[threadIdx.x & 63];
    const long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { const double d = 0.5 - x; x += d * (0.5 > x ? 0.25 : 0.125); }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x & 63] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { res[0] = c1 - c0; res[1] = w1 - w0; }
}
int main()
{
    double *d; long long *c; (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&c, 16); (void)hipMemset(d, 0, 64 * 8);
    const int n = 1 << 16;
    for (int blocks : {1, 64, 250, 256, 1024, 4096, 16384}) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_probe, blocks, 64, 0, 0, d, c, n); (void)hipDeviceSynchronize(); }
        long long r[2]; (void)hipMemcpy(r, c, 16, hipMemcpyDeviceToHost);
        printf("%6d waves: %.1f shader cycles per step, %.1f ns per step -> %.0f MHz\n", blocks, (double)r[0] / (16.0 * n),
               (double)r[1] * 10.0 / (16.0 * n), (double)r[0] / ((double)r[1] * 10.0) * 1000.0);
    }
    return 0;
}
