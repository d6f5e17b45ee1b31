// Achievable vector FP64 FMA rate on gfx950 as a function of independent chains per lane and waves per SIMD (hipEvent timing).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void __launch_bounds__(256) k_dfma(double *out, double a, double b, int n)
{
    double x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = out[(threadIdx.x + c) & 63];
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = fma(x[c], b, a);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c];
    if (s == 12345.678) out[threadIdx.x & 63] = s;
}
template <int CH> static void run(int blocks, double *d)
{
    const int n = 4096;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_dfma<CH>, blocks, 256, 0, 0, d, 1e-9, 0.999999, n);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_dfma<CH>, blocks, 256, 0, 0, d, 1e-9, 0.999999, n);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * blocks * 256.0 * n * 8.0 * CH;
    printf("chains/lane %d, %5d blocks x 256 (%4.1f waves/SIMD): %.3f ms -> %.1f TFLOP/s\n", CH, blocks, blocks * 4.0 / 1024.0, ms, flops / ms / 1e9);
}
int main()
{
    double *d; (void)hipMalloc(&d, 64 * 8); (void)hipMemset(d, 0, 64 * 8);
    for (int blocks : {256, 512, 1024, 2048}) { run<1>(blocks, d); run<4>(blocks, d); run<8>(blocks, d); }
    return 0;
}
