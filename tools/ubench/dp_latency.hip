// Dependent-issue latency of f64 / f32 VALU ops on gfx950: one wave, a chain of N dependent operations, s_memtime around it.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k_chain(double *out, long long *cyc, double a, double b, int n)
{
    double x = out[threadIdx.x];
    float xf = (float)x, af = (float)a, bf = (float)b;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) x = __dadd_rn(x, a);
            else if (OP == 1) x = __dmul_rn(x, b);
            else if (OP == 2) x = fma(x, b, a);
            else if (OP == 3) xf = __fadd_rn(xf, af);
            else if (OP == 4) xf = fmaf(xf, bf, af);
            else if (OP == 5) { const double d = a - x; x += fmax(d * 0.25, d * 0.125); }          // follower step, branch-free
            else if (OP == 6) { const double d = a - x; x += d * (a > x ? 0.25 : 0.125); }           // follower step, select
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x + xf;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    double *d; long long *c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 8); hipMemset(d, 0, 64 * 8);
    const char *names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_add_f32", "v_fma_f32", "follower step (max form)", "follower step (select form)"};
    const int n = 4096;
    for (int op = 0; op < 7; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (op) {
            case 0: hipLaunchKernelGGL(k_chain<0>, 1, 64, 0, 0, d, c, 1e-9, 1.0000001, n); break;
            case 1: hipLaunchKernelGGL(k_chain<1>, 1, 64, 0, 0, d, c, 1e-9, 1.0000001, n); break;
            case 2: hipLaunchKernelGGL(k_chain<2>, 1, 64, 0, 0, d, c, 1e-9, 1.0000001, n); break;
            case 3: hipLaunchKernelGGL(k_chain<3>, 1, 64, 0, 0, d, c, 1e-9, 1.0000001, n); break;
            case 4: hipLaunchKernelGGL(k_chain<4>, 1, 64, 0, 0, d, c, 1e-9, 1.0000001, n); break;
            case 5: hipLaunchKernelGGL(k_chain<5>, 1, 64, 0, 0, d, c, 0.5, 1.0000001, n); break;
            default: hipLaunchKernelGGL(k_chain<6>, 1, 64, 0, 0, d, c, 0.5, 1.0000001, n); break;
            }
            hipDeviceSynchronize();
        }
        long long cy = 0; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        printf("%-28s %6.1f cycles per dependent step (s_memtime ticks / %d)\n", names[op], (double)cy / (16.0 * n), 16 * n);
    }
    return 0;
}
