// Does one long-running single-wave kernel keep OTHER streams' kernels from running?  (MI355X, ROCm 7.2)
//   stream A: a kernel that spins for ~1.5 s in one wave; thread B: short kernels + hipStreamSynchronize on its own stream, timed.
// Variants: A on a normal / high-priority stream; NS extra idle streams created first (ROCclr maps streams onto GPU_MAX_HW_QUEUES queues).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/lkb tools/ubench/long_kernel_blocks.hip -lpthread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(long long cycles, int *out) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) { } if (out) *out = 1; }
__global__ void tiny(int *p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const int prio_mode = argc > 1 ? atoi(argv[1]) : 0;      // 0 normal, 1 high, 2 low
    const int extra = argc > 2 ? atoi(argv[2]) : 0;
    int plo = 0, phi = 0; CK(hipDeviceGetStreamPriorityRange(&plo, &phi));
    std::vector<hipStream_t> idle((size_t)extra);
    for (auto &s : idle) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *d0; CK(hipMalloc(&d0, 64)); CK(hipMemset(d0, 0, 64));
    for (auto &s : idle) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d0); CK(hipStreamSynchronize(s)); }
    hipStream_t a, b;
    if (prio_mode == 0) CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    else CK(hipStreamCreateWithPriority(&a, hipStreamNonBlocking, prio_mode == 1 ? phi : plo));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    int *d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d); CK(hipStreamSynchronize(b));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 1000LL, d + 1); CK(hipStreamSynchronize(a));
    const double t0 = now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 150000000LL, d + 1);       // 1.5 s at 100 MHz wall clock
    double worst = 0; int n = 0;
    std::vector<double> worst_s(idle.size(), 0.0);
    std::thread tb([&] {
        while (now() - t0 < 1.0) {
            const double t1 = now();
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d); CK(hipStreamSynchronize(b));
            const double dt = now() - t1; worst = dt > worst ? dt : worst; ++n;
            // every other stream in turn (they have all run something before: each owns a hardware queue slot by now)
            for (size_t k = 0; k < idle.size() && now() - t0 < 1.0; ++k) {
                const double t2 = now();
                hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, idle[k], d); CK(hipStreamSynchronize(idle[k]));
                const double d2 = now() - t2; worst_s[k] = d2 > worst_s[k] ? d2 : worst_s[k];
            }
        }
    });
    tb.join();
    CK(hipStreamSynchronize(a));
    int blocked = 0; for (double w : worst_s) blocked += w > 0.1;
    if (!idle.empty()) printf("  of the %zu other (active) streams, %d waited more than 100 ms for a launch + sync\n", idle.size(), blocked);
    printf("A priority mode %d (range %d..%d), %d idle streams: long kernel %.2f s; B ran %d launch+sync pairs beside it, worst %.3f ms\n",
           prio_mode, plo, phi, extra, now() - t0, n, worst * 1e3);
    return 0;
}
