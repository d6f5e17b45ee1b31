// How much host CPU does each kind of wait burn behind a ~40 ms kernel?  (hipcc --offload-arch=gfx950 wait_cpu.hip -o wait_cpu.bin)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
#include <chrono>
__global__ void spin(long long cycles, int *out) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} if (out) *out = 1; }
static double cpu_ms() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int *d; hipMalloc(&d, 4); int *pin; hipHostMalloc(&pin, 4);
    hipEvent_t e_spin, e_block; hipEventCreateWithFlags(&e_spin, hipEventDisableTiming); hipEventCreateWithFlags(&e_block, hipEventDisableTiming | hipEventBlockingSync);
    const long long cyc = 4000000;   // 100 MHz wall clock: 40 ms
    auto run = [&](const char *what, int mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(spin, 1, 1, 0, s, cyc, d);
            double c0 = cpu_ms(), w0 = wall_ms(); int host = 0;
            switch (mode) {
            case 0: hipStreamSynchronize(s); break;
            case 1: hipEventRecord(e_spin, s); hipEventSynchronize(e_spin); break;
            case 2: hipEventRecord(e_block, s); hipEventSynchronize(e_block); break;
            case 3: hipMemcpyAsync(&host, d, 4, hipMemcpyDeviceToHost, s); break;                      // pageable destination
            case 4: hipMemcpyAsync(pin, d, 4, hipMemcpyDeviceToHost, s); hipEventRecord(e_block, s); hipEventSynchronize(e_block); break;
            case 5: hipMemcpyAsync(d, &host, 4, hipMemcpyHostToDevice, s); hipEventRecord(e_block, s); hipEventSynchronize(e_block); break;   // pageable source
            case 6: hipMemsetAsync(d, 0, 4, s); hipEventRecord(e_block, s); hipEventSynchronize(e_block); break;
            }
            if (rep) printf("%-44s wall %6.1f ms  cpu %6.1f ms\n", what, wall_ms() - w0, cpu_ms() - c0);
            hipStreamSynchronize(s);
        }
    };
    run("hipStreamSynchronize", 0);
    run("hipEventSynchronize (default event)", 1);
    run("hipEventSynchronize (hipEventBlockingSync)", 2);
    run("hipMemcpyAsync D2H to a stack variable", 3);
    run("D2H to pinned + blocking event", 4);
    run("H2D from a stack variable + blocking event", 5);
    run("memset + blocking event", 6);
    hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    printf("-- after hipSetDeviceFlags(hipDeviceScheduleBlockingSync)\n");
    run("hipStreamSynchronize", 0);
    run("hipEventSynchronize (default event)", 1);
    run("hipMemcpyAsync D2H to a stack variable", 3);
    return 0;
}
