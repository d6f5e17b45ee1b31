// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against known byte counts, per access width (MI355X_MICROARCH.md, HBM:
// "FETCH_SIZE reports 1/2 of a 16 B/lane coalesced streaming read; other widths and WRITE_SIZE are uncalibrated").
// hipcc -O2 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib.bin;  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib.bin
// Every kernel moves exactly 2 GiB (larger than the 256 MiB Infinity Cache), coalesced: lane i of a wave touches element i of the row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <typename T> __global__ void k_read(const T *__restrict__ p, size_t n, T *sink)
{
    T acc{};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        const unsigned char *b = reinterpret_cast<const unsigned char *>(&v);
        unsigned char *a = reinterpret_cast<unsigned char *>(&acc);
        for (unsigned k = 0; k < sizeof(T); ++k) a[k] ^= b[k];
    }
    if (reinterpret_cast<unsigned char *>(&acc)[0] == 0x5a && threadIdx.x == 999) *sink = acc;     // (never true: keeps the loads)
}
template <typename T> __global__ void k_write(T *__restrict__ p, size_t n, T v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// a lane-serial reader: every lane walks its own contiguous chunk (how the IIR / follower kernels would read without LDS staging)
__global__ void k_read_lane_chunks(const float *__restrict__ p, size_t n, size_t chunk, float *sink)
{
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (size_t i = c * chunk; i < (c + 1) * chunk && i < n; ++i) acc += p[i];
    if (acc == 12345.678f) *sink = acc;
}
int main()
{
    const size_t bytes = (size_t)2 << 30;
    void *buf; hipMalloc(&buf, bytes + 64); hipMemset(buf, 1, bytes); void *sink; hipMalloc(&sink, 64);
    hipDeviceSynchronize();
    const int grid = 256 * 32, block = 256;
    k_read<float><<<grid, block>>>((const float *)buf, bytes / 4, (float *)sink);
    k_read<float2><<<grid, block>>>((const float2 *)buf, bytes / 8, (float2 *)sink);
    k_read<float4><<<grid, block>>>((const float4 *)buf, bytes / 16, (float4 *)sink);
    k_read<short><<<grid, block>>>((const short *)buf, bytes / 2, (short *)sink);
    k_read_lane_chunks<<<bytes / 4 / 4096 / 64, 64>>>((const float *)buf, bytes / 4, 4096, (float *)sink);
    k_write<float><<<grid, block>>>((float *)buf, bytes / 4, 1.f);
    k_write<float2><<<grid, block>>>((float2 *)buf, bytes / 8, make_float2(1.f, 2.f));
    k_write<float4><<<grid, block>>>((float4 *)buf, bytes / 16, make_float4(1.f, 2.f, 3.f, 4.f));
    k_write<short><<<grid, block>>>((short *)buf, bytes / 2, (short)3);
    hipDeviceSynchronize();
    printf("each kernel moved %zu bytes\n", bytes);
    return 0;
}
