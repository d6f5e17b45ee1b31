// global_load_lds_dwordx4 on gfx950: where do the 64 x 16 bytes of one instruction land in LDS?  (expected: base + 16 * lane)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, float *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) reinterpret_cast<float *>(sm)[i] = -1.f;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + 4 * (63 - lane)),
                                     (__attribute__((address_space(3))) void *)(sm + 1024), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = reinterpret_cast<float *>(sm)[i];
}
int main()
{
    float h[256], *d, *o, r[1024];
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, sizeof(r));
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 4096, 0, d, o);
    (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    // lane l loaded floats 4*(63-l) .. +3; report the LDS word offsets where lane 0's and lane 1's data landed
    for (int i = 0; i < 1024; ++i) if (r[i] == 252.f || r[i] == 248.f || r[i] == 0.f) printf("word %d = %g\n", i, r[i]);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) if (r[256 + 4 * l + e] != (float)(4 * (63 - l) + e)) bad++;
    printf("layout base+16*lane: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return 0;
}
