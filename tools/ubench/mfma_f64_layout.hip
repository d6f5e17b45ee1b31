// Which lane holds which element of v_mfma_f64_16x16x4_f64's operands (gfx950)?  A[i][k] = 1000 + 10*i + k is encoded so that the
// result identifies the layout: with B = unit vectors the output reveals (i, k) per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(double *out, int mode)
{
    const int l = threadIdx.x;
    // hypothesis: A lane l = (i = l % 16, k = l / 16); B lane l = (j = l % 16, k = l / 16); D lane l reg r = (i = 4 * (l / 16) + r, j = l % 16)
    const int i = l % 16, kk = l / 16;
    double a = 100.0 * i + kk + 1;            // A[i][k]
    double b = (double)((l % 16) == mode && kk == 0 ? 1.0 : 0.0);   // B[k=0][j=mode] = 1 -> D[i][mode] = A[i][0]
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main()
{
    double *d; hipMalloc(&d, 64 * 4 * 8); double h[256];
    for (int mode = 0; mode < 16; mode += 5) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("B = e(k=0, j=%d): nonzero outputs (lane, reg) -> value (= A[i][0] = 100 i + 1):\n", mode);
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (h[l * 4 + r] != 0) printf("  lane %2d reg %d : %g\n", l, r, h[l * 4 + r]);
    }
    return 0;
}
