// Instruction cost table for gfx950 (wave64): ns per wave-instruction per SIMD with 1 and 4 waves per SIMD, measured with HIP events
// around long kernels of unrolled inline-asm blocks (4 independent chains unless the name says "dep").
// hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate.bin && ./tools/ubench/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 256
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define D(v) (*(double *)&v)
#define OPS(X) \
    X(0,  "v_add_f32",            "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4", F) \
    X(1,  "v_fma_f32",            "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4", F) \
    X(2,  "v_pk_add_f32",         "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4", P) \
    X(3,  "v_pk_fma_f32",         "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4", P) \
    X(4,  "v_min3_f32",           "v_min3_f32 %0, %0, %4, %4\n v_min3_f32 %1, %1, %4, %4\n v_min3_f32 %2, %2, %4, %4\n v_min3_f32 %3, %3, %4, %4", F) \
    X(5,  "v_exp_f32",            "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3", F) \
    X(6,  "v_cndmask_b32 vcc",    "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc", F) \
    X(7,  "v_cndmask_b32 s[10:11]", "v_cndmask_b32 %0, %0, %4, s[10:11]\n v_cndmask_b32 %1, %1, %4, s[10:11]\n v_cndmask_b32 %2, %2, %4, s[10:11]\n v_cndmask_b32 %3, %3, %4, s[10:11]", F) \
    X(8,  "v_add_f64",            "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4", P) \
    X(9,  "v_mul_f64",            "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4", P) \
    X(10, "v_fma_f64",            "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4", P) \
    X(11, "v_fma_f64 dep",        "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %0, %0, %4, %4", P) \
    X(12, "v_rcp_f64",            "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3", P) \
    X(13, "v_cmp_lt_f64 vcc",     "v_cmp_lt_f64 vcc, %0, %4\n v_cmp_lt_f64 vcc, %1, %4\n v_cmp_lt_f64 vcc, %2, %4\n v_cmp_lt_f64 vcc, %3, %4", P) \
    X(14, "v_mov_b32 dpp row_shr", "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf", F) \
    X(15, "v_mov_b32 dpp wave_shr", "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf", F) \
    X(16, "v_readlane_b32",       "v_readlane_b32 s10, %0, 3\n v_readlane_b32 s11, %1, 5\n v_readlane_b32 s12, %2, 7\n v_readlane_b32 s13, %3, 9", F) \
    X(17, "v_readfirstlane+use",  "v_readfirstlane_b32 s10, %0\n v_add_f32 %1, s10, %1\n v_readfirstlane_b32 s11, %2\n v_add_f32 %3, s11, %3", F) \
    X(18, "v_mov_b32",            "v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4", F) \
    X(19, "v_mov_b64",            "v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4", P) \
    X(20, "v_add_u32",            "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4", F) \
    X(21, "v_lshl_add_u32",       "v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4", F) \
    X(22, "v_max_f64",            "v_max_f64 %0, %0, %4\n v_max_f64 %1, %1, %4\n v_max_f64 %2, %2, %4\n v_max_f64 %3, %3, %4", P) \
    X(23, "v_cmp_lt_f32 vcc",     "v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4", F) \
    X(24, "cmp vcc + cndmask vcc (2)", "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc", F) \
    X(25, "cmp sgpr + cndmask sgpr (2)", "v_cmp_lt_f32 s[10:11], %0, %4\n v_cndmask_b32 %1, %1, %4, s[10:11]\n v_cmp_lt_f32 s[12:13], %2, %4\n v_cndmask_b32 %3, %3, %4, s[12:13]", F) \
    X(26, "cmp vcc + 2 cndmask vcc (3)", "v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_add_f32 %3, %3, %4", F) \
    X(27, "cndmask vcc, 3 adds between", "v_cndmask_b32 %0, %0, %4, vcc\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4", F) \
    X(28, "v_cndmask_b32_e64 vcc", "v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc", F) \
    X(29, "v_addc_co_u32 vcc",   "v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4", F) \
    X(30, "v_div_fmas_f32 (vcc)", "v_div_fmas_f32 %0, %0, %4, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4", F) \
    X(31, "ds_bpermute+wait",    "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4", F)

template <int OP>
__global__ void k(unsigned long long *out, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float b0 = seed, b1 = seed + 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#define F(n, txt) REP64(asm volatile(txt : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11", "s12", "s13");)
#define P(n, txt) REP64(asm volatile(txt : "+v"(D(a0)), "+v"(D(a2)), "+v"(D(a4)), "+v"(D(a6)) : "v"(D(b0)) : "vcc", "s10", "s11", "s12", "s13");)
#define X(n, name, txt, kind) if (OP == n) { kind(n, txt) }
        OPS(X)
#undef X
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b1 == 12345.f) out[0] = 0;
}
template <int OP> void run(const char *name)
{
    unsigned long long *d; hipMalloc(&d, 8 * 4096);
    double ns[2], ticks = 0;
    int c = 0;
    for (int wpb : {256, 1024}) {       // 256 threads = 1 wave per SIMD, 1024 = 4 per SIMD (one workgroup per CU: grid 256)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(wpb), 0, 0, d, 1.0f);
        hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(256), dim3(wpb), 0, 0, d, 1.0f); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        const double n = (double)ITER * 64 * 4;
        if (wpb == 256) ticks = h[1] / n;
        ns[c++] = ms * 1e6 / (n * (wpb / 256));
    }
    printf("%-28s 1 wave/SIMD %6.2f ns/instr (%5.2f s_memtime ticks)   4 waves/SIMD %6.2f ns/instr\n", name, ns[0], ticks, ns[1]);
    fflush(stdout); hipFree(d);
}
int main()
{
#define X(n, name, txt, kind) run<n>(name);
    OPS(X)
    return 0;
}
