// k_nlm.hip — anlmdn (FFmpeg af_anlmdn.c; filters.go:95-100,811-816: anlmdn=s=0.00001:p=0.0060:r=0.0020:m=3)
// for gfx950.  This is the one ALU-bound stage of the path (~2S patch comparisons per sample).
//
// Mapping.  FFmpeg processes hops of H = 2K+1 output samples; inside a hop it keeps, for each of the 2S search
// offsets, a running patch SSD (`cache[j]`) that is seeded exactly at the hop start and then updated with two
// squared differences per step — a recurrence ALONG TIME for fixed offset — and then, for each output sample,
// accumulates the weighted neighbours sequentially ACROSS offsets.  One workgroup = one hop, one thread = one
// offset: each thread carries its SSD in a register through the hop (identical f32 operation order to the C
// code, no FMA contraction), writes its weight for TI consecutive outputs into a padded LDS tile, and the tile
// is then reduced one-thread-per-output in ascending-offset order — i.e. the same summation order as FFmpeg,
// so the result is bit-identical to a scalar build.  The hop's input window (H + 2(K+S) floats) is staged in
// LDS once; HBM traffic is 1 read (+halo) and 1 write per sample.
#include "jt_internal.h"

constexpr int NLM_TI = 64;     // outputs per weight tile

__global__ void k_anlmdn(const float *__restrict__ in, float *__restrict__ out, int64_t n, int K, int S,
                         float sw, float smooth, float lut_scale, int64_t nhops)
{
    extern __shared__ float smem[];
    const int H = 2 * K + 1;
    const int NW = H + 2 * (K + S);
    const int S2 = 2 * S;
    float *win = smem;                       // [NW]
    float *wt = smem + ((NW + 3) & ~3);      // [NLM_TI][S2 + 1]
    const int tid = threadIdx.x;             // offset index j in [0, 2S)
    const int64_t hop = (int64_t)blockIdx.x;
    const int64_t hs = hop * H - (K + S);    // first output sample of this hop
    for (int w = tid; w < NW; w += blockDim.x) {
        int64_t k = hs - (K + S) + w;
        win[w] = (k >= 0 && k < n) ? in[k] : 0.f;
    }
    __syncthreads();
    const float *f = win + K;                // f[i], i in [-K, N-K)
    const int dj = tid - S + (tid >= S ? 1 : 0);   // neighbour offset relative to the centre
    float cache = 0.f;
    const int wstride = S2 + 1;
    for (int i0 = S; i0 < H + S; i0 += NLM_TI) {
        const int ti_n = min(NLM_TI, H + S - i0);
        if (tid < S2) {
            for (int ii = 0; ii < ti_n; ++ii) {
                const int i = i0 + ii;
                if (i == S) {
                    // compute_distance_ssd(f + i, f + j, K), j = i + dj
                    float dist = 0.f;
                    const float *f1 = f + i, *f2 = f + i + dj;
                    for (int k = -K; k <= K; ++k) {
                        float dd = __fsub_rn(f1[k], f2[k]);
                        dist = __fadd_rn(dist, __fmul_rn(dd, dd));
                    }
                    cache = dist;
                } else {
                    // compute_cache(): cache += -(f[i-K-1]-f[j-K-1])^2 + (f[i+K]-f[j+K])^2
                    const int j = i + dj;
                    float a = __fsub_rn(f[i - K - 1], f[j - K - 1]);
                    float b = __fsub_rn(f[i + K], f[j + K]);
                    float t = __fadd_rn(-__fmul_rn(a, a), __fmul_rn(b, b));
                    cache = __fadd_rn(cache, t);
                }
                float distance = cache;
                if (distance < 0.f) cache = distance = 0.f;
                float w = __fmul_rn(distance, sw);
                float weight = 0.f;
                if (!(w >= smooth)) {
                    unsigned idx = (unsigned)__fmul_rn(w, lut_scale);
                    weight = expf(-(float)idx / lut_scale);      // weight_lut[idx]
                }
                wt[ii * wstride + tid] = weight;
            }
        }
        __syncthreads();
        if (tid < ti_n) {
            const int i = i0 + tid;
            const float *wr = wt + tid * wstride;
            float P = 0.f, Q = 0.f;
            for (int j = 0; j < S2; ++j) {
                float w = wr[j];
                if (w != 0.f) {       // skipped offsets contribute nothing (w >= smooth -> continue)
                    P = __fadd_rn(P, __fmul_rn(w, f[i - S + j + (j >= S ? 1 : 0)]));
                    Q = __fadd_rn(Q, w);
                }
            }
            P = __fadd_rn(P, f[i]);
            Q = __fadd_rn(Q, 1.f);
            int64_t o = hs + (i - S);
            if (o >= 0 && o < n) out[o] = P / Q;
        }
        __syncthreads();
    }
}

void launch_anlmdn(const float *in, float *out, int64_t n, int K, int S, float sw, float smooth, float lut_scale, hipStream_t s)
{
    if (n <= 0) return;
    const int H = 2 * K + 1;
    const int NW = H + 2 * (K + S);
    int64_t nhops = (n + (K + S) + H - 1) / H;
    int threads = ((2 * S + 63) / 64) * 64;
    if (threads < NLM_TI) threads = NLM_TI;
    JT_REQUIRE(threads <= 1024, JT_E_UNSUPPORTED, "anlmdn: research radius too large for one workgroup");
    size_t smem = sizeof(float) * (((NW + 3) & ~3) + (size_t)NLM_TI * (2 * S + 1));
    JT_REQUIRE(smem <= 160 * 1024, JT_E_UNSUPPORTED, "anlmdn: patch/research window exceeds LDS");
    JT_HIP(hipFuncSetAttribute((const void *)k_anlmdn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_anlmdn, dim3((unsigned)nhops), dim3(threads), smem, s, in, out, n, K, S, sw, smooth, lut_scale, nhops);
}
