// k_resample.hip — libswresample-equivalent polyphase kaiser-sinc resampling on gfx950.
//   * 48k/96k -> 44.1k + f64->s16 (aformat=sample_rates=44100:sample_fmts=s16, filters.go:706-710)
//   * x -> 192 kHz true-peak scan fused with a per-100ms max reduce (ebur128 peak=true, filters.go:626);
//     the oversampled signal is never materialised
//   * 44.1k -> 192k fused with K-weighting and block energies (loudnorm measurement, normalise.go:256-264)
// One thread per output sample for the FIR kernels (gather of <=36 contiguous inputs, taps from a
// phase bank that stays L2-resident); HBM traffic = 1 input read + 1 output write.
#include "jt_internal.h"

__device__ inline double in_at_f(const float *in, int64_t n, int64_t k, int flush)
{
    if (k < 0) k = -k;                       // swr invert_initial_buffer(): in[-j] = in[j]
    if (k >= n) { if (!flush) return 0.0; k = 2 * n - 1 - k; if (k < 0) return 0.0; }   // resample_flush(): in[n+j] = in[n-1-j]
    return (double)in[k];
}
__device__ inline double in_at_d(const double *in, int64_t n, int64_t k, int flush)
{
    if (k < 0) k = -k;
    if (k >= n) { if (!flush) return 0.0; k = 2 * n - 1 - k; if (k < 0) return 0.0; }
    return in[k];
}

// f32 in (the dbl->flt->dbl rounded signal) -> DBLP resample -> s16
__global__ void k_resample_to_s16(const float *__restrict__ in, int64_t n, const double *__restrict__ bank, int phase_count,
                                  int L, int center, int64_t step, int16_t *__restrict__ out, int64_t m_total)
{
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_total) return;
    int64_t idx = m * step;
    int64_t si = idx / phase_count;
    int ph = (int)(idx - si * phase_count);
    const double *f = bank + (size_t)ph * L;
    double val = 0.0;
    int64_t k0 = si - center;
    if (k0 >= 0 && k0 + L <= n) {
        for (int i = 0; i < L; ++i) val += (double)in[k0 + i] * f[i];
    } else {
        for (int i = 0; i < L; ++i) val += in_at_f(in, n, k0 + i, 1) * f[i];
    }
    double r = rint(val * 32768.0);
    r = r < -32768.0 ? -32768.0 : (r > 32767.0 ? 32767.0 : r);
    out[m] = (int16_t)r;
}
void launch_resample_to_s16(const float *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                            int64_t step, int16_t *out, int64_t m, hipStream_t s)
{
    if (m <= 0) return;
    unsigned grid = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(k_resample_to_s16, dim3(grid), dim3(256), 0, s, in, n, bank, phase_count, filter_length, center, step, out, m);
}

// True peak: streaming swr (never flushed): output m exists iff its last tap index <= n-1; it becomes visible to
// ebur128 in the 100 ms frame that contains that last tap.  block_tp[b] = max |out| over outputs visible in frame b
// (b == nblocks_full collects the trailing partial frame).  Non-negative doubles order like their bit patterns.
template <typename TIn>
__global__ void k_true_peak(const TIn *__restrict__ in, int64_t n, const double *__restrict__ bank, int phase_count,
                            int L, int center, int64_t step, int blk, unsigned long long *__restrict__ block_tp,
                            int64_t nblocks_alloc, int64_t m_total)
{
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double a = 0.0; int64_t b = -1;
    if (m < m_total) {
        int64_t idx = m * step;
        int64_t si = idx / phase_count;
        int ph = (int)(idx - si * phase_count);
        int64_t k0 = si - center, last = k0 + L - 1;
        if (last <= n - 1) {
            const double *f = bank + (size_t)ph * L;
            double val = 0.0;
            if (k0 >= 0) { for (int i = 0; i < L; ++i) val += (double)in[k0 + i] * f[i]; }
            else { for (int i = 0; i < L; ++i) { int64_t k = k0 + i; if (k < 0) k = -k; val += (double)in[k] * f[i]; } }
            a = fabs(val);
            b = last / blk;
            if (b >= nblocks_alloc) b = nblocks_alloc - 1;
        }
    }
    // wave-level combine when the whole wave maps to one block (the common case)
    int64_t b0 = __shfl(b, 0, 64);
    bool uniform = __all(b == b0);
    if (uniform) {
        for (int off = 32; off > 0; off >>= 1) a = fmax(a, __shfl_down(a, off, 64));
        if ((threadIdx.x & 63) == 0 && b0 >= 0) atomicMax(&block_tp[b0], (unsigned long long)__double_as_longlong(a));
    } else if (b >= 0) {
        atomicMax(&block_tp[b], (unsigned long long)__double_as_longlong(a));
    }
}
void launch_true_peak_f32(const float *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                          int64_t step, int blk, double *block_tp, int64_t nblocks_alloc, int64_t m_total, hipStream_t s)
{
    if (m_total <= 0) return;
    unsigned grid = (unsigned)((m_total + 255) / 256);
    hipLaunchKernelGGL(k_true_peak<float>, dim3(grid), dim3(256), 0, s, in, n, bank, phase_count, filter_length, center, step, blk,
                       (unsigned long long *)block_tp, nblocks_alloc, m_total);
}
void launch_true_peak_f64(const double *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                          int64_t step, int blk, double *block_tp, int64_t nblocks_alloc, int64_t m_total, hipStream_t s)
{
    if (m_total <= 0) return;
    unsigned grid = (unsigned)((m_total + 255) / 256);
    hipLaunchKernelGGL(k_true_peak<double>, dim3(grid), dim3(256), 0, s, in, n, bank, phase_count, filter_length, center, step, blk,
                       (unsigned long long *)block_tp, nblocks_alloc, m_total);
}

// ------------------------------------------------------------------ fused resample -> K-weight -> block energies
// Lane-serial over the OUTPUT-rate time axis: each lane owns `bpl` 100 ms blocks of the 192 kHz stream plus a
// warm-up halo, computes every oversampled sample with the polyphase FIR straight from the input-rate signal
// (aresample flushes: right edge mirrored), runs the K-weighting recurrence (libavfilter/ebur128.c, combined
// 4th-order response == the two cascaded biquads) and accumulates z^2 per block and max|x| per block.
// FLT variant (s16 in, no limiter prefix): float taps, float accumulation (swr int_sample_fmt FLTP).
// DBL variant (after the alimiter prefix): double taps (DBLP).
template <typename TIn, typename TAcc, typename TBank>
__global__ void __launch_bounds__(64)
k_resample_kweight(const TIn *__restrict__ in, int64_t n, const TBank *__restrict__ bank, int phase_count, int L, int center,
                   int64_t step, int64_t m_total, int blk, int bpl, int64_t halo, BiquadF64 pre, BiquadF64 rlb,
                   double *__restrict__ block_sums, double *__restrict__ block_peaks, int64_t nblocks_alloc, int64_t nchunks,
                   double in_scale)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const int64_t chunk = (int64_t)blk * bpl;
    int64_t m_lo = c * chunk, m_hi = min(m_lo + chunk, m_total);
    int64_t m0 = max((int64_t)0, m_lo - halo);
    double x1 = 0, x2 = 0, y0 = 0, y1 = 0, y2 = 0, z0 = 0, z1 = 0, z2 = 0;
    double acc = 0.0, pk = 0.0; int cnt = 0; int64_t bidx = c * bpl;
    for (int64_t m = m0; m < m_hi; ++m) {
        int64_t idx = m * step;
        int64_t si = idx / phase_count;
        int ph = (int)(idx - si * phase_count);
        const TBank *f = bank + (size_t)ph * L;
        int64_t k0 = si - center;
        TAcc val = (TAcc)0;
        if (k0 >= 0 && k0 + L <= n) {
#pragma unroll 8
            for (int i = 0; i < L; ++i) val += (TAcc)((TAcc)in[k0 + i] * (TAcc)in_scale) * (TAcc)f[i];
        } else {
            for (int i = 0; i < L; ++i) {
                int64_t k = k0 + i;
                if (k < 0) k = -k;
                TAcc xv = (TAcc)0;
                if (k >= n) { k = 2 * n - 1 - k; if (k >= 0) xv = (TAcc)in[k] * (TAcc)in_scale; }
                else xv = (TAcc)in[k] * (TAcc)in_scale;
                val += xv * (TAcc)f[i];
            }
        }
        double x0 = (double)val;
        y2 = y1; y1 = y0;
        y0 = x0 * pre.b0 + x1 * pre.b1 + x2 * pre.b2 - y1 * pre.a1 - y2 * pre.a2;
        x2 = x1; x1 = x0;
        z2 = z1; z1 = z0;
        z0 = y0 * rlb.b0 + y1 * rlb.b1 + y2 * rlb.b2 - z1 * rlb.a1 - z2 * rlb.a2;
        if (m >= m_lo) {
            acc += z0 * z0;
            pk = fmax(pk, fabs(x0));
            if (++cnt == blk) {
                if (bidx < nblocks_alloc) { block_sums[bidx] = acc; block_peaks[bidx] = pk; }
                acc = 0.0; pk = 0.0; cnt = 0; ++bidx;
            }
        }
    }
    if (cnt > 0 && bidx < nblocks_alloc) { block_sums[bidx] = acc; block_peaks[bidx] = pk; }
}

static void rk_geometry(int64_t m_total, int blk, int *bpl, int64_t *halo, int64_t *nchunks)
{
    int64_t h = (int64_t)std::ceil(8192.0 * (blk * 10.0) / 48000.0);
    *halo = h;
    int b = 1;                       // one 100 ms block per lane: maximum parallelism, halo overhead ~1.7x
    *bpl = b;
    int64_t chunk = (int64_t)blk * b;
    *nchunks = (m_total + chunk - 1) / chunk;
}

void launch_resample_kweight_s16(const int16_t *in, int64_t n, const float *bankf, int phase_count, int filter_length, int center,
                                 int64_t step, int64_t m_total, int blk, BiquadF64 pre, BiquadF64 rlb,
                                 double *block_sums, double *block_peaks, int64_t nblocks_alloc, hipStream_t s)
{
    if (m_total <= 0) return;
    int bpl; int64_t halo, nchunks;
    rk_geometry(m_total, blk, &bpl, &halo, &nchunks);
    unsigned grid = (unsigned)((nchunks + 63) / 64);
    hipLaunchKernelGGL((k_resample_kweight<int16_t, float, float>), dim3(grid), dim3(64), 0, s, in, n, bankf, phase_count,
                       filter_length, center, step, m_total, blk, bpl, halo, pre, rlb, block_sums, block_peaks, nblocks_alloc,
                       nchunks, 1.0 / 32768.0);
}
void launch_resample_kweight_f64(const double *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                                 int64_t step, int64_t m_total, int blk, BiquadF64 pre, BiquadF64 rlb,
                                 double *block_sums, double *block_peaks, int64_t nblocks_alloc, hipStream_t s)
{
    if (m_total <= 0) return;
    int bpl; int64_t halo, nchunks;
    rk_geometry(m_total, blk, &bpl, &halo, &nchunks);
    unsigned grid = (unsigned)((nchunks + 63) / 64);
    hipLaunchKernelGGL((k_resample_kweight<double, double, double>), dim3(grid), dim3(64), 0, s, in, n, bank, phase_count,
                       filter_length, center, step, m_total, blk, bpl, halo, pre, rlb, block_sums, block_peaks, nblocks_alloc,
                       nchunks, 1.0);
}
