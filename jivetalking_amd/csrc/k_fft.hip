// k_fft.hip — LDS-resident FFT kernels for gfx950:
//   * aspectralstats (win_size=2048, hann, hop 1024; filters.go:625, analyser_output.go:18): one 256-thread
//     workgroup walks a run of consecutive hops, FFT-2048 in LDS, 13 spectral statistics by wave-shuffle reductions,
//     previous magnitudes kept in LDS for flux.
//   * afftdn (filters.go:830-861; af_afftdn.c process_frame, tn=0): overlapped STFT (window 3A, hop A = rate/80,
//     FFT 2048 @48 kHz), per-bin decision-directed gain with bark-band masking limits, inverse FFT, overlap-add.
//     The frame-to-frame recurrences (prior[], prior_band_excit[]) are contractions, so the frame axis is split
//     into chunks with warm-up frames.
// f32 butterflies without FMA contraction (compiled -ffp-contract=off) so rounding follows a scalar C build.
#include "jt_internal.h"
#include <cfloat>

constexpr int FT = 256;   // threads per workgroup

__device__ inline unsigned brev(unsigned x, int bits) { return __brev(x) >> (32 - bits); }

// in-place radix-2 DIT over LDS arrays holding bit-reversed input; tw[k] = exp(-2*pi*i*k/N), k < N/2
template <int LOG2N>
__device__ inline void fft_lds(float *re, float *im, const float2 *__restrict__ tw)
{
    constexpr int N = 1 << LOG2N;
#pragma unroll 1
    for (int s = 1; s <= LOG2N; ++s) {
        const int half = 1 << (s - 1);
        const int tstride = N >> s;
        for (int b = threadIdx.x; b < N / 2; b += FT) {
            int k = b & (half - 1);
            int i = ((b >> (s - 1)) << s) + k;
            int j = i + half;
            float2 w = tw[k * tstride];
            float rj = re[j], ij = im[j], ri = re[i], ii = im[i];
            float xr = rj * w.x - ij * w.y;
            float xi = rj * w.y + ij * w.x;
            re[j] = ri - xr; im[j] = ii - xi;
            re[i] = ri + xr; im[i] = ii + xi;
        }
        __syncthreads();
    }
}

template <int K>
__device__ inline void block_reduce_sum(float (&v)[K], float *scratch /* [K][4] */)
{
#pragma unroll
    for (int k = 0; k < K; ++k)
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < K; ++k) scratch[k * 4 + w] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = scratch[k * 4 + 0] + scratch[k * 4 + 1] + scratch[k * 4 + 2] + scratch[k * 4 + 3];
    __syncthreads();
}

// ------------------------------------------------------------------ aspectralstats
constexpr int SP_LOG2 = 11, SP_N = 1 << SP_LOG2, SP_HALF = SP_N / 2, SP_RUN = 32;

// sel_blk == 0: every hop (out[h]).  sel_blk > 0: only the hops whose props survive ebur128's 100 ms re-framing, i.e. for
// output frame k the hop containing sample k*sel_blk (+ its predecessor for flux); out[k].  A run of SP_RUN consecutive
// frames per workgroup; a hop already in LDS as "previous magnitudes" is not recomputed.
__global__ void __launch_bounds__(FT)
k_aspectralstats(const float *__restrict__ in, int64_t n, int sr, const float2 *__restrict__ tw, const float *__restrict__ hann,
                 jt_spectral *__restrict__ hops, int64_t nhops, int sel_blk, int64_t nframes)
{
    __shared__ float re[SP_N], im[SP_N];
    __shared__ float mag[SP_HALF], prev[SP_HALF];
    __shared__ float scratch[16 * 4];
    __shared__ float psum[FT];
    __shared__ int roll_idx;
    const int tid = threadIdx.x;
    const int64_t u0 = (int64_t)blockIdx.x * SP_RUN;     // first hop (all-hops mode) or first frame (selected mode)
    const float max_freq = (float)(sr / 2);
    const float scale = max_freq / (float)SP_HALF;
    const float fscale = 1.f / SP_N;
    for (int i = tid; i < SP_HALF; i += FT) prev[i] = 0.f;
    __syncthreads();
    const int64_t nunits = sel_blk > 0 ? nframes : nhops;
    int64_t have_prev = -2;                           // hop whose magnitudes sit in prev[] (-2: none, prev[] = zeros for hop -1)
    for (int64_t unit = u0; unit < u0 + SP_RUN && unit < nunits; ++unit) {
      int64_t htarget = sel_blk > 0 ? min((unit * (int64_t)sel_blk) / SP_HALF, nhops - 1) : unit;
      for (int64_t h = (have_prev == htarget - 1 || htarget == 0) ? htarget : htarget - 1; h <= htarget; ++h) {
        const bool emit = (h == htarget);
        if (h == 0 && have_prev != -2) { for (int i = tid; i < SP_HALF; i += FT) prev[i] = 0.f; __syncthreads(); }
        // window after pushing hop h = samples [(h+1)*1024 - 2048, (h+1)*1024)
        const int64_t w0 = (h + 1) * (int64_t)SP_HALF - SP_N;
        for (int i = tid; i < SP_N; i += FT) {
            int64_t k = w0 + i;
            float x = (k >= 0 && k < n) ? in[k] : 0.f;
            unsigned r = brev((unsigned)i, SP_LOG2);
            re[r] = x * hann[i];
            im[r] = 0.f;
        }
        __syncthreads();
        fft_lds<SP_LOG2>(re, im, tw);
        for (int i = tid; i < SP_HALF; i += FT) mag[i] = hypotf(re[i] * fscale, im[i] * fscale);
        __syncthreads();
        if (emit) {
            // pass 1
            float v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            float mx = 0.f;
            const float m0 = mag[0];
            for (int i = tid; i < SP_HALF; i += FT) {
                float m = mag[i];
                v[0] += m;                                   // sum mag
                v[1] += m * i * scale;                       // centroid numerator
                float me = FLT_EPSILON + m;
                v[2] += logf(me);                            // flatness log-sum
                v[3] += me;                                  // flatness den
                v[4] += m * logf(m + FLT_EPSILON);           // entropy
                float df = m - prev[i];
                v[5] += df * df;                             // flux
                if (i >= 1) { v[6] += (m - m0) / i; v[7] += m; }   // decrease
                mx = fmaxf(mx, m);
            }
            block_reduce_sum<9>(v, scratch);
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, 64));
            if ((tid & 63) == 0) scratch[tid >> 6] = mx;
            __syncthreads();
            mx = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
            __syncthreads();
            const float sum = v[0];
            const float mean = sum / SP_HALF;
            const float centroid = sum <= FLT_EPSILON ? 1.f : v[1] / sum;
            // pass 2
            float u[6] = {0, 0, 0, 0, 0, 0};
            const float mm = SP_HALF * 0.5f;
            for (int i = tid; i < SP_HALF; i += FT) {
                float m = mag[i];
                float dm = m - mean;
                u[0] += dm * dm;
                float d = i * scale - centroid;
                u[1] += m * d * d;
                u[2] += m * d * d * d;
                u[3] += m * d * d * d * d;
                float a = (i - mm) / mm;
                u[4] += a * dm;
                u[5] += a * a;
            }
            block_reduce_sum<6>(u, scratch);
            // rolloff: contiguous 4-bin partial sums -> block scan
            float p4 = 0.f;
            for (int q = 0; q < 4; ++q) p4 += mag[tid * 4 + q];
            psum[tid] = p4;
            if (tid == 0) roll_idx = 0;
            __syncthreads();
            if (tid == 0) {   // sequential inclusive scan over 256 partials (exact left-to-right order)
                float run = 0.f;
                for (int t = 0; t < FT; ++t) { float s0 = run; run += psum[t]; psum[t] = s0; }
            }
            __syncthreads();
            {
                const float norm = sum * 0.85f;
                float run = psum[tid];
                int found = -1;
                for (int q = 0; q < 4; ++q) { run += mag[tid * 4 + q]; if (found < 0 && run >= norm) found = tid * 4 + q; }
                // first thread (lowest index) whose range crosses the threshold wins
                if (found >= 0 && psum[tid] < norm) roll_idx = found;
            }
            __syncthreads();
            if (tid == 0) {
                jt_spectral o;
                float spread = sum <= FLT_EPSILON ? 1.f : sqrtf(u[1] / sum);
                float d3 = sum * spread * spread * spread;
                float d4 = d3 * spread;
                o.mean = mean;
                o.variance = u[0] / SP_HALF;
                o.centroid = centroid;
                o.spread = spread;
                o.skewness = d3 <= FLT_EPSILON ? 1.f : u[2] / d3;
                o.kurtosis = d4 <= FLT_EPSILON ? 1.f : u[3] / d4;
                o.entropy = -v[4] / logf((float)SP_HALF);
                float fnum = expf(v[2] / SP_HALF), fden = v[3] / SP_HALF;
                o.flatness = fden <= FLT_EPSILON ? 0.f : fnum / fden;
                o.crest = mean <= FLT_EPSILON ? 0.f : mx / mean;
                o.flux = sqrtf(v[5]);
                o.slope = fabsf(u[5]) <= FLT_EPSILON ? 0.f : u[4] / u[5];
                o.decrease = v[7] <= FLT_EPSILON ? 0.f : v[6] / v[7];
                o.rolloff = roll_idx * scale;
                hops[sel_blk > 0 ? unit : h] = o;
            }
            __syncthreads();
        }
        for (int i = tid; i < SP_HALF; i += FT) prev[i] = mag[i];
        __syncthreads();
        have_prev = h;
      }
    }
}

void launch_aspectralstats(const float *in, int64_t n, int sr, int win_size, const float2 *twiddle, const float *hann,
                           jt_spectral *hops, int64_t nhops, int sel_blk, int64_t nframes, hipStream_t s)
{
    if (nhops <= 0) return;
    JT_REQUIRE(win_size == SP_N, JT_E_UNSUPPORTED, "aspectralstats: only win_size=2048 is built");
    const int64_t units = sel_blk > 0 ? nframes : nhops;
    if (units <= 0) return;
    unsigned grid = (unsigned)((units + SP_RUN - 1) / SP_RUN);
    hipLaunchKernelGGL(k_aspectralstats, dim3(grid), dim3(FT), 0, s, in, n, sr, twiddle, hann, hops, nhops, sel_blk, nframes);
}

// ------------------------------------------------------------------ afftdn
__device__ inline double limit_gain(double a, double b)
{
    if (a > 1.0) return (b * a - 1.0) / (b + a - 2.0);
    if (a < 1.0) return (b * a - 2.0 * a + 1.0) / (b - a);
    return 1.0;
}

constexpr int AF_MAXBANDS = 48;
constexpr int AF_BPT = 5;      // bins per thread for L = 2048 (1025 bins over 256 threads)

template <int LOG2N>
__global__ void __launch_bounds__(FT)
k_afftdn(const float *__restrict__ in, float *__restrict__ out, int64_t n, AfftdnDev d, int frames_per_chunk, int warm_frames,
         int64_t nframes)
{
    constexpr int L = 1 << LOG2N;
    constexpr int BINS = L / 2 + 1;
    constexpr int BPT = (BINS + FT - 1) / FT;
    extern __shared__ unsigned char smem_raw[];
    // layout: re[L] im[L] re2[L] im2[L] (float) | clean[BINS] acc[W] (double) | band arrays
    float *re = reinterpret_cast<float *>(smem_raw);
    float *im = re + L;
    float *re2 = im + L;
    float *im2 = re2 + L;
    double *clean = reinterpret_cast<double *>(im2 + L);
    double *acc = clean + ((BINS + 1) & ~1);
    double *band_excit = acc + ((d.W + 1) & ~1);
    double *prior_band = band_excit + AF_MAXBANDS;
    double *band_amt = prior_band + AF_MAXBANDS;
    int *band_lo = reinterpret_cast<int *>(band_amt + AF_MAXBANDS);
    int *band_hi = band_lo + AF_MAXBANDS;

    const int tid = threadIdx.x;
    const int A = d.A, W = d.W, nb = d.nbands;
    const int64_t t_lo = (int64_t)blockIdx.x * frames_per_chunk;
    const int64_t t_hi = min(t_lo + frames_per_chunk, nframes);
    const int64_t t0 = max((int64_t)0, t_lo - warm_frames);

    // band bin ranges (bin2band is non-decreasing)
    if (tid < nb) { band_lo[tid] = BINS; band_hi[tid] = 0; }
    __syncthreads();
    for (int i = tid; i < BINS; i += FT) {
        int b = d.bin2band[i];
        if (i == 0 || d.bin2band[i - 1] != b) band_lo[b] = i;
        if (i == BINS - 1 || d.bin2band[i + 1] != b) band_hi[b] = i + 1;
    }
    if (tid < nb) prior_band[tid] = 0.0;
    for (int m = tid; m < W; m += FT) acc[m] = 0.0;
    double prior[BPT];
#pragma unroll
    for (int q = 0; q < BPT; ++q) prior[q] = 0.0;
    __syncthreads();

    for (int64_t t = t0; t < t_hi; ++t) {
        const int64_t start = t * A - (W - A);
        for (int m = tid; m < L; m += FT) {
            float v = 0.f;
            if (m < W) {
                int64_t k = start + m;
                float x = (k >= 0 && k < n) ? in[k] : 0.f;
                v = (float)(d.window[m] * x * 8388608.0);
            }
            unsigned r = brev((unsigned)m, LOG2N);
            re[r] = v; im[r] = 0.f;
        }
        __syncthreads();
        fft_lds<LOG2N>(re, im, d.twiddle);
        // per-bin gains (process_frame); the very first frame of the stream uses ratio = 1
        const double ratio = (t == 0) ? 1.0 : 0.5, rratio = 1.0 - ratio;
        double gain_r[BPT];
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            int i = tid + q * FT;
            gain_r[q] = 0.0;
            if (i < BINS) {
                double mag = hypot((double)re[i], (double)im[i]);
                double power = mag * mag;
                double mav = power / d.abs_var[i];
                double nmav = ratio * prior[q] + rratio * fmax(mav - 1.0, 0.0);
                double ng = nmav / (1.0 + nmav);
                double sq = ng * ng;
                prior[q] = mav * sq;
                clean[i] = power * sq;
                gain_r[q] = ng;
            }
        }
        __syncthreads();
        if (tid < nb) {
            double e = 0.0;
            for (int i = band_lo[tid]; i < band_hi[tid]; ++i) e += clean[i];
            e = fmax(e, d.alpha[tid] * e + d.beta[tid] * prior_band[tid]);
            prior_band[tid] = e;
            band_excit[tid] = e;
        }
        __syncthreads();
        if (tid < nb) {
            double a = 0.0;
            const double *sp = d.spread + (size_t)tid * nb;
            for (int k = 0; k < nb; ++k) a += sp[k] * band_excit[k];
            band_amt[tid] = a;
        }
        __syncthreads();
        const bool need_out = (t + 2 >= t_lo);     // frames whose overlap-add reaches the emitted range
#pragma unroll
        for (int q = 0; q < BPT; ++q) {
            int i = tid + q * FT;
            if (i < BINS) {
                double amt = band_amt[d.bin2band[i]];
                double g = gain_r[q];
                double av = d.abs_var[i];
                if (amt > av) g = 1.0;
                else if (amt > d.min_abs_var[i]) g = limit_gain(g, sqrt(av / amt));
                else g = limit_gain(g, d.max_gain);
                float gf = (float)g;
                re[i] *= gf; im[i] *= gf;
            }
        }
        __syncthreads();
        if (need_out) {
            // inverse real transform: rebuild the conjugate half, conjugate, forward FFT (unnormalised inverse)
            for (int k = tid; k < L; k += FT) {
                float r, ii;
                if (k == 0) { r = re[0]; ii = 0.f; }
                else if (k == L / 2) { r = re[L / 2]; ii = 0.f; }
                else if (k < L / 2) { r = re[k]; ii = im[k]; }
                else { r = re[L - k]; ii = -im[L - k]; }
                unsigned rr = brev((unsigned)k, LOG2N);
                re2[rr] = r; im2[rr] = -ii;
            }
            __syncthreads();
            fft_lds<LOG2N>(re2, im2, d.twiddle);
            for (int m = tid; m < W; m += FT) acc[m] += d.window[m] * (double)re2[m] / 8388608.0;
            __syncthreads();
            if (t >= t_lo) {
                for (int m = tid; m < A; m += FT) {
                    int64_t k = start + m;
                    if (k >= 0 && k < n) out[k] = (float)acc[m];
                }
            }
            __syncthreads();
            // shift the accumulator by one hop
            double tmp[16];
            int cnt = 0;
            for (int m = tid; m < W; m += FT) { tmp[cnt++] = (m + A < W) ? acc[m + A] : 0.0; }
            __syncthreads();
            cnt = 0;
            for (int m = tid; m < W; m += FT) acc[m] = tmp[cnt++];
            __syncthreads();
        }
    }
}

void launch_afftdn(const float *in, float *out, int64_t n, const AfftdnDev &d, int frames_per_chunk, int warm_frames, hipStream_t s)
{
    if (n <= 0) return;
    JT_REQUIRE(d.nbands <= AF_MAXBANDS, JT_E_UNSUPPORTED, "afftdn: too many bark bands");
    JT_REQUIRE(d.W <= 16 * FT, JT_E_UNSUPPORTED, "afftdn: window too long for this build");
    int64_t nframes = (n + d.A - 1) / d.A + (d.W - d.A) / d.A;
    unsigned grid = (unsigned)((nframes + frames_per_chunk - 1) / frames_per_chunk);
    size_t bins = d.L / 2 + 1;
    size_t smem = sizeof(float) * 4 * d.L + sizeof(double) * (((bins + 1) & ~(size_t)1) + ((d.W + 1) & ~1) + 3 * AF_MAXBANDS)
                + sizeof(int) * 2 * AF_MAXBANDS;
    if (d.L == 2048) {
        JT_HIP(hipFuncSetAttribute((const void *)k_afftdn<11>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(k_afftdn<11>, dim3(grid), dim3(FT), smem, s, in, out, n, d, frames_per_chunk, warm_frames, nframes);
    } else if (d.L == 4096) {
        JT_HIP(hipFuncSetAttribute((const void *)k_afftdn<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(k_afftdn<12>, dim3(grid), dim3(FT), smem, s, in, out, n, d, frames_per_chunk, warm_frames, nframes);
    } else if (d.L == 1024) {
        JT_HIP(hipFuncSetAttribute((const void *)k_afftdn<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(k_afftdn<10>, dim3(grid), dim3(FT), smem, s, in, out, n, d, frames_per_chunk, warm_frames, nframes);
    } else {
        throw JtError{JT_E_UNSUPPORTED, "afftdn: unsupported FFT length"};
    }
}
