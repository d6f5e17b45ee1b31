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

// In-place Stockham FFT of H = 2^LOG2H complex points by ONE wave (H/256 radix-4 butterflies per lane and stage, a radix-2 stage when
// LOG2H is odd): every point of a stage is in registers before the first is written back, and the LDS operations of a wave are
// ordered, so no barrier is needed.  Same butterflies, twiddle values and operation order as fft_stockham (bit-identical results).
// What differs is where things sit in LDS, because the plain layout is bank-conflict bound (8-way on the stores of the stages with
// Ns = 1, 4 and on the twiddle loads of Ns = 4..64):
//  * point i is kept at fw_sk(i) = i + 4*(i >> 4) (four float2 of padding per sixteen points): the 16-point groups a stage with
//    Ns = 4 scatters to then start 40 dwords apart, eight distinct bank groups;
//  * the stage with Ns = 1 writes a lane's four consecutive outputs as two 16-byte stores;
//  * each stage has its own twiddle rows (w, w^2, w^3 of k, contiguous: fw_tw_fill), read once per stage while k does not depend on
//    the butterfly (Ns <= 64).
__device__ __forceinline__ int fw_sk(int i) { return i + ((i >> 4) << 2); }
template <int LOG2H> struct FwLayout {
    static constexpr int H = 1 << LOG2H;
    static constexpr int SKH = H + (H >> 2);                               // float2 per frame buffer
    static constexpr int N4 = LOG2H / 2;                                   // radix-4 stages
    static constexpr int TW4 = (((1 << (2 * N4)) - 4) / 3);                // 4 + 16 + ... + 4^(N4-1) rows of three
    static constexpr int TWN = 3 * TW4 + ((LOG2H & 1) ? H / 2 : 0);        // float2 in the table
    static constexpr int row0(int Ns) { return 3 * ((Ns - 4) / 3); }      // first float2 of the stage with this Ns (4 + ... + Ns/4 rows before)
};
// tws[row0(Ns) + 3 k + c] = exp(-2 pi i (c + 1) k / (4 Ns)); the radix-2 rows follow.  Every thread of the workgroup helps.
template <int LOG2H, bool F32ANGLE = false>
__device__ inline void fw_tw_fill(float2 *tws, int tid, int nthreads)
{
    using Lp = FwLayout<LOG2H>;
    constexpr int H = Lp::H;
    for (int i = tid; i < Lp::TWN; i += nthreads) {
        int q;                                                             // angle index: exp(-2 pi i q / H)
        if (i < 3 * Lp::TW4) {
            const int r = i / 3, c = i - 3 * r;                            // row r of the concatenated stages: Ns = 4: rows 0..3, Ns = 16: 4..19, ...
            int Ns = 4, base = 0;
            while (r >= base + Ns) { base += Ns; Ns <<= 2; }
            q = (c + 1) * (r - base) * (H / 4 / Ns);
        } else q = (i - 3 * Lp::TW4);                                      // radix-2 stage: Ns = H/2, step 1
        if (F32ANGLE) { float sn, cs; sincospif(2.0f * q / H, &sn, &cs); tws[i] = make_float2(cs, -sn); }       // (aspectralstats' table)
        else { double sn, cs; sincospi(2.0 * q / H, &sn, &cs); tws[i] = make_float2((float)cs, (float)-sn); }
    }
}
// One radix-4 stage of the wave transform (H = 1024 or 2048: B4 = H / 256 butterflies per lane).  With j = lane + 64 q every padded
// index splits into a per-lane base and a compile-time offset (lane < 64 never carries into the padding term), so the LDS operations
// use immediate offsets -- computed the obvious way, the 150-odd loop-invariant addresses of a transform are hoisted into registers
// and spill:
//   reads            fw_sk(j + c H/4)                = fw_sk(lane) + 80 q + (5 H / 16) c
//   writes, Ns = 1   fw_sk(4 j + c)                  = 4 lane + 4 (lane >> 2) + c + 320 q
//           Ns = 4   k = lane & 3:   fw_sk(o + 4 c)  = 4 (lane - k) + k + 4 (lane >> 2) + 4 c + 320 q
//           Ns = 16  k = lane & 15:  fw_sk(o + 16 c) = 4 (lane - k) + k + 16 (lane >> 4) + 20 c + 320 q
//           Ns = 64  k = lane:       fw_sk(o + 64 c) = fw_sk(lane) + 80 c + 320 q
//           Ns = 256 k = lane + 64 (q & 3):  fw_sk(o + 256 c) = fw_sk(lane) + 80 (q & 3) + 1280 (q >> 2) + 320 c
template <int LOG2H, bool INV, int NS>
__device__ __forceinline__ void fw_stage4(float2 *a, const float2 *__restrict__ tws, int lane)
{
    static_assert(LOG2H == 10 || LOG2H == 11, "immediate-offset layout derived for 1024 and 2048 points");
    static_assert(NS <= 256, "stages up to Ns = 256");
    using Lp = FwLayout<LOG2H>;
    constexpr int B4 = Lp::H / 256, RC = 5 * Lp::H / 16;
    const float2 *al = a + fw_sk(lane);
    float2 v[B4][4];
#pragma unroll
    for (int q = 0; q < B4; ++q) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[q][c] = al[80 * q + RC * c];
    }
    float2 w1, w2, w3;
    if (NS > 1 && NS <= 64) { const float2 *r = tws + Lp::row0(NS) + 3 * (lane & (NS - 1)); w1 = r[0]; w2 = r[1]; w3 = r[2]; }
    const int kl = lane & (NS - 1);
    float2 *wb = NS == 1 ? a + 4 * lane + 4 * (lane >> 2)
               : NS == 4 ? a + 4 * (lane - kl) + kl + 4 * (lane >> 2)
               : NS == 16 ? a + 4 * (lane - kl) + kl + 16 * (lane >> 4)
               : a + fw_sk(lane);
    constexpr int WC = NS == 1 ? 1 : NS == 4 ? 4 : NS == 16 ? 20 : NS == 64 ? 80 : 320;      // offset per output c
#pragma unroll
    for (int q = 0; q < B4; ++q) {
        const int wq = NS == 256 ? 80 * (q & 3) + 1280 * (q >> 2) : 320 * q;                 // offset of butterfly q (compile time)
        float2 v0 = v[q][0], v1 = v[q][1], v2 = v[q][2], v3 = v[q][3];
        if (NS > 1) {
            if (NS > 64) { const float2 *r = tws + Lp::row0(NS) + 3 * lane + 192 * (q & 3); w1 = r[0]; w2 = r[1]; w3 = r[2]; }   // k = lane + 64 (q & 3)
            float2 x1 = w1, x2 = w2, x3 = w3;
            if (INV) { x1.y = -x1.y; x2.y = -x2.y; x3.y = -x3.y; }
            float2 t;
            t.x = v1.x * x1.x - v1.y * x1.y; t.y = v1.x * x1.y + v1.y * x1.x; v1 = t;
            t.x = v2.x * x2.x - v2.y * x2.y; t.y = v2.x * x2.y + v2.y * x2.x; v2 = t;
            t.x = v3.x * x3.x - v3.y * x3.y; t.y = v3.x * x3.y + v3.y * x3.x; v3 = t;
        }
        const float2 s02 = make_float2(v0.x + v2.x, v0.y + v2.y), d02 = make_float2(v0.x - v2.x, v0.y - v2.y);
        const float2 s13 = make_float2(v1.x + v3.x, v1.y + v3.y), d13 = make_float2(v1.x - v3.x, v1.y - v3.y);
        const float2 r13 = INV ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);
        const float2 y0 = make_float2(s02.x + s13.x, s02.y + s13.y), y1 = make_float2(d02.x + r13.x, d02.y + r13.y);
        const float2 y2 = make_float2(s02.x - s13.x, s02.y - s13.y), y3 = make_float2(d02.x - r13.x, d02.y - r13.y);
        if (NS == 1) {                                                     // four consecutive points, 32-byte aligned in the padded layout
            float4 *dst = reinterpret_cast<float4 *>(wb + wq);
            dst[0] = make_float4(y0.x, y0.y, y1.x, y1.y); dst[1] = make_float4(y2.x, y2.y, y3.x, y3.y);
        } else { wb[wq] = y0; wb[wq + WC] = y1; wb[wq + 2 * WC] = y2; wb[wq + 3 * WC] = y3; }
    }
}
template <int LOG2H, bool INV>
__device__ inline void fft_wave(float2 *a, const float2 *__restrict__ tws, int lane)
{
    using Lp = FwLayout<LOG2H>;
    constexpr int H = Lp::H, B2 = H / 2 / 64;
    fw_stage4<LOG2H, INV, 1>(a, tws, lane);
    if (Lp::N4 > 1) fw_stage4<LOG2H, INV, 4>(a, tws, lane);
    if (Lp::N4 > 2) fw_stage4<LOG2H, INV, 16>(a, tws, lane);
    if (Lp::N4 > 3) fw_stage4<LOG2H, INV, 64>(a, tws, lane);
    if (Lp::N4 > 4) fw_stage4<LOG2H, INV, 256>(a, tws, lane);
    if (LOG2H & 1) {
        // the radix-2 stage (Ns = H/2, k = j = lane + 64 q): points j and j + H/2 at fw_sk(lane) + 80 q (+ 5 H / 8), written back in place
        float2 *al = a + fw_sk(lane);
        const float2 *tl = tws + 3 * Lp::TW4 + lane;
        constexpr int HO = 5 * H / 8;
        float2 v[B2][2];
#pragma unroll
        for (int q = 0; q < B2; ++q) { v[q][0] = al[80 * q]; v[q][1] = al[80 * q + HO]; }
#pragma unroll
        for (int q = 0; q < B2; ++q) {
            float2 w = tl[64 * q];
            if (INV) w.y = -w.y;
            const float2 v0 = v[q][0], v1 = v[q][1];
            float2 t; t.x = v1.x * w.x - v1.y * w.y; t.y = v1.x * w.y + v1.y * w.x;
            al[80 * q] = make_float2(v0.x + t.x, v0.y + t.y);
            al[80 * q + HO] = make_float2(v0.x - t.x, v0.y - t.y);
        }
    }
}

// ------------------------------------------------------------------ aspectralstats
// One WAVE per analysis unit, four independent waves per workgroup, no workgroup barriers after the table setup.
// A unit = one hop whose statistics are wanted (+ its predecessor, whose magnitudes feed the flux term): two real FFTs of
// 2048 points, each through a 1024-point complex Stockham radix-4 transform held in the wave's own LDS slab (in place:
// a stage's 16 points per lane are read into registers before any is written back; LDS operations of one wave are ordered).
// The 13 statistics are two strided passes over the 1024 magnitudes with wave-shuffle reductions, and the roll-off is a
// wave prefix scan over 16-bin lane runs.
constexpr int SP_LOG2 = 11, SP_N = 1 << SP_LOG2, SP_HALF = SP_N / 2, SP_Q = SP_HALF / 2, SP_RUN = 8, SP_WAVES = 4;

__device__ inline float wsumf(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ inline int sp_skew(int i) { return i + (i >> 4); }

// magnitudes |X[i]| / 2048, i < 1024, of the hann-windowed frame ending at sample (h+1)*1024, into mag[sp_skew(i)]
__device__ inline void sp_frame_mags(const float *__restrict__ in, int64_t n, int64_t h, float2 *zbuf, const float2 *tw, const float *hann,
                                     const float2 (&wk)[8], float *mag, int lane)
{
    const int64_t w0 = (h + 1) * (int64_t)SP_HALF - SP_N;
    const int zl = fw_sk(lane), zm0 = fw_sk(SP_HALF - lane);         // padded positions: point lane + 64 q at zl + 80 q, its mirror at zm0 - 80 q
    float xv[32];
#pragma unroll
    for (int q = 0; q < 16; ++q) {                                   // clamped addresses: all the loads of the frame in flight at once
        const int64_t k0 = w0 + 2 * (lane + 64 * q), k1 = k0 + 1;
        const float a0 = in[min(max(k0, (int64_t)0), n - 1)], a1 = in[min(max(k1, (int64_t)0), n - 1)];
        xv[2 * q] = (k0 >= 0 && k0 < n) ? a0 : 0.f; xv[2 * q + 1] = (k1 >= 0 && k1 < n) ? a1 : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int m = lane + 64 * q;
        const int i0 = 2 * m, i1 = 2 * m + 1;
        zbuf[zl + 80 * q] = make_float2(xv[2 * q] * hann[i0 < SP_HALF ? i0 : SP_N - 1 - i0], xv[2 * q + 1] * hann[i1 < SP_HALF ? i1 : SP_N - 1 - i1]);
    }
    fft_wave<SP_LOG2 - 1, false>(zbuf, tw, lane);
    const float fscale = 1.f / SP_N;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = lane + 64 * q;                       // pair (k, 1024 - k), k < 512
        const float2 zk = zbuf[zl + 80 * q], zm = zbuf[(q == 0 && lane == 0) ? 0 : zm0 - 80 * q];
        const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);          // E = (Zk + conj Zm)/2
        const float orr = 0.5f * (zk.y + zm.y), oi = -0.5f * (zk.x - zm.x);        // O = -i (Zk - conj Zm)/2
        if (k == 0) mag[sp_skew(0)] = fabsf((er + orr) * fscale);                 // X[0] = Re Z0 + Im Z0 (real)
        else {
            const float cr = wk[q].x * orr - wk[q].y * oi, ci = wk[q].x * oi + wk[q].y * orr;   // W^k O
            mag[sp_skew(k)] = hypotf((er + cr) * fscale, (ei + ci) * fscale);                   // X[k] = E + W^k O
            mag[sp_skew(SP_HALF - k)] = hypotf((er - cr) * fscale, (ei - ci) * fscale);         // X[1024-k] = conj(E - W^k O)
        }
    }
    if (lane == 0) { const float2 z = zbuf[fw_sk(SP_Q)]; mag[sp_skew(SP_Q)] = hypotf(z.x * fscale, z.y * fscale); }   // X[512] = conj(Z[512])
}

// sel_blk == 0: every hop (out[h]).  sel_blk > 0: only the hops whose props survive ebur128's 100 ms re-framing, i.e. for
// output frame k the hop containing sample k*sel_blk; out[k].
__global__ void __launch_bounds__(64 * SP_WAVES)
k_aspectralstats(const float *__restrict__ in, int64_t n, int sr, const float *__restrict__ hann_g,
                 jt_spectral *__restrict__ hops, int64_t nhops, int sel_blk, int64_t nframes)
{
    using Lp = FwLayout<SP_LOG2 - 1>;
    __shared__ float2 tw[Lp::TWN];                   // per-stage twiddle rows (fw_tw_fill, f32 angles)
    __shared__ float hann[SP_HALF];                  // the f32 hann table is exactly symmetric: w[i] == w[N-1-i]
    __shared__ float2 zb[SP_WAVES][Lp::SKH];         // padded layout (fw_sk)
    __shared__ float magb[SP_WAVES][SP_HALF + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fw_tw_fill<SP_LOG2 - 1, true>(tw, tid, 64 * SP_WAVES);
    for (int q = tid; q < SP_HALF; q += 64 * SP_WAVES) hann[q] = hann_g[q];
    __syncthreads();
    float2 wk[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { double sn, cs; sincospi(2.0 * (lane + 64 * q) / SP_N, &sn, &cs); wk[q] = make_float2((float)cs, (float)-sn); }
    float2 *zbuf = zb[wave]; float *mag = magb[wave];
    const float max_freq = (float)(sr / 2);
    const float scale = max_freq / (float)SP_HALF;
    const int64_t nunits = sel_blk > 0 ? nframes : nhops;
    const int64_t u0 = ((int64_t)blockIdx.x * SP_WAVES + wave) * SP_RUN;
    for (int64_t unit = u0; unit < u0 + SP_RUN && unit < nunits; ++unit) {
        const int64_t h = sel_blk > 0 ? min((unit * (int64_t)sel_blk) / SP_HALF, nhops - 1) : unit;
        // previous hop's magnitudes (zeros before the first hop), kept in registers: bins lane + 64 q
        float pv[16];
        if (h == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) pv[q] = 0.f;
        } else {
            sp_frame_mags(in, n, h - 1, zbuf, tw, hann, wk, mag, lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) pv[q] = mag[sp_skew(lane + 64 * q)];
        }
        sp_frame_mags(in, n, h, zbuf, tw, hann, wk, mag, lane);
        // pass 1
        float v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        float mx = 0.f;
        const float m0 = mag[sp_skew(0)];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = lane + 64 * q;
            const float m = mag[sp_skew(i)];
            v[0] += m;                                   // sum mag
            v[1] += m * i * scale;                       // centroid numerator
            const float me = FLT_EPSILON + m;
            v[2] += logf(me);                            // flatness log-sum
            v[3] += me;                                  // flatness den
            v[4] += m * logf(m + FLT_EPSILON);           // entropy
            const float df = m - pv[q];
            v[5] += df * df;                             // flux
            if (i >= 1) { v[6] += (m - m0) / i; v[7] += m; }   // decrease
            mx = fmaxf(mx, m);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = wsumf(v[q]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const float sum = v[0];
        const float mean = sum / SP_HALF;
        const float centroid = sum <= FLT_EPSILON ? 1.f : v[1] / sum;
        // pass 2
        float u[6] = {0, 0, 0, 0, 0, 0};
        const float mm = SP_HALF * 0.5f;
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int i = lane + 64 * q;
            const float m = mag[sp_skew(i)];
            const float dm = m - mean;
            u[0] += dm * dm;
            const float d = i * scale - centroid;
            u[1] += m * d * d;
            u[2] += m * d * d * d;
            u[3] += m * d * d * d * d;
            const float a = (i - mm) / mm;
            u[4] += a * dm;
            u[5] += a * a;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) u[q] = wsumf(u[q]);
        // roll-off: first bin where the running sum reaches 85 % of the total; lane l owns bins [16 l, 16 l + 16)
        int roll_idx = 0;
        {
            float loc = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) loc += mag[sp_skew(16 * lane + q)];
            float inc = loc;
            for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            float run = inc - loc;                       // exclusive prefix
            const float norm = sum * 0.85f;
            int found = -1;
#pragma unroll
            for (int q = 0; q < 16; ++q) { run += mag[sp_skew(16 * lane + q)]; if (found < 0 && run >= norm) found = 16 * lane + q; }
            const unsigned long long bal = __ballot(found >= 0);
            if (bal) { const int first = __ffsll((long long)bal) - 1; roll_idx = __shfl(found, first, 64); }
        }
        if (lane == 0) {
            jt_spectral o;
            const float spread = sum <= FLT_EPSILON ? 1.f : sqrtf(u[1] / sum);
            const float d3 = sum * spread * spread * spread;
            const float d4 = d3 * spread;
            o.mean = mean;
            o.variance = u[0] / SP_HALF;
            o.centroid = centroid;
            o.spread = spread;
            o.skewness = d3 <= FLT_EPSILON ? 1.f : u[2] / d3;
            o.kurtosis = d4 <= FLT_EPSILON ? 1.f : u[3] / d4;
            o.entropy = -v[4] / logf((float)SP_HALF);
            const float fnum = expf(v[2] / SP_HALF), fden = v[3] / SP_HALF;
            o.flatness = fden <= FLT_EPSILON ? 0.f : fnum / fden;
            o.crest = mean <= FLT_EPSILON ? 0.f : mx / mean;
            o.flux = sqrtf(v[5]);
            o.slope = fabsf(u[5]) <= FLT_EPSILON ? 0.f : u[4] / u[5];
            o.decrease = v[7] <= FLT_EPSILON ? 0.f : v[6] / v[7];
            o.rolloff = roll_idx * scale;
            hops[sel_blk > 0 ? unit : h] = o;
        }
    }
}

void launch_aspectralstats(const float *in, int64_t n, int sr, int win_size, const float2 *twiddle, const float *hann,
                           jt_spectral *hops, int64_t nhops, int sel_blk, int64_t nframes, hipStream_t s)
{
    (void)twiddle;
    if (nhops <= 0) return;
    JT_REQUIRE(win_size == SP_N, JT_E_UNSUPPORTED, "aspectralstats: only win_size=2048 is built");
    const int64_t units = sel_blk > 0 ? nframes : nhops;
    if (units <= 0) return;
    const int per_wg = SP_WAVES * SP_RUN;
    unsigned grid = (unsigned)((units + per_wg - 1) / per_wg);
    hipLaunchKernelGGL(k_aspectralstats, dim3(grid), dim3(64 * SP_WAVES), 0, s, in, n, sr, hann, hops, nhops, sel_blk, nframes);
}

// ------------------------------------------------------------------ afftdn
__device__ inline double limit_gain(double a, double b)
{
    if (a > 1.0) return (b * a - 1.0) / (b + a - 2.0);
    if (a < 1.0) return (b * a - 2.0 * a + 1.0) / (b - a);
    return 1.0;
}

constexpr int AF_MAXBANDS = 48;

// Stockham autosort FFT of H = 2^LOG2H complex points in LDS (radix-4 stages, one radix-2 stage when LOG2H is odd):
// natural order in and out, ping-pong between `a` and `b`; returns the buffer holding the result.
// tw[q] = exp(-2*pi*i*q/H), q < H.  INV uses conjugated twiddles (unnormalised inverse).
template <int LOG2H, bool INV>
__device__ inline float2 *fft_stockham(float2 *a, float2 *b, const float2 *__restrict__ tw)
{
    constexpr int H = 1 << LOG2H;
    int Ns = 1;
#pragma unroll 1
    for (int st = 0; st < LOG2H / 2; ++st) {
        for (int j = threadIdx.x; j < H / 4; j += FT) {
            const int k = j & (Ns - 1);
            const int tq = k * (H / 4 / Ns);                       // twiddle step: angle = -2 pi k t / (4 Ns)
            float2 v0 = a[j], v1 = a[j + H / 4], v2 = a[j + H / 2], v3 = a[j + 3 * H / 4];
            if (Ns > 1) {
                float2 w1 = tw[tq], w2 = tw[2 * tq], w3 = tw[3 * tq];
                if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
                float2 t;
                t.x = v1.x * w1.x - v1.y * w1.y; t.y = v1.x * w1.y + v1.y * w1.x; v1 = t;
                t.x = v2.x * w2.x - v2.y * w2.y; t.y = v2.x * w2.y + v2.y * w2.x; v2 = t;
                t.x = v3.x * w3.x - v3.y * w3.y; t.y = v3.x * w3.y + v3.y * w3.x; v3 = t;
            }
            const float2 s02 = make_float2(v0.x + v2.x, v0.y + v2.y), d02 = make_float2(v0.x - v2.x, v0.y - v2.y);
            const float2 s13 = make_float2(v1.x + v3.x, v1.y + v3.y), d13 = make_float2(v1.x - v3.x, v1.y - v3.y);
            // forward: -i*d13 = (d13.y, -d13.x); inverse: +i*d13 = (-d13.y, d13.x)
            const float2 r13 = INV ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);
            const int o = ((j - k) << 2) + k;
            b[o] = make_float2(s02.x + s13.x, s02.y + s13.y);
            b[o + Ns] = make_float2(d02.x + r13.x, d02.y + r13.y);
            b[o + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
            b[o + 3 * Ns] = make_float2(d02.x - r13.x, d02.y - r13.y);
        }
        __syncthreads();
        float2 *t = a; a = b; b = t;
        Ns <<= 2;
    }
    if (LOG2H & 1) {
        for (int j = threadIdx.x; j < H / 2; j += FT) {
            const int k = j & (Ns - 1);
            float2 v0 = a[j], v1 = a[j + H / 2];
            float2 w = tw[k * (H / 2 / Ns)];
            if (INV) w.y = -w.y;
            float2 t; t.x = v1.x * w.x - v1.y * w.y; t.y = v1.x * w.y + v1.y * w.x;
            const int o = ((j - k) << 1) + k;
            b[o] = make_float2(v0.x + t.x, v0.y + t.y);
            b[o + Ns] = make_float2(v0.x - t.x, v0.y - t.y);
        }
        __syncthreads();
        float2 *t = a; a = b; b = t;
    }
    return a;
}

__device__ inline double fast_rcp(double a)
{
    double r = __builtin_amdgcn_rcp(a);
    r = fma(fma(-a, r, 1.0), r, r);
    r = fma(fma(-a, r, 1.0), r, r);
    return r;
}

// af_afftdn.c process_frame(), tn = 0.  Real FFT of length N = 2^LOG2N through an N/2-point complex transform
// (z[m] = x[2m] + i x[2m+1]; X[k] = E + W^k O, X[N/2-k] = conj(E - W^k O)), so a thread that owns the pair (k, N/2-k) owns
// both bins for the whole chunk: their decision-directed priors and noise constants live in registers.
// Band excitation sums: lanes hold consecutive bins, so each wave does a segmented shuffle reduction per bark band and the
// <= 8 per-segment partials of a band are added in ascending-bin order by the band's thread (deterministic).
// Overlap-add accumulator is a circular double buffer in LDS (no per-hop shifting).
template <int LOG2N, int MODE>
__global__ void __launch_bounds__(FT)
k_afftdn(const float *__restrict__ in, float *__restrict__ out, int64_t n, AfftdnDev d, int frames_per_chunk, int warm_frames,
         int64_t nframes)
{
    constexpr int N = 1 << LOG2N, H = N / 2, HH = H / 2;
    constexpr int QP = (H + FT - 1) / FT;           // complex points per thread (load / overlap-add)
    constexpr int KP = (HH + FT - 1) / FT;          // bin pairs per thread
    constexpr int FSEG = (HH + 63) / 64;            // 64-bin segments per direction
    constexpr int NSEG = 2 * FSEG + 1;
    extern __shared__ unsigned char smem_raw[];
    float2 *bufA = reinterpret_cast<float2 *>(smem_raw);
    float2 *bufB = bufA + H;
    float2 *tw = bufB + H;
    double *acc = reinterpret_cast<double *>(tw + H);
    const int Wp = (d.W + 1) & ~1;
    double *part = acc + Wp;                                  // [NSEG][AF_MAXBANDS]
    double *spread = part + NSEG * AF_MAXBANDS;               // [nb][nb]
    double *band_excit = spread + d.nbands * d.nbands;
    double *prior_band = band_excit + AF_MAXBANDS;
    double *band_rs = prior_band + AF_MAXBANDS;               // 1/sqrt(band_amt)
    double *band_amt = band_rs + AF_MAXBANDS;
    int *seg_bmin = reinterpret_cast<int *>(band_amt + AF_MAXBANDS);   // [NSEG] lowest band id in the segment
    int *band_s0 = seg_bmin + NSEG + 1;                       // first / last segment touching the band
    int *band_s1 = band_s0 + AF_MAXBANDS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int A = d.A, W = d.W, nb = d.nbands;
    const int64_t t_lo = (int64_t)blockIdx.x * frames_per_chunk;
    const int64_t t_hi = min(t_lo + frames_per_chunk, nframes);
    if (t_lo >= nframes) return;
    const int64_t t0 = MODE == 2 ? t_lo : max((int64_t)0, t_lo - warm_frames);       // (the tracker's vote of a frame has no history)

    // ---- per-chunk constants
    for (int q = tid; q < H; q += FT) { double sn, cs; sincospi(2.0 * q / H, &sn, &cs); tw[q] = make_float2((float)cs, (float)-sn); }
    for (int m = tid; m < Wp; m += FT) acc[m] = 0.0;
    for (int i = tid; i < nb * nb; i += FT) spread[i] = d.spread[i];
    if (tid < AF_MAXBANDS) { prior_band[tid] = 0.0; band_s0[tid] = NSEG; band_s1[tid] = -1; }
    // segment ids ascend with bin index: forward rows 0..FSEG-1, the self-paired bin H/2, mirrored rows in reverse
    if (tid < NSEG) {
        int lo;
        if (tid < FSEG) lo = 64 * tid;
        else if (tid == FSEG) lo = HH;
        else { const int j = FSEG - 1 - (tid - FSEG - 1); lo = max(H - 64 * j - 63, HH + 1); }
        seg_bmin[tid] = d.bin2band[lo];
    }
    double win[2 * QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        const int m = tid + q * FT;
        win[2 * q] = (2 * m < W) ? d.window[2 * m] : 0.0;
        win[2 * q + 1] = (2 * m + 1 < W) ? d.window[2 * m + 1] : 0.0;
    }
    // bins owned: pair p -> k = tid + p*FT (bins k and H-k); slot 2*KP = bin H/2 (thread 0 only)
    constexpr int NBIN = 2 * KP + 1;
    double prior[NBIN], av[NBIN], inv_av[NBIN], sqrt_av[NBIN], rel[NBIN];
    int bband[NBIN], bseg[NBIN]; unsigned same[NBIN]; bool head[NBIN], valid[NBIN];
    float2 wk[KP];
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int k = tid + p * FT;
        double sn, cs; sincospi(2.0 * k / N, &sn, &cs); wk[p] = make_float2((float)cs, (float)-sn);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int e = 2 * p + side;
            const int bin = side == 0 ? k : H - k;
            valid[e] = (k < HH) && !(side == 1 && false);
            const int bi = valid[e] ? bin : 0;
            prior[e] = 0.0; av[e] = d.abs_var[bi]; inv_av[e] = 1.0 / av[e]; sqrt_av[e] = sqrt(av[e]);
            rel[e] = MODE == 1 ? d.rel_var[bi] : 0.0;
            bband[e] = valid[e] ? d.bin2band[bi] : -1 - e;
            const int row = (k >> 6);                                      // forward wave-row
            bseg[e] = side == 0 ? row : FSEG + 1 + (FSEG - 1 - row);
        }
    }
    {
        const int e = 2 * KP;
        valid[e] = (tid == 0);
        prior[e] = 0.0; av[e] = d.abs_var[HH]; inv_av[e] = 1.0 / av[e]; sqrt_av[e] = sqrt(av[e]);
        rel[e] = MODE == 1 ? d.rel_var[HH] : 0.0;
        bband[e] = valid[e] ? d.bin2band[HH] : -100;
        bseg[e] = FSEG;
    }
#pragma unroll
    for (int e = 0; e < NBIN; ++e) {
        unsigned msk = 0;
#pragma unroll
        for (int o = 1, bit = 0; o < 64; o <<= 1, ++bit) {
            const int other = __shfl_down(bband[e], o, 64);
            if (lane + o < 64 && other == bband[e] && valid[e]) msk |= 1u << bit;
        }
        same[e] = msk;
        const int prev = __shfl_up(bband[e], 1, 64);
        head[e] = valid[e] && (lane == 0 || prev != bband[e]);
    }
    __syncthreads();
    // which segments touch each band (band_s0/band_s1), from the head lanes
#pragma unroll
    for (int e = 0; e < NBIN; ++e)
        if (head[e]) { atomicMin(&band_s0[bband[e]], bseg[e]); atomicMax(&band_s1[bband[e]], bseg[e]); }
    __syncthreads();
    const double gain_scale = 1.0 / (d.max_gain * d.max_gain);
    const double max_gain = d.max_gain;

    int off = (int)(((t0 % 3) * (int64_t)A) % W);            // circular overlap-add origin: slot of frame sample 0
    // the frame's samples are fetched one frame ahead so the HBM round trip hides behind the previous frame's transforms
    float xr[2 * QP];
    auto fetch = [&](int64_t t) {
        const int64_t start = t * A - (W - A);
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const int m = tid + q * FT;
            const int64_t k0 = start + 2 * m, k1 = k0 + 1;
            xr[2 * q] = (m < H && 2 * m < W && k0 >= 0 && k0 < n) ? in[k0] : 0.f;
            xr[2 * q + 1] = (m < H && 2 * m + 1 < W && k1 >= 0 && k1 < n) ? in[k1] : 0.f;
        }
    };
    fetch(t0);
    for (int64_t t = t0; t < t_hi; ++t) {
        const int64_t start = t * A - (W - A);
        // ---- windowed frame -> packed complex
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const int m = tid + q * FT;
            if (m < H) bufA[m] = make_float2((float)(win[2 * q] * xr[2 * q] * 8388608.0), (float)(win[2 * q + 1] * xr[2 * q + 1] * 8388608.0));
        }
        if (t + 1 < t_hi) fetch(t + 1);
        __syncthreads();
        float2 *Z = fft_stockham<LOG2N - 1, false>(bufA, bufB, tw);
        float2 *Zo = (Z == bufA) ? bufB : bufA;
        // ---- split to real-signal bins, first-stage gains (decision-directed prior), clean power
        const double ratio = (t == 0) ? 1.0 : 0.5, rratio = 1.0 - ratio;
        float2 X[NBIN]; double g1[NBIN], clean[NBIN];
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            const int k = tid + p * FT;
            if (k < HH) {
                const float2 zk = Z[k], zm = Z[(H - k) & (H - 1)];
                const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);          // E = (Zk + conj Zm)/2
                const float orr = 0.5f * (zk.y + zm.y), oi = -0.5f * (zk.x - zm.x);        // O = -i (Zk - conj Zm)/2
                if (k == 0) { X[2 * p] = make_float2(er + orr, 0.f); X[2 * p + 1] = make_float2(er - orr, 0.f); }
                else {
                    const float cr = wk[p].x * orr - wk[p].y * oi, ci = wk[p].x * oi + wk[p].y * orr;   // W^k O
                    X[2 * p] = make_float2(er + cr, ei + ci);
                    X[2 * p + 1] = make_float2(er - cr, -(ei - ci));
                }
            } else { X[2 * p] = make_float2(0.f, 0.f); X[2 * p + 1] = make_float2(0.f, 0.f); }
        }
        { const float2 zk = Z[HH]; X[2 * KP] = make_float2(zk.x, -zk.y); }
        if (MODE == 2) {
            // af_afftdn.c track_noise: spectral_flatness() over the magnitudes above s->floor, floor_offset() over all of them.
            // Block reduction (count, sum of logs, sum, max, min); thread 0 writes the frame's vote.
            double cnt = 0.0, slog = 0.0, ssum = 0.0, mx = 0.0, mn = 1e300;
#pragma unroll
            for (int e = 0; e < NBIN; ++e) {
                if (!valid[e]) continue;
                const double mag = hypot((double)X[e].x, (double)X[e].y);
                if (mag > d.floor) { cnt += 1.0; slog += log(mag); ssum += mag; }
                mx = fmax(mx, mag); mn = fmin(mn, mag);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                cnt += __shfl_down(cnt, o, 64); slog += __shfl_down(slog, o, 64); ssum += __shfl_down(ssum, o, 64);
                mx = fmax(mx, __shfl_down(mx, o, 64)); mn = fmin(mn, __shfl_down(mn, o, 64));
            }
            __syncthreads();                                     // (part[] is free: nothing of this frame has used it)
            if (lane == 0) { double *pp = part + (tid >> 6) * 5; pp[0] = cnt; pp[1] = slog; pp[2] = ssum; pp[3] = mx; pp[4] = mn; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < FT / 64; ++w) { const double *pp = part + w * 5; cnt += pp[0]; slog += pp[1]; ssum += pp[2]; mx = fmax(mx, pp[3]); mn = fmin(mn, pp[4]); }
                const double size = fmax(cnt, 1.0);
                const double num = exp(slog / size), den = ssum / size;
                double vote = NAN;
                if (num / den > 0.8) {
                    const double offset = fmax(fabs(mx - den), fabs(mn - den)) / den;        // floor_offset option fo = 1.0
                    vote = fmin(fmax(10.0 * log10(den) - 100.0 + offset, -90.0), -20.0);
                }
                d.track_out[t] = vote;
            }
            __syncthreads();
            continue;
        }
        double mv_post = 0.0;
        if (MODE == 1) {
            // variances of this frame: the first-stage gains still see the floor as the previous frame left it, the masking
            // limits the one this frame's vote produced (set_parameters() runs between the two loops of process_frame())
            const double mv_pre = d.mvseq[t]; mv_post = d.mvseq[t + 1];
#pragma unroll
            for (int e = 0; e < NBIN; ++e) inv_av[e] = 1.0 / fmax(mv_pre * rel[e], 1.0);
        }
#pragma unroll
        for (int e = 0; e < NBIN; ++e) {
            const double xr = (double)X[e].x, xi = (double)X[e].y;
            const double power = fma(xr, xr, xi * xi);
            const double mav = power * inv_av[e];
            const double nmav = ratio * prior[e] + rratio * fmax(mav - 1.0, 0.0);
            const double ng = nmav * fast_rcp(1.0 + nmav);
            const double sq = ng * ng;
            if (valid[e]) prior[e] = mav * sq;
            clean[e] = valid[e] ? power * sq : 0.0;
            g1[e] = ng;
        }
        // ---- band sums of the clean power: segmented wave reduction, partials per (segment, band)
#pragma unroll
        for (int e = 0; e < NBIN; ++e) {
            if (e == 2 * KP) { if (valid[e]) part[bseg[e] * AF_MAXBANDS + 0] = clean[e]; continue; }
            double v = clean[e];
#pragma unroll
            for (int o = 1, bit = 0; o < 64; o <<= 1, ++bit) {
                const double other = __shfl_down(v, o, 64);
                if (same[e] & (1u << bit)) v += other;
            }
            if (head[e]) part[bseg[e] * AF_MAXBANDS + (bband[e] - seg_bmin[bseg[e]])] = v;
        }
        __syncthreads();
        if (tid < nb) {
            double e = 0.0;
            for (int sg = band_s0[tid]; sg <= band_s1[tid]; ++sg) e += part[sg * AF_MAXBANDS + (tid - seg_bmin[sg])];
            e = fmax(e, d.alpha[tid] * e + d.beta[tid] * prior_band[tid]);
            prior_band[tid] = e;
            band_excit[tid] = e;
        }
        __syncthreads();
        // masking amounts: spread (nb x nb) times the band excitations, 8 lanes per band + a 3-step shuffle reduction
        for (int b = tid >> 3; b < nb; b += FT / 8) {
            double a = 0.0;
            const double *sp = spread + b * nb;
            for (int k = tid & 7; k < nb; k += 8) a += sp[k] * band_excit[k];
            a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
            if ((tid & 7) == 0) { band_amt[b] = a; band_rs[b] = 1.0 / sqrt(a); }
        }
        __syncthreads();
        const bool need_out = (t + 2 >= t_lo);     // frames whose overlap-add reaches the emitted range
        if (need_out) {
            // ---- masking-limited gains, then fold the pair back into the packed inverse transform input
#pragma unroll
            for (int e = 0; e < NBIN; ++e) {
                if (!valid[e]) continue;
                if (MODE == 1) { av[e] = fmax(mv_post * rel[e], 1.0); sqrt_av[e] = sqrt(av[e]); }
                const double amt = band_amt[bband[e]];
                double g = g1[e];
                double b;
                if (amt > av[e]) g = 1.0;
                else {
                    b = (amt > gain_scale * av[e]) ? sqrt_av[e] * band_rs[bband[e]] : max_gain;
                    if (g < 1.0) g = (b * g - 2.0 * g + 1.0) * fast_rcp(b - g);
                    else if (g > 1.0) g = (b * g - 1.0) * fast_rcp(b + g - 2.0);
                }
                const float gf = (float)g;
                X[e].x *= gf; X[e].y *= gf;
            }
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                const int k = tid + p * FT;
                if (k < HH) {
                    const float2 yk = X[2 * p], ym = X[2 * p + 1];
                    if (k == 0) Zo[0] = make_float2(yk.x + ym.x, yk.x - ym.x);
                    else {
                        const float ar = yk.x + ym.x, ai = yk.y - ym.y;           // A = Yk + conj Ym
                        const float br = yk.x - ym.x, bi = yk.y + ym.y;           // B = Yk - conj Ym
                        const float cr = wk[p].x * br + wk[p].y * bi, ci = wk[p].x * bi - wk[p].y * br;   // C = conj(W) B
                        Zo[k] = make_float2(ar - ci, ai + cr);                    // A + iC
                        Zo[H - k] = make_float2(ar + ci, -(ai - cr));             // conj(A - iC)
                    }
                }
            }
            if (tid == 0) { const float2 y = X[2 * KP]; Zo[HH] = make_float2(2.f * y.x, -2.f * y.y); }
            __syncthreads();
            float2 *zt = fft_stockham<LOG2N - 1, true>(Zo, Z, tw);
            // ---- overlap-add into the circular accumulator, emit the first hop, clear it
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                const int m = tid + q * FT;
                if (m < H) {
                    const float2 v = zt[m];
                    const int m0 = 2 * m, m1 = m0 + 1;
                    if (m0 < W) { int sl = m0 + off; if (sl >= W) sl -= W; acc[sl] += win[2 * q] * (double)v.x / 8388608.0; }
                    if (m1 < W) { int sl = m1 + off; if (sl >= W) sl -= W; acc[sl] += win[2 * q + 1] * (double)v.y / 8388608.0; }
                }
            }
            __syncthreads();
            for (int m = tid; m < A; m += FT) {
                int sl = m + off; if (sl >= W) sl -= W;
                const int64_t k = start + m;
                if (t >= t_lo && k >= 0 && k < n) out[k] = (float)acc[sl];
                acc[sl] = 0.0;
            }
        }
        off += A; if (off >= W) off -= W;
        __syncthreads();
    }
}

// af_afftdn.c process_frame() for G frames at a time (G waves per workgroup).  What couples the frames of a chunk is small: the
// decision-directed prior of each bin (its own history only), the band excitations (nb values) and the overlap-add.  So wave w runs
// both transforms of frame t + w by itself (fft_wave: no workgroup barrier inside), and the threads walk the G frames in order only
// for the per-bin gains -- 7 workgroup barriers per G frames instead of 17 per frame.  The overlap-add accumulator is G*64 threads x
// NSL slots held in REGISTERS (a thread owns its circular slots for the whole chunk and adds the frames in order).  Every sum is
// taken in k_afftdn's order: the two kernels give bit-identical output.
#ifdef JT_AF_PROFILE
__device__ unsigned long long af_prof[16];
#define AF_MARK(i) do { const unsigned long long c_ = __builtin_readcyclecounter(); pc[i] += c_ - plast; plast = c_; } while (0)
#else
#define AF_MARK(i) do {} while (0)
#endif
template <int LOG2N, int MODE, int G>
__global__ void __launch_bounds__(64 * G)
k_afftdn_grp(const float *__restrict__ in, float *__restrict__ out, int64_t n, AfftdnDev d, int frames_per_chunk, int warm_frames,
             int64_t nframes)
{
    constexpr int N = 1 << LOG2N, H = N / 2, HH = H / 2, NT = 64 * G;
    constexpr int KP = HH / NT;                     // bin pairs per thread: pair p = bins k and H - k, k = tid + NT p
    static_assert(KP * NT == HH && NT % 16 == 0, "every thread owns KP whole pairs");
    constexpr int NB2 = 2 * KP;                     // paired bins per thread; bin H/2 pairs with itself and belongs to thread 0 (hh[])
    constexpr int FSEG = HH / 64;                   // 64-bin segments per direction
    constexpr int NSEG = 2 * FSEG + 1;
    constexpr int NSL = (N + NT - 1) / NT;          // accumulator slots per thread (W <= N)
    constexpr int PW = H / 64;                      // packed points per lane of a frame
    extern __shared__ unsigned char smem_raw[];
    using Lp = FwLayout<LOG2N - 1>;
    constexpr int ZS = Lp::SKH;                                 // float2 per frame buffer (padded layout, fw_sk)
    constexpr int ZP = NT + NT / 4;                             // fw_sk(i + NT) - fw_sk(i)
    float2 *zb = reinterpret_cast<float2 *>(smem_raw);          // [G][ZS]
    float2 *tw = zb + G * ZS;                                   // per-stage twiddle rows [Lp::TWN (+1)]
    const int nb = d.nbands, pst = (d.seg_span > 0 ? d.seg_span : nb) | 1;          // (a segment's bands are consecutive)
    double *part = reinterpret_cast<double *>(tw + ((Lp::TWN + 1) & ~1));          // [G][NSEG][pst]   (MODE 2: [G][G][5], G <= 8 < NSEG)
    double *spread = part + G * NSEG * pst;                     // [nb][nb]
    double *band_excit = spread + nb * nb;                      // [G][AF_MAXBANDS]
    double *band_amt = band_excit + G * AF_MAXBANDS;
    double *band_rs = band_amt + G * AF_MAXBANDS;               // 1/sqrt(band_amt)
    double *hh = band_rs + G * AF_MAXBANDS;                     // bin H/2: prior, abs_var, 1/abs_var, sqrt, rel_var, then g1 of the G frames
    double *winl = hh + 6 + G + ((6 + G) & 1);                  // the analysis / synthesis window [(W + 1) & ~1]
    int *seg_bmin = reinterpret_cast<int *>(winl + ((d.W + 1) & ~1));     // [NSEG] lowest band id in the segment
    int *band_s0 = seg_bmin + NSEG + 1;                         // first / last segment touching the band
    int *band_s1 = band_s0 + AF_MAXBANDS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A = d.A, W = d.W;
    const int64_t t_lo = (int64_t)blockIdx.x * frames_per_chunk;
    const int64_t t_hi = min(t_lo + frames_per_chunk, nframes);
    if (t_lo >= nframes) return;
    const int64_t t0 = MODE == 2 ? t_lo : max((int64_t)0, t_lo - warm_frames);

    fw_tw_fill<LOG2N - 1>(tw, tid, NT);
    for (int i = tid; i < nb * nb; i += NT) spread[i] = d.spread[i];
    for (int i = tid; i < ((d.W + 1) & ~1); i += NT) winl[i] = i < d.W ? d.window[i] : 0.0;
    if (tid < AF_MAXBANDS) { band_s0[tid] = NSEG; band_s1[tid] = -1; }
    if (tid < NSEG) {
        int lo;
        if (tid < FSEG) lo = 64 * tid;
        else if (tid == FSEG) lo = HH;
        else { const int j = FSEG - 1 - (tid - FSEG - 1); lo = max(H - 64 * j - 63, HH + 1); }
        seg_bmin[tid] = d.bin2band[lo];
    }
    const int band_hh = d.bin2band[HH];
    if (tid == 0) {
        const double a = d.abs_var[HH];
        hh[0] = 0.0; hh[1] = a; hh[2] = 1.0 / a; hh[3] = sqrt(a); hh[4] = MODE == 1 ? d.rel_var[HH] : 0.0;
    }
    double prior[NB2], av[NB2], inv_av[NB2], sqrt_av[NB2], rel[NB2];
    int bband[NB2], pidx[NB2]; unsigned same[NB2]; bool head[NB2];
    float2 wk[KP];
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int k = tid + p * NT;
        double sn, cs; sincospi(2.0 * k / N, &sn, &cs); wk[p] = make_float2((float)cs, (float)-sn);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int e = 2 * p + side;
            const int bin = side == 0 ? k : H - k;
            prior[e] = 0.0; av[e] = d.abs_var[bin]; inv_av[e] = 1.0 / av[e]; sqrt_av[e] = sqrt(av[e]);
            rel[e] = MODE == 1 ? d.rel_var[bin] : 0.0;
            bband[e] = d.bin2band[bin];
        }
    }
#pragma unroll
    for (int e = 0; e < NB2; ++e) {
        unsigned msk = 0;
#pragma unroll
        for (int o = 1, bit = 0; o < 64; o <<= 1, ++bit) {
            const int other = __shfl_down(bband[e], o, 64);
            if (lane + o < 64 && other == bband[e]) msk |= 1u << bit;
        }
        same[e] = msk;
        const int prev = __shfl_up(bband[e], 1, 64);
        head[e] = lane == 0 || prev != bband[e];
    }
    __syncthreads();
    // which segments touch each band (band_s0/band_s1) from the head lanes; where a head lane puts its (segment, band) partial
#pragma unroll
    for (int e = 0; e < NB2; ++e) {
        const int row = wave + (NT / 64) * (e >> 1);                                  // forward wave-row of bin k (wave-uniform)
        const int sg = (e & 1) == 0 ? row : FSEG + 1 + (FSEG - 1 - row);
        pidx[e] = sg * pst + (bband[e] - seg_bmin[sg]);
        if (head[e]) { atomicMin(&band_s0[bband[e]], sg); atomicMax(&band_s1[bband[e]], sg); }
    }
    if (tid == 0) { atomicMin(&band_s0[band_hh], FSEG); atomicMax(&band_s1[band_hh], FSEG); }
    __syncthreads();
    const double gain_scale = 1.0 / (d.max_gain * d.max_gain);
    const double max_gain = d.max_gain;
    double prior_band = 0.0;                                    // of band tid / G (threads below G nb): the last frame's excitation
    double accr[NSL];
#pragma unroll
    for (int i = 0; i < NSL; ++i) accr[i] = 0.0;

    // padded positions (fw_sk) of what a thread touches, as a base plus compile-time offsets: point lane + 64 q (+ 80 q); bin
    // tid + NT p (+ ZP p) and its mirror H - tid - NT p (- ZP p; bin 0 pairs with itself)
    const int zl = fw_sk(lane), zk0 = fw_sk(tid), zm0 = fw_sk(H - tid);
    constexpr int ZHH = HH + (HH >> 2);                         // fw_sk(H/2)
    // the paired real-signal bins a thread owns, from the packed transform of a frame
    auto split = [&](const float2 *Z, float2 (&X)[NB2]) {
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            const float2 zk = Z[zk0 + ZP * p], zm = Z[(p == 0 && tid == 0) ? 0 : zm0 - ZP * p];
            const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);          // E = (Zk + conj Zm)/2
            const float orr = 0.5f * (zk.y + zm.y), oi = -0.5f * (zk.x - zm.x);        // O = -i (Zk - conj Zm)/2
            if (p == 0 && tid == 0) { X[0] = make_float2(er + orr, 0.f); X[1] = make_float2(er - orr, 0.f); }
            else {
                const float cr = wk[p].x * orr - wk[p].y * oi, ci = wk[p].x * oi + wk[p].y * orr;   // W^k O
                X[2 * p] = make_float2(er + cr, ei + ci);
                X[2 * p + 1] = make_float2(er - cr, -(ei - ci));
            }
        }
    };

    float xr[2 * PW];
    auto fetch = [&](int64_t t) {                               // (clamped addresses: all 2 PW loads are in flight at once)
        const int64_t start = t * A - (W - A);
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int m = lane + 64 * q;
            const int64_t k0 = start + 2 * m, k1 = k0 + 1;
            const float v0 = in[min(max(k0, (int64_t)0), n - 1)], v1 = in[min(max(k1, (int64_t)0), n - 1)];
            xr[2 * q] = (2 * m < W && k0 >= 0 && k0 < n) ? v0 : 0.f;
            xr[2 * q + 1] = (2 * m + 1 < W && k1 >= 0 && k1 < n) ? v1 : 0.f;
        }
    };
#ifdef JT_AF_PROFILE
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = __builtin_readcyclecounter();
#endif
    for (int64_t tg = t0; tg < t_hi; tg += G) {
        AF_MARK(7);
        // ---- wave w: windowed frame tg + w -> packed complex -> forward transform
        const int64_t tf = tg + wave;
        float2 *zw = zb + wave * ZS;
        if (tf < t_hi) {
            fetch(tf);
#pragma unroll
            for (int q = 0; q < PW; ++q) {
                const int m = lane + 64 * q;
                const double w0 = (2 * m < W) ? winl[2 * m] : 0.0, w1 = (2 * m + 1 < W) ? winl[2 * m + 1] : 0.0;
                zw[zl + 80 * q] = make_float2((float)(w0 * xr[2 * q] * 8388608.0), (float)(w1 * xr[2 * q + 1] * 8388608.0));
            }
            AF_MARK(0);
            fft_wave<LOG2N - 1, false>(zw, tw, lane);
        }
        AF_MARK(1);
        __syncthreads();
        AF_MARK(6);
        if (MODE == 2) {
            // af_afftdn.c track_noise: spectral_flatness() over the magnitudes above s->floor, floor_offset() over all of them
#pragma unroll
            for (int f = 0; f < G; ++f) {
                if (tg + f >= t_hi) continue;
                const float2 *Z = zb + f * ZS;
                float2 X[NB2]; split(Z, X);
                double cnt = 0.0, slog = 0.0, ssum = 0.0, mx = 0.0, mn = 1e300;
                auto take = [&](float2 x) {
                    const double mag = hypot((double)x.x, (double)x.y);
                    if (mag > d.floor) { cnt += 1.0; slog += log(mag); ssum += mag; }
                    mx = fmax(mx, mag); mn = fmin(mn, mag);
                };
#pragma unroll
                for (int e = 0; e < NB2; ++e) take(X[e]);
                if (wave == 0) { if (tid == 0) { const float2 zk = Z[ZHH]; take(make_float2(zk.x, -zk.y)); } }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    cnt += __shfl_down(cnt, o, 64); slog += __shfl_down(slog, o, 64); ssum += __shfl_down(ssum, o, 64);
                    mx = fmax(mx, __shfl_down(mx, o, 64)); mn = fmin(mn, __shfl_down(mn, o, 64));
                }
                if (lane == 0) { double *pp = part + (f * G + wave) * 5; pp[0] = cnt; pp[1] = slog; pp[2] = ssum; pp[3] = mx; pp[4] = mn; }
            }
            __syncthreads();
            if (tid < G && tg + tid < t_hi) {
                const double *pp = part + (tid * G) * 5;
                double cnt = pp[0], slog = pp[1], ssum = pp[2], mx = pp[3], mn = pp[4];
                for (int w = 1; w < G; ++w) { const double *pw = pp + w * 5; cnt += pw[0]; slog += pw[1]; ssum += pw[2]; mx = fmax(mx, pw[3]); mn = fmin(mn, pw[4]); }
                const double size = fmax(cnt, 1.0);
                const double num = exp(slog / size), den = ssum / size;
                double vote = NAN;
                if (num / den > 0.8) {
                    const double offset = fmax(fabs(mx - den), fabs(mn - den)) / den;        // floor_offset option fo = 1.0
                    vote = fmin(fmax(10.0 * log10(den) - 100.0 + offset, -90.0), -20.0);
                }
                d.track_out[tg + tid] = vote;
            }
            __syncthreads();
            continue;
        }
        // ---- first-stage gains of the G frames in order (decision-directed prior), band partials of the clean power
        double g1[G][NB2], clean[G][NB2];
#pragma unroll
        for (int f = 0; f < G; ++f) {
            const int64_t t = tg + f;
            if (t >= t_hi) {
#pragma unroll
                for (int e = 0; e < NB2; ++e) { clean[f][e] = 0.0; g1[f][e] = 0.0; }
                continue;
            }
            const float2 *Z = zb + f * ZS;
            float2 X[NB2]; split(Z, X);
            const double ratio = (t == 0) ? 1.0 : 0.5, rratio = 1.0 - ratio;
            double mv_pre = 0.0;
            if (MODE == 1) {
                // variances of this frame: the first-stage gains still see the floor as the previous frame left it, the masking
                // limits the one this frame's vote produced (set_parameters() runs between the two loops of process_frame())
                mv_pre = d.mvseq[t];
#pragma unroll
                for (int e = 0; e < NB2; ++e) inv_av[e] = 1.0 / fmax(mv_pre * rel[e], 1.0);
            }
#pragma unroll
            for (int e = 0; e < NB2; ++e) {
                const double xr_ = (double)X[e].x, xi = (double)X[e].y;
                const double power = fma(xr_, xr_, xi * xi);
                const double mav = power * inv_av[e];
                const double nmav = ratio * prior[e] + rratio * fmax(mav - 1.0, 0.0);
                const double ng = nmav * fast_rcp(1.0 + nmav);
                const double sq = ng * ng;
                prior[e] = mav * sq;
                clean[f][e] = power * sq;
                g1[f][e] = ng;
            }
        }
        if (wave == 0) {
            if (tid == 0) {
                // bin H/2 of the G frames: what does not depend on the previous frame first (G independent sequences), then the
                // decision-directed chain in registers
                double pw[G], mavs[G], bterm[G];
#pragma unroll
                for (int f = 0; f < G; ++f) {
                    const int64_t t = tg + f;
                    const float2 zk = zb[f * ZS + ZHH];
                    const double xr_ = (double)zk.x, xi = (double)-zk.y;
                    pw[f] = fma(xr_, xr_, xi * xi);
                    double inv = hh[2];
                    if (MODE == 1) inv = 1.0 / fmax(d.mvseq[min(t, t_hi - 1)] * hh[4], 1.0);
                    mavs[f] = pw[f] * inv;
                    bterm[f] = ((t == 0) ? 0.0 : 0.5) * fmax(mavs[f] - 1.0, 0.0);
                }
                double pr = hh[0];
#pragma unroll
                for (int f = 0; f < G; ++f) {
                    const int64_t t = tg + f;
                    if (t >= t_hi) continue;
                    const double nmav = ((t == 0) ? 1.0 : 0.5) * pr + bterm[f];
                    const double ng = nmav * fast_rcp(1.0 + nmav);
                    const double sq = ng * ng;
                    pr = mavs[f] * sq;
                    part[f * NSEG * pst + FSEG * pst] = pw[f] * sq;
                    hh[6 + f] = ng;
                }
                hh[0] = pr;
            }
        }
        // band sums of the clean power: segmented wave reduction, partials per (segment, band); the 2 KP G chains of a thread are
        // independent, so their shuffle latencies overlap
#pragma unroll
        for (int o = 1, bit = 0; o < 64; o <<= 1, ++bit) {
#pragma unroll
            for (int f = 0; f < G; ++f) {
#pragma unroll
                for (int e = 0; e < NB2; ++e) {
                    const double other = __shfl_down(clean[f][e], o, 64);
                    if (same[e] & (1u << bit)) clean[f][e] += other;
                }
            }
        }
#pragma unroll
        for (int f = 0; f < G; ++f) {
            if (tg + f >= t_hi) continue;
#pragma unroll
            for (int e = 0; e < NB2; ++e) if (head[e]) part[f * NSEG * pst + pidx[e]] = clean[f][e];
        }
        AF_MARK(2);
        __syncthreads();
        AF_MARK(6);
        if (tid < G * nb) {
            // excitation of band b in frame f: G adjacent lanes per band; the partials are added in ascending-bin order, then the
            // decay recurrence e = max(raw, alpha raw + beta e_prev) walks the G lanes
            const int b = tid / G, f = tid % G;
            const int nvalid = (int)min((int64_t)G, t_hi - tg);
            const double *pf = part + f * NSEG * pst;
            double raw = 0.0;
            if (f < nvalid) for (int sg = band_s0[b]; sg <= band_s1[b]; ++sg) raw += pf[sg * pst + (b - seg_bmin[sg])];
            const double al = d.alpha[b], be = d.beta[b];
            double e = prior_band;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const double prev = __shfl_up(e, 1, G);                             // lane f - 1 of the band's group (lane 0: its own)
                if (f == i && i < nvalid) e = fmax(raw, al * raw + be * (i == 0 ? prior_band : prev));
            }
            if (f < nvalid) band_excit[f * AF_MAXBANDS + b] = e;
            prior_band = __shfl(e, nvalid - 1, G);                                  // every lane of the group keeps the last frame's value
        }
        __syncthreads();
        // masking amounts: spread (nb x nb) times the band excitations, 8 lanes per (frame, band) + a 3-step shuffle reduction
        for (int idx = tid >> 3; idx < G * nb; idx += NT / 8) {
            const int f = idx / nb, b = idx - f * nb;
            double a = 0.0;
            const double *sp = spread + b * nb, *ex = band_excit + f * AF_MAXBANDS;
            for (int k = tid & 7; k < nb; k += 8) a += sp[k] * ex[k];
            a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
            if ((tid & 7) == 0) { band_amt[f * AF_MAXBANDS + b] = a; band_rs[f * AF_MAXBANDS + b] = 1.0 / sqrt(a); }
        }
        __syncthreads();
        AF_MARK(3);
        // ---- masking-limited gains, the pair folded back into the packed inverse-transform input (in place: a thread owns its slots)
        auto limit = [&](double g, double amt, double rs, double ave, double sqe) {
            if (amt > ave) return 1.0;
            const double b = (amt > gain_scale * ave) ? sqe * rs : max_gain;
            if (g < 1.0) g = (b * g - 2.0 * g + 1.0) * fast_rcp(b - g);
            else if (g > 1.0) g = (b * g - 1.0) * fast_rcp(b + g - 2.0);
            return g;
        };
#pragma unroll
        for (int f = 0; f < G; ++f) {
            const int64_t t = tg + f;
            if (t >= t_hi || t + 2 < t_lo) continue;            // (frames whose overlap-add never reaches the emitted range)
            float2 *Z = zb + f * ZS;
            float2 X[NB2]; split(Z, X);
            double mv_post = 0.0;
            if (MODE == 1) mv_post = d.mvseq[t + 1];
            const double *amtf = band_amt + f * AF_MAXBANDS, *rsf = band_rs + f * AF_MAXBANDS;
#pragma unroll
            for (int e = 0; e < NB2; ++e) {
                double ave = av[e], sqe = sqrt_av[e];
                if (MODE == 1) { ave = fmax(mv_post * rel[e], 1.0); sqe = sqrt(ave); }
                const float gf = (float)limit(g1[f][e], amtf[bband[e]], rsf[bband[e]], ave, sqe);
                X[e].x *= gf; X[e].y *= gf;
            }
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                const float2 yk = X[2 * p], ym = X[2 * p + 1];
                if (p == 0 && tid == 0) Z[0] = make_float2(yk.x + ym.x, yk.x - ym.x);
                else {
                    const float ar = yk.x + ym.x, ai = yk.y - ym.y;           // A = Yk + conj Ym
                    const float br = yk.x - ym.x, bi = yk.y + ym.y;           // B = Yk - conj Ym
                    const float cr = wk[p].x * br + wk[p].y * bi, ci = wk[p].x * bi - wk[p].y * br;   // C = conj(W) B
                    Z[zk0 + ZP * p] = make_float2(ar - ci, ai + cr);          // A + iC
                    Z[zm0 - ZP * p] = make_float2(ar + ci, -(ai - cr));       // conj(A - iC)
                }
            }
        }
        if (wave == 0) {
            const int64_t t = tg + lane;
            if (lane < G && t < t_hi && t + 2 >= t_lo) {                      // bin H/2: lane f takes frame f
                double ave = hh[1], sqe = hh[3];
                if (MODE == 1) { ave = fmax(d.mvseq[t + 1] * hh[4], 1.0); sqe = sqrt(ave); }
                const float gf = (float)limit(hh[6 + lane], band_amt[lane * AF_MAXBANDS + band_hh], band_rs[lane * AF_MAXBANDS + band_hh], ave, sqe);
                float2 *Z = zb + lane * ZS;
                const float2 zk = Z[ZHH];
                const float2 y = make_float2(zk.x * gf, -zk.y * gf);
                Z[ZHH] = make_float2(2.f * y.x, -2.f * y.y);
            }
        }
        AF_MARK(4);
        __syncthreads();
        AF_MARK(6);
        if (tf < t_hi && tf + 2 >= t_lo) fft_wave<LOG2N - 1, true>(zw, tw, lane);
        AF_MARK(1);
        __syncthreads();
        AF_MARK(6);
        // ---- overlap-add, frame after frame, into the slots this thread owns; a frame's first hop is emitted and cleared
#pragma unroll
        for (int f = 0; f < G; ++f) {
            const int64_t t = tg + f;
            if (t >= t_hi || t + 2 < t_lo) continue;
            const float2 *zt = zb + f * ZS;
            const int off = (int)((t * A) % W);                  // circular origin: slot of the frame's sample 0
            const int64_t start = t * A - (W - A);
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int sl = tid + NT * i;
                if (sl < W) {
                    int m0 = sl - off; if (m0 < 0) m0 += W;
                    const float2 v = zt[fw_sk(m0 >> 1)];
                    accr[i] += winl[m0] * (double)((m0 & 1) ? v.y : v.x) / 8388608.0;
                    if (m0 < A) {
                        const int64_t k = start + m0;
                        if (t >= t_lo && k >= 0 && k < n) out[k] = (float)accr[i];
                        accr[i] = 0.0;
                    }
                }
            }
        }
        AF_MARK(5);
        __syncthreads();
        AF_MARK(6);
    }
#ifdef JT_AF_PROFILE
    if (tid == JT_AF_PROFILE * 64) for (int i = 0; i < 8; ++i) atomicAdd(&af_prof[i], pc[i]);
#endif
}

int64_t jt_afftdn_nframes(int64_t n, int A, int W) { return (n + A - 1) / A + (W - A) / A; }

template <int LOG2N>
static void launch_afftdn_n(unsigned grid, size_t smem, int mode, const float *in, float *out, int64_t n, const AfftdnDev &d, int frames_per_chunk,
                            int warm_frames, int64_t nframes, hipStream_t s)
{
#define AF_GO(M) do { JT_HIP(hipFuncSetAttribute((const void *)k_afftdn<LOG2N, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((k_afftdn<LOG2N, M>), dim3(grid), dim3(FT), smem, s, in, out, n, d, frames_per_chunk, warm_frames, nframes); } while (0)
    if (mode == 1) AF_GO(1); else if (mode == 2) AF_GO(2); else AF_GO(0);
#undef AF_GO
}

void launch_afftdn(const float *in, float *out, int64_t n, const AfftdnDev &d, int frames_per_chunk, int warm_frames, hipStream_t s, const JtOpts &o, int mode)
{
    if (n <= 0) return;
    JT_REQUIRE(d.nbands <= AF_MAXBANDS, JT_E_UNSUPPORTED, "afftdn: too many bark bands");
    int64_t nframes = jt_afftdn_nframes(n, d.A, d.W);
    const bool auto_chunk = frames_per_chunk <= 0;
    if (auto_chunk) {
        // one resident round of workgroups: the kernel's register footprint admits 2 workgroups per CU (256 CUs), so 512 chunks;
        // never shorter than 128 frames (warm-up overhead) nor longer than 1024
        frames_per_chunk = (int)std::min<int64_t>(1024, std::max<int64_t>(128, (nframes + 511) / 512));
    }
    unsigned grid = (unsigned)((nframes + frames_per_chunk - 1) / frames_per_chunk);
    if ((d.L == 2048 || d.L == 4096) && d.W <= d.L && !JT_AB_ON(o.afftdn_old)) {
        // several frames at a time, one wave per frame's transforms (k_afftdn_grp): one workgroup per CU (132 KB of LDS with eight
        // 2048-point frames, 150 KB with four 4096-point frames), so one resident round is 256 chunks
        const bool big = d.L == 4096;
        const int G = big ? 4 : 8, Hh = d.L / 2, NSEGh = 2 * (Hh / 2 / 64) + 1;
        if (auto_chunk) frames_per_chunk = (int)std::min<int64_t>(4096, std::max<int64_t>(128, (nframes + 255) / 256));
        grid = (unsigned)((nframes + frames_per_chunk - 1) / frames_per_chunk);
        const int pst = (d.seg_span > 0 ? d.seg_span : d.nbands) | 1;
        const size_t skh = big ? FwLayout<11>::SKH : FwLayout<10>::SKH, twn = big ? FwLayout<11>::TWN : FwLayout<10>::TWN;
        const size_t smem = sizeof(float2) * (G * skh + ((twn + 1) & ~(size_t)1))
                          + sizeof(double) * ((size_t)G * NSEGh * pst + (size_t)d.nbands * d.nbands + 3 * G * AF_MAXBANDS + (6 + G + ((6 + G) & 1)) + ((d.W + 1) & ~1))
                          + sizeof(int) * (NSEGh + 1 + 2 * AF_MAXBANDS);
        JT_REQUIRE(smem <= 160 * 1024, JT_E_UNSUPPORTED, "afftdn: window too long for this build");
#define AF_GRP(LG, M, GG) do { JT_HIP(hipFuncSetAttribute((const void *)k_afftdn_grp<LG, M, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((k_afftdn_grp<LG, M, GG>), dim3(grid), dim3(64 * GG), smem, s, in, out, n, d, frames_per_chunk, warm_frames, nframes); } while (0)
        if (big) { if (mode == 1) AF_GRP(12, 1, 4); else if (mode == 2) AF_GRP(12, 2, 4); else AF_GRP(12, 0, 4); }
        else { if (mode == 1) AF_GRP(11, 1, 8); else if (mode == 2) AF_GRP(11, 2, 8); else AF_GRP(11, 0, 8); }
#undef AF_GRP
#ifdef JT_AF_PROFILE
        { unsigned long long pr[16]; JT_HIP(hipStreamSynchronize(s)); JT_HIP(hipMemcpyFromSymbol(pr, HIP_SYMBOL(af_prof), sizeof pr)); unsigned long long z[16] = {0};
          JT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(af_prof), z, sizeof z));
          fprintf(stderr, "afftdn_grp mode %d clocks (load, fft, stage1, band, final, ola, barrier, loop):", mode); for (int i = 0; i < 8; ++i) fprintf(stderr, " %llu", pr[i] / grid); fprintf(stderr, "\n"); }
#endif
        return;
    }
    const size_t H = d.L / 2, nseg = 2 * ((H / 2 + 63) / 64) + 1;
    size_t smem = sizeof(float2) * 3 * H + sizeof(double) * (((d.W + 1) & ~1) + nseg * AF_MAXBANDS + (size_t)d.nbands * d.nbands + 4 * AF_MAXBANDS)
                + sizeof(int) * (nseg + 1 + 2 * AF_MAXBANDS);
    JT_REQUIRE(smem <= 160 * 1024, JT_E_UNSUPPORTED, "afftdn: window too long for this build");
    // (the frame-at-a-time kernel serves the 1024-point window; its 2048 / 4096-point instances are superseded by k_afftdn_grp: JT_AB build)
#ifdef JT_AB
    if (d.L == 2048) launch_afftdn_n<11>(grid, smem, mode, in, out, n, d, frames_per_chunk, warm_frames, nframes, s);
    else if (d.L == 4096) launch_afftdn_n<12>(grid, smem, mode, in, out, n, d, frames_per_chunk, warm_frames, nframes, s);
    else
#endif
    if (d.L == 1024) launch_afftdn_n<10>(grid, smem, mode, in, out, n, d, frames_per_chunk, warm_frames, nframes, s);
    else throw JtError{JT_E_UNSUPPORTED, "afftdn: unsupported FFT length"};
}
