// k_loudnorm.hip — af_loudnorm.c's DYNAMIC mode for one channel at its internal 192 kHz (libavfilter/af_loudnorm.c; the fallback the
// reference only notices from the stats: normalise.go:687-693; spec built at normalise.go:1269-1291, followed by aresample back to the
// source rate :1293-1310).  Linear mode — every normal file — is a single gain and never comes here.
//
// The filter is one long state machine: per 100 ms frame a gain from a 21-tap gaussian over the 30 most recent frame gains (each from
// the INPUT's short-term / integrated loudness), applied with a ramp into a 210 ms ring, and a look-ahead true-peak limiter
// (OUT / ATTACK / SUSTAIN / RELEASE) that edits that ring in place 10 ms ahead of the read position.  What is parallel in it:
//   * the input meter (K-weighting + 100 ms energies) does not depend on the gains: the existing K-weighting kernels measure the
//     whole stream first and the host turns the energies into the per-frame short-term / integrated / relative-threshold series the
//     frame loop consumes;
//   * inside a frame, the ring fill (19 200 products), every envelope segment (attack / sustain / release multiply a contiguous ring
//     range by a closed-form envelope) and the clamp-and-copy to the output are elementwise: 64 lanes;
//   * the peak detector only has work where |sample| exceeds the ceiling: a 64-lane max over the scan range rejects a frame without
//     one; otherwise candidates are located 64 at a time and only the filter's order-dependent detail (prev_smp is NOT updated after a
//     rejected candidate) is walked serially, for the one or two samples behind such a candidate.
// The only feedback from the output is the "not yet above threshold" phase of a file that starts quietly, where the gain ramps by
// 1.0058 per frame until the OUTPUT's short-term loudness reaches the target: during it one lane K-weights the frame just produced.
// One wave runs the whole stream: a few tens of microseconds per quiet frame, up to ~1 ms for a frame the limiter works on.  That is
// slow next to the rest of the path and irrelevant next to never delivering the file.
#include "jt_internal.h"

namespace {
constexpr int LN_F100 = 19200, LN_F3000 = 576000, LN_LBS = 40320, LN_ATT = 1920, LN_REL = 19200;      // frame_size(192000, 100 / 3000 / 210 / 10 / 100)
enum { LIM_OUT = 0, LIM_ATTACK, LIM_SUSTAIN, LIM_RELEASE };

__device__ inline double ln_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void ln_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int ln_wrap(int i) { return i >= LN_LBS ? i - LN_LBS : i; }
__device__ inline double ln_wave_max(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d, 64));
    return v;
}

struct LnState {
    double gr0, gr1, prev_smp;
    int lbi, state, peak_index, env_index, env_cnt, attack_length;
    bool first;
};

// max |ring[base .. base + count)| (ring indices wrap)
__device__ double ln_ring_absmax(const double *ring, int base, int count, int lane)
{
    double m = 0.0;
    constexpr int LN_NB8 = 16;
    for (int j0 = lane; j0 < count; j0 += 64 * LN_NB8) {
        double t[LN_NB8];
#pragma unroll
        for (int u = 0; u < LN_NB8; ++u) { const int j = j0 + 64 * u; t[u] = j < count ? fabs(ln_ld(&ring[(base + j) % LN_LBS])) : 0.0; }
#pragma unroll
        for (int u = 0; u < LN_NB8; ++u) m = fmax(m, t[u]);
    }
    return ln_wave_max(m);
}

// detect_peak(): first n in (0, nb) that is a local maximum above the ceiling with no larger sample among the next ten; the filter's
// prev_smp bookkeeping (stale after a rejected candidate) is reproduced exactly.  All lanes return the same values.
__device__ int ln_detect_peak(const double *ring, LnState &s, int offset, int nb, double ceiling, double *peak_value, int lane)
{
    int base = s.lbi + offset + LN_ATT;
    base %= LN_LBS; if (base < 0) base += LN_LBS;
    auto at = [&](int n) -> double { int i = (base + n) % LN_LBS; if (i < 0) i += LN_LBS; return fabs(ln_ld(&ring[i])); };
    if (s.first) s.prev_smp = at(-1);
    if (nb <= 0) return -1;
    s.prev_smp = at(0);                                                   // n = 0 is never a candidate
    if (nb == 1) return -1;
    int n_start = 1;
    for (;;) {
        // first n >= n_start with |x[n-1]| <= |x[n]| >= |x[n+1]| and |x[n]| > ceiling (prev_smp is the true neighbour from n_start on).
        // The range is walked in blocks of 1024: a block whose maximum stays under the ceiling holds no candidate (one batched load per
        // lane), and the walk stops at the first block that yields one -- the limiter calls this once per peak while it holds a signal down
        int n1 = -1;
        for (int b0 = n_start; b0 < nb && n1 < 0; b0 += 1024) {
            const int bc = min(1024, nb - b0);
            double t16[16]; double mx = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int j = lane + 64 * u; t16[u] = j < bc ? at(b0 + j) : 0.0; }
#pragma unroll
            for (int u = 0; u < 16; ++u) mx = fmax(mx, t16[u]);
            if (!(ln_wave_max(mx) > ceiling)) continue;
            for (int c0 = b0; c0 < b0 + bc && n1 < 0; c0 += 64) {
                const int n = c0 + lane;
                bool cand = false;
                if (n < b0 + bc) { const double t = at(n); cand = t > ceiling && at(n - 1) <= t && at(n + 1) <= t; }
                const unsigned long long bal = __ballot(cand);
                if (bal) n1 = c0 + (__ffsll((long long)bal) - 1);
            }
        }
        if (n1 < 0) { s.prev_smp = at(nb - 1); return -1; }
        double pv = at(n1 - 1);
        int m = n1;
        for (;;) {                                                           // candidate at m with prev = pv
            const double t = at(m);
            bool detected = true;
            for (int i = 2; i < 12; ++i) if (at(m + i) > t) { detected = false; break; }
            if (detected) { s.prev_smp = t; s.peak_index = (base + m) % LN_LBS; *peak_value = t; return m; }
            // rejected: prev_smp keeps the value it had; the next sample is tested against that stale value
            ++m;
            if (m >= nb) { s.prev_smp = pv; return -1; }
            const double t2 = at(m);
            if (pv <= t2 && at(m + 1) <= t2 && t2 > ceiling) continue;      // a candidate again, still with the stale value
            pv = t2; ++m;
            break;
        }
        s.prev_smp = pv;
        if (m >= nb) return -1;
        n_start = m;
    }
}

// Elementwise passes over a range: sixteen loads in flight per lane before the first store (one load -> one store per trip left every
// trip waiting a full memory latency: 300 trips per 100 ms frame and pass).
constexpr int LN_NB = 16;
// ring[env_index + j] *= env(j), j in [0, cnt)
template <typename F>
__device__ void ln_ring_scale(double *ring, int env_index, int cnt, F env, int lane)
{
    for (int j0 = lane; j0 < cnt; j0 += 64 * LN_NB) {
        double t[LN_NB];
#pragma unroll
        for (int u = 0; u < LN_NB; ++u) { const int j = j0 + 64 * u; t[u] = j < cnt ? ln_ld(&ring[(env_index + j) % LN_LBS]) : 0.0; }
#pragma unroll
        for (int u = 0; u < LN_NB; ++u) { const int j = j0 + 64 * u; if (j < cnt) ln_st(&ring[(env_index + j) % LN_LBS], t[u] * env(j)); }
    }
    __threadfence();
}
// ring[base + j] = f(src[j], j), j in [0, cnt)   (src is read-only input)
template <typename F>
__device__ void ln_ring_fill(double *ring, int base, const double *__restrict__ src, int64_t src_len, int cnt, F f, int lane)
{
    for (int j0 = lane; j0 < cnt; j0 += 64 * LN_NB) {
        double t[LN_NB];
#pragma unroll
        for (int u = 0; u < LN_NB; ++u) { const int j = j0 + 64 * u; t[u] = (j < cnt && j < src_len) ? src[j] : 0.0; }
#pragma unroll
        for (int u = 0; u < LN_NB; ++u) { const int j = j0 + 64 * u; if (j < cnt) ln_st(&ring[(base + j) % LN_LBS], j < src_len ? f(t[u], j) : 0.); }
    }
    __threadfence();
}

__device__ void ln_true_peak_limiter(double *ring, LnState &s, double *out, int nb, double ceiling, int lane)
{
    const int index0 = s.lbi;
    int smp_cnt = 0;
    if (s.first) {
        const double mx = ln_ring_absmax(ring, 0, LN_ATT, lane);
        if (mx > ceiling) {
            s.gr1 = ceiling / mx;
            s.state = LIM_SUSTAIN;
            const double g = s.gr1;
            ln_ring_scale(ring, 0, LN_ATT, [&](int) { return g; }, lane);
        }
    }
    do {
        switch (s.state) {
        case LIM_OUT: {
            double pkv = 0.0;
            const int pd = ln_detect_peak(ring, s, smp_cnt, nb - smp_cnt, ceiling, &pkv, lane);
            if (pd != -1) {
                s.env_cnt = 0;
                smp_cnt += (pd - s.attack_length);
                s.gr0 = 1.; s.gr1 = ceiling / pkv;
                s.state = LIM_ATTACK;
                s.env_index = s.peak_index - s.attack_length;
                if (s.env_index < 0) s.env_index += LN_LBS;
                s.env_index += s.env_cnt;
                if (s.env_index > LN_LBS) s.env_index -= LN_LBS;
            } else smp_cnt = nb;
            break; }
        case LIM_ATTACK: {
            int cnt = s.attack_length - s.env_cnt; if (cnt > nb - smp_cnt) cnt = nb - smp_cnt; if (cnt < 0) cnt = 0;
            const double g0 = s.gr0, g1 = s.gr1; const int c0 = s.env_cnt, al = s.attack_length;
            ln_ring_scale(ring, s.env_index, cnt, [&](int j) { return g0 - ((double)(c0 + j) / (al - 1) * (g0 - g1)); }, lane);
            s.env_index = (s.env_index + cnt) % LN_LBS; s.env_cnt += cnt; smp_cnt += cnt;
            if (smp_cnt < nb) { s.env_cnt = 0; s.attack_length = LN_ATT; s.state = LIM_SUSTAIN; }
            break; }
        case LIM_SUSTAIN: {
            double pkv = 0.0;
            const int pd = ln_detect_peak(ring, s, smp_cnt, nb, ceiling, &pkv, lane);
            if (pd == -1) { s.state = LIM_RELEASE; s.gr0 = s.gr1; s.gr1 = 1.; s.env_cnt = 0; break; }
            const double gr = ceiling / pkv;
            if (gr < s.gr1) {
                s.state = LIM_ATTACK;
                s.attack_length = pd; if (s.attack_length <= 1) s.attack_length = 2;
                s.gr0 = s.gr1; s.gr1 = gr; s.env_cnt = 0;
                break;
            }
            int cnt = pd; if (cnt > nb - smp_cnt) cnt = nb - smp_cnt; if (cnt < 0) cnt = 0;
            const double g = s.gr1;
            ln_ring_scale(ring, s.env_index, cnt, [&](int) { return g; }, lane);
            s.env_index = (s.env_index + cnt) % LN_LBS; s.env_cnt = cnt; smp_cnt += cnt;
            break; }
        case LIM_RELEASE: {
            int cnt = LN_REL - s.env_cnt; if (cnt > nb - smp_cnt) cnt = nb - smp_cnt; if (cnt < 0) cnt = 0;
            const double g0 = s.gr0, g1 = s.gr1; const int c0 = s.env_cnt;
            ln_ring_scale(ring, s.env_index, cnt, [&](int j) { return g0 + (((double)(c0 + j) / (LN_REL - 1)) * (g1 - g0)); }, lane);
            s.env_index = (s.env_index + cnt) % LN_LBS; s.env_cnt += cnt; smp_cnt += cnt;
            if (smp_cnt < nb) { s.env_cnt = 0; s.state = LIM_OUT; }
            break; }
        }
    } while (smp_cnt < nb);
    for (int n0 = lane; n0 < nb; n0 += 64 * LN_NB) {
        double t[LN_NB];
#pragma unroll
        for (int u = 0; u < LN_NB; ++u) { const int n = n0 + 64 * u; t[u] = n < nb ? ln_ld(&ring[(index0 + n) % LN_LBS]) : 0.0; }
#pragma unroll
        for (int u = 0; u < LN_NB; ++u) {
            const int n = n0 + 64 * u;
            double v = t[u];
            if (fabs(v) > ceiling) v = ceiling * (v < 0 ? -1 : 1);
            if (n < nb) out[n] = v;
        }
    }
    __threadfence();
}

// gaussian_filter(s, index)
__device__ double ln_gaussian(const double *delta, const double *w, int index)
{
    double result = 0.;
    index = index - 10 > 0 ? index - 10 : index + 20;
    for (int i = 0; i < 21; i++) result += delta[((index + i) < 30) ? (index + i) : (index + i - 30)] * w[i];
    return result;
}

__global__ void __launch_bounds__(64)
k_loudnorm_dynamic(const double *__restrict__ x, int64_t n, LoudnormDynParams P, const double *__restrict__ series, double *__restrict__ ring,
                   double *__restrict__ y, double *__restrict__ dbg)
{
    __shared__ double delta[30], w[21], oe[30], tile[1024];
    const int lane = threadIdx.x;
    if (lane < 30) { delta[lane] = P.delta0; oe[lane] = 0.0; }
    if (lane < 21) w[lane] = P.weights[lane];
    __syncthreads();
    const double ceiling = P.target_tp_lin, offset = P.offset_lin;
    LnState s; s.gr0 = 1.; s.gr1 = 1.; s.prev_smp = 0.; s.lbi = 0; s.state = LIM_OUT; s.peak_index = 0; s.env_index = 0; s.env_cnt = 0;
    s.attack_length = LN_ATT; s.first = true;
    int index = 1, above = P.above0;
    double prev_delta = P.delta0;
    double kv1 = 0, kv2 = 0, kv3 = 0, kv4 = 0;                                  // the output meter's filter state (lane 0)
    int oe_pos = 0;
    // K-weighted energy of an output frame (only while the stream has not yet reached the target: one lane, the filter is a recurrence)
    auto out_energy = [&](const double *src, int cnt) {
        double sum = 0.0;
        for (int c0 = 0; c0 < cnt; c0 += 1024) {
            const int m = min(1024, cnt - c0);
            __syncthreads();
            for (int j = lane; j < m; j += 64) tile[j] = src[c0 + j];
            __syncthreads();
            if (lane == 0) {
                for (int j = 0; j < m; ++j) {
                    const double v0 = tile[j] - P.kwa[1] * kv1 - P.kwa[2] * kv2 - P.kwa[3] * kv3 - P.kwa[4] * kv4;
                    const double o = P.kwb[0] * v0 + P.kwb[1] * kv1 + P.kwb[2] * kv2 + P.kwb[3] * kv3 + P.kwb[4] * kv4;
                    kv4 = kv3; kv3 = kv2; kv2 = kv1; kv1 = v0;
                    sum += o * o;
                }
            }
        }
        __syncthreads();
        if (lane == 0) { oe[oe_pos] = sum; }
        oe_pos = (oe_pos + 1) % 30;
        __syncthreads();
    };
    // ---- FIRST_FRAME: the first 210 ms with the initial gain, 100 ms out
    { const double d0 = P.delta0; ln_ring_fill(ring, 0, x, n, LN_LBS, [&](double v, int) { return v * d0 * offset; }, lane); }
    ln_true_peak_limiter(ring, s, y, LN_F100, ceiling, lane);
    s.first = false;
    if (!above) out_energy(y, LN_F100);
    int64_t produced = LN_F100, abs_in = LN_LBS;
    // ---- INNER_FRAMEs
    for (int64_t k = 0; k < P.n_inner; ++k) {
        const int nb = (int)min<int64_t>(LN_F100, n - LN_F3000 - k * LN_F100);
        const double gain = ln_gaussian(delta, w, index + 10 < 30 ? index + 10 : index + 10 - 30);
        const double gain_next = ln_gaussian(delta, w, index + 11 < 30 ? index + 11 : index + 11 - 30);
        ln_ring_fill(ring, s.lbi, x + abs_in, n - abs_in, nb, [&](double v, int j) { return v * (gain + (((double)j / nb) * (gain_next - gain))) * offset; }, lane);
        s.lbi = (s.lbi + nb) % LN_LBS;
        s.lbi = (s.lbi + (LN_F100 - nb)) % LN_LBS;
        ln_true_peak_limiter(ring, s, y + produced, nb, ceiling, lane);
        const double shortterm = series[3 * k], global = series[3 * k + 1], relthr = series[3 * k + 2];
        if (above == 0) {
            if (shortterm > P.measured_thresh) prev_delta *= 1.0058;
            out_energy(y + produced, nb);
            double e = 0.0; for (int q = 0; q < 30; ++q) e += oe[q];
            e = e * (P.dual_mono ? 2.0 : 1.0) / (double)LN_F3000;
            const double st_out = e <= 0.0 ? -HUGE_VAL : 10 * (log(e) / log(10.0)) - 0.691;
            if (st_out >= P.target_i) above = 1;
        }
        double d;
        if (shortterm < relthr || shortterm <= -70. || above == 0) d = prev_delta;
        else {
            const double env_global = fabs(shortterm - global) < (P.target_lra / 2.) ? shortterm - global
                                                                                     : (P.target_lra / 2.) * ((shortterm - global) < 0 ? -1 : 1);
            const double env_shortterm = P.target_i - shortterm;
            d = pow(10., (env_global + env_shortterm) / 20.);
        }
        __syncthreads();
        if (lane == 0) delta[index] = d;
        __syncthreads();
        prev_delta = d;
        index++; if (index >= 30) index -= 30;
        produced += nb; abs_in += nb;
    }
    // ---- FINAL_FRAME: the last 2.9 s again from the look-ahead buffer, one gain
    {
        const double gain = ln_gaussian(delta, w, index + 10 < 30 ? index + 10 : index + 10 - 30);
        const double *src = x + (n - P.final_len);
        s.lbi = 0;
        ln_ring_fill(ring, 0, src, P.final_len, LN_LBS, [&](double v, int) { return v * gain * offset; }, lane);
        int64_t src_index = LN_LBS;
        for (int i = 0; i < P.final_len / LN_F100; ++i) {
            ln_true_peak_limiter(ring, s, y + produced, LN_F100, ceiling, lane);
            ln_ring_fill(ring, s.lbi, src + src_index, P.final_len - src_index, LN_F100, [&](double v, int) { return v * gain * offset; }, lane);
            src_index = min<int64_t>(src_index + LN_F100, P.final_len);
            s.lbi = (s.lbi + LN_F100) % LN_LBS;
            produced += LN_F100;
        }
    }
    if (lane == 0 && dbg) { dbg[0] = (double)produced; dbg[1] = (double)above; dbg[2] = (double)s.state; dbg[3] = prev_delta; }
}

__global__ void k_scale_f64(const double *__restrict__ in, double *__restrict__ out, int64_t n, double g)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * g;
}
// swr dbl -> dbl at any ratio, one output per thread, taps ascending, multiply then add (resample_template.c's order): the plain form of
// k_polyphase for the one place whose ratio (192 kHz -> 44.1 kHz, step 640) does not fit that kernel's LDS tile.  Flush mode.
__global__ void k_swr_plain_f64(const double *__restrict__ in, int64_t n, const double *__restrict__ bank, int P, int L, int center, int64_t step,
                                int64_t m_total, double *__restrict__ out)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_total) return;
    const int64_t idx = m * step;
    const int ph = (int)(idx % P);
    const int64_t si = idx / P - center;
    const double *f = bank + (size_t)ph * L;
    double val = 0.0;
    for (int i = 0; i < L; ++i) {
        int64_t g = si + i;
        double v = 0.0;
        if (g < 0) g = -g;                                           // invert_initial_buffer(): in[-j] = in[j]
        if (g < n) v = in[g];
        else { const int64_t r = 2 * n - 1 - g; if (r >= 0 && r < n) v = in[r]; }   // resample_flush()
        val += v * f[i];
    }
    out[m] = val;
}
}  // namespace

void launch_swr_plain_f64(const double *in, int64_t n, const double *bank, int P, int L, int center, int64_t step, int64_t m_total, double *out, hipStream_t s)
{
    if (m_total > 0) hipLaunchKernelGGL(k_swr_plain_f64, dim3((unsigned)((m_total + 255) / 256)), dim3(256), 0, s, in, n, bank, P, L, center, step, m_total, out);
}
void launch_loudnorm_dynamic(const double *x, int64_t n, const LoudnormDynParams &P, const double *series, double *ring, double *y, double *dbg, hipStream_t s)
{
    // 96 KB of (unused) dynamic LDS: the workgroup then has a CU to itself.  Several files in flight each run one such wave, and the
    // dispatcher packed them onto the first CU with room -- the same SIMDs -- where each ran 1.5x slower than alone.
    static const int reserve = getenv("JT_DYN_NO_CU_RESERVE") ? 0 : 96 * 1024;
    if (reserve) (void)hipFuncSetAttribute((const void *)k_loudnorm_dynamic, hipFuncAttributeMaxDynamicSharedMemorySize, reserve);
    hipLaunchKernelGGL(k_loudnorm_dynamic, dim3(1), dim3(64), reserve, s, x, n, P, series, ring, y, dbg);
}
void launch_scale_f64(const double *in, double *out, int64_t n, double g, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_scale_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n, g);
}
