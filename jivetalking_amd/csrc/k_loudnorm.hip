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

// =====================================================================================================================
// The same state machine run by a WORKGROUP of sixteen waves with the frame the limiter is editing held in LDS.
// The one-wave kernel above keeps the 210 ms ring in global memory and pays a memory round trip for every dependent step: a frame the
// limiter works on is ~130 such steps (44 peaks a frame on a file driven into the ceiling: detect, attack ramp, sustain, ...), 0.8 ms.
// What the limiter touches in one call is the frame it is about to hand out plus the 1932 samples behind it (the envelope segments are
// applied 10 ms ahead of the output position, the peak test reads 12 samples further).  So: that range (less its first 832 samples:
// LDS holds 20 300 doubles) is loaded into LDS when the call starts, every access to it goes to LDS, anything outside falls through
// to the ring in global memory, and the window is written back (ring) and the frame out (clamped) when the call ends.  All threads carry the same
// state and take the same decisions (every decision reads values all threads see alike); the elementwise passes, the block maximum
// and the search for the first candidate are spread over the 1024 threads.  Same arithmetic on the same values in the same order per
// element as the one-wave kernel, hence the same output bit for bit (tests/test_gpu_round3.py).
#ifndef JT_LN_WG
#define JT_LN_WG 512
#endif
constexpr int LN_WG = JT_LN_WG;
#ifdef JT_LN_PROFILE
// phase clocks (tools/prof_dynamic_phases.sh): accumulated in LDS by thread 0 (a GLOBAL read-modify-write per event cost more than the
// events it measured) and stored once per kernel; shader cycles (clock64)
__device__ unsigned long long g_ln_prof[16];        // [0] fill, [1] window load, [2] limiter loop, [3] write-back + out, [4] detect, [7] scale; [5] scale passes, [6] detect calls, [8] / [9] left the window / general walks
__shared__ unsigned long long lnp_acc[16];
#define LNP_T0 const unsigned long long lnp_t0_ = clock64();
#define LNP_ADD(i) do { if (threadIdx.x == 0) lnp_acc[i] += clock64() - lnp_t0_; } while (0)
#define LNP_CNT(i) do { if (threadIdx.x == 0) lnp_acc[i] += 1; } while (0)
#else
#define LNP_T0
#define LNP_ADD(i)
#define LNP_CNT(i)
#endif
// LDS window: ring positions index0 + [LN_LO, LN_LO + LN_CACHE), index0 = the frame about to go out.  LN_LO = 1920: the peak detector
// reads from 1920 samples (the 10 ms look-ahead) into the frame on, 19 200 samples ahead of the output position at most and a few
// hundred as a rule, so a window of 20 300 doubles (what 160 KB of LDS hold beside the other arrays) that STARTS there keeps the
// detector inside LDS up to the last sample of the frame; started at the frame's first sample it ended 245 samples short of that, and
// the 3 % of the detector's calls that then walked the ring in global memory took two thirds of the kernel's time.  The first 1920
// samples of the frame are only touched by the first envelope segments of a call and stay in global memory (batched read-modify-write).
constexpr int LN_LO = LN_ATT, LN_CACHE = 19960;
// (round 5: 340 entries fewer than the 20 300 that fit, for the frame's detected-peak bitmap; the window still reaches 760 samples behind
//  the frame's last one, and a held-down signal has its next peak within a couple of hundred)
constexpr int LN_BMW = (LN_CACHE + 63) / 64;                      // bitmap words: bit b of word k = window entry 64 k + b is a detected peak
struct LnRing {
    double *ring, *cache; int index0;
    __device__ int rel(int i) const { const int r = i - index0; return r < 0 ? r + LN_LBS : r; }
    __device__ double rd(int i) const { const unsigned c = (unsigned)(rel(i) - LN_LO); return c < (unsigned)LN_CACHE ? cache[c] : ln_ld(&ring[i]); }
    __device__ void wr(int i, double v) const { const unsigned c = (unsigned)(rel(i) - LN_LO); if (c < (unsigned)LN_CACHE) cache[c] = v; else ln_st(&ring[i], v); }
};
__device__ inline int ln_mod(int i) { i %= LN_LBS; return i < 0 ? i + LN_LBS : i; }

// Envelope segment, wave 0 only (no barriers: one wave's LDS operations execute in order)
template <typename F>
__device__ __forceinline__ void lnv_ring_scale(const LnRing &R, int env_index, int cnt, F env, int lane)
{
    LNP_CNT(5);
    LNP_T0
    const int r0 = R.rel(env_index % LN_LBS) - LN_LO;
    if (r0 >= 0 && r0 + cnt <= LN_CACHE) {                          // the whole segment lies in the window: LDS only
        double *c = R.cache + r0;
        for (int j0 = lane; j0 < cnt; j0 += 64 * 8) {                 // eight reads in flight per lane
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + 64 * u; t[u] = j < cnt ? c[j] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + 64 * u; if (j < cnt) c[j] = t[u] * env(j); }
        }
    } else {
        for (int j0 = lane; j0 < cnt; j0 += 64 * 8) {                 // (window or ring, element by element; eight reads in flight)
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + 64 * u; t[u] = j < cnt ? R.rd((env_index + j) % LN_LBS) : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + 64 * u; if (j < cnt) R.wr((env_index + j) % LN_LBS, t[u] * env(j)); }
        }
        __threadfence();
    }
    __builtin_amdgcn_wave_barrier();
    LNP_ADD(7);
}
// cnt <= 40320: at most NBF elements per thread, all loads of a thread in flight before its first store
constexpr int LN_NBF = 20;
template <typename F>
// `hot[k]`: ring samples 64 k .. 64 k + 63 may hold one above the ceiling.  Set here from the values written (base and the waves' strides
// are multiples of 64: a wave writes one aligned block per step); everything the limiter does afterwards multiplies by gains <= 1, so a
// clear flag stays true.  The detector's walk outside the LDS window skips blocks whose flag is clear without touching memory.
__device__ __forceinline__ void lnw_ring_fill(double *ring, int base, const double *__restrict__ src, int64_t src_len, int cnt, F f, int tid,
                                              unsigned char *hot, double ceiling)
{
    LNP_T0
    for (int j0 = tid; j0 < cnt; j0 += LN_WG * LN_NBF) {
        double t[LN_NBF];
#pragma unroll
        for (int u = 0; u < LN_NBF; ++u) { const int j = j0 + LN_WG * u; t[u] = (j < cnt && j < src_len) ? src[j] : 0.0; }
#pragma unroll
        for (int u = 0; u < LN_NBF; ++u) {
            const int j = j0 + LN_WG * u;
            const bool wr = j < cnt;
            const double v = (wr && j < src_len) ? f(t[u], j) : 0.;
            const int i = (base + j) % LN_LBS;
            if (wr) ln_st(&ring[i], v);
            const unsigned long long act = __ballot(wr), h = __ballot(wr && fabs(v) > ceiling);
            if (wr && (tid & 63) == 0) hot[i >> 6] = act == ~0ull ? (unsigned char)(h != 0) : (unsigned char)(hot[i >> 6] | (h != 0));
        }
    }
    __threadfence();
    __syncthreads();
    LNP_ADD(0);
}
__device__ __forceinline__ double lnv_ring_absmax(const LnRing &R, int base, int count, int lane)
{
    double m = 0.0;
    for (int j = lane; j < count; j += 64) m = fmax(m, fabs(R.rd((base + j) % LN_LBS)));
    return ln_wave_max(m);
}
// ln_detect_peak() on the LDS window, wave 0 only.  Two functions.
//   lnv_detect_window: everything it reads lies in the window (q[n] = |sample base + n| for -1 <= n < e).  No accessor, no branch with a
//   memory load on its other side, and every value the wave agrees on -- positions, the candidate's value, its neighbours -- is moved
//   to scalar registers explicitly (readfirstlane): left to the compiler, the "uniform" control flow of the walk was vector compares
//   and exec-mask save / restore around every read, 2 000 shader cycles per 64-position step.  Returns the peak's position, -1 (no peak
//   before nb), or -2 with *resume set when the walk would leave the window: the general function then redoes that search step.
//   lnv_detect_general: the same walk through the accessor (window or ring), from any starting position.
// Same walk as ln_detect_peak: candidates in ascending order (64 positions at a time, the next 256 first, then 1024-position blocks
// rejected by their maximum), the order-dependent tail serially; the "no larger sample among the next ten" test is one 10-lane read.
__device__ __forceinline__ double ln_rfl(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ double ln_rl(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ int lnv_detect_window(const double *q, int e, LnState &s, int nb, double ceiling, double *peak_value, int base, int lane, int *resume)
{
    // first candidate among the 256 positions from c on (all inside the window: c + 257 <= e): twelve reads in flight, four ballots.
    // (With one 64-position step at a time and the three reads of a position behind each other's compare -- what `a && b && c` compiles
    // to -- a step was three LDS round trips, 500 shader cycles, and the walk is nothing but such steps.)
    auto group = [&](int c, int lim) -> int {
        double t[4], l[4], r[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { const int n = c + 64 * g + lane; t[g] = fabs(q[n]); l[g] = fabs(q[n - 1]); r[g] = fabs(q[n + 1]); }
        int found = -1;
#pragma unroll
        for (int g = 3; g >= 0; --g) {
            const int n = c + 64 * g + lane;
            const bool cand = (n < lim) & (t[g] > ceiling) & (l[g] <= t[g]) & (r[g] <= t[g]);
            const unsigned long long bal = __ballot(cand);
            if (bal) found = c + 64 * g + (__ffsll((long long)bal) - 1);
        }
        return found;
    };
    int n_start = *resume;                                          // (1, or the first position the SUSTAIN batch has not judged yet)
#ifdef JT_LN_PROFILE
    unsigned long long pc_ = clock64();
#define LNW_SEC(i) { const unsigned long long c_ = clock64(); if (threadIdx.x == 0) lnp_acc[i] += c_ - pc_; pc_ = c_; }
#else
#define LNW_SEC(i)
#endif
    for (;;) {
        int n1 = -1;
        int b0 = n_start;
        // a signal held down by the limiter has its next peak within a couple of hundred samples: look there first
        for (int k = 0; k < 2 && b0 < nb && n1 < 0; ++k, b0 += 256) {
            if (b0 + 257 > e) { *resume = n_start; return -2; }
            n1 = group(b0, nb);
        }
        for (; b0 < nb && n1 < 0; b0 += 1024) {
            if (b0 + 1025 > e) { *resume = n_start; return -2; }
            const int bc = min(1024, nb - b0);
            double mx = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int j = lane + 64 * u; const double t = fabs(q[b0 + j]); mx = fmax(mx, j < bc ? t : 0.0); }
            if (!__any(mx > ceiling)) continue;
            for (int c = b0; c < b0 + bc && n1 < 0; c += 256) n1 = group(c, b0 + bc);
        }
        LNW_SEC(11)
        if (n1 < 0) { s.prev_smp = ln_rfl(fabs(q[nb - 1])); return -1; }
        int m = n1;
        if (m + 15 > e) { *resume = n_start; return -2; }
        double pv = ln_rfl(fabs(q[m - 1]));
        for (;;) {
            // one read serves the whole step: lane i holds |x[m - 1 + i]|, i = 0 .. 15
            if (m + 15 > e) { *resume = n_start; return -2; }
            const double v = fabs(q[m - 1 + (lane & 15)]);
            const double t = ln_rl(v, 1);
            const bool larger = lane >= 3 && lane < 13 && v > t;                 // x[m + 2] .. x[m + 11]
            if (__ballot(larger) == 0ull) { s.prev_smp = t; s.peak_index = ln_mod(base + m); *peak_value = t; LNW_SEC(12) return m; }
            // rejected: prev_smp keeps the value it had; the next sample is tested against that stale value
            ++m;
            if (m >= nb) { s.prev_smp = pv; return -1; }
            const double t2 = ln_rl(v, 2);                                       // x[m] (the new m)
            if (pv <= t2 && ln_rl(v, 3) <= t2 && t2 > ceiling) continue;          // a candidate again, still with the stale value
            pv = t2; ++m;
            break;
        }
        s.prev_smp = pv;
        LNW_SEC(12)
        if (m >= nb) return -1;
        n_start = m;
    }
}
__device__ __forceinline__ int lnv_detect_general(const LnRing &R, const unsigned char *hot, LnState &s, int nb, double ceiling, double *peak_value,
                                                  int base, int lane, int n_start)
{
    auto at = [&](int n) -> double { return fabs(R.rd(ln_mod(base + n))); };
    for (;;) {
        int n1 = -1;
        for (int c = n_start; c < nb && n1 < 0; c += 64) {
            // the 64 positions from c on lie in (at most) two aligned ring blocks: neither hot, no candidate, no read
            const int i0 = ln_mod(base + c), i1 = i0 + 63 >= LN_LBS ? i0 + 63 - LN_LBS : i0 + 63;
            if (!(hot[i0 >> 6] | hot[i1 >> 6])) continue;
            const int n = c + lane;
            bool cand = false;
            if (n < nb) { const double t = at(n), l = at(n - 1), r = at(n + 1); cand = (t > ceiling) & (l <= t) & (r <= t); }
            const unsigned long long bal = __ballot(cand);
            if (bal) n1 = c + (__ffsll((long long)bal) - 1);
        }
        if (n1 < 0) { s.prev_smp = at(nb - 1); return -1; }
        double pv = at(n1 - 1);
        int m = n1;
        for (;;) {
            const double t = at(m);
            const bool larger = lane >= 2 && lane < 12 && at(m + lane) > t;
            if (__ballot(larger) == 0ull) { s.prev_smp = t; s.peak_index = ln_mod(base + m); *peak_value = t; return m; }
            ++m;
            if (m >= nb) { s.prev_smp = pv; return -1; }
            const double t2 = at(m);
            if (pv <= t2 && at(m + 1) <= t2 && t2 > ceiling) continue;
            pv = t2; ++m;
            break;
        }
        s.prev_smp = pv;
        if (m >= nb) return -1;
        n_start = m;
    }
}
__device__ __forceinline__ int lnv_detect_peak_(const LnRing &R, const unsigned char *hot, LnState &s, int offset, int nb, double ceiling, double *peak_value, int lane, int n_first)
{
    LNP_CNT(6);
    const int base = ln_mod(s.lbi + offset + LN_ATT);
    const int c0 = __builtin_amdgcn_readfirstlane(R.rel(base) - LN_LO);
    auto at = [&](int n) -> double { return fabs(R.rd(ln_mod(base + n))); };
    if (s.first) s.prev_smp = at(-1);
    if (nb <= 0) return -1;
    s.prev_smp = at(0);
    if (nb == 1) return -1;
    int n_start = n_first;
    if (c0 >= 0) {                                                   // base is inside the window (the walk reads base + n - 1, n >= 1)
        const int r = lnv_detect_window(R.cache + c0, LN_CACHE - c0, s, nb, ceiling, peak_value, base, lane, &n_start);
        if (r != -2) return r;
        LNP_CNT(8);
    }
    LNP_CNT(9);
    LNP_T0
    const int rg = lnv_detect_general(R, hot, s, nb, ceiling, peak_value, base, lane, n_start);
    LNP_ADD(10);
    return rg;
}
__device__ __forceinline__ int lnv_detect_peak(const LnRing &R, const unsigned char *hot, LnState &s, int offset, int nb, double ceiling, double *peak_value, int lane, int n_first = 1)
{
    LNP_T0
    const int r = lnv_detect_peak_(R, hot, s, offset, nb, ceiling, peak_value, lane, n_first);
    LNP_ADD(4);
    return r;
}
// SUSTAIN, all harmless peaks of the frame's remainder at once (round 5).  While the limiter holds a signal down, detect_peak() finds a
// peak every few hundred samples and almost every one of them needs no more reduction than is already applied (ceiling / peak >= gr1:
// within a stretch of continuous limiting gr1 only ever falls, so only a new record peak is "harmful"): the filter then multiplies the
// samples up to that peak by gr1 and looks for the next one -- ~600 times a second, one dependent detector call each.  The batch finds,
// among the positions n = 1 .. R (R = what is left of the frame; q[n] = |sample scan_start + n|, all inside the LDS window),
//   h = the first DETECTED peak whose reduction would be stronger (ceiling / q[n] < gr1), and
//   l = the last detected peak before h (or before the frame's end when there is no h),
// so that the caller can apply gr1 up to l in ONE segment and resume the filter's own statements there: with h it takes the SUSTAIN ->
// ATTACK transition the detector call from l would have led to (peak_delta = h - l), without h that call starts behind the positions
// the batch has judged already (what lies behind the frame's end decides between sustain, attack and release, as before).
// "Detected" is the filter's own outcome, stated without its prev_smp bookkeeping:
//   |q[n-1]| <= |q[n]| >= |q[n+1]|,  |q[n]| > ceiling,  no |q[n+i]| > |q[n]| for i = 2 .. 11.
// detect_peak() leaves prev_smp stale after a REJECTED candidate m (a local maximum with a larger sample L at l' in m+2 .. m+11), which
// can make positions m+1 .. l'-2 candidates that a fresh prev_smp would not, or the other way round -- but every one of those has l' inside
// its own window of ten and is rejected either way, and at l'-1 the candidate test fails on `next <= this` under both (q[l'] > q[m] >=
// q[l'-1]), which refreshes prev_smp before l'.  The detected SET is therefore the clean predicate's; a call that starts at a detected peak
// (n = 0 there) cannot return that peak again, which is "n >= 1" here.  Returns l (0: none), *lprev = the detected peak before l (0: none).
__device__ __forceinline__ int lnv_sustain_batch(const double *q, int R, double ceiling, double gr1, int lane, int *lprev, int *harm_n, double *harm_v)
{
    int l = 0, lp = 0;
    bool stop = false;
    *harm_n = 0; *harm_v = 0.0;
    // 256 positions per trip: the twelve reads of four 64-position groups are in flight together (one LDS round trip per trip instead of
    // one per group: the walk is nothing but such trips), the groups are then judged in order
    for (int c0 = 1; c0 <= R && !stop; c0 += 256) {
        double tt[4], aa[4], bb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { const int n = c0 + 64 * g + lane; tt[g] = fabs(q[n]); aa[g] = fabs(q[n - 1]); bb[g] = fabs(q[n + 1]); }
        bool cand[4]; bool anyc = false;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = c0 + 64 * g + lane;
            cand[g] = (n <= R) & (tt[g] > ceiling) & (aa[g] <= tt[g]) & (bb[g] <= tt[g]);
            anyc |= cand[g];
        }
        if (__ballot(anyc) == 0ull) continue;
        // the ten samples behind every position of the trip (not only behind its candidates: no divergent branch, so all forty reads are
        // in flight together -- judged group by group behind their own branch, a trip waited for up to four more LDS round trips)
        double mx[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = c0 + 64 * g + lane;
            double m = 0.0;
#ifdef JT_LN_BATCH_ALLGROUPS
            {
#else
            if (__ballot(cand[g])) {                                   // (wave-uniform: only the groups that hold a candidate)
#endif
                double v[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) v[i] = q[n + 2 + i];
#pragma unroll
                for (int i = 0; i < 10; ++i) m = fmax(m, fabs(v[i]));
            }
            mx[g] = m;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (stop) break;
            const int c = c0 + 64 * g;
            const double t = tt[g];
            const bool det = cand[g] && !(mx[g] > t);
            const unsigned long long dm = __ballot(det);
            if (dm == 0ull) continue;
            const bool harm = det && (ceiling / t < gr1);               // the filter's own comparison: gain_reduction < s->gain_reduction[1]
            const unsigned long long hm = __ballot(harm);
            unsigned long long before = dm;
            if (hm) before = dm & ((1ull << (__ffsll((long long)hm) - 1)) - 1ull);
            if (before) {
                const int hi = 63 - __clzll((long long)before);
                const unsigned long long rest = before & ~(1ull << hi);
                lp = rest ? c + (63 - __clzll((long long)rest)) : l;
                l = c + hi;
            }
            if (hm) { const int f = __ffsll((long long)hm) - 1; *harm_n = c + f; *harm_v = ln_rl(t, f); stop = true; }
        }
    }
    *lprev = lp;
    return l;
}
// SUSTAIN segments (one constant gain over a stretch of the window) that wave 0 has decided but not applied: all waves multiply them
// into the window side by side when the state machine has finished the frame -- one wave's read-modify-write of up to 19 200 LDS entries
// per segment was a third of the limiter's time.  Allowed while nothing reads those entries before the frame goes out: the detector works
// ahead of every segment (bm_ok), and an envelope segment that would overlap a pending one is applied after the pending ones.
struct LnPending { int n; int start[4], len[4]; double g[4]; };
// ---- the frame's detected peaks as a bitmap (round 5).  "Detected" as in lnv_sustain_batch's comment: a pure function of the samples,
// so all the workgroup's waves can judge the window's entries side by side before the one wave that walks the state machine starts: a word
// per 64 consecutive entries, one ballot each.  Entry w needs w - 1 and w + 1 .. w + 11 inside the window.
__device__ __forceinline__ void lnw_build_bitmap(const double *cache, unsigned long long *bm, double ceiling, int tid)
{
    const int lane = tid & 63, wv = tid >> 6;
    for (int k = wv; k < LN_BMW; k += LN_WG / 64) {
        const int w = 64 * k + lane;
        const bool in = w >= 1 && w + 11 < LN_CACHE;
        const double t = in ? fabs(cache[w]) : 0.0, a = in ? fabs(cache[w - 1]) : 0.0, b = in ? fabs(cache[w + 1]) : 0.0;
        bool det = in & (t > ceiling) & (a <= t) & (b <= t);
        if (__ballot(det)) {
            double m = 0.0;
            if (in) {
                double v[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) v[i] = cache[w + 2 + i];
#pragma unroll
                for (int i = 0; i < 10; ++i) m = fmax(m, fabs(v[i]));
            }
            det = det && !(m > t);
        }
        const unsigned long long mk = __ballot(det);
        if (lane == 0) bm[k] = mk;
    }
}
// first detected entry in [from, to] (window entries, from >= 1); -1: none.  64 words per trip.
__device__ __forceinline__ int lnv_bitmap_first(const unsigned long long *bm, int from, int to, int lane)
{
    if (to >= LN_CACHE - 12) to = LN_CACHE - 13;
    for (int k0 = from >> 6; 64 * k0 <= to; k0 += 64) {
        const int k = k0 + lane;
        unsigned long long m = (k < LN_BMW && 64 * k <= to) ? bm[k] : 0ull;
        if (64 * k < from) m &= ~0ull << (from - 64 * k);             // (only the first word of the range can start inside it)
        if (64 * k + 63 > to) { const int keep = to - 64 * k + 1; m = keep >= 64 ? m : (keep <= 0 ? 0ull : m & ((1ull << keep) - 1ull)); }
        const unsigned long long any = __ballot(m != 0ull);
        if (any) {
            const int fl = __ffsll((long long)any) - 1;
            const int lo = __builtin_amdgcn_readlane((int)(unsigned)m, fl), hi = __builtin_amdgcn_readlane((int)(unsigned)(m >> 32), fl);
            const unsigned long long mw = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
            return 64 * (k0 + fl) + (__ffsll((long long)mw) - 1);
        }
    }
    return -1;
}
// The SUSTAIN batch on the bitmap: window entries [from, to]; returns l (last harmless detected entry before the first harmful one, -1:
// none), *hn / *hv the first harmful entry and its magnitude (-1: none), *lprev the detected entry before l (-1: none).
__device__ __forceinline__ int lnv_sustain_batch_bm(const double *cache, const unsigned long long *bm, int from, int to, double ceiling, double gr1, int lane,
                                                    int *lprev, int *hn, double *hv)
{
    int l = -1, lp = -1;
    *hn = -1; *hv = 0.0;
    for (int k0 = from >> 6; 64 * k0 <= to; k0 += 64) {
        const int k = k0 + lane;
        unsigned long long m = (k < LN_BMW && 64 * k <= to) ? bm[k] : 0ull;
        if (64 * k < from) m &= ~0ull << (from - 64 * k);
        if (64 * k + 63 > to) { const int keep = to - 64 * k + 1; m = keep >= 64 ? m : (keep <= 0 ? 0ull : m & ((1ull << keep) - 1ull)); }
        if (__ballot(m != 0ull) == 0ull) continue;
        // a lane walks the few peaks of its own word in order
        int first_h = -1, last_ok = -1, prev_ok = -1; double hval = 0.0;
        while (m) {
            const int pos = 64 * k + (__ffsll((long long)m) - 1);
            m &= m - 1ull;
            const double t = fabs(cache[pos]);
            if (ceiling / t < gr1) { first_h = pos; hval = t; break; }      // the filter's own comparison: gain_reduction < s->gain_reduction[1]
            prev_ok = last_ok; last_ok = pos;
        }
        const unsigned long long hb = __ballot(first_h >= 0);
        const int fl = hb ? __ffsll((long long)hb) - 1 : 63;
        const unsigned long long okb = __ballot(last_ok >= 0) & (fl >= 63 ? ~0ull : ((2ull << fl) - 1ull));
        if (okb) {
            const int hl = 63 - __clzll((long long)okb);
            const int nl = __builtin_amdgcn_readlane(last_ok, hl), np = __builtin_amdgcn_readlane(prev_ok, hl);
            const unsigned long long below = okb & ~(1ull << hl);
            lp = np >= 0 ? np : (below ? __builtin_amdgcn_readlane(last_ok, 63 - __clzll((long long)below)) : l);
            l = nl;
        }
        if (hb) { *hn = __builtin_amdgcn_readlane(first_h, fl); *hv = ln_rl(hval, fl); break; }
    }
    *lprev = lp;
    return l;
}
__device__ __forceinline__ void lnw_true_peak_limiter(double *ring, double *cache, const unsigned char *hot, LnState &s, double *out, int nb, double ceiling, int tid, bool batch_off,
                                                      unsigned long long *bm, LnPending *pend)
{
    const int index0 = s.lbi;
    LnRing R{ring, cache, index0};
    __syncthreads();
    unsigned long long lnp_a_ = 0, lnp_b_ = 0;
#ifdef JT_LN_PROFILE
    lnp_a_ = clock64();
#endif
    for (int n0 = tid; n0 < LN_CACHE; n0 += LN_WG * LN_NBF) {
        double t[LN_NBF];
#pragma unroll
        for (int u = 0; u < LN_NBF; ++u) { const int n = n0 + LN_WG * u; t[u] = n < LN_CACHE ? ln_ld(&ring[(index0 + LN_LO + n) % LN_LBS]) : 0.0; }
#pragma unroll
        for (int u = 0; u < LN_NBF; ++u) { const int n = n0 + LN_WG * u; if (n < LN_CACHE) cache[n] = t[u]; }
    }
    __syncthreads();
#ifdef JT_LN_PROFILE
    lnp_b_ = clock64(); if (tid == 0) lnp_acc[1] += lnp_b_ - lnp_a_;
#endif
    if (!batch_off) { lnw_build_bitmap(cache, bm, ceiling, tid); __syncthreads(); }
#ifdef JT_LN_PROFILE
    { const unsigned long long c_ = clock64(); if (tid == 0) lnp_acc[11] += c_ - lnp_b_; lnp_b_ = c_; }
#endif
    // the state machine itself is ONE wave's work (every step depends on the one before): wave 0 runs it on the window, the other
    // fifteen wait at the barrier below -- run by all sixteen alike it took four times as long, four waves sharing each SIMD
    if (tid < 64) {
    const int lane = tid;
    int smp_cnt = 0;
    // the bitmap describes the window as loaded; it stays true for the entries the scan has not reached as long as every envelope
    // segment is applied AT or BEHIND the scan position (the rule; see the batch's comment for the exception)
    bool bm_ok = !batch_off;
    // A segment [env_index, env_index + cnt) leaves alone what the scan has still to read when it starts at or behind the scan position --
    // r = its offset from the frame's first ring position, the scan stands at offset smp_cnt + 1920 -- or wholly behind the window, and
    // its wrapped part (a start far behind reaches round the ring) ends before the scan position too.  (Measured from the scan position
    // alone, "30 000 behind" passed for "behind": it is 10 320 AHEAD, inside the window.)
    auto seg_keeps_bitmap = [&](int cnt) {
        const int r = R.rel(s.env_index % LN_LBS);
        return (r <= smp_cnt + LN_LO || r - LN_LO >= LN_CACHE) && (r + cnt - LN_LBS <= smp_cnt + LN_LO);
    };
    int npend = 0;                                                  // (wave-uniform copy of pend->n)
    auto flush_pending = [&]() {
        for (int k = 0; k < npend; ++k) { const double g = pend->g[k]; lnv_ring_scale(R, ln_mod(index0 + LN_LO + pend->start[k]), pend->len[k], [&](int) { return g; }, lane); }
        npend = 0;
    };
    // a constant-gain segment: deferred when it lies inside the window and the detector never looks back (bm_ok), applied now otherwise
    auto sustain_scale = [&](int env_index, int cnt, double g) {
        if (cnt <= 0) return;
        const int r0 = R.rel(env_index % LN_LBS) - LN_LO;
        if (bm_ok && r0 >= 0 && r0 + cnt <= LN_CACHE && npend < 4) {
            if (lane == 0) { pend->start[npend] = r0; pend->len[npend] = cnt; pend->g[npend] = g; }
            ++npend;
            return;
        }
        flush_pending();                                               // (order: what was decided first is multiplied first)
        lnv_ring_scale(R, env_index, cnt, [&](int) { return g; }, lane);
    };
    // an envelope segment that is applied at once: pending segments it overlaps (it never does on the paths seen so far) go first
    auto before_immediate = [&](int env_index, int cnt) {
        if (!npend) return;
        const int a = R.rel(env_index % LN_LBS) - LN_LO;
        bool hit = false;
        for (int k = 0; k < npend; ++k) hit |= a < pend->start[k] + pend->len[k] && a + cnt > pend->start[k];
        hit |= a + cnt > LN_LBS - LN_LO;                               // (wraps round the ring: rare enough not to be worth the arithmetic)
        if (hit) flush_pending();
    };
    if (s.first) {
        const double mx = lnv_ring_absmax(R, 0, LN_ATT, lane);
        if (mx > ceiling) {
            s.gr1 = ceiling / mx;
            s.state = LIM_SUSTAIN;
            const double g = s.gr1;
            lnv_ring_scale(R, 0, LN_ATT, [&](int) { return g; }, lane);
        }
    }
    do {
        int n_first = 1;
        double pkv = 0.0; int pd = -1; bool have_pd = false;
        if (s.state == LIM_OUT && bm_ok) {
            // detect_peak(smp_cnt, nb - smp_cnt): the first detected entry among n = 1 .. nb - smp_cnt - 1, all inside the window
            const int e = lnv_bitmap_first(bm, smp_cnt + 1, nb - 1, lane);
            have_pd = true;
            if (e >= 0) { pd = e - smp_cnt; pkv = fabs(R.cache[e]); s.prev_smp = pkv; s.peak_index = ln_mod(s.lbi + LN_ATT + e); }
        }
        if (s.state == LIM_SUSTAIN && !batch_off) {
            // every harmless peak of the frame's remainder in one segment; the scan start is ring position lbi + smp_cnt + 1920 = window
            // entry smp_cnt.  Two conditions make the one-segment form the filter's own result.  (1) What the batch scans must be what the
            // per-peak walk would have scanned: the segment [env_index, env_index + l) must not reach into positions the scan still has to
            // read.  As a rule env_index IS the scan start (or trails it by 1920: the first frame's episode), and every sample is read
            // before it is scaled; but FINAL_FRAME refills the ring from position 0 with the limiter's state untouched, after which
            // env_index can stand AHEAD of the scan -- there the walk stays per peak.  (2) A detector call returns a peak at n <= nb - 1 only.
            const int Rb = min(nb - smp_cnt, nb - 1);
            if (seg_keeps_bitmap(Rb)) {
                int lprev = 0, lb = 0, hn = 0; double hv = 0.0;
                const int scan0 = smp_cnt;
                if (bm_ok) {
                    int lpe = -1, hne = -1;
                    const int le = lnv_sustain_batch_bm(R.cache, bm, scan0 + 1, scan0 + Rb, ceiling, s.gr1, lane, &lpe, &hne, &hv);
                    lb = le >= 0 ? le - scan0 : 0; lprev = lpe >= 0 ? lpe - scan0 : 0; hn = hne >= 0 ? hne - scan0 : 0;
                } else lb = lnv_sustain_batch(R.cache + smp_cnt, Rb, ceiling, s.gr1, lane, &lprev, &hn, &hv);
                n_first = Rb - lb + 1;                                                  // (nothing detected in (l, Rb])
                if (lb > 0) {
                    sustain_scale(s.env_index, lb, s.gr1);
                    s.prev_smp = fabs(R.cache[smp_cnt + lb]);
                    s.peak_index = ln_mod(s.lbi + smp_cnt + LN_ATT + lb);
                    s.env_index = (s.env_index + lb) % LN_LBS; s.env_cnt = lb - lprev; smp_cnt += lb;
                }
                if (hn > 0) {
                    // the detector call from l returns h: peak_delta = h - l, and its reduction is the stronger one (the batch's own test)
                    const int pdh = hn - lb;
                    s.prev_smp = hv; s.peak_index = ln_mod(s.lbi + scan0 + LN_ATT + hn);
                    s.state = LIM_ATTACK;
                    s.attack_length = pdh; if (s.attack_length <= 1) s.attack_length = 2;
                    s.gr0 = s.gr1; s.gr1 = ceiling / hv; s.env_cnt = 0;
                    continue;
                }
                if (smp_cnt >= nb) continue;
                if (bm_ok) {
                    // the call from l scans n = 1 .. nb - 1 from there: what lies behind the frame's remainder, as far as the window reaches
                    const int lim = smp_cnt + nb - 1, wlim = min(lim, LN_CACHE - 13);
                    const int e = lnv_bitmap_first(bm, scan0 + Rb + 1, wlim, lane);
                    if (e >= 0) { have_pd = true; pd = e - smp_cnt; pkv = fabs(R.cache[e]); s.prev_smp = pkv; s.peak_index = ln_mod(s.lbi + LN_ATT + e); }
                    else if (wlim == lim) have_pd = true;                               // (pd = -1: release)
                    else n_first = wlim - smp_cnt + 1;
                }
            }
        }
        // (one call site for the detector: OUT scans what is left of the frame, SUSTAIN a frame's length from where it stands)
        if (!have_pd && (s.state == LIM_OUT || s.state == LIM_SUSTAIN)) pd = lnv_detect_peak(R, hot, s, smp_cnt, s.state == LIM_OUT ? nb - smp_cnt : nb, ceiling, &pkv, lane, n_first);
        switch (s.state) {
        case LIM_OUT: {
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
            if (!seg_keeps_bitmap(cnt)) { bm_ok = false; flush_pending(); }
            before_immediate(s.env_index, cnt);
            lnv_ring_scale(R, s.env_index, cnt, [&](int j) { return g0 - ((double)(c0 + j) / (al - 1) * (g0 - g1)); }, lane);
            s.env_index = (s.env_index + cnt) % LN_LBS; s.env_cnt += cnt; smp_cnt += cnt;
            if (smp_cnt < nb) { s.env_cnt = 0; s.attack_length = LN_ATT; s.state = LIM_SUSTAIN; }
            break; }
        case LIM_SUSTAIN: {
            if (pd == -1) { s.state = LIM_RELEASE; s.gr0 = s.gr1; s.gr1 = 1.; s.env_cnt = 0; break; }
            const double gr = ceiling / pkv;
            if (gr < s.gr1) {
                s.state = LIM_ATTACK;
                s.attack_length = pd; if (s.attack_length <= 1) s.attack_length = 2;
                s.gr0 = s.gr1; s.gr1 = gr; s.env_cnt = 0;
                break;
            }
            int cnt = pd; if (cnt > nb - smp_cnt) cnt = nb - smp_cnt; if (cnt < 0) cnt = 0;
            if (!seg_keeps_bitmap(cnt)) { bm_ok = false; flush_pending(); }
            sustain_scale(s.env_index, cnt, s.gr1);
            s.env_index = (s.env_index + cnt) % LN_LBS; s.env_cnt = cnt; smp_cnt += cnt;
            break; }
        case LIM_RELEASE: {
            int cnt = LN_REL - s.env_cnt; if (cnt > nb - smp_cnt) cnt = nb - smp_cnt; if (cnt < 0) cnt = 0;
            const double g0 = s.gr0, g1 = s.gr1; const int c0 = s.env_cnt;
            if (!seg_keeps_bitmap(cnt)) { bm_ok = false; flush_pending(); }
            before_immediate(s.env_index, cnt);
            lnv_ring_scale(R, s.env_index, cnt, [&](int j) { return g0 + (((double)(c0 + j) / (LN_REL - 1)) * (g1 - g0)); }, lane);
            s.env_index = (s.env_index + cnt) % LN_LBS; s.env_cnt += cnt; smp_cnt += cnt;
            if (smp_cnt < nb) { s.env_cnt = 0; s.state = LIM_OUT; }
            break; }
        }
    } while (smp_cnt < nb);
    if (lane == 0) pend->n = npend;
    }
    __syncthreads();
    {   // the pending SUSTAIN segments, all waves (disjoint stretches of the window: any order among them gives the same products)
        const int np = pend->n;
        for (int k = 0; k < np; ++k) {
            const int st = pend->start[k], len = pend->len[k]; const double g = pend->g[k];
            for (int j = tid; j < len; j += LN_WG) cache[st + j] *= g;
        }
        if (np) __syncthreads();
    }
#ifdef JT_LN_PROFILE
    lnp_a_ = clock64(); if (tid == 0) lnp_acc[2] += lnp_a_ - lnp_b_;
#endif
    // the window goes back into the ring -- the envelope segments run 1920 samples AHEAD of the output position, so the look-ahead part
    // has been edited too -- and the frame goes out, clamped
    for (int n = tid; n < LN_CACHE; n += LN_WG) ln_st(&ring[(index0 + LN_LO + n) % LN_LBS], cache[n]);
    for (int n0 = tid; n0 < nb; n0 += LN_WG * LN_NBF) {
        double t[LN_NBF];
#pragma unroll
        for (int u = 0; u < LN_NBF; ++u) {
            const int n = n0 + LN_WG * u;
            t[u] = n >= nb ? 0.0 : (n >= LN_LO ? cache[n - LN_LO] : ln_ld(&ring[(index0 + n) % LN_LBS]));
        }
#pragma unroll
        for (int u = 0; u < LN_NBF; ++u) {
            const int n = n0 + LN_WG * u;
            if (n >= nb) continue;
            double v = t[u];
            if (fabs(v) > ceiling) v = ceiling * (v < 0 ? -1 : 1);
            out[n] = v;
        }
    }
    __threadfence();
    __syncthreads();
#ifdef JT_LN_PROFILE
    if (tid == 0) lnp_acc[3] += clock64() - lnp_a_;
#endif
    (void)lnp_a_; (void)lnp_b_;
}

__global__ void __launch_bounds__(LN_WG)
k_loudnorm_dynamic_wg(const double *__restrict__ x, int64_t n, LoudnormDynParams P, const double *__restrict__ series, double *__restrict__ ring,
                      double *__restrict__ y, double *__restrict__ dbg, double *__restrict__ carry, int64_t it_begin, int64_t it_end)
{
    // One launch walks the steps [it_begin, it_end) of the file (step 0 = FIRST_FRAME, then the INNER_FRAMEs, then the FINAL_FRAME's 29)
    // and hands everything it knows to the next launch through `carry` (LN_CARRY doubles).  A single launch for the whole file ran for
    // a second and more, and a kernel that long holds up every stream ROCclr has mapped onto its hardware queue (other files' passes on
    // the same GPU); a stream of the high-priority pool avoided that but brought the process to 24 hardware queues, which the driver
    // time-slices: everything else on the GPU ran at half speed from then on.
    extern __shared__ double ln_cache[];                            // [LN_CACHE]; the output meter's tile aliases its head
    __shared__ double delta[30], w[21], oe[30];
    __shared__ unsigned char hot[LN_LBS / 64];
    __shared__ LnPending ln_pend;
    __shared__ unsigned long long ln_bm[LN_BMW];                    // the frame's detected peaks (lnw_build_bitmap)
    double *tile = ln_cache;
    const int tid = threadIdx.x;
    if (it_begin > 0) {
        // carry[84] = the next step to run: the stream path (below) may have covered this launch's steps, or some of them, meanwhile
        const int64_t next_it = (int64_t)carry[84];
        if (it_end <= next_it) return;
        if (it_begin < next_it) it_begin = next_it;
    }
    const bool fresh = it_begin == 0;
    if (tid < 30) { delta[tid] = fresh ? P.delta0 : carry[16 + tid]; oe[tid] = fresh ? 0.0 : carry[46 + tid]; }
    if (tid < 21) w[tid] = P.weights[tid];
    if (!fresh) for (int k = tid; k < LN_LBS / 64; k += LN_WG) hot[k] = reinterpret_cast<const unsigned char *>(carry + 96)[k];
#ifdef JT_LN_PROFILE
    if (tid < 16) lnp_acc[tid] = 0;
#endif
    __syncthreads();
    const double ceiling = P.target_tp_lin, offset = P.offset_lin;
    LnState s; s.gr0 = 1.; s.gr1 = 1.; s.prev_smp = 0.; s.lbi = 0; s.state = LIM_OUT; s.peak_index = 0; s.env_index = 0; s.env_cnt = 0;
    s.attack_length = LN_ATT; s.first = true;
    int index = 1, above = P.above0;
    double prev_delta = P.delta0;
    double kv1 = 0, kv2 = 0, kv3 = 0, kv4 = 0;                      // the output meter's filter state (thread 0)
    int oe_pos = 0;
    int64_t produced = 0, abs_in = LN_LBS, src_index = LN_LBS;
    double gain_final = 0.0;
    if (!fresh) {
        s.gr0 = carry[0]; s.gr1 = carry[1]; s.prev_smp = carry[2]; s.lbi = (int)carry[3]; s.state = (int)carry[4]; s.peak_index = (int)carry[5];
        s.env_index = (int)carry[6]; s.env_cnt = (int)carry[7]; s.attack_length = (int)carry[8]; s.first = carry[9] != 0.0;
        index = (int)carry[10]; above = (int)carry[11]; prev_delta = carry[12]; oe_pos = (int)carry[13];
        kv1 = carry[76]; kv2 = carry[77]; kv3 = carry[78]; kv4 = carry[79];
        produced = (int64_t)carry[80]; abs_in = (int64_t)carry[81]; src_index = (int64_t)carry[82]; gain_final = carry[83];
    }
    auto out_energy = [&](const double *src, int cnt) {
        double sum = 0.0;
        for (int c0 = 0; c0 < cnt; c0 += 1024) {
            const int m = min(1024, cnt - c0);
            __syncthreads();
            for (int j = tid; j < m; j += LN_WG) tile[j] = src[c0 + j];
            __syncthreads();
            if (tid == 0) {
                for (int j = 0; j < m; ++j) {
                    const double v0 = tile[j] - P.kwa[1] * kv1 - P.kwa[2] * kv2 - P.kwa[3] * kv3 - P.kwa[4] * kv4;
                    const double o = P.kwb[0] * v0 + P.kwb[1] * kv1 + P.kwb[2] * kv2 + P.kwb[3] * kv3 + P.kwb[4] * kv4;
                    kv4 = kv3; kv3 = kv2; kv2 = kv1; kv1 = v0;
                    sum += o * o;
                }
            }
        }
        __syncthreads();
        if (tid == 0) { oe[oe_pos] = sum; }
        oe_pos = (oe_pos + 1) % 30;
        __syncthreads();
    };
    // FIRST_FRAME (the first 210 ms with the initial gain, 100 ms out), the INNER_FRAMEs, the FINAL_FRAME's 29 steps: ONE loop with ONE
    // call of the limiter -- inlined at three call sites the kernel was 108 KB of code, more than the instruction cache holds, and the
    // one wave that walks the state machine waited for instruction fetches
    const int nfinal = (int)(P.final_len / LN_F100);
    const int64_t total = 1 + P.n_inner + nfinal;
    const double *fsrc = x + (n - P.final_len);
    for (int64_t it = it_begin; it < total && it < it_end; ++it) {
        int nb = LN_F100;
        if (it == 0) {
            const double d0 = P.delta0;
            lnw_ring_fill(ring, 0, x, n, LN_LBS, [&](double v, int) { return v * d0 * offset; }, tid, hot, ceiling);
        } else if (it <= P.n_inner) {
            const int64_t k = it - 1;
            nb = (int)min<int64_t>(LN_F100, n - LN_F3000 - k * LN_F100);
            const double gain = ln_gaussian(delta, w, index + 10 < 30 ? index + 10 : index + 10 - 30);
            const double gain_next = ln_gaussian(delta, w, index + 11 < 30 ? index + 11 : index + 11 - 30);
            lnw_ring_fill(ring, s.lbi, x + abs_in, n - abs_in, nb, [&](double v, int j) { return v * (gain + (((double)j / nb) * (gain_next - gain))) * offset; }, tid, hot, ceiling);
            s.lbi = (s.lbi + nb) % LN_LBS;
            s.lbi = (s.lbi + (LN_F100 - nb)) % LN_LBS;
        } else if (it == P.n_inner + 1) {
            // FINAL_FRAME: the last 2.9 s again from the look-ahead buffer, one gain
            gain_final = ln_gaussian(delta, w, index + 10 < 30 ? index + 10 : index + 10 - 30);
            const double gain = gain_final;
            s.lbi = 0;
            lnw_ring_fill(ring, 0, fsrc, P.final_len, LN_LBS, [&](double v, int) { return v * gain * offset; }, tid, hot, ceiling);
        }
        lnw_true_peak_limiter(ring, ln_cache, hot, s, y + produced, nb, ceiling, tid, P.no_batch != 0, ln_bm, &ln_pend);
        if (it == 0) {
            s.first = false;
            if (!above) out_energy(y, LN_F100);
            produced = LN_F100;
        } else if (it <= P.n_inner) {
            const int64_t k = it - 1;
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
            if (tid == 0) delta[index] = d;
            __syncthreads();
            prev_delta = d;
            index++; if (index >= 30) index -= 30;
            produced += nb; abs_in += nb;
        } else {
            const double gain = gain_final;
            lnw_ring_fill(ring, s.lbi, fsrc + src_index, P.final_len - src_index, LN_F100, [&](double v, int) { return v * gain * offset; }, tid, hot, ceiling);
            src_index = min<int64_t>(src_index + LN_F100, P.final_len);
            s.lbi = (s.lbi + LN_F100) % LN_LBS;
            produced += LN_F100;
        }
    }
    if (tid == 0 && dbg) { dbg[0] = (double)produced; dbg[1] = (double)above; dbg[2] = (double)s.state; dbg[3] = prev_delta; }
    // hand-over to the next launch (thread 0 is in the wave that ran the limiter: its copy of the state is the live one)
    __syncthreads();
    if (tid == 0) {
        carry[0] = s.gr0; carry[1] = s.gr1; carry[2] = s.prev_smp; carry[3] = s.lbi; carry[4] = s.state; carry[5] = s.peak_index;
        carry[6] = s.env_index; carry[7] = s.env_cnt; carry[8] = s.attack_length; carry[9] = s.first ? 1.0 : 0.0;
        carry[10] = index; carry[11] = above; carry[12] = prev_delta; carry[13] = oe_pos;
        carry[76] = kv1; carry[77] = kv2; carry[78] = kv3; carry[79] = kv4;
        carry[80] = (double)produced; carry[81] = (double)abs_in; carry[82] = (double)src_index; carry[83] = gain_final;
        carry[84] = (double)min<int64_t>(total, it_end);
    }
    if (tid < 30) { carry[16 + tid] = delta[tid]; carry[46 + tid] = oe[tid]; }
    for (int k = tid; k < LN_LBS / 64; k += LN_WG) reinterpret_cast<unsigned char *>(carry + 96)[k] = hot[k];
#ifdef JT_LN_PROFILE
    __syncthreads();
    if (tid < 16) g_ln_prof[tid] = lnp_acc[tid];
#endif
}

// =====================================================================================================================
// Round 5, the STREAM PATH: the steady state of the dynamic mode as data-parallel sweeps around a state machine that touches no samples.
//
// Once the "not yet above threshold" phase is over (above_threshold is a latch) nothing the limiter does feeds back into the frame gains:
// delta[k] is a function of the INPUT meter's series alone.  And between the first frame's special case and the flush the limiter's ring
// is a sliding window over ONE linear stream L[t] = x[t] * gain(t) * offset: ring position = t mod 40320, the envelope position is always
// the scan position (10 ms ahead of the output position), every detector call reads samples no envelope has reached yet.  So for the
// full INNER frames [ka, kb) after a launch of the workgroup kernel has handed its state over:
//   k_lns_begin    the frame gains of all those frames (the candidate deltas side by side, the "keep the previous one" chain by one thread)
//   k_lns_fill     L into the OUTPUT buffer, whole GPU (the ring's current 110 ms are copied in front of it: they may carry envelope already)
//   k_lns_bitmap / _scan / _scatter   the detected peaks of L -- a pure function of the samples (lnv_sustain_batch's comment) -- as a
//                  sorted list (time, magnitude)
//   k_lns_machine  ONE wave walks af_loudnorm's OUT / ATTACK / SUSTAIN / RELEASE machine frame by frame over that LIST (64 entries at a
//                  time in registers) and writes the envelope segments it decides on to a second list: no sample is read or written
//   k_lns_apply    the segments multiplied into L, whole GPU.  A sample is multiplied at most twice -- by the tail of a RELEASE and then
//                  by the head of the ATTACK of a peak detected less than 10 ms after the release ended -- and floating-point products
//                  do not commute, so those attacks are a second layer applied after everything else
//   k_lns_finish   clamp, and the window the next frame needs goes back into the ring for the workgroup kernel (last partial frame, flush)
// What the list cannot know: a detector call of SUSTAIN that starts in a frame's last ten samples and finds nothing for 100 ms reads up to
// 11 samples PAST the ring's 210 ms -- the filter wraps to the frame's first samples, which the envelope has already edited.  The machine
// checks that corner (is any sample of those last twelve above the ceiling?) and, if so, stops BEFORE that frame: what it has is
// committed, the workgroup kernel carries on from there, and a later attempt takes over again.  Same for a full segment or peak list.
// Everything is enqueued without a host round trip: every kernel of an attempt looks at the control block the first one fills.
constexpr int LNS_BW = 64;                                   // bitmap words per block of k_lns_bitmap (4096 samples)
constexpr int LNS_MAXF = 100000;                             // frames per attempt: 1.92e9 samples, what the list's 32-bit relative times hold
__device__ __forceinline__ long long lns_rl64(long long v, int l)
{
    return ((long long)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, l);
}
__global__ void __launch_bounds__(256)
k_lns_begin(const double *__restrict__ carry, const double *__restrict__ series, LoudnormDynParams P, LnsBufs B, int ka, int kb)
{
    __shared__ int go;
    __shared__ double chunk[256];
    __shared__ double w[21];
    __shared__ double prev_s;
    const int tid = threadIdx.x;
    if (tid == 0) {
        LnsCtl *c = B.ctl;
        const int lbi = (int)carry[3], state = (int)carry[4];
        bool ok = carry[84] == (double)(ka + 1) && carry[11] == 1.0 && carry[9] == 0.0;
        ok = ok && lbi == (int)(((long long)ka * LN_F100) % LN_LBS);                             // (the frame's fill advances it to its output position)
        if (state != LIM_OUT) ok = ok && (int)carry[6] == (lbi + LN_F100 + LN_ATT) % LN_LBS;     // the envelope stands at the scan position
        if (state != LIM_ATTACK) ok = ok && (int)carry[8] == LN_ATT;
        c->active = ok ? 1 : 0; c->ok = 0; c->attempts += 1;
        if (ok) { c->ka = ka; c->kbe = ka; c->npk = 0; c->nseg = 0; c->why = 0; c->tbase = (long long)(ka + 1) * LN_F100; }   // (the diagnostics are the last ACTIVE attempt's)
        else if (c->frames == 0) { c->why = 1; c->why_mask |= 2; }
        go = ok ? 1 : 0;
        prev_s = carry[12];
    }
    if (tid < 21) w[tid] = P.weights[tid];
    __syncthreads();
    if (!go) return;
    const int index = (int)carry[10], nf = kb - ka;
    if (tid < 30) B.E[tid] = carry[16 + (index + tid) % 30];                                  // the 30 most recent deltas, oldest first
    // delta[k]: the INNER_FRAME branch of the workgroup kernel with above == 1 (a candidate, or "keep the previous one")
    for (int j0 = 0; j0 < nf; j0 += 256) {
        const int j = j0 + tid;
        double d = -1.0;
        if (j < nf) {
            const int k = ka + j;
            const double shortterm = series[3 * (long long)k], global = series[3 * (long long)k + 1], relthr = series[3 * (long long)k + 2];
            if (!(shortterm < relthr || shortterm <= -70.)) {
                const double env_global = fabs(shortterm - global) < (P.target_lra / 2.) ? shortterm - global
                                                                                         : (P.target_lra / 2.) * ((shortterm - global) < 0 ? -1 : 1);
                const double env_shortterm = P.target_i - shortterm;
                d = pow(10., (env_global + env_shortterm) / 20.);
            }
        }
        chunk[tid] = d;
        __syncthreads();
        if (tid == 0) {
            double prev = prev_s;
            const int m = min(256, nf - j0);
            for (int q = 0; q < m; ++q) { double v = chunk[q]; if (v < 0.0) v = prev; chunk[q] = v; prev = v; }
            prev_s = prev;
        }
        __syncthreads();
        if (j < nf) B.E[30 + j] = chunk[tid];
        __syncthreads();
    }
    __threadfence();
    __syncthreads();
    // gain / gain_next of frame ka + j: ln_gaussian over the ring as it stands BEFORE the frame = E[j .. j + 30), start at its oldest entry
    for (int j = tid; j < nf; j += 256) {
        double g = 0., gn = 0.;
        for (int i = 0; i < 21; i++) g += B.E[j + i] * w[i];
        for (int i = 0; i < 21; i++) gn += B.E[j + 1 + i] * w[i];
        B.G[ka + j] = g; B.Gn[ka + j] = gn;
    }
}
// L[t] for the fills of frames [ka, kb) (frame k fills t = LBS + k F100 + j with the ramp from gain to gain_next), and in front of it the
// ring as the workgroup kernel left it: t in [(ka + 1) F100, (ka + 1) F100 + LBS - F100)
__global__ void __launch_bounds__(256)
k_lns_fill(const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ ring, LnsBufs B, int ka, int kb, double offset)
{
    if (!B.ctl->active) return;
    const long long nfill = (long long)(kb - ka) * LN_F100, nring = LN_LBS - LN_F100;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nfill + nring; i += (long long)gridDim.x * 256)
    if (i < nfill) {
        const int k = ka + (int)(i / LN_F100), j = (int)(i % LN_F100);
        const int nb = LN_F100;
        const double gain = B.G[k], gain_next = B.Gn[k];
        const long long t = (long long)LN_LBS + (long long)ka * LN_F100 + i;
        y[t] = x[t] * (gain + (((double)j / nb) * (gain_next - gain))) * offset;
    } else if (i < nfill + nring) {
        const long long t = (long long)(ka + 1) * LN_F100 + (i - nfill);
        y[t] = ring[t % LN_LBS];
    }
}
// detected peaks of L[tb0, tb1) as a bitmap (bit b of word w: t = 64 w + b), the words' popcounts as offsets inside their block of 64
// words and the block's total; lnw_build_bitmap's test on the linear stream
__global__ void __launch_bounds__(256)
k_lns_bitmap(const double *__restrict__ y, LnsBufs B, long long tb0, long long tb1, double ceiling, long long blk0, int nblk)
{
    if (!B.ctl->active) return;
    __shared__ int wc[LNS_BW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int bq = blockIdx.x; bq < nblk; bq += gridDim.x) {
    const long long blk = blk0 + bq;
    __syncthreads();
    for (int i = 0; i < LNS_BW / 4; ++i) {
        const int wl = wv * (LNS_BW / 4) + i;
        const long long wd = blk * LNS_BW + wl, t = wd * 64 + lane;
        const bool in = t >= tb0 && t < tb1;
        const double v = in ? fabs(y[t]) : 0.0, a = in ? fabs(y[t - 1]) : 0.0, b = in ? fabs(y[t + 1]) : 0.0;
        bool det = in & (v > ceiling) & (a <= v) & (b <= v);
        if (__ballot(det)) {
            double m = 0.0;
            if (in) {
                double u[10];
#pragma unroll
                for (int q = 0; q < 10; ++q) u[q] = y[t + 2 + q];
#pragma unroll
                for (int q = 0; q < 10; ++q) m = fmax(m, fabs(u[q]));
            }
            det = det && !(m > v);
        }
        const unsigned long long mk = __ballot(det);
        if (lane == 0) { B.bm[wd] = mk; wc[wl] = __popcll(mk); }
    }
    __syncthreads();
    if (tid < 64) {
        const int c = wc[tid];
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (tid >= d) incl += o; }
        B.woff[blk * LNS_BW + tid] = (unsigned short)(incl - c);
        if (tid == 63) B.bcnt[bq] = incl;
    }
    }
}
__global__ void __launch_bounds__(1024)
k_lns_scan(LnsBufs B, int nblk)
{
    if (!B.ctl->active) return;
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (nblk + 1023) / 1024, lo = min(nblk, tid * per), hi = min(nblk, lo + per);
    int sum = 0;
    for (int q = lo; q < hi; ++q) sum += B.bcnt[q];
    part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const int o = tid >= d ? part[tid - d] : 0; __syncthreads(); part[tid] += o; __syncthreads(); }
    int run = part[tid] - sum;
    for (int q = lo; q < hi; ++q) { B.boff[q] = run; run += B.bcnt[q]; }
    if (tid == 1023) {
        const int total = part[1023];
        B.ctl->npk = total;
        if (total > B.pk_cap) { B.ctl->active = 0; B.ctl->why = 2; B.ctl->why_mask |= 4; }
    }
}
__global__ void __launch_bounds__(256)
k_lns_scatter(const double *__restrict__ y, LnsBufs B, long long blk0, int nblk, double ceiling)
{
    if (!B.ctl->active) return;
    const long long tbase = B.ctl->tbase;                          // (list times are relative to the attempt's first output sample: 32 bits)
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < (long long)nblk * LNS_BW; q += (long long)gridDim.x * 256) {   // q: word of the range
    const long long wd = blk0 * LNS_BW + q;
    unsigned long long m = B.bm[wd];
    long long off = (long long)B.boff[q / LNS_BW] + B.woff[wd];
    while (m) {
        const long long t = wd * 64 + (__ffsll((long long)m) - 1);
        m &= m - 1ull;
        const double v = fabs(y[t]);
        B.pk_t[off] = (int)(t - tbase); B.pk_v[off] = v; B.pk_g[off] = ceiling / v; ++off;                // (ceiling / peak_value: the filter's gain reduction of this peak)
    }
    }
}
// per frame of the attempt: what its scan range holds of the list (LnsFrame).  One thread per frame, two binary searches and a walk over the
// frame's own ~64 entries.
__global__ void __launch_bounds__(256)
k_lns_frames(LnsBufs B, int ka, int kb)
{
    if (!B.ctl->active) return;
    const int npk = B.ctl->npk;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < kb - ka; f += gridDim.x * 256) {
        const int ts = f * LN_F100 + LN_ATT;                          // relative to tbase
        auto first_after = [&](int t) {                               // first entry with time > t
            int lo = 0, hi = npk;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (B.pk_t[mid] > t) hi = mid; else lo = mid + 1; }
            return lo;
        };
        LnsFrame r;
        r.fi = first_after(ts); r.fe = first_after(ts + LN_F100 - 1);
        double m = 2.0;
        for (int i = r.fi; i < r.fe; ++i) m = fmin(m, B.pk_g[i]);
        r.ming = m;
        r.l_t = r.fe > r.fi ? B.pk_t[r.fe - 1] : -1;
        r.e_t = r.fe < npk ? B.pk_t[r.fe] : 0x7fffffff;
        r.e_g = r.fe < npk ? B.pk_g[r.fe] : 2.0;
        B.fr[f] = r;
    }
}
// the limiter's state machine over the peak list: lnw_true_peak_limiter's loop with the bitmap queries answered by the list and the ring
// operations written down as segments.  Times are sample positions relative to the attempt's first output sample (32 bits: an attempt
// covers at most LNS_MAXF frames); the frame with output position T0 scans from T0 + c + 1920
// (c = the filter's smp_cnt), which is also where its envelope stands.
__global__ void __launch_bounds__(64)
k_lns_machine(double *__restrict__ carry, LoudnormDynParams P, LnsBufs B, const double *__restrict__ y, int ka, int kb)
{
    LnsCtl *ctl = B.ctl;
    if (!ctl->active) return;
    const int lane = threadIdx.x;
    const int npk = ctl->npk;
    const double ceiling = P.target_tp_lin;
    const int nb = LN_F100, stop_every = P.stream_stop, seg_cap = B.seg_cap;
    const long long tbase = ctl->tbase;
    double gr0 = carry[0], gr1 = carry[1];
    int state = (int)carry[4], env_cnt = (int)carry[7], att = (int)carry[8];
    int last_pk = -1;                                                    // list entry of the last detected peak (prev_smp / peak_index of the hand-over)
    bool layer2 = false;
    int nseg = 0;
    // One wave runs alone and issues an instruction every five cycles: the machine is bound by its instruction count (2 400 cycles per step
    // with 64-bit times and a division per window; a memory round trip hidden or not made no difference).  Hence 32-bit times (relative
    // to the attempt's first output sample; an attempt covers at most LNS_MAXF frames), the peaks' gain reductions ceiling / |peak| precomputed by the
    // scatter kernel (the only thing the machine ever does with a magnitude), and the list through LDS: 4096 entries, refilled 1024 at a
    // time with all of a refill's loads in flight together.  One wave: its LDS operations execute in order, no barrier.
    constexpr int LCAP = 2048, LCH = 512, TINF = 0x7fffffff;
    __shared__ int lt[LCAP];
    __shared__ double lg[LCAP];
    // the frames' records (k_lns_frames), 512 at a time the same way: a frame SUSTAIN passes costs the machine one of these and no list entry
    constexpr int FCH = 512;
    __shared__ LnsFrame lfr[FCH];
    int fr_lo = 0, fr_hi = 0;                                            // records of frames [fr_lo, fr_hi) (relative to ka) are in LDS
    int wbase = 0, wt = 0, cur = 0, lhi = 0;                             // window: entry wbase + lane in this lane's registers; [.., lhi) in LDS
    double wg = 0.0;
    int n_refill = 0, n_iter = 0;
    const long long clk0 = clock64();
    auto refill = [&](int b) {
        while (b + 64 > lhi) {
            // (b >= cur: everything before b is dead, so [lhi, lhi + LCH) fits whenever lhi - b <= LCAP - LCH, and lhi < b + 64 here)
            int tt[LCH / 64]; double gg[LCH / 64];
#pragma unroll
            for (int u = 0; u < LCH / 64; ++u) { const int i = lhi + 64 * u + lane; tt[u] = i < npk ? B.pk_t[i] : TINF; gg[u] = i < npk ? B.pk_g[i] : 2.0; }
#pragma unroll
            for (int u = 0; u < LCH / 64; ++u) { const int q = (lhi + 64 * u + lane) & (LCAP - 1); lt[q] = tt[u]; lg[q] = gg[u]; }
            lhi += LCH; ++n_refill;
        }
        const int q = (b + lane) & (LCAP - 1);
        wt = lt[q]; wg = lg[q]; wbase = b;
    };
    refill(0);
    // cur := the first entry later than ts (cur never goes back: ts does not between detector calls)
    auto advance = [&](int ts) {
        for (;;) {
            if (cur >= wbase + 64 || cur < wbase) refill(cur);
            const int off = cur - wbase;
            const unsigned long long m = __ballot(wt > ts) & (~0ull << off);
            if (m) { cur = wbase + (__ffsll((long long)m) - 1); return; }
            cur = wbase + 64;
        }
    };
    bool ovf = false;
    auto emit = [&](int kind, int t, int len, double g0, double g1, int c0, int al) {
        if (len <= 0) return;
        if (nseg >= seg_cap) { ovf = true; return; }
        if (lane == 0) { LnsSeg sg; sg.t = t; sg.len = len; sg.kind = kind | (layer2 && kind == 1 ? 256 : 0); sg.c0 = c0; sg.al = al; sg.g0 = g0; sg.g1 = g1; B.seg[nseg] = sg; }
        ++nseg;
    };
    int k = ka, why = 0;
    for (; k < kb; ++k) {
        if (stop_every > 0 && k > ka && k % stop_every == 0) { why = 5; break; }              // (test switch)
        const int T0 = (k - ka) * LN_F100;                                                     // relative to tbase, like the list's times
        // the frame is taken whole or not at all
        const double s_gr0 = gr0, s_gr1 = gr1; const int s_state = state, s_lp = last_pk, s_ec = env_cnt, s_att = att, s_nseg = nseg;
        const bool s_l2 = layer2;
        bool hazard = false;
        int c = 0, guard = 0;
        if (state == LIM_SUSTAIN) {
            // The frame SUSTAIN passes: its scan range [ts + 1, ts + nb - 1] holds peaks, all harmless (batch: one segment up to the last
            // one, l), the next peak lies within nb - 1 of l and is harmless too (the detector call behind the remainder: the gain held to
            // the frame's end), and that call stays clear of the ring's last twelve samples.  Two segments of the same gain that meet are
            // one: the whole frame at gr1.  Everything it needs is the frame's record.
            const int f = k - ka;
            if (f >= fr_hi) {
                fr_lo = f; fr_hi = min(kb - ka, f + FCH);
                for (int q = lane; q < fr_hi - fr_lo; q += 64) lfr[q] = B.fr[fr_lo + q];
            }
            const LnsFrame r = lfr[f - fr_lo];
            const int ts = T0 + LN_ATT, lb = r.l_t - ts;
            if (r.fe > r.fi && r.ming >= gr1 && lb < LN_F100 - 11 && r.e_t <= r.l_t + nb - 1 && r.e_g >= gr1) {
                emit(0, ts, nb, gr1, 0., 0, 0);
                last_pk = r.fe; env_cnt = nb - lb; cur = r.fe;
                c = nb; ++n_iter;
            }
        }
        if (c < nb)
        do {
            const int ts = T0 + c + LN_ATT;
            ++n_iter;
            // (every step consumes a sample or changes state towards one that does: a frame cannot take more steps than it has samples.
            //  A frame that does is handed to the workgroup kernel like the ring-end corner -- never a spinning wave)
            if (++guard > 2 * LN_F100) { hazard = true; break; }
            switch (state) {
            case LIM_OUT: {
                // detect_peak(c, nb - c): the first detected peak among n = 1 .. nb - c - 1
                advance(ts);
                const int off = cur - wbase;
                const int e = __builtin_amdgcn_readlane(wt, off);
                if (e <= ts + (nb - c) - 1) {
                    const int pd = e - ts;
                    last_pk = cur;
                    env_cnt = 0;
                    layer2 = pd < att;                                                          // the attack starts before the scan position: over a release's tail
                    c += pd - att;
                    gr0 = 1.; gr1 = ln_rl(wg, off);                                              // ceiling / peak_value
                    state = LIM_ATTACK;
                } else c = nb;
                break; }
            case LIM_ATTACK: {
                int cnt = att - env_cnt; if (cnt > nb - c) cnt = nb - c; if (cnt < 0) cnt = 0;
                emit(1, ts, cnt, gr0, gr1, env_cnt, att);
                env_cnt += cnt; c += cnt;
                if (c < nb) { env_cnt = 0; att = LN_ATT; state = LIM_SUSTAIN; layer2 = false; }
                break; }
            case LIM_SUSTAIN: {
                // the batch (lnv_sustain_batch_bm): the harmless peaks of [ts + 1, ts + Rb] in one segment, the first harmful one into ATTACK
                const int Rb = min(nb - c, nb - 1);
                int l_t = -1, lp_t = -1, h_t = -1, l_i = -1, h_i = -1; double h_g = 0.0;
                advance(ts);
                for (int scan = cur;;) {
                    if (scan >= wbase + 64 || scan < wbase) refill(scan);
                    const int off = scan - wbase;
                    const bool inr = wt <= ts + Rb && lane >= off;
                    const bool harm = inr && (wg < gr1);                                          // the filter's own comparison: gain_reduction < s->gain_reduction[1]
                    const unsigned long long im = __ballot(inr), hm = __ballot(harm);
                    const unsigned long long okm = hm ? im & ((1ull << (__ffsll((long long)hm) - 1)) - 1ull) : im;
                    if (okm) {
                        const int hi = 63 - __clzll((long long)okm);
                        const unsigned long long rest = okm & ~(1ull << hi);
                        lp_t = rest ? __builtin_amdgcn_readlane(wt, 63 - __clzll((long long)rest)) : l_t;
                        l_t = __builtin_amdgcn_readlane(wt, hi); l_i = wbase + hi;
                    }
                    // (cur follows the scan: what it has passed lies at or before the next scan position, and a window left behind
                    //  would have to be fetched again)
                    if (hm) { const int f = __ffsll((long long)hm) - 1; h_t = __builtin_amdgcn_readlane(wt, f); h_g = ln_rl(wg, f); h_i = wbase + f; cur = h_i; break; }
                    if (im != (~0ull << off)) { cur = wbase + off + __popcll(im); break; }          // the range ends inside this window
                    scan = wbase + 64; cur = scan;
                }
                const int lb = l_t >= 0 ? l_t - ts : 0, lprev = lp_t >= 0 ? lp_t - ts : 0;
                if (lb > 0) {
                    emit(0, ts, lb, gr1, 0., 0, 0);
                    last_pk = l_i;
                    env_cnt = lb - lprev; c += lb;
                }
                if (h_t >= 0) {
                    const int pdh = (h_t - ts) - lb;
                    last_pk = h_i;
                    state = LIM_ATTACK;
                    att = pdh; if (att <= 1) att = 2;
                    gr0 = gr1; gr1 = h_g; env_cnt = 0;
                    break;
                }
                if (c >= nb) break;
                // the call from there scans n = 1 .. nb - 1: what lies behind the frame's remainder
                const int ts2 = T0 + c + LN_ATT, lim = ts2 + nb - 1, zone = T0 + LN_LBS - 12;
                advance(ts + Rb);
                const int off = cur - wbase;
                const int e = __builtin_amdgcn_readlane(wt, off);
                bool found = e <= lim;
                if (lim >= zone && !(found && e < zone)) {
                    // the scan reaches the ring's last twelve samples, whose test reads past its end (the filter wraps to samples the
                    // envelope has edited): only harmless when none of them can be a candidate at all
                    const int t = zone + lane;
                    const bool hot = lane < 12 && t <= lim && t < T0 + LN_LBS && fabs(y[tbase + t]) > ceiling;
                    if (__ballot(hot)) { hazard = true; break; }
                    found = false;
                }
                if (found) {
                    const double gr = ln_rl(wg, off);
                    const int pd = e - ts2;
                    last_pk = cur;
                    if (gr < gr1) {
                        state = LIM_ATTACK;
                        att = pd; if (att <= 1) att = 2;
                        gr0 = gr1; gr1 = gr; env_cnt = 0;
                        break;
                    }
                    int cnt = pd; if (cnt > nb - c) cnt = nb - c; if (cnt < 0) cnt = 0;
                    emit(0, ts2, cnt, gr1, 0., 0, 0);
                    env_cnt = cnt; c += cnt;
                } else { state = LIM_RELEASE; gr0 = gr1; gr1 = 1.; env_cnt = 0; }
                break; }
            case LIM_RELEASE: {
                int cnt = LN_REL - env_cnt; if (cnt > nb - c) cnt = nb - c; if (cnt < 0) cnt = 0;
                emit(2, ts, cnt, gr0, gr1, env_cnt, 0);
                env_cnt += cnt; c += cnt;
                if (c < nb) { env_cnt = 0; state = LIM_OUT; }
                break; }
            }
        } while (c < nb && !hazard && !ovf);
        if (hazard || ovf) {
            gr0 = s_gr0; gr1 = s_gr1; state = s_state; last_pk = s_lp; env_cnt = s_ec; att = s_att; nseg = s_nseg; layer2 = s_l2;
            why = ovf ? 3 : 4;
            break;
        }
    }
    const int kbe = k;
    if (lane == 0) { ctl->iters = n_iter; ctl->refills = n_refill; ctl->cycles = clock64() - clk0; }
    if (lane == 0) { ctl->kbe = kbe; ctl->nseg = nseg; ctl->why = why; ctl->why_mask |= 1 << why; ctl->ok = kbe > ka ? 1 : 0; if (kbe > ka) ctl->frames += kbe - ka; }
    if (kbe == ka) return;
    // hand-over to the workgroup kernel: the state at the start of inner frame kbe (step kbe + 1)
    const int index_ka = (int)carry[10], index_new = (index_ka + (kbe - ka)) % 30;
    const int lbi = (int)(((long long)kbe * LN_F100) % LN_LBS);
    const double dv = lane < 30 ? B.E[(kbe - ka) + lane] : 0.0;
    const double pdl = B.E[30 + (kbe - ka) - 1];
    __builtin_amdgcn_wave_barrier();
    if (lane < 30) carry[16 + (index_new + lane) % 30] = dv;
    if (lane == 0) {
        carry[0] = gr0; carry[1] = gr1; carry[3] = lbi; carry[4] = state;
        if (last_pk >= 0) { carry[2] = B.pk_v[last_pk]; carry[5] = (double)((tbase + B.pk_t[last_pk]) % LN_LBS); }      // prev_smp, peak_index
        carry[6] = (lbi + LN_F100 + LN_ATT) % LN_LBS; carry[7] = env_cnt; carry[8] = att; carry[9] = 0.0;
        carry[10] = index_new; carry[12] = pdl;
        carry[80] = (double)((long long)(kbe + 1) * LN_F100); carry[81] = (double)((long long)LN_LBS + (long long)kbe * LN_F100);
        carry[84] = (double)(kbe + 1);
    }
}
__global__ void __launch_bounds__(256)
k_lns_apply(double *__restrict__ y, LnsBufs B, int layer)
{
    if (!B.ctl->ok) return;
    const int nseg = B.ctl->nseg;
    for (int q = blockIdx.x; q < nseg; q += gridDim.x) {
        const LnsSeg sg = B.seg[q];
        if ((sg.kind >> 8) != layer) continue;
        const int kind = sg.kind & 255;
        double *p = y + B.ctl->tbase + sg.t;
        const double g0 = sg.g0, g1 = sg.g1; const int c0 = sg.c0, al = sg.al;
        if (kind == 0) for (int j = threadIdx.x; j < sg.len; j += 256) p[j] *= g0;
        else if (kind == 1) for (int j = threadIdx.x; j < sg.len; j += 256) p[j] *= g0 - ((double)(c0 + j) / (al - 1) * (g0 - g1));
        else for (int j = threadIdx.x; j < sg.len; j += 256) p[j] *= g0 + (((double)(c0 + j) / (LN_REL - 1)) * (g1 - g0));
    }
}
// the frames go out (clamped); the 110 ms behind them go back into the ring, with the flags of its 64-sample blocks (lnw_ring_fill)
__global__ void __launch_bounds__(256)
k_lns_finish(double *__restrict__ y, double *__restrict__ ring, double *__restrict__ carry, LnsBufs B, double ceiling)
{
    if (!B.ctl->ok) return;
    const long long t0 = (long long)(B.ctl->ka + 1) * LN_F100, t1 = (long long)(B.ctl->kbe + 1) * LN_F100;
    unsigned char *hot = reinterpret_cast<unsigned char *>(carry + 96);
    const long long ntot = max((long long)LN_LBS, t1 - t0);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ntot; i += (long long)gridDim.x * 256) {
    if (i < LN_LBS) {
        const long long t = t1 + i;
        const int r = (int)(t % LN_LBS);
        bool h = true;
        if (i < LN_LBS - LN_F100) { const double v = y[t]; ring[r] = v; h = fabs(v) > ceiling; }
        const unsigned long long hm = __ballot(h);
        if ((threadIdx.x & 63) == 0) hot[r >> 6] = (unsigned char)(hm != 0);
    }
    const long long t = t0 + i;
    if (t < t1) { double v = y[t]; if (fabs(v) > ceiling) v = ceiling * (v < 0 ? -1 : 1); y[t] = v; }
    }
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
// The same sums for a block of 256 consecutive outputs with the inputs they share in LDS (256 step / P + L samples: 10 KB at 192 -> 44.1 kHz,
// loaded once, coalesced, the edges' reflections resolved on the way in) and the taps from the TRANSPOSED bank bankT[i][phase]: the lanes
// of a wave sit on 64 different phases, which in the bank's own layout are 64 different cache lines per tap and in the transposed one
// the ten lines of one row.  Same products in the same order per output (taps ascending, multiply then add).  The one-thread-per-output
// kernel above took 16.6 ms for ten minutes -- the longest kernel of a file that takes the dynamic mode once its limiter was out of the way.
__global__ void __launch_bounds__(256)
k_swr_tile_f64(const double *__restrict__ in, int64_t n, const double *__restrict__ bankT, int P, int L, int center, int64_t step, int64_t m_total,
               double *__restrict__ out)
{
    extern __shared__ double swr_tile[];
    const int tid = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * 256, mlast = min(m_total - 1, m0 + 255);
    const int64_t s0 = (m0 * step) / P - center, s1 = (mlast * step) / P - center + L;
    const int cnt = (int)(s1 - s0);
    for (int j = tid; j < cnt; j += 256) {
        int64_t g = s0 + j;
        double v = 0.0;
        if (g < 0) g = -g;                                           // invert_initial_buffer(): in[-j] = in[j]
        if (g < n) v = in[g];
        else { const int64_t r = 2 * n - 1 - g; if (r >= 0 && r < n) v = in[r]; }   // resample_flush()
        swr_tile[j] = v;
    }
    __syncthreads();
    const int64_t m = m0 + tid;
    if (m >= m_total) return;
    const int64_t idx = m * step;
    const int ph = (int)(idx % P);
    const double *t = swr_tile + (int)(idx / P - center - s0);
    const double *f = bankT + ph;
    double val = 0.0;
#pragma unroll 4
    for (int i = 0; i < L; ++i) val += t[i] * f[(size_t)i * P];
    out[m] = val;
}
// Four outputs per thread.  Outputs P apart share their phase, i.e. their taps: thread t of a block computes m0 + t, m0 + t + P, m0 + t + 2P,
// m0 + t + 3P with ONE tap fetch per tap and four independent sums (k_swr_tile_f64: one fetch and one dependent multiply-add chain per
// output).  The block's inputs ((4P - 1) step / P + L samples: 22 KB at 192 -> 44.1 kHz) go through LDS as before.  Same sums.
// (The tap rows in LDS too -- 49 phases x 5 outputs per block -- was slower than either: 4.2 ms against 3.4, the block spends as long
//  filling 78 KB of LDS as it spends computing.)
constexpr int SWR_K = 4;
__global__ void __launch_bounds__(256)
k_swr_tile4_f64(const double *__restrict__ in, int64_t n, const double *__restrict__ bankT, int P, int L, int center, int64_t step, int64_t m_total,
                double *__restrict__ out, int tile_len)
{
    extern __shared__ double swr_tile4[];
    const int tid = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * SWR_K * P;
    const int64_t s0 = (m0 * step) / P - center;
    for (int e = tid; e < tile_len; e += blockDim.x) {
        int64_t g = s0 + e;
        double v = 0.0;
        if (g < 0) g = -g;                                           // invert_initial_buffer(): in[-j] = in[j]
        if (g < n) v = in[g];
        else { const int64_t r = 2 * n - 1 - g; if (r >= 0 && r < n) v = in[r]; }   // resample_flush()
        swr_tile4[e] = v;
    }
    __syncthreads();
    if (tid >= P) return;
    const int64_t m = m0 + tid;
    if (m >= m_total) return;
    const int64_t idx = m * step;
    const int ph = (int)(idx % P);
    const double *t = swr_tile4 + (int)(idx / P - center - s0);     // output m + k P starts k * step samples further on
    const double *f = bankT + ph;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
#pragma unroll 2
    for (int i = 0; i < L; ++i) {
        const double c = f[(size_t)i * P];
        v0 += t[i] * c; v1 += t[i + step] * c; v2 += t[i + 2 * step] * c; v3 += t[i + 3 * step] * c;
    }
    out[m] = v0;
    if (m + P < m_total) out[m + P] = v1;
    if (m + 2 * (int64_t)P < m_total) out[m + 2 * (int64_t)P] = v2;
    if (m + 3 * (int64_t)P < m_total) out[m + 3 * (int64_t)P] = v3;
}
}  // namespace

void launch_swr_plain_f64(const double *in, int64_t n, const double *bank, int P, int L, int center, int64_t step, int64_t m_total, double *out, hipStream_t s,
                          const double *bankT)
{
    if (m_total <= 0) return;
    if (bankT && P >= 64 && P <= 256) {
        const int tile_len = (int)(((int64_t)(SWR_K * P - 1) * step) / P + L + 2);
        const size_t lds = sizeof(double) * (size_t)tile_len;
        if (lds <= 48 * 1024) {
            const int64_t per_block = (int64_t)SWR_K * P;
            hipLaunchKernelGGL(k_swr_tile4_f64, dim3((unsigned)((m_total + per_block - 1) / per_block)), dim3((unsigned)((P + 63) / 64 * 64)), lds, s,
                               in, n, bankT, P, L, center, step, m_total, out, tile_len);
            return;
        }
    }
    const size_t smem = sizeof(double) * (size_t)((255 * step) / P + L + 2);
    if (bankT && smem <= 48 * 1024) {
        hipLaunchKernelGGL(k_swr_tile_f64, dim3((unsigned)((m_total + 255) / 256)), dim3(256), smem, s, in, n, bankT, P, L, center, step, m_total, out);
        return;
    }
    hipLaunchKernelGGL(k_swr_plain_f64, dim3((unsigned)((m_total + 255) / 256)), dim3(256), 0, s, in, n, bank, P, L, center, step, m_total, out);
}
size_t jt_lns_scratch_bytes(int64_t n, int64_t n_inner, LnsBufs *B, unsigned char *base)
{
    // one allocation, carved up (256-byte aligned pieces); base == nullptr: size only
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *p = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
    const int64_t nwords = n / 64 + 2 * LNS_BW, nblk = nwords / LNS_BW + 2;
    const int64_t pk_cap = n / 48 + 1024, seg_cap = 64 * n_inner + 1024;
    LnsBufs b{};
    b.ctl = reinterpret_cast<LnsCtl *>(take(sizeof(LnsCtl)));
    b.G = reinterpret_cast<double *>(take(sizeof(double) * (size_t)(n_inner + 64)));
    b.Gn = reinterpret_cast<double *>(take(sizeof(double) * (size_t)(n_inner + 64)));
    b.E = reinterpret_cast<double *>(take(sizeof(double) * (size_t)(n_inner + 128)));
    b.bm = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * (size_t)nwords));
    b.woff = reinterpret_cast<unsigned short *>(take(sizeof(unsigned short) * (size_t)nwords));
    b.bcnt = reinterpret_cast<int *>(take(sizeof(int) * (size_t)nblk));
    b.boff = reinterpret_cast<int *>(take(sizeof(int) * (size_t)nblk));
    b.pk_t = reinterpret_cast<int *>(take(sizeof(int) * (size_t)pk_cap));
    b.pk_v = reinterpret_cast<double *>(take(sizeof(double) * (size_t)pk_cap));
    b.pk_g = reinterpret_cast<double *>(take(sizeof(double) * (size_t)pk_cap));
    b.seg = reinterpret_cast<LnsSeg *>(take(sizeof(LnsSeg) * (size_t)seg_cap));
    b.fr = reinterpret_cast<LnsFrame *>(take(sizeof(LnsFrame) * (size_t)(n_inner + 64)));
    b.pk_cap = (int)std::min<int64_t>(pk_cap, 0x7fffffff); b.seg_cap = (int)std::min<int64_t>(seg_cap, 0x7fffffff);
    if (B) *B = b;
    return off;
}
// one attempt of the stream path at inner frames [ka, kb): every kernel looks at the control block k_lns_begin fills, nothing waits for the host
static void lns_attempt(const double *x, const LoudnormDynParams &P, const double *series, double *ring, double *y, double *carry, const LnsBufs &B,
                        int ka, int kb, hipStream_t s)
{
    const int64_t t_start = (int64_t)(ka + 1) * LN_F100, l_end = (int64_t)LN_LBS + (int64_t)kb * LN_F100;
    const int64_t tb0 = t_start + 1, tb1 = l_end - 11;
    const int64_t blk0 = tb0 / (64 * LNS_BW), blk1 = (tb1 - 1) / (64 * LNS_BW);
    const int nblk = (int)(blk1 - blk0 + 1);
    const int64_t nfill = (int64_t)(kb - ka) * LN_F100 + (LN_LBS - LN_F100);
    hipLaunchKernelGGL(k_lns_begin, dim3(1), dim3(256), 0, s, carry, series, P, B, ka, kb);
    // (grids of a fixed size that stride over their ranges: an attempt that turns out not to run costs eight empty launches, not eight big ones)
    auto grid = [](int64_t items) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, (items + 255) / 256))); };
    hipLaunchKernelGGL(k_lns_fill, grid(nfill), dim3(256), 0, s, x, y, ring, B, ka, kb, P.offset_lin);
    hipLaunchKernelGGL(k_lns_bitmap, dim3((unsigned)std::min(nblk, 8192)), dim3(256), 0, s, y, B, (long long)tb0, (long long)tb1, P.target_tp_lin, (long long)blk0, nblk);
    hipLaunchKernelGGL(k_lns_scan, dim3(1), dim3(1024), 0, s, B, nblk);
    hipLaunchKernelGGL(k_lns_scatter, grid((int64_t)nblk * LNS_BW), dim3(256), 0, s, y, B, (long long)blk0, nblk, P.target_tp_lin);
    hipLaunchKernelGGL(k_lns_frames, grid(kb - ka), dim3(256), 0, s, B, ka, kb);
    hipLaunchKernelGGL(k_lns_machine, dim3(1), dim3(64), 0, s, carry, P, B, y, ka, kb);
    hipLaunchKernelGGL(k_lns_apply, dim3(4096), dim3(256), 0, s, y, B, 0);
    hipLaunchKernelGGL(k_lns_apply, dim3(4096), dim3(256), 0, s, y, B, 1);
    const int64_t nfin = std::max<int64_t>(LN_LBS, (int64_t)(kb - ka) * LN_F100);
    hipLaunchKernelGGL(k_lns_finish, grid(nfin), dim3(256), 0, s, y, ring, carry, B, P.target_tp_lin);
}
void launch_loudnorm_dynamic(const double *x, int64_t n, const LoudnormDynParams &P, const double *series, double *ring, double *y, double *dbg, hipStream_t s,
                             double *carry, const JtOpts &o, const LnsBufs *stream)
{
    (void)o;
    if (!JT_AB_ON(o.dyn_one_wave)) {
        // eight waves, the limiter's window in LDS (159 KB: the workgroup has its CU to itself); LN_STEPS steps of the file per launch
        const int smem = LN_CACHE * (int)sizeof(double);
        JT_HIP(hipFuncSetAttribute((const void *)k_loudnorm_dynamic_wg, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        const int64_t total = 1 + P.n_inner + P.final_len / LN_F100;
        const int64_t steps = JT_AB_ON(o.dyn_steps > 0) ? o.dyn_steps : 64;
        const bool use_stream = stream && !P.no_batch;
        const int64_t nb_last = n - LN_F3000 - (P.n_inner - 1) * LN_F100, kb = nb_last == LN_F100 ? P.n_inner : P.n_inner - 1;
        int chunk = 0;
        for (int64_t it = 0, end = 0; it < total; it = end, ++chunk) {
            // with the stream path behind it the workgroup kernel normally runs the first 8 steps and the last 30: launches of 8, 8, 16,
            // 32, ... 256 steps (each skips what the stream path has covered meanwhile: carry[84]), an attempt of the stream path behind
            // the first seven and then every fourth -- a file that starts quietly latches above_threshold late, and an attempt that
            // stopped at one of its corners leaves the rest to a later one
            const int64_t len = !use_stream ? steps : (chunk < 2 ? 8 : chunk >= 7 ? 256 : (int64_t)8 << (chunk - 1));      // (8, 8, 16, .. 256, 256, ..)
            end = std::min(total, it + len);
            hipLaunchKernelGGL(k_loudnorm_dynamic_wg, dim3(1), dim3(LN_WG), smem, s, x, n, P, series, ring, y, dbg, carry, it, end);
            if (use_stream && (chunk <= 6 || chunk % 4 == 0) && kb - (end - 1) >= 32 && end < total)
                lns_attempt(x, P, series, ring, y, carry, *stream, (int)(end - 1), (int)std::min<int64_t>(kb, end - 1 + LNS_MAXF), s);      // (the rest: a later attempt)
#ifdef JT_LN_PROFILE
            if (chunk == 8 || end >= total) {   // phase clocks of the ninth launch (inner frames) and of the last one (the flush): tools/prof_dynamic_phases.sh
                unsigned long long pr[16];
                JT_HIP(hipStreamSynchronize(s)); JT_HIP(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_ln_prof), sizeof pr));
                const double st = (double)(end - it);
                fprintf(stderr, "loudnorm dynamic, launch at step %lld (%g steps), kcycles per step: fill %.1f, window load %.1f, limiter %.1f (detect %.1f, scale %.1f), write-back + out %.1f; "
                                "per step: %.1f scale passes, %.1f detector calls\n", (long long)it, st, pr[0] / st / 1e3, pr[1] / st / 1e3, pr[2] / st / 1e3, pr[4] / st / 1e3,
                        pr[7] / st / 1e3, pr[3] / st / 1e3, pr[5] / st, pr[6] / st);
            }
#endif
        }
        return;
    }
    // 96 KB of (unused) dynamic LDS: the workgroup then has a CU to itself.  Several files in flight each run one such wave, and the
    // dispatcher packed them onto the first CU with room -- the same SIMDs -- where each ran 1.5x slower than alone.
#ifdef JT_AB
    const int reserve = o.dyn_no_cu_reserve ? 0 : 96 * 1024;
    if (reserve) JT_HIP(hipFuncSetAttribute((const void *)k_loudnorm_dynamic, hipFuncAttributeMaxDynamicSharedMemorySize, reserve));
    hipLaunchKernelGGL(k_loudnorm_dynamic, dim3(1), dim3(64), reserve, s, x, n, P, series, ring, y, dbg);
#endif
}
void launch_scale_f64(const double *in, double *out, int64_t n, double g, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_scale_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n, g);
}
