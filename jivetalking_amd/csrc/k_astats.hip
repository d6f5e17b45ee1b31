// k_astats.hip — astats (FFmpeg af_astats.c; filters.go:624 "astats=metadata=1:measure_perchannel=all") on gfx950 as a
// handful of fully parallel, coalesced kernels instead of one sequential per-sample state machine:
//   R  : order-independent reductions (min/max/sums/differences/zero-crossings/entropy histogram), grid-stride, float4-wide
//   M  : global min/max finalisation on device (no host round trip)
//   RUN: Peak_count / Flat_factor run statistics at the global extrema (run starts scan forward)
//   GP : van Herk / Gil-Werman block prefix & suffix maxima of |x| (one thread per 50 ms block)
//   NF : sliding 50 ms local-peak minimum + count (Noise_floor, Noise_floor_count)
//   ZS / SCAN / SIG: exponentially averaged power (RMS_peak / RMS_trough) as an exact linear scan:
//        per-chunk zero-state response, sequential carry over chunks, then extrema with the carried-in state
// All double-precision sums; only the summation ORDER differs from the sequential filter (1e-15 relative).
#include "jt_internal.h"
#include <cfloat>

constexpr int AS_T = 256;

struct AsPartial {
    double min, max, min_nz, sx, sx2, mind, maxd, d1, d2;
    unsigned long long zero_runs, mask_or, mask_and, count;
};
struct AsRuns { double min_count, min_runs, max_count, max_runs; };
struct AsNF { double nf; unsigned long long cnt; };

__device__ inline double wsum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; }
__device__ inline double wmin(double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64)); return v; }
__device__ inline double wmax(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64)); return v; }
__device__ inline unsigned long long wsumu(unsigned long long v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64); return v; }
__device__ inline unsigned long long woru(unsigned long long v) { for (int o = 32; o > 0; o >>= 1) v |= __shfl_down(v, o, 64); return v; }
__device__ inline unsigned long long wandu(unsigned long long v) { for (int o = 32; o > 0; o >>= 1) v &= __shfl_down(v, o, 64); return v; }

// HIST = false: without the entropy histogram (the announced regions' samples report levels only: no LDS atomics, no zeroed buffer)
template <bool HIST>
__global__ void __launch_bounds__(AS_T)
k_as_reduce(const float *__restrict__ x, int64_t n, AsPartial *__restrict__ part, unsigned long long *__restrict__ ehist)
{
    __shared__ unsigned int lh[HIST ? 8192 : 1];
    __shared__ AsPartial sp[AS_T / 64];
    const int tid = threadIdx.x;
    if (HIST) {
        for (int i = tid; i < 8192; i += AS_T) lh[i] = 0;
        __syncthreads();
    }
    double mn = DBL_MAX, mx = -DBL_MAX, mnz = DBL_MAX, sx = 0, sx2 = 0, mind = DBL_MAX, maxd = 0, d1 = 0, d2 = 0;
    unsigned long long zr = 0, mor = 0, mand = ~0ull, cnt = 0;
    const int64_t stride = (int64_t)gridDim.x * AS_T;
    auto step = [&](int64_t i, float xf, float xp) {
        const double d = (double)xf;
        mn = fmin(mn, d); mx = fmax(mx, d);
        const double ad = fabs(d);
        if (d != 0 && ad < mnz) mnz = ad;
        sx += d; sx2 += d * d;
        if (i > 0) {
            const double p = (double)xp;
            const double df = fabs(d - p);
            mind = fmin(mind, df); maxd = fmax(maxd, df); d1 += df; d2 += (d - p) * (d - p);
        }
        if (d != 0) {
            // FFSIGN of the previous non-zero sample (NaN before the first one: FFSIGN(NaN) = -1)
            int ps;
            if (xp != 0.f && i > 0) ps = xp > 0.f ? 1 : -1;
            else {
                int64_t j = i - 1;
                while (j >= 0 && x[j] == 0.f) --j;
                ps = (j >= 0 && x[j] > 0.f) ? 1 : -1;
            }
            const int cs = d > 0 ? 1 : -1;
            zr += (cs != ps);
        }
        if (HIST) {
            int h = (int)rint(fmin(fmax(ad, 0.0), 1.0) * 8191.0);
            h = h < 0 ? 0 : (h > 8191 ? 8191 : h);
            atomicAdd(&lh[h], 1u);
        }
        const long long iv = (long long)llrint(d * 2147483648.0);
        mor |= (unsigned long long)iv; mand &= (unsigned long long)iv;
        cnt++;
    };
    // four grid-strided samples per round: the eight loads are issued together (one HBM round trip instead of four)
    int64_t i = (int64_t)blockIdx.x * AS_T + tid;
    for (; i + 3 * stride < n; i += 4 * stride) {
        float xv[4], xq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int64_t k = i + q * stride; xv[q] = x[k]; xq[q] = k > 0 ? x[k - 1] : 0.f; }
#pragma unroll
        for (int q = 0; q < 4; ++q) step(i + q * stride, xv[q], xq[q]);
    }
    for (; i < n; i += stride) step(i, x[i], i > 0 ? x[i - 1] : 0.f);
    mn = wmin(mn); mx = wmax(mx); mnz = wmin(mnz); sx = wsum(sx); sx2 = wsum(sx2); mind = wmin(mind); maxd = wmax(maxd);
    d1 = wsum(d1); d2 = wsum(d2); zr = wsumu(zr); mor = woru(mor); mand = wandu(mand); cnt = wsumu(cnt);
    if ((tid & 63) == 0) sp[tid >> 6] = AsPartial{mn, mx, mnz, sx, sx2, mind, maxd, d1, d2, zr, mor, mand, cnt};
    __syncthreads();
    if (tid == 0) {
        AsPartial a = sp[0];
        for (int w = 1; w < AS_T / 64; ++w) {
            const AsPartial &b = sp[w];
            a.min = fmin(a.min, b.min); a.max = fmax(a.max, b.max); a.min_nz = fmin(a.min_nz, b.min_nz);
            a.sx += b.sx; a.sx2 += b.sx2; a.mind = fmin(a.mind, b.mind); a.maxd = fmax(a.maxd, b.maxd); a.d1 += b.d1; a.d2 += b.d2;
            a.zero_runs += b.zero_runs; a.mask_or |= b.mask_or; a.mask_and &= b.mask_and; a.count += b.count;
        }
        part[blockIdx.x] = a;
    }
    if (HIST) for (int i = tid; i < 8192; i += AS_T) if (lh[i]) atomicAdd(&ehist[i], (unsigned long long)lh[i]);
}

// run statistics at the global extrema: min_count = #samples == min; min_runs = sum over maximal runs of len^2
// (the global extrema come from the reduce sweep's partials: every workgroup folds them itself -- at most 2048 entries -- instead of a
// one-workgroup launch in between; workgroup 0 leaves them in mm[] for the host)
__global__ void __launch_bounds__(AS_T)
k_as_runs(const float *__restrict__ x, int64_t n, const AsPartial *__restrict__ rpart, int nrparts, double *__restrict__ mm, AsRuns *__restrict__ part)
{
    __shared__ AsRuns sp[AS_T / 64];
    __shared__ double smm[2][AS_T / 64];
    {
        double mn = DBL_MAX, mx = -DBL_MAX;
        for (int i = threadIdx.x; i < nrparts; i += AS_T) { mn = fmin(mn, rpart[i].min); mx = fmax(mx, rpart[i].max); }
        mn = wmin(mn); mx = wmax(mx);
        if ((threadIdx.x & 63) == 0) { smm[0][threadIdx.x >> 6] = mn; smm[1][threadIdx.x >> 6] = mx; }
        __syncthreads();
    }
    double gmn = smm[0][0], gmx = smm[1][0];
    for (int w = 1; w < AS_T / 64; ++w) { gmn = fmin(gmn, smm[0][w]); gmx = fmax(gmx, smm[1][w]); }
    if (blockIdx.x == 0 && threadIdx.x == 0) { mm[0] = gmn; mm[1] = gmx; }
    const float gmin = (float)gmn, gmax = (float)gmx;
    double c0 = 0, r0 = 0, c1 = 0, r1 = 0;
    const int64_t stride = (int64_t)gridDim.x * AS_T;
    auto look = [&](int64_t i, float v) {
        if (v == gmin) {
            c0 += 1;
            if (i == 0 || x[i - 1] != gmin) { int64_t j = i + 1; while (j < n && x[j] == gmin) ++j; double len = (double)(j - i); r0 += len * len; }
        }
        if (v == gmax) {
            c1 += 1;
            if (i == 0 || x[i - 1] != gmax) { int64_t j = i + 1; while (j < n && x[j] == gmax) ++j; double len = (double)(j - i); r1 += len * len; }
        }
    };
    // (round 6) eight loads of a thread in flight per trip: a sweep that almost never finds anything was 5 % active and 91 % waiting with one.
    // Counts and squared run lengths are integers in doubles: the sums are exact in any order
    int64_t i = (int64_t)blockIdx.x * AS_T + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = x[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) if (v[k] == gmin || v[k] == gmax) look(i + k * stride, v[k]);
    }
    for (; i < n; i += stride) look(i, x[i]);
    c0 = wsum(c0); r0 = wsum(r0); c1 = wsum(c1); r1 = wsum(r1);
    if ((threadIdx.x & 63) == 0) sp[threadIdx.x >> 6] = AsRuns{c0, r0, c1, r1};
    __syncthreads();
    if (threadIdx.x == 0) {
        AsRuns a = sp[0];
        for (int w = 1; w < AS_T / 64; ++w) { a.min_count += sp[w].min_count; a.min_runs += sp[w].min_runs; a.max_count += sp[w].max_count; a.max_runs += sp[w].max_runs; }
        part[blockIdx.x] = a;
    }
}

// per 50 ms block: suffix maxima G[i] = max|x[i..blockend]| and prefix maxima P[i] = max|x[blockstart..i]|
__global__ void k_as_gp(const float *__restrict__ x, float *__restrict__ g, float *__restrict__ p, int64_t n, int w)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t lo = b * w;
    if (lo >= n) return;
    const int64_t hi = min(lo + (int64_t)w, n);
    float m = 0.f;
    for (int64_t i = lo; i < hi; ++i) { m = fmaxf(m, fabsf(x[i])); p[i] = m; }
    m = 0.f;
    for (int64_t i = hi - 1; i >= lo; --i) { m = fmaxf(m, fabsf(x[i])); g[i] = m; }
}

// sliding local peak over the last tc samples (valid once i >= tc-1): min over time and the number of times it occurs
__global__ void __launch_bounds__(AS_T)
k_as_noise_floor(const float *__restrict__ g, const float *__restrict__ p, int64_t n, int tc, AsNF *__restrict__ part)
{
    __shared__ AsNF sp[AS_T / 64];
    double nf = DBL_MAX; unsigned long long cnt = 0;
    const int64_t stride = (int64_t)gridDim.x * AS_T;
    for (int64_t i = (int64_t)blockIdx.x * AS_T + threadIdx.x + (tc - 1); i < n; i += stride) {
        const int64_t j = i - tc + 1;
        const float wm = (j % tc == 0) ? g[j] : fmaxf(g[j], p[i]);
        const double lp = (double)wm;
        if (lp < nf) { nf = lp; cnt = 1; } else if (lp == nf) cnt++;
    }
    // (min, count) merge
    for (int o = 32; o > 0; o >>= 1) {
        double onf = __shfl_down(nf, o, 64); unsigned long long oc = __shfl_down(cnt, o, 64);
        if (onf < nf) { nf = onf; cnt = oc; } else if (onf == nf) cnt += oc;
    }
    if ((threadIdx.x & 63) == 0) sp[threadIdx.x >> 6] = AsNF{nf, cnt};
    __syncthreads();
    if (threadIdx.x == 0) {
        AsNF a = sp[0];
        for (int w = 1; w < AS_T / 64; ++w) { if (sp[w].nf < a.nf) a = sp[w]; else if (sp[w].nf == a.nf) a.cnt += sp[w].cnt; }
        part[blockIdx.x] = a;
    }
}

// wave64 inclusive prefix maximum of non-negative values (gfx9 DPP: row_shr 1/2/4/8, row_bcast 15/31; lanes shifted in from
// outside a row read 0)
#define JT_DPPF(v, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (rmask), 0xf, false))
__device__ inline float wave_prefix_max(float v)
{
    v = fmaxf(v, JT_DPPF(v, 0x111, 0xf));
    v = fmaxf(v, JT_DPPF(v, 0x112, 0xf));
    v = fmaxf(v, JT_DPPF(v, 0x114, 0xf));
    v = fmaxf(v, JT_DPPF(v, 0x118, 0xf));
    v = fmaxf(v, JT_DPPF(v, 0x142, 0xa));
    v = fmaxf(v, JT_DPPF(v, 0x143, 0xc));
    return v;
}


// Noise floor without LDS: sliding-window peak from registers.  Window [s, s+tc) with s = 64 bs + l (lane l) ends at
// e = 64 (bs + q) + l + r, q = (tc-1)/64, r = (tc-1)%64.  Its maximum is
//     max( suffix-max of block bs from lane l,  max of the whole blocks in between,  prefix-max of the end block up to its lane ),
// the whole blocks being bs+1 .. bs+q-1 (one more, bs+q, for the lanes whose end wraps into block bs+q+1).  Block maxima and
// their (q-1)-wide running maxima are two tiny pre-passes; the main kernel is three DPP scans and three lane permutes per
// 64 outputs.  Maxima are order independent, so the result equals the sequential filter's.
__global__ void __launch_bounds__(256)
k_nf_blockmax(const float *__restrict__ x, int64_t n, float *__restrict__ bm, int64_t nblocks, unsigned *__restrict__ ub_min)
{
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) *ub_min = 0x7f800000u;          // +inf: k_nf_runmax (next on this stream) takes minima into it
    const int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    // (round 6) a lane takes FOUR consecutive samples (one 16-byte load), a wave 256 = four blocks, two such groups per trip: a block's
    // maximum is a 4-step exchange inside its 16 lanes.  One 4-byte load, six dependent shuffles and a store per block left the sweep
    // 11 % active and 85 % waiting.  Maxima: the same values whatever the order.
    const int64_t ngroups = (nblocks + 3) / 4;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        for (int64_t g = w; g < ngroups; g += 2 * nw) {
            float4 v[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int64_t i = (g + k * nw) * 256 + 4 * lane;
                if (g + k * nw < ngroups && i + 3 < n) v[k] = *reinterpret_cast<const float4 *>(x + i);
                else { v[k].x = i < n ? x[i] : 0.f; v[k].y = i + 1 < n ? x[i + 1] : 0.f; v[k].z = i + 2 < n ? x[i + 2] : 0.f; v[k].w = 0.f; if (g + k * nw >= ngroups) v[k].x = v[k].y = v[k].z = 0.f; }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float m = fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                const int64_t blk = (g + k * nw) * 4 + (lane >> 4);
                if ((lane & 15) == 0 && g + k * nw < ngroups && blk < nblocks) bm[blk] = m;
            }
        }
        return;
    }
    for (int64_t b = w; b < nblocks; b += nw) {
        const int64_t i = b * 64 + lane;
        float v = i < n ? fabsf(x[i]) : 0.f;
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        if (lane == 0) bm[b] = v;
    }
}
__global__ void __launch_bounds__(256)
k_nf_runmax(const float *__restrict__ bm, float *__restrict__ rq, int64_t nblocks, int width, int64_t nstartblocks, unsigned *__restrict__ ub_min)
{
    // Round 6, branch and bound for the noise floor (the MINIMUM over all 50 ms windows of the window's peak, and how many windows reach it):
    // every window that starts in block b lies inside blocks b .. b + width + 2, so UB(b) = max bm[b .. b + width + 2] bounds its peak from
    // above, and the minimum of UB over the start blocks bounds the noise floor from above -- from block maxima alone.  k_nf_main then
    // skips every start block whose windows all contain a whole block louder than that (rq[b + 1] > bound: they can neither be the
    // minimum nor tie with it) without touching its samples.  Exact: a window that reaches the minimum is never skipped.
    // the workgroup's 256 + width + 3 block maxima through LDS (one coalesced read; a thread then walks its width + 3 neighbours there: read
    // straight from memory, the walk's 39 dependent L2 round trips made this the longest kernel of the chain once k_nf_main was pruned).
    // Block maxima are >= 0, so the zeros behind the last block change no maximum.
    extern __shared__ float nf_tile[];
    const int64_t b0 = (int64_t)blockIdx.x * 256;
    const int span = 256 + width + 3;
    for (int i = threadIdx.x; i < span; i += 256) { const int64_t g = b0 + i; nf_tile[i] = g < nblocks ? bm[g] : 0.f; }
    __syncthreads();
    const int64_t b = b0 + threadIdx.x;
    float ub = __uint_as_float(0x7f800000u);
    if (b < nblocks) {
        const float *t = nf_tile + threadIdx.x;
        float m = 0.f;
        for (int k = 0; k < width; ++k) m = fmaxf(m, t[k]);
        rq[b] = m;
        if (b < nstartblocks) {
            for (int k = width; k < width + 3; ++k) m = fmaxf(m, t[k]);
            ub = m;
        }
    }
    for (int o = 32; o > 0; o >>= 1) ub = fminf(ub, __shfl_xor(ub, o, 64));
    // (non-negative floats order like their bits.)  A wave that cannot lower the bound does not touch it: 42 000 atomics on one word were
    // most of this kernel's time; after the first few hundred waves hardly any improves the minimum
    if ((threadIdx.x & 63) == 0 && ub < __uint_as_float(__hip_atomic_load(ub_min, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicMin(ub_min, __float_as_uint(ub));
}
__global__ void __launch_bounds__(256)
k_nf_main(const float *__restrict__ x, int64_t n, int tc, const float *__restrict__ bm, const float *__restrict__ rq, int64_t nstartblocks,
          AsNF *__restrict__ part, const unsigned *__restrict__ ub_min)
{
    __shared__ AsNF sp[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = (tc - 1) >> 6, r = (tc - 1) & 63;
    const int64_t w0 = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
    double nf = DBL_MAX; unsigned long long cnt = 0;
    // a wave takes NF_U consecutive start blocks at a time: their end blocks overlap (block bs+q+1 of one is bs+q of the next), so
    // 2 NF_U + 1 loads are in flight instead of 3 and each end block's prefix scan is shared
    constexpr int NF_U = 4;
    const float bound = ub_min ? __uint_as_float(*ub_min) : __uint_as_float(0x7f800000u);
    for (int64_t b0 = w0 * NF_U; b0 < nstartblocks; b0 += nw * NF_U) {
        if (q >= 2 && ub_min) {
            // (wave-uniform) every start block of the group holds, in each of its windows, a whole block louder than the bound: nothing here
            // can be the minimum or tie with it
            bool skip = true;
#pragma unroll
            for (int u = 0; u < NF_U; ++u) if (b0 + u < nstartblocks && !(rq[b0 + u + 1] > bound)) skip = false;
            if (skip) continue;
        }
        float xr[NF_U], xe[NF_U + 1], rqv[NF_U], bmv[NF_U];
#pragma unroll
        for (int u = 0; u < NF_U; ++u) {
            const int64_t ir = (b0 + u) * 64 + (63 - lane);
            xr[u] = ir < n ? fabsf(x[ir]) : 0.f;                                         // block bs, reversed lane order
            const bool have = b0 + u < nstartblocks;
            rqv[u] = (have && q >= 2) ? rq[b0 + u + 1] : 0.f;                            // blocks bs+1 .. bs+q-1
            bmv[u] = (have && q >= 1) ? bm[b0 + u + q] : 0.f;                            // block bs+q, whole when the end wraps
        }
#pragma unroll
        for (int u = 0; u <= NF_U; ++u) {
            const int64_t ie = (b0 + q + u) * 64 + lane;
            xe[u] = ie < n ? fabsf(x[ie]) : 0.f;
        }
        float pe[NF_U + 1];
#pragma unroll
        for (int u = 0; u <= NF_U; ++u) pe[u] = wave_prefix_max(xe[u]);
        const int el = lane + r;
#pragma unroll
        for (int u = 0; u < NF_U; ++u) {
            const int64_t bs = b0 + u;
            if (bs >= nstartblocks) break;
            const float grev = wave_prefix_max(xr[u]);                                   // lane j: max of block positions 63-j .. 63
            const float g = __shfl(grev, 63 - lane, 64);                                 // suffix max from position `lane`
            const float pa = __shfl(pe[u], el & 63, 64), pb = __shfl(pe[u + 1], el & 63, 64);
            float m = fmaxf(g, el < 64 ? pa : pb);
            if (q >= 2) m = fmaxf(m, rqv[u]);
            if (el >= 64 && q >= 1) m = fmaxf(m, bmv[u]);
            const int64_t e = bs * 64 + lane + tc - 1;
            if (e < n) {
                const double lp = (double)m;
                if (lp < nf) { nf = lp; cnt = 1; } else if (lp == nf) cnt++;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        double onf = __shfl_down(nf, o, 64); unsigned long long oc = __shfl_down(cnt, o, 64);
        if (onf < nf) { nf = onf; cnt = oc; } else if (onf == nf) cnt += oc;
    }
    if (lane == 0) sp[wave] = AsNF{nf, cnt};
    __syncthreads();
    if (threadIdx.x == 0) {
        AsNF a = sp[0];
        for (int w = 1; w < 4; ++w) { if (sp[w].nf < a.nf) a = sp[w]; else if (sp[w].nf == a.nf) a.cnt += sp[w].cnt; }
        part[blockIdx.x] = a;
    }
}

// exponential power average: zero-state response of each chunk (one thread per chunk, coalesced through LDS rows)
constexpr int ZC = 1024;     // chunk length
// tile step `pos` of the 64 chunks from c0: row q = chunk c0 + q, lane = sample pos + lane of it (coalesced rows, 64 loads in flight)
__device__ inline void as_rows_load(float (&v)[64], const float *__restrict__ x, int64_t n, int64_t c0, int pos, int lane)
{
#pragma unroll
    for (int q = 0; q < 64; ++q) { const int64_t idx = (c0 + q) * ZC + pos + lane; v[q] = x[idx < n ? idx : n - 1]; }
}
__device__ inline void as_rows_commit(float (*tile)[65], const float (&v)[64], int64_t n, int64_t c0, int pos, int lane, int nrows)
{
#pragma unroll
    for (int q = 0; q < 64; ++q) { const int64_t idx = (c0 + q) * ZC + pos + lane; tile[q][lane] = (q < nrows && idx < n) ? v[q] : 0.f; }
}
__global__ void __launch_bounds__(64)
k_as_zs(const float *__restrict__ x, int64_t n, double mult, double mult_chunk, double *__restrict__ zs, double *__restrict__ blkA,
        double *__restrict__ blkB, int64_t nchunks)
{
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const int nrows = (int)min((int64_t)64, nchunks - c0);
    double z = 0.0;
    const double om = 1.0 - mult;
    // (round 4) the next tile's 64 row loads are in flight while this one is consumed, and a lane reads its 64 samples of the tile into
    // registers before the recurrence starts: written as "load, store, barrier, read a sample, use it", every step of the chain waited
    // for an LDS round trip and every tile for an HBM one (0.40 ms for an hour alone, five f64 operations per sample)
    float v[64];
    as_rows_load(v, x, n, c0, 0, lane);
    for (int pos = 0; pos < ZC; pos += 64) {
        as_rows_commit(tile, v, n, c0, pos, lane, nrows);
        __syncthreads();
        if (pos + 64 < ZC) as_rows_load(v, x, n, c0, pos + 64, lane);
        if (lane < nrows) {
            const int64_t base = (c0 + lane) * ZC + pos;
            if (base + 64 <= n) {
                float xs[64];
#pragma unroll
                for (int j = 0; j < 64; ++j) xs[j] = tile[lane][j];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 64; ++j) { double d = (double)xs[j]; z = z * mult + om * d * d; }
            } else {
                for (int j = 0; j < 64; ++j) {
                    if (base + j < n) { double d = (double)tile[lane][j]; z = z * mult + om * d * d; }
                }
            }
        }
        __syncthreads();
    }
    if (lane < nrows) zs[c0 + lane] = z;
    // affine map of this block's 64 chunks (s_out = A s_in + B), composed in chunk order by a wave scan
    double A = 1.0, B = 0.0;
    if (lane < nrows) {
        const int64_t len = min((int64_t)ZC, n - (c0 + lane) * ZC);
        A = (len == ZC) ? mult_chunk : pow(mult, (double)len);
        B = z;
    }
    for (int o = 1; o < 64; o <<= 1) {
        const double Ap = __shfl_up(A, o, 64), Bp = __shfl_up(B, o, 64);
        if (lane >= o) { B = A * Bp + B; A = A * Ap; }
    }
    if (lane == 63) { blkA[blockIdx.x] = A; blkB[blockIdx.x] = B; }
}

// (the carry scan over the per-block affine maps -- the state entering every block of 64 chunks -- is done by each k_as_sigma workgroup for
// its own block: the wave-level inclusive scan in rounds of 64 maps that a one-wave kernel used to run in between, stopped at the block)
__global__ void __launch_bounds__(64)
k_as_sigma(const float *__restrict__ x, int64_t n, int tc, double mult, double mult_chunk, const double *__restrict__ zs,
           const double *__restrict__ blkA, const double *__restrict__ blkB, double *__restrict__ out_min, double *__restrict__ out_max, int64_t nchunks)
{
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * 64;
    const int nrows = (int)min((int64_t)64, nchunks - c0);
    // state entering this block: the maps of blocks 0 .. blockIdx.x - 1 composed in order.  Lane l composes its run of ceil(nb / 64)
    // consecutive maps serially (its loads all in flight at once), one wave scan joins the 64 runs: a round of 64 maps per scan (41 scans
    // for an hour, each behind its own loads) was 40 % of a workgroup's life -- the prologue is latency, and nothing else was running yet
    double block_in = 0.0;
    float v[64];
    as_rows_load(v, x, n, c0, 0, lane);                  // (the first tile's rows are on their way meanwhile)
    {
        const int64_t nb = blockIdx.x;
        const int64_t K = (nb + 63) / 64, lo = (int64_t)lane * K, hi = min(nb, lo + K);
        double A = 1.0, B = 0.0;
        for (int64_t j = lo; j < hi; ++j) { const double a = blkA[j], b = blkB[j]; B = a * B + b; A = a * A; }
        for (int o = 1; o < 64; o <<= 1) {
            const double Ap = __shfl_up(A, o, 64), Bp = __shfl_up(B, o, 64);
            if (lane >= o) { B = A * Bp + B; A = A * Ap; }
        }
        block_in = __shfl(B, 63, 64);                    // (the composite applied to the state 0 the file starts from)
    }
    // state entering this lane's chunk: the block's entering state pushed through the preceding chunks of the block
    double avg;
    {
        double A = 1.0, B = 0.0;
        if (lane < nrows) { A = mult_chunk; B = zs[c0 + lane]; }     // only a file's last chunk is short, and nothing follows it
        for (int o = 1; o < 64; o <<= 1) {
            const double Ap = __shfl_up(A, o, 64), Bp = __shfl_up(B, o, 64);
            if (lane >= o) { B = A * Bp + B; A = A * Ap; }
        }
        const double cin = block_in;
        const double s_after = A * cin + B;
        avg = __shfl_up(s_after, 1, 64);
        if (lane == 0) avg = cin;
    }
    double mn = DBL_MAX, mx = 0.0;
    const double om = 1.0 - mult;
    for (int pos = 0; pos < ZC; pos += 64) {
        as_rows_commit(tile, v, n, c0, pos, lane, nrows);
        __syncthreads();
        if (pos + 64 < ZC) as_rows_load(v, x, n, c0, pos + 64, lane);
        if (lane < nrows) {
            const int64_t base = (c0 + lane) * ZC + pos;
            if (base + 64 <= n && base >= tc) {
                float xs[64];
#pragma unroll
                for (int j = 0; j < 64; ++j) xs[j] = tile[lane][j];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    double d = (double)xs[j];
                    avg = avg * mult + om * d * d;
                    mx = fmax(mx, avg); mn = fmin(mn, avg);
                }
            } else {
                for (int j = 0; j < 64; ++j) {
                    const int64_t idx = base + j;
                    if (idx < n) {
                        double d = (double)tile[lane][j];
                        avg = avg * mult + om * d * d;
                        if (idx >= tc) { mx = fmax(mx, avg); mn = fmin(mn, avg); }
                    }
                }
            }
        }
        __syncthreads();
    }
    mn = wmin(mn); mx = wmax(mx);
    if (lane == 0) { out_min[blockIdx.x] = mn; out_max[blockIdx.x] = mx; }
}

// ------------------------------------------------------------------ host driver (enqueue / finish)
size_t jt_arena_bytes_for(int64_t n)
{
    // astats partials + entropy histogram + K-weighting chunk sums (chunks are > 1000 samples, 16 B each) + per-100ms records
    return (size_t)(2048 * (sizeof(AsPartial) + sizeof(AsRuns) + sizeof(AsNF)) + 8192 * 8 + (size_t)(n / ZC + 64) * 16 / 32
                    + (size_t)(n / 512 + 64) * 16 + (size_t)(n / 800 + 64) * (sizeof(jt_spectral) + 64) + (1u << 20));
}

void jt_astats_enqueue(jt_ctx *h, const float *x, int64_t n, int sr, AstatsJob *job, hipStream_t sA, hipStream_t sB, hipStream_t sC,
                       unsigned long long *ehist, bool levels_only)
{
    const double time_constant = 0.05;
    const double mult = std::exp((-1 / time_constant / sr));
    const int tc = (int)std::max(time_constant * sr + .5, 1.0);
    const int nparts = (int)std::min<int64_t>((n + AS_T - 1) / AS_T, 2048);
    const int64_t nchunks = (n + ZC - 1) / ZC;
    const int nsig = (int)((nchunks + 63) / 64);
    // scratch layout (bytes): [head: partials | runs | mm[2] | entropy histogram[8192] | nf | smin[nsig] | smax[nsig]] | zs[nchunks] | carry[nchunks]
    // (what one chain writes is contiguous: chain A partials .. histogram, chain B nf, chain C smin | smax -- one copy per chain)
    const size_t o_part = 0, o_runs = o_part + sizeof(AsPartial) * nparts, o_mm = o_runs + sizeof(AsRuns) * nparts, o_eh = o_mm + 16,
                 o_nf = o_eh + 8 * (size_t)8192, o_smin = o_nf + sizeof(AsNF) * nparts, o_smax = o_smin + 8 * (size_t)nsig,
                 head = o_smax + 8 * (size_t)nsig, o_zs = head, o_carry = o_zs + 8 * (size_t)nchunks,
                 total = o_carry + 8 * 3 * (size_t)nsig;      // blkA | blkB | blk_in
    unsigned char *base = h->as_take(total);
    AsPartial *d_part = reinterpret_cast<AsPartial *>(base + o_part);
    AsRuns *d_runs = reinterpret_cast<AsRuns *>(base + o_runs);
    AsNF *d_nf = reinterpret_cast<AsNF *>(base + o_nf);
    double *d_mm = reinterpret_cast<double *>(base + o_mm), *d_zs = reinterpret_cast<double *>(base + o_zs),
           *d_carry = reinterpret_cast<double *>(base + o_carry), *d_smin = reinterpret_cast<double *>(base + o_smin),
           *d_smax = reinterpret_cast<double *>(base + o_smax);
    (void)ehist;                                                       // every job has its own histogram inside its scratch block
    ehist = reinterpret_cast<unsigned long long *>(base + o_eh);
    if (levels_only) {
        // a region sample reports the RMS / peak levels and the crest factor only (regions_finish): one sweep instead of nine launches,
        // its few partials written straight into the pinned arena (device-visible host memory: no copy to queue)
        unsigned char *hb = h->pin.take<unsigned char>(o_runs);
        hipLaunchKernelGGL(k_as_reduce<false>, dim3(nparts), dim3(AS_T), 0, sA, x, n, reinterpret_cast<AsPartial *>(hb + o_part), ehist);
        *job = AstatsJob{};
        job->hb = hb; job->o_part = o_part; job->nparts = nparts; job->n = n; job->levels_only = true;
        return;
    } else {
        JT_HIP(hipMemsetAsync(ehist, 0, sizeof(unsigned long long) * 8192, sA));
        hipLaunchKernelGGL(k_as_reduce<true>, dim3(nparts), dim3(AS_T), 0, sA, x, n, d_part, ehist);
    }
    hipLaunchKernelGGL(k_as_runs, dim3(nparts), dim3(AS_T), 0, sA, x, n, (const AsPartial *)d_part, nparts, d_mm, d_runs);
    const bool have_nf = n >= tc;
    int nf_parts = nparts;
    if (have_nf) {
        if (tc >= 65) {
            // register/DPP sliding maximum: block maxima, their running maxima, then the per-output combination
            const int64_t nblocks = (n + 63) / 64 + 2;
            const int q = (tc - 1) >> 6;
            float *d_bm = reinterpret_cast<float *>(h->as_take(sizeof(float) * 2 * (size_t)nblocks)), *d_rq = d_bm + nblocks;
            unsigned *d_ub = reinterpret_cast<unsigned *>(h->as_take(256));
            const int64_t nstart = (n - tc + 1 + 63) / 64;                       // start blocks that contain a complete window start
            nf_parts = (int)std::min<int64_t>((nstart + 15) / 16, nparts);                 // 4 waves x 4 start blocks per pass
            hipLaunchKernelGGL(k_nf_blockmax, dim3((unsigned)std::min<int64_t>((nblocks + 3) / 4, 4096)), dim3(256), 0, sB, x, n, d_bm, nblocks, d_ub);
            hipLaunchKernelGGL(k_nf_runmax, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), sizeof(float) * (size_t)(256 + std::max(q - 1, 0) + 3), sB, d_bm, d_rq, nblocks, std::max(q - 1, 0), nstart, d_ub);
            hipLaunchKernelGGL(k_nf_main, dim3(nf_parts), dim3(256), 0, sB, x, n, tc, d_bm, d_rq, nstart, d_nf, h->opts.nf_unpruned ? nullptr : d_ub);
        } else {                                   // very low sample rates: van Herk arrays in HBM
            h->as_g.ensure((size_t)n); h->as_p.ensure((size_t)n);
            const int64_t nb = (n + tc - 1) / tc;
            hipLaunchKernelGGL(k_as_gp, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, sB, x, h->as_g.p, h->as_p.p, n, tc);
            hipLaunchKernelGGL(k_as_noise_floor, dim3(nparts), dim3(AS_T), 0, sB, h->as_g.p, h->as_p.p, n, tc, d_nf);
        }
    }
    const double mult_chunk = std::pow(mult, (double)ZC);
    double *d_blkA = d_carry, *d_blkB = d_carry + nsig;
    hipLaunchKernelGGL(k_as_zs, dim3((unsigned)nsig), dim3(64), 0, sC, x, n, mult, mult_chunk, d_zs, d_blkA, d_blkB, nchunks);
    hipLaunchKernelGGL(k_as_sigma, dim3((unsigned)nsig), dim3(64), 0, sC, x, n, tc, mult, mult_chunk, d_zs, (const double *)d_blkA, (const double *)d_blkB, d_smin, d_smax, nchunks);
    unsigned char *hb = h->pin.take<unsigned char>(head);
    const unsigned long long *eh = reinterpret_cast<const unsigned long long *>(hb + o_eh);
    // one copy per chain, each covering the bytes that chain wrote (the chains may run on different streams)
    JT_HIP(hipMemcpyAsync(hb + o_part, base + o_part, o_nf - o_part, hipMemcpyDeviceToHost, sA));          // partials | runs | mm | histogram
    if (have_nf) JT_HIP(hipMemcpyAsync(hb + o_nf, base + o_nf, o_smin - o_nf, hipMemcpyDeviceToHost, sB));
    JT_HIP(hipMemcpyAsync(hb + o_smin, base + o_smin, head - o_smin, hipMemcpyDeviceToHost, sC));
    job->hb = hb; job->eh = eh; job->o_part = o_part; job->o_runs = o_runs; job->o_nf = o_nf; job->o_smin = o_smin; job->o_smax = o_smax;
    job->nparts = nparts; job->nf_parts = nf_parts; job->nsig = nsig; job->have_nf = have_nf; job->n = n;
}

void jt_astats_finish(const AstatsJob *job, jt_astats *out)
{
    const int nparts = job->nparts, nf_parts = job->nf_parts, nsig = job->nsig; const bool have_nf = job->have_nf;
    const unsigned long long *eh = job->eh;
    const AsPartial *pp = reinterpret_cast<const AsPartial *>(job->hb + job->o_part);
    if (job->levels_only) {
        AsPartial a = pp[0];
        for (int i = 1; i < nparts; ++i) { const AsPartial &b = pp[i]; a.min = std::min(a.min, b.min); a.max = std::max(a.max, b.max); a.sx += b.sx; a.sx2 += b.sx2; a.count += b.count; }
        std::memset(out, 0, sizeof(*out));
        if (a.count == 0) return;
        const double count = (double)a.count;
        out->dc_offset = a.sx / count; out->min_level = a.min; out->max_level = a.max;
        out->peak_level = std::log10(std::max(-a.min, a.max)) * 20;
        out->rms_level = std::log10(std::sqrt(a.sx2 / count)) * 20;
        out->crest_factor = a.sx2 ? std::max(-a.min, a.max) / std::sqrt(a.sx2 / count) : 1;
        out->number_of_samples = count;
        return;
    }
    const AsRuns *pr = reinterpret_cast<const AsRuns *>(job->hb + job->o_runs);
    const AsNF *pn = reinterpret_cast<const AsNF *>(job->hb + job->o_nf);
    const double *smin = reinterpret_cast<const double *>(job->hb + job->o_smin), *smax = reinterpret_cast<const double *>(job->hb + job->o_smax);
    AsPartial a = pp[0];
    for (int i = 1; i < nparts; ++i) {
        const AsPartial &b = pp[i];
        a.min = std::min(a.min, b.min); a.max = std::max(a.max, b.max); a.min_nz = std::min(a.min_nz, b.min_nz);
        a.sx += b.sx; a.sx2 += b.sx2; a.mind = std::min(a.mind, b.mind); a.maxd = std::max(a.maxd, b.maxd); a.d1 += b.d1; a.d2 += b.d2;
        a.zero_runs += b.zero_runs; a.mask_or |= b.mask_or; a.mask_and &= b.mask_and; a.count += b.count;
    }
    AsRuns r{0, 0, 0, 0};
    for (int i = 0; i < nparts; ++i) { r.min_count += pr[i].min_count; r.min_runs += pr[i].min_runs; r.max_count += pr[i].max_count; r.max_runs += pr[i].max_runs; }
    double nf = DBL_MAX; unsigned long long nfc = 0;
    if (have_nf) for (int i = 0; i < nf_parts; ++i) { if (pn[i].nf < nf) { nf = pn[i].nf; nfc = pn[i].cnt; } else if (pn[i].nf == nf) nfc += pn[i].cnt; }
    double min_sig = DBL_MAX, max_sig = 0;
    for (int i = 0; i < nsig; ++i) { min_sig = std::min(min_sig, smin[i]); max_sig = std::max(max_sig, smax[i]); }
    std::memset(out, 0, sizeof(*out));
    const double count = (double)a.count;
    if (a.count == 0) return;
    auto DB = [](double v) { return std::log10(v) * 20; };
    out->dc_offset = a.sx / count; out->min_level = a.min; out->max_level = a.max;
    out->min_difference = a.mind; out->max_difference = a.maxd;
    out->mean_difference = a.d1 / (count - 1); out->rms_difference = std::sqrt(a.d2 / (count - 1));
    out->peak_level = DB(std::max(-a.min, a.max));
    out->rms_level = DB(std::sqrt(a.sx2 / count));
    out->rms_peak = DB(std::sqrt(max_sig));
    out->rms_trough = min_sig != 1 ? DB(std::sqrt(min_sig)) : 0.0;
    out->crest_factor = a.sx2 ? std::max(-a.min, a.max) / std::sqrt(a.sx2 / count) : 1;
    out->flat_factor = DB((r.min_runs + r.max_runs) / (r.min_count + r.max_count));
    out->peak_count = r.min_count + r.max_count;
    out->noise_floor = have_nf ? DB(nf) : NAN; out->noise_floor_count = (double)nfc;
    double ent = 0;
    for (int i = 0; i < 8192; ++i) { double e = eh[i] / count; if (e > 1e-8) ent += e * std::log2(e); }
    out->entropy = -ent / std::log2((double)std::min<unsigned long long>(a.count, 8192));
    out->dynamic_range = DB(2 * std::max(std::fabs(a.min), std::fabs(a.max)) / a.min_nz);
    out->zero_crossings = (double)a.zero_runs; out->zero_crossings_rate = (double)a.zero_runs / count;
    out->number_of_samples = count;
    { unsigned bits = 0; unsigned long long m = a.mask_or & 0xffffffffull; if (m) { unsigned tz = 0; while (!(m & 1)) { m >>= 1; ++tz; } bits = 32 - tz; } out->bit_depth = bits; }
}
