// k_resample.hip — libswresample-equivalent polyphase kaiser-sinc resampling on gfx950 (resample.c: kaiser beta 9,
// filter_size 32, cutoff 0.97, exact_rational phase bank).
//   * 48k/96k -> 44.1k + dbl->s16   (aformat=sample_rates=44100:sample_fmts=s16, filters.go:706-710)
//   * x -> 192 kHz true-peak scan fused with a per-100ms max reduce (ebur128 peak=true, filters.go:626);
//     the oversampled signal is never materialised
//   * 44.1k -> 192k stream for the loudnorm measurement (normalise.go:256-264), K-weighted by k_lane.hip
//
// Mapping ("phase-major"): a workgroup stages a tile of T = 64*R*step input samples (+ taps-1 halo) in LDS.  Because
// gcd(step, P) = 1, any P consecutive outputs carry the P distinct phases, and outputs of equal phase are P apart and
// read inputs `step` apart.  One wave therefore evaluates 64 outputs OF THE SAME PHASE at a time: the 32-36 taps are
// wave-uniform (scalar loads, no LDS/VGPR traffic for coefficients), the inputs come from LDS with a lane stride of
// `step` words through a skewed index (conflict-free for odd and even strides alike).  Tap order and accumulation
// order per output are exactly swresample's (ascending taps, double or float accumulate), so results are
// bit-identical to the scalar resample_common loops.  HBM traffic = 1 input read (+halo) + 1 output write.
#include "jt_internal.h"

constexpr int PP_THREADS = 256;
namespace d147 { constexpr int P = 147, STEP = 160, L = 36, RING = 40, SW = 32, REACH = 224, NOUT = 64 * P, NIN = 64 * STEP; }

__device__ inline int skew(int i) { return i + (i >> 5); }
struct PPRemap { int64_t skip_lo, skip_n, out_base; };     // k_polyphase block remap (see the kernel); out_base: first output of the s16 destination

// MODE 0: true peak (max |out| per 100 ms visibility block, streaming swr: outputs need all taps inside the input)
// MODE 1: flush-mode resample -> s16 (av_clip_int16(lrint(x*32768))), staged in LDS for coalesced stores
// MODE 2: flush-mode resample -> TAcc stream (strided stores, merged in L2)
template <typename TIn, typename TAcc, typename TTap, int MODE>
__global__ void __launch_bounds__(1024)
k_polyphase(const TIn *__restrict__ in, int64_t n, const TTap *__restrict__ bank, int P, int L, int center, int64_t step,
            int64_t m_total, int R, double in_scale, int blk, unsigned long long *__restrict__ block_tp, int64_t nblocks_alloc,
            int16_t *__restrict__ out_s16, TAcc *__restrict__ out_stream, int skip_interior, PPRemap rm)
{
    // block of the tiling this workgroup computes: launches that cover only part of it (the stream edges around k_down147's
    // interior, or the blocks of one output range) skip `skip_n` blocks from `skip_lo` on
    const int64_t bid = (int64_t)blockIdx.x < rm.skip_lo ? (int64_t)blockIdx.x : (int64_t)blockIdx.x + rm.skip_n;
    if (skip_interior) {     // 48k -> 44.1k: interior blocks belong to k_down147 (same block geometry, same interior test)
        const int64_t s0i = bid * d147::NIN, mlo = bid * d147::NOUT;
        if ((s0i - center >= 0) && (s0i - center + 63 * d147::STEP + d147::REACH <= n) && (mlo + d147::NOUT <= m_total)) return;
    }
    extern __shared__ unsigned char smem_pp[];
    TIn *xin = reinterpret_cast<TIn *>(smem_pp);            // staged in the input type (a 96 k -> 44.1 k tile is 20 k samples)
    const int T = 64 * R * (int)step;
    const int nin = T + L;
    int16_t *otile = reinterpret_cast<int16_t *>(smem_pp + (((size_t)(skew(nin) + 2) * sizeof(TIn) + 7) & ~(size_t)7));
    __shared__ unsigned long long slots[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t s0 = bid * T;
    const int64_t m_lo = bid * 64 * R * P;                          // = s0 * P / step exactly
    const int64_t nout = min((int64_t)64 * R * P, m_total - m_lo);
    if (nout <= 0) return;
    const int flush = MODE != 0;
    const int nthreads = blockDim.x;                                // 256, or 1024 when the tile leaves room for one workgroup per CU only
    for (int i = tid; i < nin; i += nthreads) {
        int64_t g = s0 - center + i;
        TIn v = (TIn)0;
        if (g < 0) g = -g;                                           // invert_initial_buffer(): in[-j] = in[j]
        if (g < n) v = in[g];
        else if (flush) { int64_t r = 2 * n - 1 - g; if (r >= 0 && r < n) v = in[r]; }   // resample_flush()
        xin[skew(i)] = v;
    }
    if (MODE == 0 && tid < 8) slots[tid] = 0ull;
    __syncthreads();
    const int64_t b_first = (s0 - center + L - 1 < 0 ? 0 : (s0 - center + L - 1)) / blk;
    // work items: (j, kc) with j = first-output offset (phase selector) and kc = 64-lane chunk of same-phase outputs
    const int nitems = P * R;
    for (int item = wave; item < nitems; item += nthreads / 64) {
        const int j = item / R, kc = item - j * R;
        const int64_t m0 = m_lo + j;
        const int64_t idx0 = m0 * step;
        const int ph = __builtin_amdgcn_readfirstlane((int)(idx0 % P));   // wave-uniform -> scalar tap loads
        const int si0 = (int)(idx0 / P - s0);                         // tile-relative input position of lane 0, chunk 0
        const int k = kc * 64 + lane;
        const int64_t m = m0 + (int64_t)P * k;
        const int si = si0 + (int)step * k;
        const TTap *f = bank + (size_t)ph * L;
        TAcc val = (TAcc)0;
        const bool live = m < m_total;
        if (live) {
            // ascending taps, one fused multiply-add per tap (swresample accumulates mul+add; the fused form differs by < 1 ulp
            // of the accumulator per tap and halves the VALU work)
            if (L == 32) {
#pragma unroll
                for (int i = 0; i < 32; ++i) val = fma((TAcc)((TAcc)xin[skew(si + i)] * (TAcc)in_scale), (TAcc)f[i], val);
            } else if (L == 36) {
#pragma unroll
                for (int i = 0; i < 36; ++i) val = fma((TAcc)((TAcc)xin[skew(si + i)] * (TAcc)in_scale), (TAcc)f[i], val);
            } else {
#pragma unroll 4
                for (int i = 0; i < L; ++i) val = fma((TAcc)((TAcc)xin[skew(si + i)] * (TAcc)in_scale), (TAcc)f[i], val);
            }
        }
        if (MODE == 0) {
            const int64_t last = s0 + si - center + L - 1;
            if (live && last <= n - 1) {
                int64_t b = last / blk;
                if (b >= nblocks_alloc) b = nblocks_alloc - 1;
                int sl = (int)(b - b_first);
                unsigned long long bits = (unsigned long long)__double_as_longlong(fabs((double)val));
                if (sl >= 0 && sl < 8) atomicMax(&slots[sl], bits);
                else atomicMax(&block_tp[b], bits);
            }
        } else if (MODE == 1) {
            if (live) {
                double r = rint((double)val * 32768.0);
                r = r < -32768.0 ? -32768.0 : (r > 32767.0 ? 32767.0 : r);
                otile[j + P * k] = (int16_t)r;
            }
        } else {
            if (live) out_stream[m] = val;
        }
    }
    __syncthreads();
    if (MODE == 0) {
        if (tid < 8 && slots[tid]) {
            int64_t b = b_first + tid;
            if (b >= nblocks_alloc) b = nblocks_alloc - 1;
            atomicMax(&block_tp[b], slots[tid]);
        }
    } else if (MODE == 1) {
        for (int64_t i = tid; i < nout; i += nthreads) out_s16[m_lo - rm.out_base + i] = otile[i];
    }
}

// ---- upsampling variant (step < P, 32 taps): "window-major".  All outputs whose first tap reads the same input sample share
// one 32-sample window, so a lane loads the window from LDS ONCE into registers and evaluates the ~P/step phases that use it
// (4 at 48k->192k, 4.35 at 44.1k->192k); the taps of each phase stay wave-uniform scalars.  LDS reads per output drop from 32
// to ~7 (QL = 1) or ~2 (step == 1, QL = 4 adjacent windows per lane), leaving the FMA pipe as the limiter.  Tap order per
// output is unchanged (ascending), so results are bit-identical to k_polyphase.
template <typename T> __device__ inline void store_quad(T *p, T a, T b, T c, T d)
{
    if constexpr (sizeof(T) == 4) { typedef T v4 __attribute__((ext_vector_type(4))); *reinterpret_cast<v4 *>(p) = v4{a, b, c, d}; }
    else { typedef T v2 __attribute__((ext_vector_type(2))); reinterpret_cast<v2 *>(p)[0] = v2{a, b}; reinterpret_cast<v2 *>(p)[1] = v2{c, d}; }
}

// Stream output (MODE 2) with P % 16 == 0 and NW windows per lane (44.1 k -> 192 k: P = 640).  Two things set the time of the plain
// item loop there, and neither is the FMAs:
//   * a lane's outputs of one window period are P consecutive samples, P samples away from the next lane's, so storing them one
//     by one makes every store instruction touch 64 cache lines for one element each;
//   * every output phase needs its own 32-tap row (an 80 KB bank: the scalar cache holds a fifth of it), and a row fetched for 64
//     windows x 32 FMAs keeps the scalar data path, not the vector ALU, busy.
// Here each wave owns a quarter of the period's outputs (j in [jw0, jw1)) for ALL window offsets that produce them and walks the
// offsets in order: a lane collects four consecutive outputs in registers before one 16-byte store (j & 3 is wave-uniform), and a
// tap row, once in SGPRs, serves the NW windows (64 periods apart) a lane keeps in registers.  Offsets straddling a quarter boundary
// are evaluated by both neighbours, each keeping its own outputs.  Tap order per output is unchanged.
// Output path (measured: 6.7 GB written for a 2.8 GB stream with one 16-byte store per lane and quad -- the 64 lanes of a store sit in
// 64 different cache lines, a line is completed by eight stores ~4 phase rows apart, and with every CU holding hundreds of open lines
// L2 evicts them half-filled; the kernel ran at the HBM write rate, whatever fed the FMAs).  A wave therefore collects 16 consecutive
// outputs of each of its 64 x NW windows in an LDS tile ([NW][64][17]) and writes them out line-wise: lane = 4 * window-in-sixteen +
// quarter, one store instruction covers sixteen windows x 64 contiguous bytes (f32; 128 for f64).
template <typename TIn, typename TAcc, typename TTap, int NW, int NWV = 4, int CHN = 16>
__device__ inline void upsample32_stream_quads(const TIn *xin, const TTap *__restrict__ bank, int P, int step, int64_t m_lo, int64_t m_total,
                                               double in_scale, int skewed, TAcc *__restrict__ out_stream, TAcc *otile_all)
{
    constexpr int L = 32, CH = CHN, OS = CH + 1, QPW = CH / 4;         // QPW quads per window in a chunk: 64 / QPW windows per store instruction
    static_assert(CH == 16 || CH == 8, "chunk length");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // uniform: scalar taps
    TAcc *ot = otile_all + (size_t)wave * NW * 64 * OS;
    const unsigned uP = (unsigned)P, ustep = (unsigned)step;
    const int JW = P / NWV, jw0 = wave * JW, jw1 = jw0 + JW;              // JW is a multiple of 4: chunks start on 16-byte boundaries
    const int off_first = (int)(((unsigned)jw0 * ustep) / uP), off_last = (int)(((unsigned)(jw1 - 1) * ustep) / uP);
    // write-out of the chunk [jc0, jc0 + cnt): window rw of sixteen, quad c
    const int rw = lane / QPW, c4 = (lane % QPW) * 4;
    auto flush = [&](int jc0, int cnt) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
#pragma unroll
            for (int it = 0; it < QPW; ++it) {
                const int win = (64 / QPW) * it + rw;
                const TAcc *src = ot + ((size_t)w * 64 + win) * OS + c4;
                const TAcc v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
                const int64_t m_base = m_lo + (int64_t)P * (w * 64 + win);
                const int64_t left = m_total - m_base;                     // outputs j < left exist
                const int j = jc0 + c4;
                if (c4 + 3 < cnt && j + 3 < left) store_quad(out_stream + m_base + j, v0, v1, v2, v3);
                else {
                    if (c4 < cnt && j < left) out_stream[m_base + j] = v0;
                    if (c4 + 1 < cnt && j + 1 < left) out_stream[m_base + j + 1] = v1;
                    if (c4 + 2 < cnt && j + 2 < left) out_stream[m_base + j + 2] = v2;
                    if (c4 + 3 < cnt && j + 3 < left) out_stream[m_base + j + 3] = v3;
                }
            }
        }
    };
    int jc0 = jw0;                                                       // first output of the open chunk
    for (int off0 = off_first; off0 <= off_last; ++off0) {
        const int off = __builtin_amdgcn_readfirstlane(off0);
        int j_lo = (int)(((unsigned)off * uP + ustep - 1u) / ustep), j_hi = (int)(((unsigned)(off + 1) * uP + ustep - 1u) / ustep);
        j_lo = j_lo > jw0 ? j_lo : jw0; j_hi = j_hi < jw1 ? j_hi : jw1;
        TAcc xw[NW][L];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int si = off + step * (w * 64 + lane);
            if (!skewed) {
                const TIn *wp = xin + si;
                if (in_scale == 1.0) {
#pragma unroll
                    for (int i = 0; i < L; ++i) xw[w][i] = (TAcc)wp[i];
                } else {
#pragma unroll
                    for (int i = 0; i < L; ++i) xw[w][i] = (TAcc)((TAcc)wp[i] * (TAcc)in_scale);
                }
            } else {
                const int sb = si & 31, base = si + (si >> 5);
#pragma unroll
                for (int i = 0; i < L; ++i) xw[w][i] = (TAcc)((TAcc)xin[base + i + ((sb + i) >> 5)] * (TAcc)in_scale);
            }
        }
        int ph = (int)(((unsigned)j_lo * ustep) % uP);
        // the next row's taps are requested before this row's FMAs (two rows of SGPRs, used alternately): with two waves per SIMD
        // nothing else covers the scalar-cache round trip
        auto row = [&](const TTap (&tp)[L], int j) {
            TAcc val[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) val[w] = (TAcc)0;
#pragma unroll
            for (int i = 0; i < L; ++i) {
#pragma unroll
                for (int w = 0; w < NW; ++w) val[w] = fma(xw[w][i], (TAcc)tp[i], val[w]);
            }
            const int jj = j - jc0;
#pragma unroll
            for (int w = 0; w < NW; ++w) ot[((size_t)w * 64 + lane) * OS + jj] = val[w];
            if (jj == CH - 1 || j == jw1 - 1) {                            // (wave-uniform)
                __builtin_amdgcn_wave_barrier();
                flush(jc0, jj + 1);
                __builtin_amdgcn_wave_barrier();
                jc0 = j + 1;
            }
        };
        auto fetch_row = [&](TTap (&tp)[L], int phase) {
            const TTap *f = bank + (size_t)(unsigned)phase * L;
#pragma unroll
            for (int i = 0; i < L; ++i) tp[i] = f[i];
        };
        TTap ta[L], tb[L];
        if (j_lo < j_hi) fetch_row(ta, ph);
        for (int j = j_lo; j < j_hi; j += 2) {
            int ph1 = ph + step; ph1 -= ph1 >= P ? P : 0;
            int ph2 = ph1 + step; ph2 -= ph2 >= P ? P : 0;
            fetch_row(tb, ph1);                                           // (a row past the last one of this offset is simply not used)
            row(ta, j);
            if (j + 1 < j_hi) {
                fetch_row(ta, ph2);
                row(tb, j + 1);
            }
            ph = ph2;
        }
    }
}

template <typename TIn, typename TAcc, typename TTap, int MODE, int QL>
__global__ void __launch_bounds__(PP_THREADS)
k_upsample32(const TIn *__restrict__ in, int64_t n, const TTap *__restrict__ bank, int P, int center, int step,
             int64_t m_total, int R, double in_scale, int blk, unsigned long long *__restrict__ block_tp, int64_t nblocks_alloc,
             TAcc *__restrict__ out_stream, int skewed)
{
    constexpr int L = 32;
    constexpr int WL = L + QL - 1;                                    // QL > 1 only with step == 1
    extern __shared__ unsigned char smem_pp[];
    TIn *xin = reinterpret_cast<TIn *>(smem_pp);          // staged in the input type (halves the LDS footprint for f32 / s16 sources)
    const int T = 64 * R * QL * step;
    const int nin = T + L;
    __shared__ unsigned long long slots[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t s0 = (int64_t)blockIdx.x * T;
    const int64_t m_lo = (int64_t)blockIdx.x * 64 * R * QL * P;
    if (m_lo >= m_total) return;
    const int flush = MODE != 0;
    // QL > 1 (step == 1): lane windows start QL samples apart, so an UNSKEWED tile read with 16-byte loads is conflict-free (each
    // lane a distinct 16-byte segment).  QL == 1: lane windows start `step` samples apart; for an odd step the unskewed tile is
    // (nearly) conflict-free as it is and a window is 32 reads off ONE address with immediate offsets, where the skewed index costs
    // ~7 VALU instructions per read -- more than the 4.35 FMAs a staged sample feeds.  The skew remains for even steps.
    constexpr bool VEC = QL * sizeof(TIn) == 16;
    const bool flat = VEC || !skewed;
    if (s0 - center >= 0 && s0 - center + nin <= n) {
        const TIn *src = in + (s0 - center);                         // interior workgroup: no reflection, no end of stream
        for (int i = tid; i < nin; i += PP_THREADS) xin[flat ? i : skew(i)] = src[i];
    } else {
        for (int i = tid; i < nin; i += PP_THREADS) {
            int64_t g = s0 - center + i;
            TIn v = (TIn)0;
            if (g < 0) g = -g;                                       // invert_initial_buffer(): in[-j] = in[j]
            if (g < n) v = in[g];
            else if (flush) { int64_t r = 2 * n - 1 - g; if (r >= 0 && r < n) v = in[r]; }   // resample_flush()
            xin[flat ? i : skew(i)] = v;
        }
    }
    if (MODE == 0 && tid < 8) slots[tid] = 0ull;
    __syncthreads();
    if constexpr (MODE == 2 && QL == 1) {
        if ((P & 15) == 0 && (R == 1 || R == 2)) {
            // the waves' output tiles sit behind the input tile (16-byte aligned)
            TAcc *otile = reinterpret_cast<TAcc *>(smem_pp + (((size_t)(nin + (nin >> 5) + 4) * sizeof(TIn) + 15) & ~(size_t)15));
            if (R == 2) upsample32_stream_quads<TIn, TAcc, TTap, 2>(xin, bank, P, step, m_lo, m_total, in_scale, skewed, out_stream, otile);
            else upsample32_stream_quads<TIn, TAcc, TTap, 1>(xin, bank, P, step, m_lo, m_total, in_scale, skewed, out_stream, otile);
            return;
        }
    }
    const int64_t b_first = (s0 - center + L - 1 < 0 ? 0 : (s0 - center + L - 1)) / blk;
    const int nitems = step * R;
    // All index arithmetic below is 32-bit and strength-reduced: the phase advances by `step` modulo P from one output to the next,
    // a lane's outputs of one item are m_base + j, and the end of the stream is a per-lane bound on j.  (Written with 64-bit
    // products, divisions and modulos per output, this loop spent ~150 scalar instructions per 32 FMAs.)
    TAcc run_v = (TAcc)0; int run_sl = -1;                           // MODE 0: per-lane running maximum for block slot run_sl
    auto flush_run = [&]() {
        if (run_sl < 0) return;
        TAcc v = run_v;
#pragma unroll
        for (int mm = 1; mm < 64; mm <<= 1) v = fmax(v, __shfl_xor(v, mm, 64));
        if (lane == 0 && v > (TAcc)0) atomicMax(&slots[run_sl], (unsigned long long)__double_as_longlong((double)v));
        run_v = (TAcc)0; run_sl = -1;
    };
    for (int item0 = wave; item0 < nitems; item0 += PP_THREADS / 64) {
        const int item = __builtin_amdgcn_readfirstlane(item0);
        const int off = item / R, kc = item - off * R;
        const unsigned uP = (unsigned)P, ustep = (unsigned)step;
        const int j_lo = (int)(((unsigned)off * uP + ustep - 1u) / ustep);
        int j_hi = (int)(((unsigned)(off + 1) * uP + ustep - 1u) / ustep);
        j_hi = j_hi < P ? j_hi : P;
        const int qq = (kc * 64 + lane) * QL;                       // first window of this lane (tile-relative q)
        const int si = off + step * qq;
        TAcc xw[WL];
        if (VEC) {
            // si = QL * window index: 16-byte aligned in the unskewed tile; the last vector reads up to QL - 1 staged words past
            // the window (inside the tile's slack)
            constexpr int E = 16 / (int)sizeof(TIn), NV = (WL + E - 1) / E;
            typedef TIn vecT __attribute__((ext_vector_type(E)));
            const vecT *vp = reinterpret_cast<const vecT *>(xin + si);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const vecT t = vp[k];
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (k * E + e < WL) { const TAcc v = (TAcc)t[e]; xw[k * E + e] = MODE == 0 ? v : (TAcc)(v * (TAcc)in_scale); }
            }
        } else if (!skewed) {
            const TIn *wp = xin + si;
            if (MODE == 0 || in_scale == 1.0) {                       // (s16 sources arrive with the scale folded into the taps)
#pragma unroll
                for (int i = 0; i < WL; ++i) xw[i] = (TAcc)wp[i];
            } else {
#pragma unroll
                for (int i = 0; i < WL; ++i) xw[i] = (TAcc)((TAcc)wp[i] * (TAcc)in_scale);
            }
        } else {
            const int sb = si & 31, base = si + (si >> 5);          // skew(si + i) = base + i + ((sb + i) >> 5)
#pragma unroll
            for (int i = 0; i < WL; ++i) {
                const TAcc v = (TAcc)xin[base + i + ((sb + i) >> 5)];
                xw[i] = MODE == 0 ? v : (TAcc)(v * (TAcc)in_scale);   // the true-peak instances run with in_scale == 1
            }
        }
        TAcc vmax[QL];
#pragma unroll
        for (int u = 0; u < QL; ++u) vmax[u] = (TAcc)0;
        const int64_t m_base = m_lo + (int64_t)P * qq;               // output index of (this lane's first window, phase 0)
        int jlim[QL];                                                // outputs j < jlim[u] exist (m < m_total)
#pragma unroll
        for (int u = 0; u < QL; ++u) {
            const int64_t left = m_total - (m_base + (int64_t)P * u);
            jlim[u] = left <= 0 ? 0 : (left >= (int64_t)P ? P : (int)left);
        }
        int ph = (int)(((unsigned)j_lo * ustep) % uP);
        auto row = [&](const TTap (&tp)[L], int j) {
            // tap-major, window-minor: the QL accumulation chains advance together (each chain keeps its ascending tap order), so
            // consecutive FMAs are independent
            TAcc val[QL];
#pragma unroll
            for (int u = 0; u < QL; ++u) val[u] = (TAcc)0;
#pragma unroll
            for (int i = 0; i < L; ++i) {
#pragma unroll
                for (int u = 0; u < QL; ++u) val[u] = fma(xw[u + i], (TAcc)tp[i], val[u]);
            }
#pragma unroll
            for (int u = 0; u < QL; ++u) {
                if (MODE == 0) vmax[u] = j < jlim[u] ? fmax(vmax[u], fabs(val[u])) : vmax[u];
                else if (j < jlim[u]) out_stream[m_base + (int64_t)P * u + j] = val[u];
            }
        };
        auto fetch_row = [&](TTap (&tp)[L], int phase) {
            const TTap *f = bank + (size_t)(unsigned)phase * L;
#pragma unroll
            for (int i = 0; i < L; ++i) tp[i] = f[i];
        };
        // (requesting the next row's taps before this row's FMAs, as the stream variant does, made these instances SLOWER: 3.0 -> 3.5 ms
        // and 3.5 -> 3.7 ms -- with four waves per SIMD the scalar round trip is already covered and the second row of SGPRs spills)
        for (int j = j_lo; j < j_hi; ++j) {
            TTap tp[L];
            fetch_row(tp, ph);
            row(tp, j);
            ph += step; ph -= ph >= P ? P : 0;
        }
        if (MODE == 0) {
            // 100 ms block of every window's last input sample, relative to the tile's first block (32-bit: the tile spans a few
            // blocks at most).  A wave's windows almost always sit in ONE block: then the wave reduces its maxima in registers and
            // a single lane does the LDS atomic (64 lanes x QL atomics on one address serialise in the LDS unit otherwise).
            const int64_t base0 = s0 - center + L - 1;                              // `last` of tile sample 0 (block-uniform)
            const int64_t bq = (base0 < 0 ? base0 - (blk - 1) : base0) / blk;        // floor division, once per workgroup value
            const unsigned brem = (unsigned)(base0 - bq * blk);
            const bool smallq = (unsigned)(T + L) < 2u * (unsigned)blk;
            int sl[QL]; TAcc vm[QL];
#pragma unroll
            for (int u = 0; u < QL; ++u) {
                const unsigned t = brem + (unsigned)(si + u * step);
                const int64_t last = base0 + si + u * step;
                // t < blk + T + L: with a tile shorter than two blocks the quotient is 0, 1 or 2 (two compares instead of a division)
                const unsigned tq = smallq ? (unsigned)(t >= (unsigned)blk) + (unsigned)(t >= 2u * (unsigned)blk) : t / (unsigned)blk;
                int64_t b = bq + (int64_t)tq;
                if (b >= nblocks_alloc) b = nblocks_alloc - 1;
                sl[u] = (int)(b - b_first);
                vm[u] = (last >= 0 && last <= n - 1) ? vmax[u] : (TAcc)0;
            }
            const int sl_lo = __builtin_amdgcn_readfirstlane(sl[0]), sl_hi = __builtin_amdgcn_readlane(sl[QL - 1], 63);
            if (sl_lo == sl_hi && sl_lo >= 0 && sl_lo < 8) {
                // consecutive items of a wave almost always land in the same block too: the lanes keep running maxima and the
                // wave reduces them once per block it touches (flush_run), not once per item
                if (run_sl != sl_lo) { flush_run(); run_sl = sl_lo; }
#pragma unroll
                for (int u = 0; u < QL; ++u) run_v = fmax(run_v, vm[u]);
            } else {
#pragma unroll
                for (int u = 0; u < QL; ++u) {
                    if (vm[u] > (TAcc)0) {
                        const unsigned long long bits = (unsigned long long)__double_as_longlong((double)vm[u]);
                        if (sl[u] >= 0 && sl[u] < 8) atomicMax(&slots[sl[u]], bits);
                        else atomicMax(&block_tp[b_first + sl[u]], bits);
                    }
                }
            }
        }
    }
    if (MODE == 0) {
        flush_run();
        __syncthreads();
        if (tid < 8 && slots[tid]) {
            int64_t b = b_first + tid;
            if (b >= nblocks_alloc) b = nblocks_alloc - 1;
            atomicMax(&block_tp[b], slots[tid]);
        }
    }
}

// The stream upsampler with eight waves per workgroup, for sources whose tile leaves room for one workgroup per CU only (f64: a tile
// of 64 windows x 147 samples is 75 KB).  With one window per lane a row is 32 dependent FMAs, so a SIMD needs a second wave to keep
// issuing: four waves per CU ran at 10 ms for the hour-long stream, eight at half that, SIXTEEN (four per SIMD; the output chunks
// halved to eight samples so that sixteen [64][9] tiles fit beside the input tile, 149 KB) at 3.3 ms instead of 4.3.  93 VGPRs.
template <typename TIn, typename TAcc, typename TTap, int NWV = 8, int CHN = 16>
__global__ void __launch_bounds__(64 * NWV)
k_upsample32_stream8(const TIn *__restrict__ in, int64_t n, const TTap *__restrict__ bank, int P, int center, int step, int64_t m_total,
                     double in_scale, TAcc *__restrict__ out_stream)
{
    constexpr int L = 32, NT = 64 * NWV;
    extern __shared__ unsigned char smem_pp[];
    TIn *xin = reinterpret_cast<TIn *>(smem_pp);
    const int T = 64 * step, nin = T + L;
    const int tid = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * T;
    const int64_t m_lo = (int64_t)blockIdx.x * 64 * P;
    if (m_lo >= m_total) return;
    if (s0 - center >= 0 && s0 - center + nin <= n) {
        const TIn *src = in + (s0 - center);                         // interior workgroup: no reflection, no end of stream
        for (int i = tid; i < nin; i += NT) xin[i] = src[i];
    } else {
        for (int i = tid; i < nin; i += NT) {
            int64_t g = s0 - center + i;
            TIn v = (TIn)0;
            if (g < 0) g = -g;                                       // invert_initial_buffer(): in[-j] = in[j]
            if (g < n) v = in[g];
            else { int64_t r = 2 * n - 1 - g; if (r >= 0 && r < n) v = in[r]; }   // resample_flush()
            xin[i] = v;
        }
    }
    __syncthreads();
    TAcc *otile = reinterpret_cast<TAcc *>(smem_pp + (((size_t)(nin + 4) * sizeof(TIn) + 15) & ~(size_t)15));
    upsample32_stream_quads<TIn, TAcc, TTap, 1, NWV, CHN>(xin, bank, P, step, m_lo, m_total, in_scale, 0, out_stream, otile);
}

// ------------------------------------------------------------------ Pass 3 in one sweep: swr -> 192 kHz, K-weighting, sample peak, chunk partials
// loudnorm's first pass (normalise.go:226-346) resamples the Pass-2 output to 192 kHz and meters it.  As two kernels the 192 kHz stream
// (4.35 samples per input sample, f64 behind the limiter prefix: 5.5 GB for an hour) was written to HBM by the upsampler and read back by
// the K-weighting sweep -- 15 GB of the step's 77 GB of traffic, 4.5 ms on the critical path.  Here the stream never
// leaves the registers: a thread of k_upsample32_stream8's layout produces a contiguous run of JW = P / NWV outputs of one polyphase
// period (lane = period, wave = j-range) in ascending order, so it runs the K-weighting recurrence over that run FROM ZERO STATE as
// the samples appear and keeps what k_kw1 keeps per chunk: the end state, sum zs^2, max |x| and the four sums zs_j g_j[k].  The filter is
// linear, so the NWV runs of a period fold into the period's own zero-state results exactly: with s_w the state the period's zero-state
// run carries into run w (s_0 = 0, s_{w+1} = F^JW s_w + e_w) and r_w = c_w + G s_w (G = sum_j g_j g_j'),
//   sum y^2 = sum_w (q_w + 2 s_w . c_w + s_w' G s_w),   cross = sum_w r_w' F^(JW w),   end state = s_NWV,
// a few hundred flops per period, done by the first wave from partials parked in the (dead) input tile.  A chunk of the K-weighting job
// is then one PERIOD (640 outputs at 44.1 -> 192 kHz: 30 chunks per 100 ms block) and k_kw_fix / k_kw_blocks run unchanged.
// The meter also takes loudnorm's flush frame -- the last 556 800 samples again, behind the stream (jt_api.cpp loudnorm_meter_len): that is
// a whole number of periods, so chunk c past the end of the stream is period c - flush / P; only the chunk that holds the stream's last
// sample mixes two periods and goes to k_p3_boundary (one workgroup).  Every 192 kHz sample is the tap sum, in the tap order, of the
// stand-alone upsampler: sample peaks are bit-identical, energies agree to the rounding of a different chunking (1e-12 LU).
struct P3Merge {
    double FJ[16];          // F^JW
    double G[10];           // sum_{j < JW} g_j g_j', upper triangle row-major
    double Gt[10];          // the same over the rows of the tail chunk's last, partial run
    const double *Mpow;     // [NWV][16]: F^(JW w), device
};
struct P3Section { int64_t chunk0, chunk_end, period_delta; unsigned block0; };     // chunks [chunk0, chunk_end): source period = chunk + period_delta

#define P3_KW_STEP(X)                                                   \
    {                                                                    \
        const double xx = (X);                                           \
        const double y = fma(k.b0, xx, s1);                              \
        s1 = fma(-k.a1, y, fma(k.b1, xx, s2));                           \
        s2 = fma(-k.a2, y, k.b2 * xx);                                   \
        zz = fma(k.c0, y, t1);                                           \
        t1 = fma(-k.d1, zz, fma(k.c1, y, t2));                           \
        t2 = fma(-k.d2, zz, k.c2 * y);                                   \
    }

template <typename TIn, typename TAcc, typename TTap, int NWV>
__global__ void __launch_bounds__(64 * NWV)
k_p3_fused(const TIn *__restrict__ in, int64_t n, const TTap *__restrict__ bank, int P, int center, int step, double in_scale,
           P3Section secA, P3Section secB, int64_t tail_chunk, int tail_rows, KwCoef k, const double *__restrict__ gtab, P3Merge M,
           double *__restrict__ zs_out, double *__restrict__ csum, double *__restrict__ cpeak, double *__restrict__ cross)
{
    constexpr int L = 32, NT = 64 * NWV, PS = 10;                       // PS doubles parked per thread
    extern __shared__ unsigned char smem_pp[];
    TIn *xin = reinterpret_cast<TIn *>(smem_pp);
    const int T = 64 * step, nin = T + L;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const bool inB = blockIdx.x >= secB.block0;
    const P3Section sec = inB ? secB : secA;
    const int64_t c0 = sec.chunk0 + (int64_t)(blockIdx.x - sec.block0) * 64;        // first chunk of this workgroup
    if (c0 >= sec.chunk_end) return;
    const int nrows = (int)min((int64_t)64, sec.chunk_end - c0);
    const int64_t q0 = c0 + sec.period_delta;                                        // first source period
    const int64_t s0 = q0 * step;
    if (s0 - center >= 0 && s0 - center + nin <= n) {
        const TIn *src = in + (s0 - center);                         // interior workgroup: no reflection, no end of stream
        for (int i = tid; i < nin; i += NT) xin[i] = src[i];
    } else {
        for (int i = tid; i < nin; i += NT) {
            int64_t g = s0 - center + i;
            TIn v = (TIn)0;
            if (g < 0) g = -g;                                       // invert_initial_buffer(): in[-j] = in[j]
            if (g < n) v = in[g];
            else { int64_t r = 2 * n - 1 - g; if (r >= 0 && r < n) v = in[r]; }   // resample_flush()
            xin[i] = v;
        }
    }
    __syncthreads();
    const unsigned uP = (unsigned)P, ustep = (unsigned)step;
    const int JW = P / NWV, jw0 = wave * JW, jw1 = jw0 + JW;
    const int off_first = (int)(((unsigned)jw0 * ustep) / uP), off_last = (int)(((unsigned)(jw1 - 1) * ustep) / uP);
    // rows of this lane's chunk that exist (every row, except in the stream's last chunk)
    const int jlim = lane < nrows ? ((c0 + lane == tail_chunk) ? tail_rows : P) : 0;
    double s1 = 0, s2 = 0, t1 = 0, t2 = 0, zz = 0, acc = 0, pk = 0, x0c = 0, x1c = 0, x2c = 0, x3c = 0;
    for (int off0 = off_first; off0 <= off_last; ++off0) {
        const int off = __builtin_amdgcn_readfirstlane(off0);
        int j_lo = (int)(((unsigned)off * uP + ustep - 1u) / ustep), j_hi = (int)(((unsigned)(off + 1) * uP + ustep - 1u) / ustep);
        j_lo = j_lo > jw0 ? j_lo : jw0; j_hi = j_hi < jw1 ? j_hi : jw1;
        TAcc xw[L];
        {
            const TIn *wp = xin + (off + step * lane);
            if (in_scale == 1.0) {
#pragma unroll
                for (int i = 0; i < L; ++i) xw[i] = (TAcc)wp[i];
            } else {
#pragma unroll
                for (int i = 0; i < L; ++i) xw[i] = (TAcc)((TAcc)wp[i] * (TAcc)in_scale);
            }
        }
        int ph = (int)(((unsigned)j_lo * ustep) % uP);
        auto row = [&](const TTap (&tp)[L], int j) {
            TAcc val = (TAcc)0;
#pragma unroll
            for (int i = 0; i < L; ++i) val = fma(xw[i], (TAcc)tp[i], val);
            const double *gp = gtab + 4 * (j - jw0);                    // wave-uniform: scalar loads
            const double g0 = gp[0], g1 = gp[1], g2 = gp[2], g3 = gp[3];
            if (j < jlim) {
                const double u = (double)val;
                P3_KW_STEP(u)
                acc = fma(zz, zz, acc); pk = fmax(pk, fabs(u));
                x0c = fma(zz, g0, x0c); x1c = fma(zz, g1, x1c); x2c = fma(zz, g2, x2c); x3c = fma(zz, g3, x3c);
            }
        };
        auto fetch_row = [&](TTap (&tp)[L], int phase) {
            const TTap *f = bank + (size_t)(unsigned)phase * L;
#pragma unroll
            for (int i = 0; i < L; ++i) tp[i] = f[i];
        };
        TTap ta[L], tb[L];
        if (j_lo < j_hi) fetch_row(ta, ph);
        for (int j = j_lo; j < j_hi; j += 2) {
            int ph1 = ph + step; ph1 -= ph1 >= P ? P : 0;
            int ph2 = ph1 + step; ph2 -= ph2 >= P ? P : 0;
            fetch_row(tb, ph1);
            row(ta, j);
            if (j + 1 < j_hi) {
                fetch_row(ta, ph2);
                row(tb, j + 1);
            }
            ph = ph2;
        }
    }
    __syncthreads();                                                    // every wave is done with the input tile: park the runs' results in it
    double *park = reinterpret_cast<double *>(smem_pp);
    {
        double *pp = park + ((size_t)wave * 64 + lane) * PS;
        pp[0] = s1; pp[1] = s2; pp[2] = t1; pp[3] = t2; pp[4] = acc; pp[5] = pk; pp[6] = x0c; pp[7] = x1c; pp[8] = x2c; pp[9] = x3c;
    }
    __syncthreads();
    if (wave != 0 || lane >= nrows) return;
    // fold the NWV runs of this lane's period (in order) into the period's zero-state results
    const int w_last = jlim >= P ? NWV - 1 : (jlim - 1) / JW;           // the tail chunk ends inside run w_last
    double s[4] = {0, 0, 0, 0}, Q = 0, PK = 0, C[4] = {0, 0, 0, 0};
    for (int w = 0; w <= w_last; ++w) {
        const double *pp = park + ((size_t)w * 64 + lane) * PS;
        const double e[4] = {pp[0], pp[1], pp[2], pp[3]}, q = pp[4], c[4] = {pp[6], pp[7], pp[8], pp[9]};
        PK = fmax(PK, pp[5]);
        const double *Gm = (jlim < P && w == w_last) ? M.Gt : M.G;
        // Gs = G s (symmetric, upper triangle stored)
        double Gs[4];
        {
            const double g00 = Gm[0], g01 = Gm[1], g02 = Gm[2], g03 = Gm[3], g11 = Gm[4], g12 = Gm[5], g13 = Gm[6], g22 = Gm[7], g23 = Gm[8], g33 = Gm[9];
            Gs[0] = g00 * s[0] + g01 * s[1] + g02 * s[2] + g03 * s[3];
            Gs[1] = g01 * s[0] + g11 * s[1] + g12 * s[2] + g13 * s[3];
            Gs[2] = g02 * s[0] + g12 * s[1] + g22 * s[2] + g23 * s[3];
            Gs[3] = g03 * s[0] + g13 * s[1] + g23 * s[2] + g33 * s[3];
        }
        double lin = 0, quad = 0, r[4];
        for (int a = 0; a < 4; ++a) { lin += s[a] * c[a]; quad += s[a] * Gs[a]; r[a] = c[a] + Gs[a]; }
        Q += q + 2.0 * lin + quad;
        const double *Mw = M.Mpow + 16 * w;                             // F^(JW w), row-major: (state after JW w steps)[l] = sum_k Mw[l][k] s[k]
        for (int kk = 0; kk < 4; ++kk) C[kk] += r[0] * Mw[0 * 4 + kk] + r[1] * Mw[1 * 4 + kk] + r[2] * Mw[2 * 4 + kk] + r[3] * Mw[3 * 4 + kk];
        double ns[4];
        for (int a = 0; a < 4; ++a) ns[a] = M.FJ[a * 4 + 0] * s[0] + M.FJ[a * 4 + 1] * s[1] + M.FJ[a * 4 + 2] * s[2] + M.FJ[a * 4 + 3] * s[3] + e[a];
        for (int a = 0; a < 4; ++a) s[a] = ns[a];
    }
    const int64_t c = c0 + lane;
    zs_out[c * 4 + 0] = s[0]; zs_out[c * 4 + 1] = s[1]; zs_out[c * 4 + 2] = s[2]; zs_out[c * 4 + 3] = s[3];
    csum[c] = Q; cpeak[c] = PK;
    cross[c * 4 + 0] = C[0]; cross[c * 4 + 1] = C[1]; cross[c * 4 + 2] = C[2]; cross[c * 4 + 3] = C[3];
}

// The one chunk that holds the stream's last sample when the flush copy follows it: rows below `t1` come from period `qa`, the rest from
// period `qb` (= qa - flush / P).  One workgroup: 64 lanes evaluate the P tap sums (same taps, same order), lane 0 runs the chunk's
// zero-state recurrence over them.
template <typename TIn, typename TAcc, typename TTap>
__global__ void __launch_bounds__(64)
k_p3_boundary(const TIn *__restrict__ in, int64_t n, const TTap *__restrict__ bank, int P, int center, int step, double in_scale, int64_t qa, int64_t qb,
              int n_first, int rows, int64_t chunk, KwCoef k, const double *__restrict__ gtab, double *__restrict__ zs_out, double *__restrict__ csum,
              double *__restrict__ cpeak, double *__restrict__ cross)
{
    constexpr int L = 32;
    extern __shared__ unsigned char smem_pp[];
    double *u = reinterpret_cast<double *>(smem_pp);
    for (int j = threadIdx.x; j < rows; j += 64) {
        const int64_t q = j < n_first ? qa : qb;
        const int off = (int)(((unsigned)j * (unsigned)step) / (unsigned)P), ph = (int)(((unsigned)j * (unsigned)step) % (unsigned)P);
        const int64_t g0 = q * step + off - center;
        const TTap *f = bank + (size_t)ph * L;
        TAcc val = (TAcc)0;
        for (int i = 0; i < L; ++i) {
            int64_t g = g0 + i;
            TIn v = (TIn)0;
            if (g < 0) g = -g;
            if (g < n) v = in[g];
            else { const int64_t r = 2 * n - 1 - g; if (r >= 0 && r < n) v = in[r]; }
            const TAcc xv = in_scale == 1.0 ? (TAcc)v : (TAcc)((TAcc)v * (TAcc)in_scale);
            val = fma(xv, (TAcc)f[i], val);
        }
        u[j] = (double)val;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double s1 = 0, s2 = 0, t1 = 0, t2 = 0, zz = 0, acc = 0, pk = 0, xc[4] = {0, 0, 0, 0};
    for (int j = 0; j < rows; ++j) {
        const double x = u[j];
        P3_KW_STEP(x)
        acc = fma(zz, zz, acc); pk = fmax(pk, fabs(x));
        for (int a = 0; a < 4; ++a) xc[a] = fma(zz, gtab[4 * j + a], xc[a]);
    }
    zs_out[chunk * 4 + 0] = s1; zs_out[chunk * 4 + 1] = s2; zs_out[chunk * 4 + 2] = t1; zs_out[chunk * 4 + 3] = t2;
    csum[chunk] = acc; cpeak[chunk] = pk;
    for (int a = 0; a < 4; ++a) cross[chunk * 4 + a] = xc[a];
}
#undef P3_KW_STEP

static void p3_mat4_mul(const double *A, const double *B, double *C)
{
    double t[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int q = 0; q < 4; ++q) s += A[i * 4 + q] * B[q * 4 + j]; t[i * 4 + j] = s; }
    std::memcpy(C, t, sizeof(t));
}

#ifndef JT_P3_NWV
#define JT_P3_NWV 8
#endif
constexpr int P3_NWV = JT_P3_NWV;                                   // waves per workgroup of k_p3_fused = runs per period (8: two workgroups per CU; 16: one)
// can this (rate pair, block) be measured by the fused sweep?  (44.1 kHz -> 192 kHz: P = 640, step = 147)
bool jt_p3_fused_supported(int phase_count, int filter_length, int64_t step, int blk, int64_t flush)
{
    return filter_length == 32 && step < phase_count && (step & 1) && phase_count % 64 == 0 && blk % phase_count == 0 && flush % phase_count == 0 &&
           phase_count % P3_NWV == 0 && phase_count <= 2400 && (size_t)(64 * step + 36) * sizeof(double) <= 100 * 1024;
}

// F^(JW w), w < P3_NWV, for the fold (JW = P / P3_NWV): depends on the K-weighting coefficients and the period only -- the caller uploads it once
int jt_p3_fold_powers(const KwCoef &k, int P, double *out /* P3_NWV * 16 */)
{
    const int JW = P / P3_NWV;
    double F[16] = {-k.a1, 1, 0, 0,   -k.a2, 0, 0, 0,   k.c1 - k.d1 * k.c0, 0, -k.d1, 1,   k.c2 - k.d2 * k.c0, 0, -k.d2, 0};
    double FJ[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, Bq[16];
    std::memcpy(Bq, F, sizeof F);
    for (int e = JW; e > 0; e >>= 1) { if (e & 1) p3_mat4_mul(FJ, Bq, FJ); p3_mat4_mul(Bq, Bq, Bq); }
    double cur[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int w = 0; w < P3_NWV; ++w) { std::memcpy(out + 16 * w, cur, sizeof cur); p3_mat4_mul(FJ, cur, cur); }
    return P3_NWV * 16;
}

// The sweep of a K-weighting job over chunks of P outputs (jt_kweight_enqueue_sweep with chunk_len = P): m_total outputs of the resampler,
// followed by the last `flush` of them again (0: no flush frame).  mpow_dev: jt_p3_fold_powers' table on the device.
template <typename TIn, typename TAcc, typename TTap>
static void launch_p3_fused_t(const TIn *in, int64_t n, const TTap *bank, int P, int center, int64_t step, int64_t m_total, int64_t flush, double in_scale,
                              const KwSweep &W, const double *mpow_dev, hipStream_t s)
{
    constexpr int NWV = P3_NWV;
    const int JW = P / NWV;
    JT_REQUIRE(W.L == P && P % NWV == 0 && W.gtab && W.gtab_host, JT_E_INVAL, "fused Pass-3 sweep: chunk length must be the period");
    const int64_t m_meter = m_total + flush;
    JT_REQUIRE(W.nchunks == (m_meter + P - 1) / P, JT_E_INVAL, "fused Pass-3 sweep: chunk count");
    // constants of the fold: F^JW, its powers, the Gram matrices of the table's first JW rows
    const KwCoef &k = W.k;
    double F[16] = {-k.a1, 1, 0, 0,   -k.a2, 0, 0, 0,   k.c1 - k.d1 * k.c0, 0, -k.d1, 1,   k.c2 - k.d2 * k.c0, 0, -k.d2, 0};
    P3Merge M; std::memset(&M, 0, sizeof M);
    double FJ[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, Bq[16];
    std::memcpy(Bq, F, sizeof F);
    for (int e = JW; e > 0; e >>= 1) { if (e & 1) p3_mat4_mul(FJ, Bq, FJ); p3_mat4_mul(Bq, Bq, Bq); }
    std::memcpy(M.FJ, FJ, sizeof FJ);
    M.Mpow = mpow_dev;
    const int64_t c_b = m_total / P; const int t1 = (int)(m_total - c_b * P);
    const int64_t tail_chunk = W.nchunks - 1; const int tail_rows = (int)(m_meter - tail_chunk * P);
    const int tail_run_rows = tail_rows >= P ? JW : tail_rows - ((tail_rows - 1) / JW) * JW;
    for (int j = 0; j < JW; ++j) {
        int u = 0;
        for (int a = 0; a < 4; ++a) for (int b = a; b < 4; ++b, ++u) {
            const double p_ = W.gtab_host[(size_t)4 * j + a] * W.gtab_host[(size_t)4 * j + b];
            M.G[u] += p_; if (j < tail_run_rows) M.Gt[u] += p_;
        }
    }
    P3Section A{0, 0, 0, 0}, B{0, 0, 0, 0};
    const bool mixed = flush > 0 && t1 > 0;
    if (flush > 0) {
        A.chunk_end = c_b;                                              // whole periods of the stream itself
        B.chunk0 = c_b + (mixed ? 1 : 0); B.chunk_end = W.nchunks; B.period_delta = -(flush / P);
    } else A.chunk_end = W.nchunks;                                     // (the last chunk is the tail: t1 rows, or P)
    const unsigned nbA = (unsigned)((A.chunk_end - A.chunk0 + 63) / 64), nbB = (unsigned)((std::max<int64_t>(0, B.chunk_end - B.chunk0) + 63) / 64);
    A.block0 = 0; B.block0 = nbA;
    if (!(flush > 0)) B.block0 = nbA + 1;                               // (no block belongs to B)
    const size_t tile = (sizeof(TIn) * (size_t)(64 * step + 32 + 4) + 15) & ~(size_t)15;
    const size_t smem = std::max(tile, sizeof(double) * (size_t)NWV * 64 * 10);
    auto kf = k_p3_fused<TIn, TAcc, TTap, NWV>;
    JT_HIP(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (nbA + nbB > 0)
        hipLaunchKernelGGL(kf, dim3(nbA + (flush > 0 ? nbB : 0)), dim3(64 * NWV), smem, s, in, n, bank, P, center, (int)step, in_scale, A, B, tail_chunk, tail_rows,
                           W.k, W.gtab, M, W.zs, W.csum, W.cpeak, W.cross);
    if (mixed) {
        const int rows = c_b == tail_chunk ? tail_rows : P;
        auto kb = k_p3_boundary<TIn, TAcc, TTap>;
        hipLaunchKernelGGL(kb, dim3(1), dim3(64), sizeof(double) * (size_t)P, s, in, n, bank, P, center, (int)step, in_scale, c_b, c_b - flush / P, t1, rows, c_b,
                           W.k, W.gtab, W.zs, W.csum, W.cpeak, W.cross);
    }
}
void launch_p3_fused_s16(const int16_t *in, int64_t n, const float *bankf, const float *bankf_scaled, int P, int center, int64_t step, int64_t m_total,
                         int64_t flush, const KwSweep &W, const double *mpow_dev, hipStream_t s)
{
    launch_p3_fused_t<int16_t, float, float>(in, n, bankf_scaled ? bankf_scaled : bankf, P, center, step, m_total, flush, bankf_scaled ? 1.0 : 1.0 / 32768.0,
                                             W, mpow_dev, s);
}
void launch_p3_fused_f64(const double *in, int64_t n, const double *bank, int P, int center, int64_t step, int64_t m_total, int64_t flush, const KwSweep &W,
                         const double *mpow_dev, hipStream_t s)
{
    launch_p3_fused_t<double, double, double>(in, n, bank, P, center, step, m_total, flush, 1.0, W, mpow_dev, s);
}

// ------------------------------------------------------------------ true peak at fractional ratios (44.1 kHz -> 192 kHz) in the stream layout
// k_upsample32<.., MODE 0, QL = 1> gives a workgroup four waves and every wave whole windows: a tap row fetched into SGPRs feeds 64 x 32
// FMAs of ONE dependent chain per lane, and the kernel took twice the time of the 48 kHz instance (four chains per lane) for the same
// 22 GFMA.  This is k_p3_fused's layout without the K-weighting: lane = polyphase period, eight waves share the 64-period input tile
// (38 KB of f32: three workgroups per CU), a wave walks its eighth of the period's outputs in order with the next tap row in flight, and
// keeps the running maximum of |y| per lane for the 100 ms block its window ends in (ebur128's assignment: the block of the window's
// last input sample, as the kernel it replaces).  The same tap sums in the same order, the same block for every output: bit-identical
// block maxima.
template <typename TIn, typename TAcc, typename TTap, int NWV>
__global__ void __launch_bounds__(64 * NWV)
k_tp_stream(const TIn *__restrict__ in, int64_t n, const TTap *__restrict__ bank, int P, int center, int step, int64_t m_total, int blk,
            unsigned long long *__restrict__ block_tp, int64_t nblocks_alloc)
{
    constexpr int L = 32, NT = 64 * NWV;
    extern __shared__ unsigned char smem_pp[];
    __shared__ unsigned long long slots[8];
    TIn *xin = reinterpret_cast<TIn *>(smem_pp);
    const int T = 64 * step, nin = T + L;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int64_t s0 = (int64_t)blockIdx.x * T;
    const int64_t m_lo = (int64_t)blockIdx.x * 64 * P;
    if (m_lo >= m_total) return;
    if (s0 - center >= 0 && s0 - center + nin <= n) {
        const TIn *src = in + (s0 - center);
        for (int i = tid; i < nin; i += NT) xin[i] = src[i];
    } else {
        for (int i = tid; i < nin; i += NT) {
            int64_t g = s0 - center + i;
            TIn v = (TIn)0;
            if (g < 0) g = -g;                                       // invert_initial_buffer(): in[-j] = in[j]
            if (g < n) v = in[g];                                    // (the true-peak stream is never flushed: nothing behind the end)
            xin[i] = v;
        }
    }
    if (tid < 8) slots[tid] = 0ull;
    __syncthreads();
    const unsigned uP = (unsigned)P, ustep = (unsigned)step;
    const int JW = P / NWV, jw0 = wave * JW, jw1 = jw0 + JW;
    const int off_first = (int)(((unsigned)jw0 * ustep) / uP), off_last = (int)(((unsigned)(jw1 - 1) * ustep) / uP);
    const int64_t m_base = m_lo + (int64_t)P * lane;
    const int64_t left = m_total - m_base;
    const int jlim = left <= 0 ? 0 : (left >= (int64_t)P ? P : (int)left);   // outputs j < jlim of this lane's period exist
    const int64_t base0 = s0 - center + L - 1;                                // `last` input sample of the window at tile sample 0
    const int64_t b_first = (base0 < 0 ? 0 : base0) / blk;
    TAcc run_v = (TAcc)0; int run_sl = -1;
    auto flush_run = [&]() {
        if (run_sl >= 0 && run_v > (TAcc)0) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong((double)run_v);
            if (run_sl < 8) atomicMax(&slots[run_sl], bits); else atomicMax(&block_tp[b_first + run_sl], bits);
        }
        run_v = (TAcc)0;
    };
    for (int off0 = off_first; off0 <= off_last; ++off0) {
        const int off = __builtin_amdgcn_readfirstlane(off0);
        int j_lo = (int)(((unsigned)off * uP + ustep - 1u) / ustep), j_hi = (int)(((unsigned)(off + 1) * uP + ustep - 1u) / ustep);
        j_lo = j_lo > jw0 ? j_lo : jw0; j_hi = j_hi < jw1 ? j_hi : jw1;
        const int si = off + step * lane;
        TAcc xw[L];
        {
            const TIn *wp = xin + si;
#pragma unroll
            for (int i = 0; i < L; ++i) xw[i] = (TAcc)wp[i];
        }
        TAcc vmax = (TAcc)0;
        int ph = (int)(((unsigned)j_lo * ustep) % uP);
        auto row = [&](const TTap (&tp)[L], int j) {
            TAcc val = (TAcc)0;
#pragma unroll
            for (int i = 0; i < L; ++i) val = fma(xw[i], (TAcc)tp[i], val);
            vmax = j < jlim ? fmax(vmax, fabs(val)) : vmax;
        };
        auto fetch_row = [&](TTap (&tp)[L], int phase) {
            const TTap *f = bank + (size_t)(unsigned)phase * L;
#pragma unroll
            for (int i = 0; i < L; ++i) tp[i] = f[i];
        };
        TTap ta[L], tb[L];
        if (j_lo < j_hi) fetch_row(ta, ph);
        for (int j = j_lo; j < j_hi; j += 2) {
            int ph1 = ph + step; ph1 -= ph1 >= P ? P : 0;
            int ph2 = ph1 + step; ph2 -= ph2 >= P ? P : 0;
            fetch_row(tb, ph1);
            row(ta, j);
            if (j + 1 < j_hi) {
                fetch_row(ta, ph2);
                row(tb, j + 1);
            }
            ph = ph2;
        }
        // the 100 ms block of this window's last input sample (windows that end before the signal or behind it count for nothing)
        const int64_t last = base0 + si;
        if (last >= 0 && last <= n - 1 && vmax > (TAcc)0) {
            int64_t b = last / blk;
            if (b >= nblocks_alloc) b = nblocks_alloc - 1;
            const int sl = (int)(b - b_first);
            if (sl != run_sl) { flush_run(); run_sl = sl; }
            run_v = fmax(run_v, vmax);
        }
    }
    flush_run();
    __syncthreads();
    if (tid < 8 && slots[tid]) {
        int64_t b = b_first + tid;
        if (b >= nblocks_alloc) b = nblocks_alloc - 1;
        atomicMax(&block_tp[b], slots[tid]);
    }
}
constexpr int TP_NWV = 8;

// ---------------------------------------------------------------- true peak by branch and bound (round 4)
// ebur128's true peak reaches the host logic only as a RUNNING maximum (lavfi.r128.true_peak of frame k = the largest |y| of the
// 192 kHz stream up to frame k: analysis_finish folds the per-100 ms maxima with a prefix max).  An output is a 32-tap sum, so
// |y| <= A * max|x over its window| with A = the largest l1 norm of a tap row (times 1 + 1e-9: the 32 roundings of the FMA chain are
// below 4e-15 of that bound).  A UNIT (one polyphase period at fractional ratios, 256 consecutive windows at integer ratios) whose
// bound does not exceed an |y| that was evaluated EXACTLY in an earlier unit cannot move the running maximum at any frame and is
// not evaluated at all.  Two rounds: (1) k_tp_bounds sweeps the signal once (max |x| per unit) and names the loudest unit of every
// group of 64 as a seed, k_tp_list_* evaluates the seeds exactly (E[g], and a coarse maximum per 64 groups); (2) k_tp_select keeps
// the units whose bound exceeds the largest seed result of the groups BEFORE theirs, k_tp_list_* evaluates those.  Every evaluated
// output is the same tap sum in the same order attributed to the same 100 ms block as in k_upsample32 / k_tp_stream: the block
// maxima of evaluated units are bit-identical, the others stay zero, and the prefix maximum -- all anybody reads -- is identical.
// What is saved depends on the signal: speech keeps the units within 20 log10(A) ~ 6 dB of the loudest peak so far.
struct TpPruneDev {
    float *amax; int *seed, *list, *cnt; unsigned long long *E, *Ec;      // amax: the unit's BOUND (rounded up)
    int64_t n_units, n_groups; int G;       // periods per unit
    double A, T0, M1;                       // tap-row norms: max sum |t|, max |sum t|, max sum |t_i| |i - c| (times 1 + 1e-9)
};

__device__ inline float tp_load_abs(const float *__restrict__ in, int64_t n, int64_t g)
{
    if (g < 0) g = -g;                                               // invert_initial_buffer(): in[-j] = in[j]
    return g < n ? fabsf(in[g]) : 0.0f;                              // (the true-peak stream is never flushed)
}

// one workgroup (four waves) per group of 64 units; a wave reads a unit's span (S + 31 samples) with coalesced loads.
// Two bounds on every |y| of the unit, a = max |x|, d = max |x[i+1] - x[i]| over the span (reflection / padding included):
//   |y| = |sum t_i x_i| <= a sum |t_i|                                                      (any signal)
//   |y| = |x_c sum t_i + sum t_i (x_i - x_c)| <= a |sum t_i| + d sum |t_i| |i - c|          (smooth signals: speech peaks)
// the unit's bound is the smaller (the second is ~1.3 a on voiced speech where the first is 2 a: half as many units survive)
__global__ void __launch_bounds__(256)
k_tp_bounds(const float *__restrict__ in, int64_t n, int center, int S, TpPruneDev D)
{
    __shared__ float gmax[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t g = blockIdx.x;
    const int span = S + 31;
    const bool interior = (g * 64) * S - center >= 0 && (g * 64 + 64) * S - center + 31 <= n;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int ul = wave * 16 + k;
        const int64_t u = g * 64 + ul;
        float v = 0.0f, dv = 0.0f;
        if (u < D.n_units) {
            const int64_t base = u * S - center;
            if (interior) {
                for (int i = lane; i < span; i += 64) {
                    const float a = in[base + i], b = in[base + (i + 1 < span ? i + 1 : i)];
                    v = fmaxf(v, fabsf(a)); dv = fmaxf(dv, fabsf(b - a));
                }
            } else {
                auto at = [&](int64_t gi) -> float { if (gi < 0) gi = -gi; return gi < n ? in[gi] : 0.0f; };
                for (int i = lane; i < span; i += 64) {
                    const float a = at(base + i), b = at(base + (i + 1 < span ? i + 1 : i));
                    v = fmaxf(v, fabsf(a)); dv = fmaxf(dv, fabsf(b - a));
                }
            }
        }
#pragma unroll
        for (int mm = 1; mm < 64; mm <<= 1) { v = fmaxf(v, __shfl_xor(v, mm, 64)); dv = fmaxf(dv, __shfl_xor(dv, mm, 64)); }
        if (lane == 0) {
            gmax[ul] = v;
            if (u < D.n_units) {
                // (b - a in f32 is within half an ulp of the exact difference: 1 + 1e-6 on d covers it; NaN / Inf fall to the first bound or keep the unit)
                const double b1 = (double)v * D.A, b2 = (double)v * D.T0 + (double)dv * (1.0 + 1e-6) * D.M1 + (v > 0.0f ? 1e-30 : 0.0);   // (+ a difference flushed to zero)
                const double b = (b2 < b1) ? b2 : b1;
                D.amax[u] = __double2float_ru(b);
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
        // the loudest unit of the group (ties: the earliest)
        unsigned long long key = ((unsigned long long)__float_as_uint(gmax[lane]) << 32) | (unsigned)(63 - lane);
        if (g * 64 + lane >= D.n_units) key = 0ull;
#pragma unroll
        for (int mm = 1; mm < 64; mm <<= 1) {
            const unsigned long long o = ((unsigned long long)(unsigned)__shfl_xor((int)(key >> 32), mm, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)key, mm, 64);
            key = o > key ? o : key;
        }
        if (lane == 0) {
            D.seed[g] = (int)(g * 64 + (63 - (int)(key & 63u)));
            D.E[g] = 0ull;
            if ((g & 63) == 0) D.Ec[g >> 6] = 0ull;
            if (g == 0) *D.cnt = 0;
        }
    }
}

// round 2's list: a wave per group; lbp = the largest exact |y| of the seeds of all EARLIER groups
__global__ void __launch_bounds__(256)
k_tp_select(TpPruneDev D)
{
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= D.n_groups) return;
    const int64_t c = g >> 6;
    unsigned long long lb = 0ull;
    for (int64_t k = lane; k < c; k += 64) { const unsigned long long v = D.Ec[k]; lb = v > lb ? v : lb; }
    if (lane < (int)(g & 63)) { const unsigned long long v = D.E[c * 64 + lane]; lb = v > lb ? v : lb; }
    double lbp = __longlong_as_double((long long)lb);               // (non-negative doubles order like their bit patterns)
#pragma unroll
    for (int mm = 1; mm < 64; mm <<= 1) lbp = fmax(lbp, __shfl_xor(lbp, mm, 64));
    const int64_t u = g * 64 + lane;
    const bool keep = u < D.n_units && (int)u != D.seed[g] && (double)D.amax[u] > lbp;
    const unsigned long long m = __ballot(keep);
    const int total = __popcll(m);
    int base = 0;
    if (lane == 0 && total) base = atomicAdd(D.cnt, total);
    base = __shfl(base, 0, 64);
    if (keep) D.list[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)u;
}

// fractional ratios (k_tp_stream's layout over a GATHERED tile): lane = one listed period, eight waves share the 64 rows and walk an
// eighth of the period's outputs each; rows are step + 31 samples at an odd stride
template <int NWV>
__global__ void __launch_bounds__(64 * NWV)
k_tp_list_period(const float *__restrict__ in, int64_t n, const double *__restrict__ bank, int P, int center, int step, int64_t m_total, int blk,
                 unsigned long long *__restrict__ block_tp, int64_t nblocks_alloc, TpPruneDev D, const int *__restrict__ list, int count_host,
                 int seeds)
{
    constexpr int L = 32;
    extern __shared__ unsigned char smem_pp[];
    __shared__ unsigned long long lmax[64][2];
    __shared__ int qs[64];
    float *xin = reinterpret_cast<float *>(smem_pp);
    const int RW = step + L - 1, RS = RW | 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int count = seeds ? count_host : *D.cnt;
    const unsigned uP = (unsigned)P, ustep = (unsigned)step;
    const int JW = P / NWV, jw0 = wave * JW, jw1 = jw0 + JW;
    const int off_first = (int)(((unsigned)jw0 * ustep) / uP), off_last = (int)(((unsigned)(jw1 - 1) * ustep) / uP);
    for (int chunk = blockIdx.x; (int64_t)chunk * 64 < count; chunk += gridDim.x) {
        const int e = chunk * 64 + lane;
        const bool valid = e < count;
        if (wave == 0) { qs[lane] = list[valid ? e : count - 1]; lmax[lane][0] = 0ull; lmax[lane][1] = 0ull; }
        __syncthreads();
        for (int r = wave; r < 64; r += NWV) {
            const int64_t g0 = (int64_t)qs[r] * step - center;
            float *row = xin + (size_t)r * RS;
            if (g0 >= 0 && g0 + RW <= n) { for (int i = lane; i < RW; i += 64) row[i] = in[g0 + i]; }
            else {
                for (int i = lane; i < RW; i += 64) {
                    int64_t g = g0 + i;
                    if (g < 0) g = -g;
                    row[i] = g < n ? in[g] : 0.0f;
                }
            }
        }
        __syncthreads();
        const int64_t q = qs[lane];
        const int64_t left = m_total - q * P;
        const int jlim = left <= 0 ? 0 : (left >= (int64_t)P ? P : (int)left);
        const int64_t base0 = q * step - center + L - 1;                  // `last` input sample of the period's first window
        const int64_t b_first = (base0 < 0 ? 0 : base0) / blk;
        double run_v = 0.0; int run_sl = -1;
        auto flush_run = [&]() {
            if (run_sl >= 0 && run_v > 0.0 && valid) atomicMax(&lmax[lane][run_sl], (unsigned long long)__double_as_longlong(run_v));
            run_v = 0.0;
        };
        for (int off0 = off_first; off0 <= off_last; ++off0) {
            const int off = __builtin_amdgcn_readfirstlane(off0);
            int j_lo = (int)(((unsigned)off * uP + ustep - 1u) / ustep), j_hi = (int)(((unsigned)(off + 1) * uP + ustep - 1u) / ustep);
            j_lo = j_lo > jw0 ? j_lo : jw0; j_hi = j_hi < jw1 ? j_hi : jw1;
            double xw[L];
            {
                const float *wp = xin + off + RS * lane;
#pragma unroll
                for (int i = 0; i < L; ++i) xw[i] = (double)wp[i];
            }
            double vmax = 0.0;
            int ph = (int)(((unsigned)j_lo * ustep) % uP);
            auto row = [&](const double (&tp)[L], int j) {
                double val = 0.0;
#pragma unroll
                for (int i = 0; i < L; ++i) val = fma(xw[i], tp[i], val);
                vmax = j < jlim ? fmax(vmax, fabs(val)) : vmax;
            };
            auto fetch_row = [&](double (&tp)[L], int phase) {
                const double *f = bank + (size_t)(unsigned)phase * L;
#pragma unroll
                for (int i = 0; i < L; ++i) tp[i] = f[i];
            };
            double ta[L], tb[L];
            if (j_lo < j_hi) fetch_row(ta, ph);
            for (int j = j_lo; j < j_hi; j += 2) {
                int ph1 = ph + step; ph1 -= ph1 >= P ? P : 0;
                int ph2 = ph1 + step; ph2 -= ph2 >= P ? P : 0;
                fetch_row(tb, ph1);
                row(ta, j);
                if (j + 1 < j_hi) {
                    fetch_row(ta, ph2);
                    row(tb, j + 1);
                }
                ph = ph2;
            }
            const int64_t last = base0 + off;
            if (last >= 0 && last <= n - 1 && vmax > 0.0) {
                const int sl = (int)(last / blk - b_first);               // 0 or 1: a period is shorter than a block
                if (sl != run_sl) { flush_run(); run_sl = sl; }
                run_v = fmax(run_v, vmax);
            }
        }
        flush_run();
        __syncthreads();
        if (tid < 128) {
            const int l = tid >> 1, sl = tid & 1;
            const unsigned long long v = lmax[l][sl];
            const int64_t ql = qs[l];
            const int64_t bl0 = ql * step - center + L - 1;
            int64_t b = (bl0 < 0 ? 0 : bl0) / blk + sl;
            if (b >= nblocks_alloc) b = nblocks_alloc - 1;
            if (v && chunk * 64 + l < count) atomicMax(&block_tp[b], v);
        }
        if (seeds && tid < 64 && valid) {
            const unsigned long long a = lmax[lane][0], b = lmax[lane][1], v = a > b ? a : b;
            D.E[e] = v;
            if (v) atomicMax(&D.Ec[e >> 6], v);
        }
        __syncthreads();
    }
}

// integer ratios (step == 1; k_upsample32<.., 0, 4>'s item): a wave per listed unit of 256 consecutive windows, four windows per lane
__global__ void __launch_bounds__(256)
k_tp_list_q4(const float *__restrict__ in, int64_t n, const double *__restrict__ bank, int P, int center, int64_t m_total, int blk,
             unsigned long long *__restrict__ block_tp, int64_t nblocks_alloc, TpPruneDev D, const int *__restrict__ list, int count_host, int seeds)
{
    constexpr int L = 32, QL = 4, WL = L + QL - 1, TW = 256 + 36;
    __shared__ __attribute__((aligned(16))) float tiles[4][TW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int count = seeds ? count_host : *D.cnt;
    float *xin = tiles[wave];
    for (int e0 = blockIdx.x * 4; e0 < count; e0 += gridDim.x * 4) {
        const int e = e0 + wave;                                       // (wave-uniform)
        if (e < count) {
            const int64_t w0 = (int64_t)list[e] * 256;                 // first window of the unit (= its period index: step == 1)
            const int64_t g0 = w0 - center;
            if (g0 >= 0 && g0 + TW <= n) { for (int i = lane; i < TW; i += 64) xin[i] = in[g0 + i]; }
            else {
                for (int i = lane; i < TW; i += 64) {
                    int64_t g = g0 + i;
                    if (g < 0) g = -g;
                    xin[i] = g < n ? in[g] : 0.0f;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                               // (a wave's LDS operations complete in order)
        if (e < count) {
            const int64_t w0 = (int64_t)list[e] * 256;
            double xw[WL];
            {
                typedef float vecT __attribute__((ext_vector_type(4)));
                const vecT *vp = reinterpret_cast<const vecT *>(xin + lane * QL);
#pragma unroll
                for (int k = 0; k < (WL + 3) / 4; ++k) {
                    const vecT t = vp[k];
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (k * 4 + c < WL) xw[k * 4 + c] = (double)t[c];
                }
            }
            double vmax[QL]; int jlim[QL];
#pragma unroll
            for (int u = 0; u < QL; ++u) {
                vmax[u] = 0.0;
                const int64_t left = m_total - (w0 + lane * QL + u) * P;
                jlim[u] = left <= 0 ? 0 : (left >= (int64_t)P ? P : (int)left);
            }
            for (int j = 0; j < P; ++j) {
                double tp[L];
                const double *f = bank + (size_t)j * L;
#pragma unroll
                for (int i = 0; i < L; ++i) tp[i] = f[i];
                double val[QL];
#pragma unroll
                for (int u = 0; u < QL; ++u) val[u] = 0.0;
#pragma unroll
                for (int i = 0; i < L; ++i) {
#pragma unroll
                    for (int u = 0; u < QL; ++u) val[u] = fma(xw[u + i], tp[i], val[u]);
                }
#pragma unroll
                for (int u = 0; u < QL; ++u) vmax[u] = j < jlim[u] ? fmax(vmax[u], fabs(val[u])) : vmax[u];
            }
            // 100 ms block of each window's last input sample: a unit touches two blocks at most
            const int64_t base0 = w0 - center + L - 1;
            const int64_t b_first = (base0 < 0 ? 0 : base0) / blk;
            double v0 = 0.0, v1 = 0.0;
#pragma unroll
            for (int u = 0; u < QL; ++u) {
                const int64_t last = base0 + lane * QL + u;
                if (last >= 0 && last <= n - 1) {
                    if (last / blk == b_first) v0 = fmax(v0, vmax[u]); else v1 = fmax(v1, vmax[u]);
                }
            }
#pragma unroll
            for (int mm = 1; mm < 64; mm <<= 1) { v0 = fmax(v0, __shfl_xor(v0, mm, 64)); v1 = fmax(v1, __shfl_xor(v1, mm, 64)); }
            if (lane == 0) {
                int64_t b0 = b_first, b1 = b_first + 1;
                if (b0 >= nblocks_alloc) b0 = nblocks_alloc - 1;
                if (b1 >= nblocks_alloc) b1 = nblocks_alloc - 1;
                if (v0 > 0.0) atomicMax(&block_tp[b0], (unsigned long long)__double_as_longlong(v0));
                if (v1 > 0.0) atomicMax(&block_tp[b1], (unsigned long long)__double_as_longlong(v1));
                if (seeds) {
                    const unsigned long long v = (unsigned long long)__double_as_longlong(fmax(v0, v1));
                    D.E[e] = v;
                    if (v) atomicMax(&D.Ec[e >> 6], v);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

static inline size_t tp_al(size_t b) { return (b + 255) & ~(size_t)255; }
// periods per unit for a plan the pruned path serves, 0 otherwise
static int tp_prune_unit(int P, int L, int64_t step, int blk)
{
    if (L != 32 || step >= P) return 0;
    if (step == 1) return 256 + 36 < blk ? 256 : 0;
    if ((step & 1) && P % (2 * TP_NWV) == 0 && step + 32 < blk && ((size_t)64 * ((step + 31) | 1) * 4) <= 96 * 1024) return 1;
    return 0;
}
size_t jt_tp_prune_scratch_bytes(int64_t n, int P, int L, int64_t step, int blk)
{
    const int G = tp_prune_unit(P, L, step, blk);
    if (!G) return 0;
    const int64_t m_total = (int64_t)(((__int128)n * P + step - 1) / step);
    const int64_t n_units = ((m_total + P - 1) / P + G - 1) / G, n_groups = (n_units + 63) / 64;
    return tp_al(4 * (size_t)n_units) * 2 + tp_al(4 * (size_t)n_groups) + tp_al(8 * (size_t)n_groups) + tp_al(8 * (size_t)(n_groups / 64 + 1)) + 256;
}
// false: this plan / length is not served (the caller runs the exhaustive kernels)
bool launch_true_peak_f32_pruned(const float *in, int64_t n, const double *bank, int P, int L, int center, int64_t step, int blk, double *block_tp,
                                 int64_t nblocks_alloc, int64_t m_total, const double norms[3], void *scratch, size_t scratch_bytes, hipStream_t s,
                                 const int **kept_dev, int64_t *units, int64_t *seeds)
{
    const int G = tp_prune_unit(P, L, step, blk);
    if (!G || m_total <= 0 || !scratch || !(norms[0] > 0)) return false;
    TpPruneDev D;
    D.G = G;
    D.n_units = ((m_total + P - 1) / P + G - 1) / G; D.n_groups = (D.n_units + 63) / 64;
    if (D.n_groups < 2 || D.n_units >= (int64_t)1 << 31) return false;
    unsigned char *p = (unsigned char *)scratch; size_t o = 0;
    D.amax = (float *)(p + o); o += tp_al(4 * (size_t)D.n_units);
    D.list = (int *)(p + o); o += tp_al(4 * (size_t)D.n_units);
    D.seed = (int *)(p + o); o += tp_al(4 * (size_t)D.n_groups);
    D.E = (unsigned long long *)(p + o); o += tp_al(8 * (size_t)D.n_groups);
    D.Ec = (unsigned long long *)(p + o); o += tp_al(8 * (size_t)(D.n_groups / 64 + 1));
    D.cnt = (int *)(p + o); o += 256;
    if (o > scratch_bytes) return false;
    if (kept_dev) *kept_dev = D.cnt;
    if (units) *units = D.n_units;
    if (seeds) *seeds = D.n_groups;
    D.A = norms[0] * (1.0 + 1e-9); D.T0 = norms[1] * (1.0 + 1e-9); D.M1 = norms[2] * (1.0 + 1e-9);
    const int S = G * (int)step;
    hipLaunchKernelGGL(k_tp_bounds, dim3((unsigned)D.n_groups), dim3(256), 0, s, in, n, center, S, D);
    unsigned long long *btp = (unsigned long long *)block_tp;
    if (G == 1) {
        auto k = k_tp_list_period<TP_NWV>;
        const size_t tile = (size_t)64 * ((step + 31) | 1) * 4;
        JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile));
        const unsigned g1 = (unsigned)((D.n_groups + 63) / 64), g2 = (unsigned)std::min<int64_t>((D.n_units + 63) / 64, 4096);
        hipLaunchKernelGGL(k, dim3(g1), dim3(64 * TP_NWV), tile, s, in, n, bank, P, center, (int)step, m_total, blk, btp, nblocks_alloc, D,
                           (const int *)D.seed, (int)D.n_groups, 1);
        hipLaunchKernelGGL(k_tp_select, dim3((unsigned)((D.n_groups + 3) / 4)), dim3(256), 0, s, D);
        hipLaunchKernelGGL(k, dim3(g2), dim3(64 * TP_NWV), tile, s, in, n, bank, P, center, (int)step, m_total, blk, btp, nblocks_alloc, D,
                           (const int *)D.list, 0, 0);
    } else {
        const unsigned g1 = (unsigned)((D.n_groups + 3) / 4), g2 = (unsigned)std::min<int64_t>((D.n_units + 3) / 4, 16384);
        hipLaunchKernelGGL(k_tp_list_q4, dim3(g1), dim3(256), 0, s, in, n, bank, P, center, m_total, blk, btp, nblocks_alloc, D, (const int *)D.seed,
                           (int)D.n_groups, 1);
        hipLaunchKernelGGL(k_tp_select, dim3((unsigned)((D.n_groups + 3) / 4)), dim3(256), 0, s, D);
        hipLaunchKernelGGL(k_tp_list_q4, dim3(g2), dim3(256), 0, s, in, n, bank, P, center, m_total, blk, btp, nblocks_alloc, D, (const int *)D.list, 0, 0);
    }
    return true;
}

template <typename TIn, typename TAcc, typename TTap, int MODE>
static bool launch_upsample32(const TIn *in, int64_t n, const TTap *bank, int P, int L, int center, int64_t step, int64_t m_total,
                              double in_scale, int blk, double *block_tp, int64_t nblocks_alloc, TAcc *out, hipStream_t s, const JtOpts *o = nullptr)
{
    (void)o;
    if (L != 32 || step >= P || step > 512) return false;
    const int ql = step == 1 ? 4 : 1;
    if constexpr (MODE == 0) {
        const size_t tile = (sizeof(TIn) * (size_t)(64 * step + 32 + 4) + 15) & ~(size_t)15;
        if (ql == 1 && (step & 1) && P % (2 * TP_NWV) == 0 && tile <= 96 * 1024 && !JT_AB_ON(o && o->tp_old)) {
            auto kt = k_tp_stream<TIn, TAcc, TTap, TP_NWV>;
            JT_HIP(hipFuncSetAttribute((const void *)kt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile));
            const int64_t pb = (int64_t)64 * P;
            hipLaunchKernelGGL(kt, dim3((unsigned)((m_total + pb - 1) / pb)), dim3(64 * TP_NWV), tile, s, in, n, bank, P, center, (int)step, m_total, blk,
                               (unsigned long long *)block_tp, nblocks_alloc);
            return true;
        }
    }
    int R = (int)std::max<int64_t>(1, 4096 / (64 * ql * step));
    // stream output with per-lane window groups (upsample32_stream_quads): as many windows per lane as a <= 80 KB tile allows
    if (MODE == 2 && ql == 1 && (P & 15) == 0 && R == 1)
        for (int r = 2; r > 1; r >>= 1) if (sizeof(TIn) * (size_t)(64 * r * step + 64) <= 80 * 1024) { R = r; break; }
    if constexpr (MODE == 2) {
        if (ql == 1 && R == 1 && (P & 31) == 0 && (step & 1) && !JT_AB_ON(o && o->ups_no_stream8)) {
            const size_t tile8 = (sizeof(TIn) * (size_t)(64 * step + 32 + 4) + 15) & ~(size_t)15;
            const size_t sm16 = tile8 + sizeof(TAcc) * (size_t)16 * 64 * 9;
            if ((P % 64) == 0 && sm16 <= 150 * 1024 && !JT_AB_ON(o && o->ups_no_stream16)) {
                auto k16 = k_upsample32_stream8<TIn, TAcc, TTap, 16, 8>;
                JT_HIP(hipFuncSetAttribute((const void *)k16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm16));
                const int64_t pb = (int64_t)64 * P;
                hipLaunchKernelGGL(k16, dim3((unsigned)((m_total + pb - 1) / pb)), dim3(1024), sm16, s, in, n, bank, P, center, (int)step, m_total, in_scale, out);
                return true;
            }
            const size_t sm8 = tile8 + sizeof(TAcc) * (size_t)8 * 64 * 17;
            if (sm8 <= 150 * 1024) {
                auto k8 = k_upsample32_stream8<TIn, TAcc, TTap>;
                JT_HIP(hipFuncSetAttribute((const void *)k8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm8));
                const int64_t pb = (int64_t)64 * P;
                hipLaunchKernelGGL(k8, dim3((unsigned)((m_total + pb - 1) / pb)), dim3(512), sm8, s, in, n, bank, P, center, (int)step, m_total, in_scale, out);
                return true;
            }
        }
    }
    const int T = 64 * R * ql * (int)step, nin = T + 32;
    size_t smem = sizeof(TIn) * (size_t)(nin + (nin >> 5) + 4);
    if (MODE == 2 && ql == 1 && (P & 15) == 0 && (R == 1 || R == 2))          // upsample32_stream_quads: + [4 waves][R][64][17] output tiles
        smem = ((smem + 15) & ~(size_t)15) + sizeof(TAcc) * (size_t)(PP_THREADS / 64) * R * 64 * 17;
    if (smem > 150 * 1024) return false;
    const int64_t per_block = (int64_t)64 * R * ql * P;
    const unsigned grid = (unsigned)((m_total + per_block - 1) / per_block);
    if (ql == 4) {
        auto k = k_upsample32<TIn, TAcc, TTap, MODE, 4>;
        JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(k, dim3(grid), dim3(PP_THREADS), smem, s, in, n, bank, P, center, (int)step, m_total, R, in_scale, blk,
                           (unsigned long long *)block_tp, nblocks_alloc, out, 0);
    } else {
        auto k = k_upsample32<TIn, TAcc, TTap, MODE, 1>;
        JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL(k, dim3(grid), dim3(PP_THREADS), smem, s, in, n, bank, P, center, (int)step, m_total, R, in_scale, blk,
                           (unsigned long long *)block_tp, nblocks_alloc, out, (step & 1) ? 0 : 1);
    }
    return true;
}

// ---- 48 kHz -> 44.1 kHz (+ s16): the default output stage (filters.go:706-710), P = 147 phases, 160 input samples per
// 147 outputs, 36 taps.  "Period-per-lane": lane l of a wave produces the 147 consecutive outputs of one polyphase period
// (lanes = 64 consecutive periods), so at every step all lanes are at the SAME phase (taps = wave-uniform scalar loads) and
// a lane's 36-sample window slides by 1 or 2 inputs per output.  The whole period is unrolled at compile time: the window
// is a 40-slot register ring with static indices (160 % 40 == 0, so the ring realigns every period), each input sample is
// read from LDS exactly once, and the inner work is 36 register-operand f64 FMAs per output -- the FMA pipe is the only
// limiter.  Inputs are staged 32 samples x 64 lanes at a time ([64][33] f32, row-coalesced loads), outputs leave through a
// [64][34] s16 tile.  Tap order per output is unchanged (ascending), so results are bit-identical to k_polyphase<.,.,.,1>.
// Interior blocks only; blocks touching the stream edges (reflection / flush) are left to k_polyphase.
namespace d147 {
__host__ __device__ constexpr int win_start(int r) { return (r * STEP) / P; }      // relative input index of the first tap of output r
__host__ __device__ constexpr int phase(int r) { return (r * STEP) % P; }
}

__global__ void __launch_bounds__(64)
k_down147(const float *__restrict__ in, int64_t n, const double *__restrict__ bank, int center, int64_t m_total,
          int16_t *__restrict__ out)
{
    using namespace d147;
    __shared__ float stage[64][SW + 1];
    __shared__ int16_t ostage[64][34];
    const int lane = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * NIN;           // first input sample of the block's first period
    const int64_t m_lo = (int64_t)blockIdx.x * NOUT;
    // interior test (must match k_polyphase's complementary early-out): every staged read is inside [0, n) and every output exists
    const bool interior = (s0 - center >= 0) && (s0 - center + 63 * STEP + REACH <= n) && (m_lo + NOUT <= m_total);
    if (!interior) return;
    // row r, relative index x -> src[r*STEP + x].  Block-uniform bases with 32-bit lane offsets: the 32 staging loads (and stores)
    // then share one scalar base instead of pinning 29 address pairs in VGPRs for the whole kernel
    const float *src = in + (s0 - center);
    int16_t *dst = out + m_lo;
    const unsigned ioff = (unsigned)(lane >> 5) * STEP + (unsigned)(lane & 31);
    const unsigned ooff = (unsigned)(lane >> 5) * P + (unsigned)(lane & 31);
    double ring[RING];
    int staged = -1;                                        // stage number currently in LDS (compile-time after unrolling)
    int have = 0;                                           // inputs [0, have) are in the ring (or were)
#pragma clang loop unroll(full)
    for (int r = 0; r < P; ++r) {
        const int a = win_start(r);
        // bring the window [a, a+L) into the ring
#pragma clang loop unroll(full)
        for (int x = have; x < a + L; ++x) {
            if (x / SW != staged) {
                staged = x / SW;
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 32; ++it) {
                    const int row = it * 2 + (lane >> 5), col = lane & 31;
                    stage[row][col] = src[ioff + (unsigned)(it * 2 * STEP + staged * SW)];
                    if (it % 16 == 15) __asm__ volatile("" ::: "memory");     // 16 loads in flight
                }
                __syncthreads();
            }
            ring[x % RING] = (double)stage[lane][x % SW];
        }
        have = a + L > have ? a + L : have;
        __asm__ volatile("" ::: "memory");          // keep each output's 36 scalar tap loads next to their FMAs (SGPR budget)
        const double *f = bank + phase(r) * L;
        double val = 0.0;
#pragma clang loop unroll(full)
        for (int t = 0; t < L; ++t) val = fma(ring[(a + t) % RING], f[t], val);
        double q = rint(val * 32768.0);
        q = q < -32768.0 ? -32768.0 : (q > 32767.0 ? 32767.0 : q);
        ostage[lane][r % 32] = (int16_t)q;
        if (r % 32 == 31 || r == P - 1) {
            const int g0 = (r / 32) * 32, cnt = r - g0 + 1;
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 32; ++it) {
                const int row = it * 2 + (lane >> 5), col = lane & 31;
                if (col < cnt) dst[ooff + (unsigned)(it * 2 * P + g0)] = ostage[row][col];
                if (it % 8 == 7) __asm__ volatile("" ::: "memory");
            }
            __syncthreads();
        }
    }
}

struct PPGeom { int R; int T; size_t smem; unsigned grid; };
// a workgroup whose tile admits one workgroup per CU runs sixteen waves: the P phases are walked by the waves one after another, each
// waiting for its scalar tap loads, and nothing else on the CU covers that (96 k -> 44.1 k: an 82 KB tile, 147 phases of 72 taps)
static inline unsigned pp_threads(size_t smem) { return smem > 40 * 1024 ? 1024u : (unsigned)PP_THREADS; }
template <typename TIn>
static PPGeom pp_geometry(int64_t n, int P, int L, int64_t step, int64_t m_total, bool otile)
{
    PPGeom g;
    int R = (int)std::max<int64_t>(1, 4096 / (64 * step));
    if (otile) while (R > 1 && (size_t)64 * R * P * 2 > 48 * 1024) R >>= 1;
    g.R = R; g.T = 64 * R * (int)step;
    int nin = g.T + L;
    g.smem = ((sizeof(TIn) * (size_t)(nin + (nin >> 5) + 4) + 7) & ~(size_t)7) + (otile ? (size_t)64 * R * P * 2 + 16 : 0);
    int64_t per_block = (int64_t)64 * R * P;
    g.grid = (unsigned)((m_total + per_block - 1) / per_block);
    (void)n;
    return g;
}

void launch_true_peak_f32(const float *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                          int64_t step, int blk, double *block_tp, int64_t nblocks_alloc, int64_t m_total, hipStream_t s, const JtOpts *o)
{
    if (m_total <= 0) return;
    if (launch_upsample32<float, double, double, 0>(in, n, bank, phase_count, filter_length, center, step, m_total, 1.0, blk, block_tp,
                                                   nblocks_alloc, (double *)nullptr, s, o)) return;
    PPGeom g = pp_geometry<float>(n, phase_count, filter_length, step, m_total, false);
    JT_REQUIRE(g.smem <= 150 * 1024, JT_E_UNSUPPORTED, "true peak: rate ratio needs too large an LDS tile");
    auto k = k_polyphase<float, double, double, 0>;
    JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.smem));
    hipLaunchKernelGGL(k, dim3(g.grid), dim3(pp_threads(g.smem)), g.smem, s, in, n, bank, phase_count, filter_length, center, step, m_total,
                       g.R, 1.0, blk, (unsigned long long *)block_tp, nblocks_alloc, (int16_t *)nullptr, (double *)nullptr, 0, PPRemap{0, 0, 0});
}

void launch_true_peak_f64(const double *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                          int64_t step, int blk, double *block_tp, int64_t nblocks_alloc, int64_t m_total, hipStream_t s, const JtOpts *o)
{
    if (m_total <= 0) return;
    if (launch_upsample32<double, double, double, 0>(in, n, bank, phase_count, filter_length, center, step, m_total, 1.0, blk, block_tp,
                                                    nblocks_alloc, (double *)nullptr, s, o)) return;
    PPGeom g = pp_geometry<double>(n, phase_count, filter_length, step, m_total, false);
    JT_REQUIRE(g.smem <= 150 * 1024, JT_E_UNSUPPORTED, "true peak: rate ratio needs too large an LDS tile");
    auto k = k_polyphase<double, double, double, 0>;
    JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.smem));
    hipLaunchKernelGGL(k, dim3(g.grid), dim3(pp_threads(g.smem)), g.smem, s, in, n, bank, phase_count, filter_length, center, step, m_total,
                       g.R, 1.0, blk, (unsigned long long *)block_tp, nblocks_alloc, (int16_t *)nullptr, (double *)nullptr, 0, PPRemap{0, 0, 0});
}

// f32 in (the dbl->flt->dbl rounded signal) -> DBLP resample -> s16
// The stream-edge blocks of the 48 k -> 44.1 k output stage (the first block, whose taps reach before sample 0, and the last one or two,
// which reach past the end), one wave per (block, output phase): in k_polyphase a block is one workgroup whose four waves walk the 147
// phases one after another, each waiting for its 36 scalar tap loads -- 0.5 ms for two blocks, on the critical path of Pass 2.
// Same sums as k_polyphase<float, double, double, 1> (ascending taps, one FMA per tap, rint and clip to s16).
__global__ void __launch_bounds__(64)
k_polyphase_edge_s16(const float *__restrict__ in, int64_t n, const double *__restrict__ bank, int P, int L, int center, int64_t step,
                     int64_t m_total, int64_t skip_lo, int64_t skip_n, int16_t *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x / P; const int j = (int)((int64_t)blockIdx.x % P);
    const int64_t bid = e < skip_lo ? e : e + skip_n;
    const int64_t m = bid * 64 * P + j + (int64_t)P * threadIdx.x;
    if (m >= m_total) return;
    const int64_t idx = m * step;
    const int ph = (int)(idx % P);
    const int64_t g0 = idx / P - center;
    const double *f = bank + (size_t)ph * L;
    double val = 0.0;
    for (int i = 0; i < L; ++i) {
        int64_t g = g0 + i;
        float v = 0.f;
        if (g < 0) g = -g;                                           // invert_initial_buffer(): in[-j] = in[j]
        if (g < n) v = in[g];
        else { const int64_t r = 2 * n - 1 - g; if (r >= 0 && r < n) v = in[r]; }   // resample_flush()
        val = fma((double)((double)v * 1.0), f[i], val);
    }
    double r = rint(val * 32768.0);
    r = r < -32768.0 ? -32768.0 : (r > 32767.0 ? 32767.0 : r);
    out[m] = (int16_t)r;
}

void launch_resample_to_s16(const float *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                            int64_t step, int16_t *out, int64_t m, hipStream_t s, const JtOpts &o)
{
    if (m <= 0) return;
    PPGeom g = pp_geometry<float>(n, phase_count, filter_length, step, m, true);
    JT_REQUIRE(g.smem <= 150 * 1024, JT_E_UNSUPPORTED, "resample: rate ratio needs too large an LDS tile");
    const bool fast = phase_count == d147::P && step == d147::STEP && filter_length == d147::L && g.R == 1;
    // interior blocks [b_lo, b_hi) (the kernels' own test, solved for the block index) go to k_down147; k_polyphase is launched
    // for the stream-edge blocks only (a full-grid launch of early-outs costs ~1 ms of an hour-long file's critical path)
    int64_t b_lo = 0, b_hi = 0;
    if (fast) {
        b_lo = center > 0 ? (center + d147::NIN - 1) / d147::NIN : 0;
        const int64_t lim_in = n + center - 63 * d147::STEP - d147::REACH;
        b_hi = std::min<int64_t>(lim_in >= 0 ? lim_in / d147::NIN + 1 : 0, m / d147::NOUT);
        b_hi = std::min<int64_t>(b_hi, g.grid);
        if (b_hi <= b_lo) b_lo = b_hi = 0;
    }
    // the few edge blocks first (the two kernels write disjoint outputs): one wave per block and phase beside k_down147's interior;
    // other geometries keep k_polyphase for every block
    if (fast && b_hi > b_lo && g.R == 1 && !JT_AB_ON(o.edge_polyphase)) {
        const int64_t nedge = g.grid - (b_hi - b_lo);
        if (nedge > 0) hipLaunchKernelGGL(k_polyphase_edge_s16, dim3((unsigned)(nedge * phase_count)), dim3(64), 0, s, in, n, bank, phase_count,
                                          filter_length, center, step, m, b_lo, b_hi - b_lo, out);
    } else {
        auto k = k_polyphase<float, double, double, 1>;
        JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.smem));
        hipLaunchKernelGGL(k, dim3((unsigned)(g.grid - (b_hi - b_lo))), dim3(pp_threads(g.smem)), g.smem, s, in, n, bank, phase_count, filter_length,
                           center, step, m, g.R, 1.0, 1, (unsigned long long *)nullptr, (int64_t)0, out, (double *)nullptr, fast ? 1 : 0,
                           PPRemap{b_lo, b_hi - b_lo, 0});
    }
    if (b_hi > b_lo) hipLaunchKernelGGL(k_down147, dim3((unsigned)b_hi), dim3(64), 0, s, in, n, bank, center, m, out);
}

// The same resample restricted to the outputs [m_first, m_first + m_count) of the m-sample result (the announced output regions
// of Pass 2): the k_polyphase blocks that cover the range write dst[0 ..), and the range starts at dst[return value].  Every
// output is the same tap sum whichever kernel or block computes it, so these samples equal the full resample's bit for bit.
int64_t launch_resample_range_to_s16(const float *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                                     int64_t step, int64_t m, int64_t m_first, int64_t m_count, int16_t *dst, int64_t dst_cap, hipStream_t s)
{
    PPGeom g = pp_geometry<float>(n, phase_count, filter_length, step, m, true);
    JT_REQUIRE(g.smem <= 150 * 1024, JT_E_UNSUPPORTED, "resample: rate ratio needs too large an LDS tile");
    const int64_t per_block = (int64_t)64 * g.R * phase_count;
    const int64_t b_a = m_first / per_block, b_b = (m_first + m_count - 1) / per_block;
    JT_REQUIRE(m_count > 0 && m_first >= 0 && m_first + m_count <= m && (b_b - b_a + 1) * per_block <= dst_cap, JT_E_INVAL, "resample range: bad range");
    auto k = k_polyphase<float, double, double, 1>;
    JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.smem));
    hipLaunchKernelGGL(k, dim3((unsigned)(b_b - b_a + 1)), dim3(pp_threads(g.smem)), g.smem, s, in, n, bank, phase_count, filter_length,
                       center, step, m, g.R, 1.0, 1, (unsigned long long *)nullptr, (int64_t)0, dst, (double *)nullptr, 0,
                       PPRemap{0, b_a, b_a * per_block});
    return m_first - b_a * per_block;
}
// destination samples launch_resample_range_to_s16 needs for a range of m_count outputs
int64_t jt_resample_range_cap(int64_t n, int phase_count, int filter_length, int64_t step, int64_t m, int64_t m_count)
{
    PPGeom g = pp_geometry<float>(n, phase_count, filter_length, step, m, true);
    const int64_t per_block = (int64_t)64 * g.R * phase_count;
    return (m_count / per_block + 2) * per_block;
}

// Pass-3 streams at 192 kHz.  FLT variant (s16 in, no limiter prefix): swr int_sample_fmt FLTP = float taps, float
// accumulation.  DBL variant (after the alimiter prefix): DBLP.
void launch_resample_stream_s16_f32(const int16_t *in, int64_t n, const float *bankf, const float *bankf_scaled, int phase_count,
                                    int filter_length, int center, int64_t step, int64_t m_total, float *out, hipStream_t s)
{
    if (m_total <= 0) return;
    // bankf_scaled (taps * 2^-15, null when that is not exact): the window then needs no multiply after the int -> float conversion
    if (launch_upsample32<int16_t, float, float, 2>(in, n, bankf_scaled ? bankf_scaled : bankf, phase_count, filter_length, center, step,
                                                   m_total, bankf_scaled ? 1.0 : 1.0 / 32768.0, 1, (double *)nullptr, (int64_t)0, out, s)) return;
    PPGeom g = pp_geometry<int16_t>(n, phase_count, filter_length, step, m_total, false);
    auto k = k_polyphase<int16_t, float, float, 2>;
    JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.smem));
    hipLaunchKernelGGL(k, dim3(g.grid), dim3(pp_threads(g.smem)), g.smem, s, in, n, bankf, phase_count, filter_length, center, step, m_total,
                       g.R, 1.0 / 32768.0, 1, (unsigned long long *)nullptr, (int64_t)0, (int16_t *)nullptr, out, 0, PPRemap{0, 0, 0});
}
void launch_resample_stream_f64(const double *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                                int64_t step, int64_t m_total, double *out, hipStream_t s, const JtOpts &o)
{
    if (m_total <= 0) return;
    if (launch_upsample32<double, double, double, 2>(in, n, bank, phase_count, filter_length, center, step, m_total, 1.0, 1,
                                                    (double *)nullptr, (int64_t)0, out, s, &o)) return;
    PPGeom g = pp_geometry<double>(n, phase_count, filter_length, step, m_total, false);
    auto k = k_polyphase<double, double, double, 2>;
    JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.smem));
    hipLaunchKernelGGL(k, dim3(g.grid), dim3(pp_threads(g.smem)), g.smem, s, in, n, bank, phase_count, filter_length, center, step, m_total,
                       g.R, 1.0, 1, (unsigned long long *)nullptr, (int64_t)0, (int16_t *)nullptr, out, 0, PPRemap{0, 0, 0});
}
