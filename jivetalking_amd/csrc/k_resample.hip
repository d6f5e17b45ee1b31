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

template <typename TIn, typename TAcc, typename TTap, int MODE>
static bool launch_upsample32(const TIn *in, int64_t n, const TTap *bank, int P, int L, int center, int64_t step, int64_t m_total,
                              double in_scale, int blk, double *block_tp, int64_t nblocks_alloc, TAcc *out, hipStream_t s, const JtOpts *o = nullptr)
{
    (void)o;
    if (L != 32 || step >= P || step > 512) return false;
    const int ql = step == 1 ? 4 : 1;
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
                          int64_t step, int blk, double *block_tp, int64_t nblocks_alloc, int64_t m_total, hipStream_t s)
{
    if (m_total <= 0) return;
    if (launch_upsample32<float, double, double, 0>(in, n, bank, phase_count, filter_length, center, step, m_total, 1.0, blk, block_tp,
                                                   nblocks_alloc, (double *)nullptr, s)) return;
    PPGeom g = pp_geometry<float>(n, phase_count, filter_length, step, m_total, false);
    JT_REQUIRE(g.smem <= 150 * 1024, JT_E_UNSUPPORTED, "true peak: rate ratio needs too large an LDS tile");
    auto k = k_polyphase<float, double, double, 0>;
    JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.smem));
    hipLaunchKernelGGL(k, dim3(g.grid), dim3(pp_threads(g.smem)), g.smem, s, in, n, bank, phase_count, filter_length, center, step, m_total,
                       g.R, 1.0, blk, (unsigned long long *)block_tp, nblocks_alloc, (int16_t *)nullptr, (double *)nullptr, 0, PPRemap{0, 0, 0});
}

void launch_true_peak_f64(const double *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                          int64_t step, int blk, double *block_tp, int64_t nblocks_alloc, int64_t m_total, hipStream_t s)
{
    if (m_total <= 0) return;
    if (launch_upsample32<double, double, double, 0>(in, n, bank, phase_count, filter_length, center, step, m_total, 1.0, blk, block_tp,
                                                    nblocks_alloc, (double *)nullptr, s)) return;
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
