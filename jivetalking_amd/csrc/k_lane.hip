// k_lane.hip — "lane-serial" kernels for gfx950: recurrent per-sample filters (IIR biquads, K-weighting,
// envelope followers, de-esser, look-ahead limiter, running statistics) mapped onto the GPU by
// splitting the time axis into chunks, one chunk per lane, each lane running the recurrence
// sequentially over its chunk (plus a warm-up halo where the recurrence forgets its state).
//
// Memory pattern: a wave owns 64 consecutive chunks.  Each step it stages a [64 lanes][TW samples]
// tile through LDS: 64 coalesced row loads (row r = 64 consecutive samples of lane r's chunk,
// 256 B for f32), the lanes then walk their own row with a (TW+1)-padded stride (conflict-free),
// results go back through the same tile as coalesced row stores.  HBM traffic is 1 read + 1 write
// per sample (plus halo re-reads); no MFMA — these are scalar recurrences.
//
// Reference filters replaced (FFmpeg 8.1, instantiated by /root/reference/internal/processor):
//   highpass/lowpass a=tdii (filters.go:740-769), ebur128 K-weighting (filters.go:626),
//   agate/acompressor/deesser (filters.go:869-932), alimiter (normalise.go:446-480),
//   astats (filters.go:624).
#include "jt_internal.h"

constexpr int LANES = 64;
constexpr int TW = 64;

// wave-uniform: all LANES rows of the TW-column tile at `pos` lie inside [0, n)
__device__ inline bool tile_interior(int64_t n, int64_t base0, int64_t rowstride, int64_t pos)
{
    return base0 + pos >= 0 && base0 + (int64_t)(LANES - 1) * rowstride + pos + TW <= n;
}

// Edge tiles: element (r, lane) sits at tile-relative index rel = r * rowstride + lane (32-bit) from the wave-uniform origin
// base0 + pos; it exists when lo <= rel < hi.
struct TileEdge { int lo, hi; };
__device__ inline TileEdge tile_edge(int64_t n, int64_t base0, int64_t pos)
{
    const int64_t o = base0 + pos, big = (int64_t)1 << 30;
    TileEdge e;
    e.lo = (int)(-o < -big ? -big : (-o > big ? big : -o));
    e.hi = (int)(n - o < -big ? -big : (n - o > big ? big : n - o));
    return e;
}

// 16-byte row loads: a lane fetches E = 16 / sizeof(T) consecutive samples of one row, so a 64 x 64 tile takes 16 (f32) or 32 (f64)
// load instructions per lane instead of 64.  A tile qualifies when it is 16-byte aligned and every row's TW-sample segment lies
// either entirely inside [0, n) or entirely outside (rows outside load nothing and stage zeros) -- which, with chunk origins that
// are multiples of TW, is every tile except the one that straddles the end of the signal.  The kernels' duration is set by their
// slowest workgroup, and that used to be the first and the last one (clamped scalar loads); with this they stage like the rest.
// v[k*E + e] = element (row k*RPL + lane/LPR, column (lane%LPR)*E + e).
template <typename TIn> struct RowVec { static constexpr int E = 16 / (int)sizeof(TIn), LPR = TW / E, RPL = 64 / LPR, NLD = LANES / RPL; };
// run-level (wave-uniform) test: every tile of rows starting at `in + base0` with this stride can use 16-byte loads
template <typename TIn>
__device__ inline bool rows_vec_aligned(const TIn *in, int64_t base0, int64_t rowstride)
{
    using RV = RowVec<TIn>;
    return (rowstride % RV::E) == 0 && rowstride <= ((int64_t)1 << 24) && (reinterpret_cast<uintptr_t>(in + base0) & 15) == 0;
}
// elements outside [0, n) (or in rows >= nrows) come back as zeros; a vector that straddles the end of the signal is fetched
// element by element, so nothing outside the caller's buffer is ever read
template <typename TIn>
__device__ inline void rows_prefetch_vec(TIn (&v)[LANES], const TIn *__restrict__ in, int64_t base0, int64_t rowstride, int64_t pos,
                                         const TileEdge &eg, int nrows, int lane)
{
    using RV = RowVec<TIn>;
    typedef TIn vecT __attribute__((ext_vector_type(RV::E)));
    const int rs = (int)rowstride, col = (lane % RV::LPR) * RV::E;
    const TIn *org = in + (base0 + pos);
#pragma unroll
    for (int k = 0; k < RV::NLD; ++k) {
        const int row = k * RV::RPL + lane / RV::LPR, rel = row * rs + col;
        vecT t = (vecT)(TIn)0;
        if (row < nrows && rel >= eg.lo && rel + RV::E <= eg.hi) t = *reinterpret_cast<const vecT *>(org + rel);
        else if (row < nrows && rel + RV::E > eg.lo && rel < eg.hi) {
#pragma unroll
            for (int e = 0; e < RV::E; ++e) if (rel + e >= eg.lo && rel + e < eg.hi) t[e] = org[rel + e];
        }
#pragma unroll
        for (int e = 0; e < RV::E; ++e) v[k * RV::E + e] = t[e];
    }
}
template <typename TIn, typename TL>
__device__ inline void rows_commit_vec(TL (*tile)[TW + 1], const TIn (&v)[LANES], int lane)
{
    using RV = RowVec<TIn>;
#pragma unroll
    for (int k = 0; k < RV::NLD; ++k)
#pragma unroll
        for (int e = 0; e < RV::E; ++e) tile[k * RV::RPL + lane / RV::LPR][(lane % RV::LPR) * RV::E + e] = (TL)v[k * RV::E + e];
}

template <typename TIn, typename TL>
__device__ inline void tile_load(TL (*tile)[TW + 1], const TIn *__restrict__ in, int64_t n,
                                 int64_t base0, int64_t rowstride, int64_t pos, int lane, int nrows)
{
    // Row loads are issued in batches of 16 with clamped (always valid) addresses and no branches, so the 16
    // HBM round trips overlap; out-of-range rows are zeroed by a select afterwards.
    constexpr int G = 16;
    {
        const TileEdge eg = tile_edge(n, base0, pos);
        if (rows_vec_aligned(in, base0 + pos, rowstride)) {
            TIn v[LANES];
            rows_prefetch_vec<TIn>(v, in, base0, rowstride, pos, eg, nrows, lane);
            rows_commit_vec<TIn, TL>(tile, v, lane);
            return;
        }
    }
    if (tile_interior(n, base0, rowstride, pos) && nrows == LANES) {
        // every element of the tile exists (all but the first and last workgroups): no clamps, no selects, 64-bit address
        // arithmetic only on the wave-uniform row base
        const TIn *p = in + (base0 + pos) + lane;
#pragma unroll 1
        for (int r0 = 0; r0 < LANES; r0 += G) {
            TIn v[G];
#pragma unroll
            for (int q = 0; q < G; ++q) v[q] = p[(int64_t)(r0 + q) * rowstride];
#pragma unroll
            for (int q = 0; q < G; ++q) tile[r0 + q][lane] = (TL)v[q];
        }
        return;
    }
#pragma unroll 1
    for (int r0 = 0; r0 < LANES; r0 += G) {
        TIn v[G];
#pragma unroll
        for (int q = 0; q < G; ++q) {
            int64_t idx = base0 + (int64_t)(r0 + q) * rowstride + pos + lane;
            int64_t ic = idx < 0 ? 0 : (idx >= n ? n - 1 : idx);
            v[q] = in[ic];
        }
#pragma unroll
        for (int q = 0; q < G; ++q) {
            int64_t idx = base0 + (int64_t)(r0 + q) * rowstride + pos + lane;
            bool ok = (r0 + q) < nrows && idx >= 0 && idx < n;
            tile[r0 + q][lane] = ok ? (TL)v[q] : (TL)0;
        }
    }
}

// store columns [pos, pos+TW) of each row for indices inside [lo_r, hi_r)
template <typename TOut, typename TL>
__device__ inline void tile_store(TL (*tile)[TW + 1], TOut *__restrict__ out, int64_t n,
                                  int64_t base0, int64_t rowstride, int64_t pos, int lane, int nrows,
                                  int64_t halo, int64_t chunk)
{
    if (nrows == LANES && pos >= halo && pos + TW <= halo + chunk && tile_interior(n, base0, rowstride, pos)) {
        TOut *p = out + (base0 + pos) + lane;
#pragma unroll 8
        for (int r = 0; r < LANES; ++r) p[(int64_t)r * rowstride] = (TOut)tile[r][lane];
        return;
    }
#pragma unroll 4
    for (int r = 0; r < nrows; ++r) {
        int64_t rel = pos + lane;                       // position inside lane r's [halo + chunk) run
        int64_t idx = base0 + (int64_t)r * rowstride + rel;
        if (rel >= halo && rel < halo + chunk && idx >= 0 && idx < n) out[idx] = (TOut)tile[r][lane];
    }
}

// Software-pipelined variant: the next tile's 64 row loads are issued into registers BEFORE the current tile is
// processed, so their HBM latency hides behind the recurrence instead of stalling every tile.
template <typename TIn>
__device__ inline void rows_prefetch(TIn (&v)[LANES], const TIn *__restrict__ in, int64_t n, int64_t base0, int64_t rowstride,
                                     int64_t pos, int lane)
{
    if (tile_interior(n, base0, rowstride, pos)) {
        const TIn *p = in + (base0 + pos) + lane;
#pragma unroll
        for (int r = 0; r < LANES; ++r) v[r] = p[(int64_t)r * rowstride];
        return;
    }
#pragma unroll
    for (int r = 0; r < LANES; ++r) {
        int64_t idx = base0 + (int64_t)r * rowstride + pos + lane;
        int64_t ic = idx < 0 ? 0 : (idx >= n ? n - 1 : idx);
        v[r] = in[ic];
    }
}
template <typename TIn, typename TL>
__device__ inline void rows_commit(TL (*tile)[TW + 1], const TIn (&v)[LANES], int64_t n, int64_t base0, int64_t rowstride,
                                   int64_t pos, int lane, int nrows)
{
    if (nrows == LANES && tile_interior(n, base0, rowstride, pos)) {
#pragma unroll
        for (int r = 0; r < LANES; ++r) tile[r][lane] = (TL)v[r];
        return;
    }
#pragma unroll
    for (int r = 0; r < LANES; ++r) {
        int64_t idx = base0 + (int64_t)r * rowstride + pos + lane;
        bool ok = r < nrows && idx >= 0 && idx < n;
        tile[r][lane] = ok ? (TL)v[r] : (TL)0;
    }
}

// ------------------------------------------------------------------ small elementwise kernels
// aformat=channel_layouts=mono of a stereo source = the aresample libavfilter inserts in front of it (libswresample/rematrix.c).
// swr_build_matrix2: FRONT_CENTER <- M_SQRT1_2 * FL + M_SQRT1_2 * FR; auto_matrix normalises the row to sum 1 only when the
// converter's OUTPUT or INTERNAL sample format is an integer one (maxval = 1.0), for float formats maxval = INT_MAX and the
// coefficients stay 1/sqrt(2).  Which formats those are is decided by the graph behind the down-mix:
//   mode 0  output fltp, internal FLTP (Pass 1, Pass 2: a float-only filter follows, whatever the source format):
//           float coefficients, mix_2_1 = fl(fl(c*L) + fl(c*R)), c = (float)M_SQRT1_2
//   mode 1  the band graphs of a 16-bit source (output s16p, internal S16P): (L*16384 + R*16384 + 16384) >> 15 on the integers
//   mode 2  the band graphs of a 24/32-bit source (output s32p, internal FLTP): coefficients 0.5, then flt -> s32 (llrintf, clip)
__global__ void k_downmix(const float *__restrict__ in, float *__restrict__ out, int64_t frames, int channels, int mode, DownmixRow row)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float c = 0.70710678118654752440f;
    for (; i < frames; i += stride) {
        if (row.stereo) {
            float2 v = reinterpret_cast<const float2 *>(in)[i];
            if (mode == 0) out[i] = __fadd_rn(__fmul_rn(c, v.x), __fmul_rn(c, v.y));
            else if (mode == 1) {
                const int li = (int)rintf(v.x * 32768.0f), ri = (int)rintf(v.y * 32768.0f);
                out[i] = (float)((li * 16384 + ri * 16384 + 16384) >> 15) * (1.0f / 32768.0f);
            } else {
                const float sm = __fadd_rn(0.5f * v.x, 0.5f * v.y);
                double q = rint((double)sm * 2147483648.0);
                q = q > 2147483647.0 ? 2147483647.0 : (q < -2147483648.0 ? -2147483648.0 : q);
                out[i] = (float)(q * (1.0 / 2147483648.0));
            }
            continue;
        }
        // any other layout (swri_rematrix for one output channel): k = 1 copy / mix_1_1, k = 2 mix_2_1, k >= 3 the generic loop
        // v = 0; v += in_j * c_j in channel order, float products and float sums (S16P: integer products, (v + 16384) >> 15)
        const float *x = in + i * channels;
        if (mode == 1) {
            int v = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < row.k) v += (int)rintf(x[row.nz[j]] * 32768.0f) * row.ci[j];
            out[i] = (float)(short)((v + 16384) >> 15) * (1.0f / 32768.0f);
            continue;
        }
        float v;
        if (row.k == 1) v = row.cf[0] == 1.0f ? x[row.nz[0]] : __fmul_rn(x[row.nz[0]], row.cf[0]);
        else if (row.k == 2) v = __fadd_rn(__fmul_rn(x[row.nz[0]], row.cf[0]), __fmul_rn(x[row.nz[1]], row.cf[1]));
        else {
            v = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j < row.k) v = __fadd_rn(v, __fmul_rn(x[row.nz[j]], row.cf[j]));
        }
        if (mode == 0) { out[i] = v; continue; }
        double q = rint((double)v * 2147483648.0);
        q = q > 2147483647.0 ? 2147483647.0 : (q < -2147483648.0 ? -2147483648.0 : q);
        out[i] = (float)(q * (1.0 / 2147483648.0));
    }
}
unsigned long long jt_default_layout(int channels)
{
    // libavutil/channel_layout.c, channel_layout_map[]: the first entry with that many channels -- mono, stereo, 2.1, 4.0, 5.0 (back),
    // 5.1 (back), 6.1, 7.1
    static const unsigned long long def[9] = {0, 0x4, 0x3, 0xB, 0x107, 0x37, 0x3F, 0x70F, 0x63F};
    return channels >= 1 && channels <= 8 ? def[channels] : 0;
}
// libswresample/rematrix.c swr_build_matrix2, defaults (center_mix_level = surround_mix_level = M_SQRT1_2, lfe_mix_level = 0), one
// FRONT_CENTER output: FL / FR M_SQRT1_2 each; FC 1.0 (center_mix_level * sqrt(2) when FL / FR are there); BC, BL / BR, SL / SR
// surround_mix_level * M_SQRT1_2 each; FLC / FRC M_SQRT1_2 each; LFE 0.  auto_matrix divides the row by the sum of its |coefficients|
// when that exceeds maxval = 1.0 -- only when the converter's output or internal format is an integer one (modes 1, 2: the band graphs
// of integer sources); float graphs keep the raw coefficients (mode 0).
bool jt_downmix_row(int channels, unsigned long long mask, int mode, DownmixRow *row)
{
    if (!mask) mask = jt_default_layout(channels);
    if (!mask || (mask >> 11) || channels < 1 || channels > 8) return false;
    int nb = 0; for (int b = 0; b < 11; ++b) nb += (int)((mask >> b) & 1);
    if (nb != channels) return false;
    const bool stereo = (mask & 3) != 0;
    double coef[8]; double sum = 0.0; int c = 0;
    for (int b = 0; b < 11; ++b) {
        if (!((mask >> b) & 1)) continue;
        double v;
        switch (b) {
        case 0: case 1: case 6: case 7: v = M_SQRT1_2; break;
        case 2: v = stereo ? M_SQRT1_2 * std::sqrt(2.0) : 1.0; break;
        case 3: v = 0.0; break;
        default: v = M_SQRT1_2 * M_SQRT1_2; break;
        }
        coef[c++] = v; sum += std::fabs(v);
    }
    if (mode != 0 && sum > 1.0) for (int i = 0; i < channels; ++i) coef[i] /= sum;
    std::memset(row, 0, sizeof *row);
    row->stereo = mask == 0x3 ? 1 : 0;                   // FL + FR: the two-channel fast path (same statements, constants folded)
    for (int i = 0; i < channels; ++i) if (coef[i] != 0.0) row->nz[row->k++] = i;
    for (int j = 0; j < row->k; ++j) row->cf[j] = (float)coef[row->nz[j]];
    if (row->k == 2) {
        // mix_2_1's native_matrix (swri_rematrix_init): lrintf's remainder is carried from one input of the row to the next
        double rem = 0.0;
        for (int i = 0; i < channels; ++i) {
            const double target = coef[i] * 32768 + rem; const int q = (int)lrintf((float)target); rem += target - q;
            for (int j = 0; j < row->k; ++j) if (row->nz[j] == i) row->ci[j] = q;
        }
    } else for (int j = 0; j < row->k; ++j) row->ci[j] = (int)lrintf((float)(coef[row->nz[j]] * 32768));
    return true;
}
void launch_downmix(const float *in, float *out, int64_t frames, int channels, int mode, const DownmixRow &row, hipStream_t s)
{
    int grid = (int)std::min<int64_t>((frames + 255) / 256, 4096);
    hipLaunchKernelGGL(k_downmix, dim3(grid), dim3(256), 0, s, in, out, frames, channels, mode, row);
}

__global__ void k_s16_to_f32(const int16_t *__restrict__ in, float *__restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (float)in[i] * (1.0f / 32768.0f);
}
void launch_s16_to_f32(const int16_t *in, float *out, int64_t n, hipStream_t s)
{
    int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_s16_to_f32, dim3(grid), dim3(256), 0, s, in, out, n);
}
__global__ void k_s16_to_f32_pair(const int16_t *__restrict__ in0, int64_t n0, const int16_t *__restrict__ in1, int64_t n1, float *__restrict__ out0,
                                  float *__restrict__ out1)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n0 + n1; i += stride) {
        if (i < n0) out0[i] = (float)in0[i] * (1.0f / 32768.0f);
        else out1[i - n0] = (float)in1[i - n0] * (1.0f / 32768.0f);
    }
}
void launch_s16_to_f32_pair(const int16_t *in0, int64_t n0, const int16_t *in1, int64_t n1, float *out0, float *out1, hipStream_t s)
{
    if (n0 + n1 <= 0) return;
    const int grid = (int)std::min<int64_t>((n0 + n1 + 255) / 256, 4096);
    hipLaunchKernelGGL(k_s16_to_f32_pair, dim3(grid), dim3(256), 0, s, in0, n0, in1, n1, out0, out1);
}

// s16 -> dbl (audioconvert: x * (1.0/(1<<15))), optional volume stage in float precision
// (af_volume.c precision=float: s16 -> flt, scale in float, then flt -> dbl).
__global__ void k_s16_to_f64(const int16_t *__restrict__ in, double *__restrict__ out, int64_t n, double gain, int gain_in_float)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (gain_in_float) {
            float v = (float)in[i] * (1.0f / 32768.0f);
            v = v * (float)gain;
            out[i] = (double)v;
        } else {
            out[i] = (double)in[i] * (1.0 / 32768.0) * gain;
        }
    }
}
void launch_s16_to_f64(const int16_t *in, double *out, int64_t n, double gain, int gain_in_float, hipStream_t s)
{
    int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_s16_to_f64, dim3(grid), dim3(256), 0, s, in, out, n, gain, gain_in_float);
}

__global__ void k_f32_to_f64(const float *__restrict__ in, double *__restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (double)in[i];
}
void launch_f32_to_f64(const float *in, double *out, int64_t n, hipStream_t s)
{
    int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_f32_to_f64, dim3(grid), dim3(256), 0, s, in, out, n);
}

// dbl -> s16 (audioconvert: av_clip_int16(lrint(x * 32768))); round_via_float models the dbl->flt->dbl
// hop FFmpeg inserts around aspectralstats (fltp) before ebur128/aformat.
__global__ void k_f64_to_s16(const double *__restrict__ in, int16_t *__restrict__ out, float *__restrict__ out_f32,
                             int64_t n, int round_via_float)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        double v = in[i];
        if (round_via_float) { float f = (float)v; v = (double)f; if (out_f32) out_f32[i] = f; }
        double r = rint(v * 32768.0);
        r = r < -32768.0 ? -32768.0 : (r > 32767.0 ? 32767.0 : r);
        if (out) out[i] = (int16_t)r;
    }
}
void launch_f64_to_s16(const double *in, int16_t *out, float *out_f32, int64_t n, int round_via_float, hipStream_t s)
{
    int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_f64_to_s16, dim3(grid), dim3(256), 0, s, in, out, out_f32, n, round_via_float);
}

// ------------------------------------------------------------------ per-decoder-frame sum(x^2), max|x|
// frameSumSquaresAndPeak (analyser_metrics.go:273-358): double accumulation over all interleaved samples.
__global__ void k_frame_stats(const float *__restrict__ in, int64_t n_total, int spf, double *__restrict__ sumsq,
                              double *__restrict__ peak, int64_t nframes)
{
    int64_t f = blockIdx.x;
    if (f >= nframes) return;
    int64_t lo = f * (int64_t)spf, hi = lo + spf;
    if (hi > n_total) hi = n_total;
    double acc = 0.0, pk = 0.0;
    if (hi - lo == 4096 && blockDim.x == 256) {
        // (round 6) a whole 4096-sample frame: the thread's sixteen loads leave together, then the same sums in the same order -- the loop
        // below waits for every load before it issues the next (7 % active, 87 % waiting: 0.6 ms beside Pass 1's analysis)
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = in[lo + threadIdx.x + 256 * k];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const double d = (double)v[k]; acc += d * d; pk = fmax(pk, fabs(d)); }
    } else
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        double v = (double)in[i];
        acc += v * v;
        pk = fmax(pk, fabs(v));
    }
    for (int off = 32; off > 0; off >>= 1) {
        acc += __shfl_down(acc, off, 64);
        pk = fmax(pk, __shfl_down(pk, off, 64));
    }
    __shared__ double sa[4], sp[4];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sa[w] = acc; sp[w] = pk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, p = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { a += sa[k]; p = fmax(p, sp[k]); }
        sumsq[f] = a; peak[f] = p;
    }
}
void launch_frame_stats(const float *in, int64_t n_total, int spf, double *sumsq, double *peak, int64_t nframes, hipStream_t s)
{
    if (nframes <= 0) return;
    hipLaunchKernelGGL(k_frame_stats, dim3((unsigned)nframes), dim3(256), 0, s, in, n_total, spf, sumsq, peak, nframes);
}

// the same for decoder frames of different lengths (a variable-blocksize FLAC stream): frame f = samples off[f] .. off[f + 1] per channel
__global__ void k_frame_stats_var(const float *__restrict__ in, int channels, const int64_t *__restrict__ off, double *__restrict__ sumsq,
                                  double *__restrict__ peak, int64_t nframes)
{
    int64_t f = blockIdx.x;
    if (f >= nframes) return;
    const int64_t lo = off[f] * channels, hi = off[f + 1] * channels;
    double acc = 0.0, pk = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        double v = (double)in[i];
        acc += v * v;
        pk = fmax(pk, fabs(v));
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_down(acc, o, 64);
        pk = fmax(pk, __shfl_down(pk, o, 64));
    }
    __shared__ double sa[4], sp[4];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sa[w] = acc; sp[w] = pk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, p = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { a += sa[k]; p = fmax(p, sp[k]); }
        sumsq[f] = a; peak[f] = p;
    }
}
void launch_frame_stats_var(const float *in, int channels, const int64_t *off, double *sumsq, double *peak, int64_t nframes, hipStream_t s)
{
    if (nframes <= 0) return;
    hipLaunchKernelGGL(k_frame_stats_var, dim3((unsigned)nframes), dim3(256), 0, s, in, channels, off, sumsq, peak, nframes);
}

// calculateFrameLevel (encoder.go:235-257) on the frames of a stage output: sum of squares of (s16 / 32768) per frame
__global__ void k_frame_sumsq_s16(const int16_t *__restrict__ in, int64_t n, int spf, double *__restrict__ sumsq, int64_t nframes)
{
    const int64_t f = blockIdx.x;
    if (f >= nframes) return;
    const int64_t lo = f * (int64_t)spf, hi = min(lo + spf, n);
    double acc = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) { const double v = (double)in[i] * (1.0 / 32768.0); acc += v * v; }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double sa[4];
    if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double a = 0; for (int k = 0; k < (int)(blockDim.x >> 6); ++k) a += sa[k]; sumsq[f] = a; }
}
void launch_frame_sumsq_s16(const int16_t *in, int64_t n, int spf, double *sumsq, int64_t nframes, hipStream_t s)
{
    if (nframes <= 0) return;
    hipLaunchKernelGGL(k_frame_sumsq_s16, dim3((unsigned)nframes), dim3(256), 0, s, in, n, spf, sumsq, nframes);
}

// ------------------------------------------------------------------ biquad cascade (f32, TDII)
// BIQUAD_TDII_FILTER(flt): out = b0*in + w1; w1 = b1*in + w2 + a1*out; w2 = b2*in + a2*out
__global__ void __launch_bounds__(64)
k_biquad_f32(const float *__restrict__ in, float *__restrict__ out, int64_t n, int64_t chunk, int64_t halo,
             int nstages, BiquadF32 s0, BiquadF32 s1, int64_t nchunks)
{
    __shared__ float tile[LANES][TW + 1];
    const int lane = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * LANES;
    const int nrows = (int)min((int64_t)LANES, nchunks - c0);
    const int64_t base0 = c0 * chunk - halo;
    float w1a = 0.f, w2a = 0.f, w1b = 0.f, w2b = 0.f;
    const int64_t total = halo + chunk;
    float v[LANES];
    rows_prefetch<float>(v, in, n, base0, chunk, 0, lane);
    for (int64_t pos = 0; pos < total; pos += TW) {
        rows_commit<float, float>(tile, v, n, base0, chunk, pos, lane, nrows);
        __syncthreads();
        if (pos + TW < total) rows_prefetch<float>(v, in, n, base0, chunk, pos + TW, lane);   // next tile's loads fly during this one
        if (lane < nrows) {
            // 16 samples at a time: their LDS reads first, the recurrence on registers, the stores after (left in one loop, every
            // sample waited for its own LDS read).  Before the file start the true state is exactly zero and so is the staged input,
            // which keeps every state at zero: no index test is needed
#pragma unroll 1
            for (int j0 = 0; j0 < TW; j0 += 16) {
                float xs[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) xs[j] = tile[lane][j0 + j];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float x = xs[j];
                    float y = __fadd_rn(__fmul_rn(s0.b0, x), w1a);
                    w1a = __fadd_rn(__fadd_rn(__fmul_rn(s0.b1, x), w2a), __fmul_rn(s0.a1, y));
                    w2a = __fadd_rn(__fmul_rn(s0.b2, x), __fmul_rn(s0.a2, y));
                    x = y;
                    if (nstages > 1) {
                        float y2 = __fadd_rn(__fmul_rn(s1.b0, x), w1b);
                        w1b = __fadd_rn(__fadd_rn(__fmul_rn(s1.b1, x), w2b), __fmul_rn(s1.a1, y2));
                        w2b = __fadd_rn(__fmul_rn(s1.b2, x), __fmul_rn(s1.a2, y2));
                        x = y2;
                    }
                    xs[j] = x;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) tile[lane][j0 + j] = xs[j];
            }
        }
        __syncthreads();
        if (pos + TW > halo) tile_store<float, float>(tile, out, n, base0, chunk, pos, lane, nrows, halo, chunk);   // warm-up tiles store nothing
        __syncthreads();
    }
}

void launch_biquad_f32(const float *in, float *out, int64_t n, int nstages, const BiquadF32 *st, hipStream_t s)
{
    if (n <= 0) return;
    // halo from the slowest pole radius: r^2 = |a2|  =>  r^halo <= 1e-10
    double rmax = 0.0;
    for (int k = 0; k < nstages; ++k) rmax = std::max(rmax, std::sqrt(std::fabs((double)st[k].a2)));
    int64_t halo = 256;
    if (rmax > 0.0 && rmax < 1.0) halo = (int64_t)std::ceil(std::log(1e-10) / std::log(rmax));
    halo = std::max<int64_t>(256, std::min<int64_t>(halo, 1 << 20));
    halo = (halo + TW - 1) / TW * TW;
    // one lane = one chunk and the launch lasts as long as a lane's serial run (halo + chunk dependent steps): chunks of half a
    // halo keep that run short; an hour of audio is still ~1700 waves, and the re-read halo stays in L2
    int64_t chunk = std::max<int64_t>(1024, halo / 2);
    chunk = (chunk + TW - 1) / TW * TW;
    int64_t nchunks = (n + chunk - 1) / chunk;
    int grid = (int)((nchunks + LANES - 1) / LANES);
    BiquadF32 z{1.f, 0.f, 0.f, 0.f, 0.f};
    hipLaunchKernelGGL(k_biquad_f32, dim3(grid), dim3(LANES), 0, s, in, out, n, chunk, halo, nstages,
                       st[0], nstages > 1 ? st[1] : z, nchunks);
}

// ------------------------------------------------------------------ K-weighting + 100 ms block energies
// BS.1770 K-weighting (f_ebur128.c / ebur128.c coefficients: pre-filter shelf + RLB high-pass) is a LINEAR 4-state filter,
// so chunk boundary states need no warm-up halo:   state_c = e_{c-1} + F^L e_{c-2} + F^2L e_{c-3} + ...
// where e_c is the zero-state response of chunk c at its end and F the homogeneous transition matrix (|eig| ~ 0.995, so the
// series is truncated once ||F^kL|| < 1e-18: exact to double precision).
//   k_kw_zs  : one lane per L-sample chunk, zero initial state -> e_c                       (N sample-steps)
//   k_kw_run : one lane per chunk, carried-in state from the series, accumulates sum(z^2) and max|x| per chunk (N sample-steps)
// Chunks are sub-divisions of the 100 ms blocks; the host adds the per-chunk partials of each block in order.
// Both biquads run in transposed direct form II with fused multiply-adds (same transfer function as the FILTER macro of
// f_ebur128.c; rounding differs at 1e-16).
// (struct KwCoef: jt_internal.h)

#define KW2_STEP(X)                                                     \
    {                                                                    \
        const double xx = (X);                                           \
        const double y = fma(k.b0, xx, s1);                              \
        s1 = fma(-k.a1, y, fma(k.b1, xx, s2));                           \
        s2 = fma(-k.a2, y, k.b2 * xx);                                   \
        zz = fma(k.c0, y, t1);                                           \
        t1 = fma(-k.d1, zz, fma(k.c1, y, t2));                           \
        t2 = fma(-k.d2, zz, k.c2 * y);                                   \
    }

template <typename TIn, bool RUN>
__global__ void __launch_bounds__(64)
k_kw(const TIn *__restrict__ in, int64_t n, int64_t L, KwCoef k, const double *__restrict__ zs_in, const double *__restrict__ fpow, int nterms,
     double *__restrict__ zs_out, double *__restrict__ csum, double *__restrict__ cpeak, int64_t nchunks)
{
    __shared__ TIn tile[LANES][TW + 1];
    const int lane = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * LANES;
    const int nrows = (int)min((int64_t)LANES, nchunks - c0);
    const int64_t base0 = c0 * L;
    const int64_t my_c = c0 + lane;
    const int64_t my_base = base0 + (int64_t)lane * L;
    double s1 = 0, s2 = 0, t1 = 0, t2 = 0;
    if (RUN && lane < nrows) {
        // carried-in state: sum_k F^(kL) e_{c-1-k}   (k = 0 term is e_{c-1} itself)
        for (int q = 0; q < nterms; ++q) {
            const int64_t cc = my_c - 1 - q;
            if (cc < 0) break;
            const double e0 = zs_in[cc * 4 + 0], e1 = zs_in[cc * 4 + 1], e2 = zs_in[cc * 4 + 2], e3 = zs_in[cc * 4 + 3];
            if (q == 0) { s1 += e0; s2 += e1; t1 += e2; t2 += e3; }
            else {
                const double *M = fpow + (size_t)(q - 1) * 16;
                s1 += M[0] * e0 + M[1] * e1 + M[2] * e2 + M[3] * e3;
                s2 += M[4] * e0 + M[5] * e1 + M[6] * e2 + M[7] * e3;
                t1 += M[8] * e0 + M[9] * e1 + M[10] * e2 + M[11] * e3;
                t2 += M[12] * e0 + M[13] * e1 + M[14] * e2 + M[15] * e3;
            }
        }
    }
    double acc = 0.0, pk = 0.0, zz = 0.0;
    TIn v[LANES];
    rows_prefetch<TIn>(v, in, n, base0, L, 0, lane);
    for (int64_t pos = 0; pos < L; pos += TW) {
        rows_commit<TIn, TIn>(tile, v, n, base0, L, pos, lane, nrows);
        __syncthreads();
        if (pos + TW < L) rows_prefetch<TIn>(v, in, n, base0, L, pos + TW, lane);
        if (lane < nrows) {
            const int64_t t0 = my_base + pos;
            if (pos + TW <= L && t0 + TW <= n) {
#pragma unroll 16
                for (int j = 0; j < TW; ++j) {
                    const double x0 = (double)tile[lane][j];
                    KW2_STEP(x0)
                    if (RUN) { acc = fma(zz, zz, acc); pk = fmax(pk, fabs(x0)); }
                }
            } else {
                for (int j = 0; j < TW; ++j) {
                    if (pos + j >= L || t0 + j >= n) break;
                    const double x0 = (double)tile[lane][j];
                    KW2_STEP(x0)
                    if (RUN) { acc = fma(zz, zz, acc); pk = fmax(pk, fabs(x0)); }
                }
            }
        }
        __syncthreads();
    }
    if (lane < nrows) {
        if (RUN) { csum[my_c] = acc; cpeak[my_c] = pk; }
        else { zs_out[my_c * 4 + 0] = s1; zs_out[my_c * 4 + 1] = s2; zs_out[my_c * 4 + 2] = t1; zs_out[my_c * 4 + 3] = t2; }
    }
}
// ONE sweep instead of two (round 3).  With zs the chunk's zero-state output and h_j = g_j . s0 the homogeneous response to the state s0
// carried into the chunk, the chunk's energy is sum (zs_j + h_j)^2 = sum zs_j^2 + 2 s0 . (sum zs_j g_j) + s0' (sum g_j g_j') s0: the sweep
// that produces the zero-state end states also accumulates sum zs^2 and the four sums zs_j g_j[k] (g_j: the output at step j of a chunk
// started from the k-th unit state with no input, a table per (rate, L) that every lane reads at the same j: scalar loads), and
// k_kw_fix puts the carried state in afterwards, one thread per chunk.  The second sweep re-read the whole signal (5.5 GB for an hour at
// 192 kHz) to do the same thing sample by sample.  Rounding differs from the two-sweep form at the 1e-13 level (the three terms cancel
// the start-up transient of the zero-state run); JT_KW_TWO_SWEEPS=1 keeps the old form.
// (round 4) The sweep ran five times slower than its 15 f64 operations per sample cost: the compiler put every LDS read -- the sample,
// and the table row, which sat in LDS too -- directly in front of its use behind an s_waitcnt lgkmcnt(0), two or three exposed LDS round
// trips per sample.  Now four samples and their four table rows are read from LDS into registers a block ahead, with scheduling
// barriers between "request the next block" and "compute this one".  Same statements per sample, in the same order.  (The rows are
// wave-uniform, but as scalar loads two blocks of them are 64 SGPRs beside 20 of coefficients: the allocator spilled the lot.)
template <typename TIn>
__global__ void __launch_bounds__(64)
k_kw1(const TIn *__restrict__ in, int64_t n, int64_t L, KwCoef k, const double *__restrict__ g, double *__restrict__ zs_out,
      double *__restrict__ csum, double *__restrict__ cpeak, double *__restrict__ cross, int64_t nchunks)
{
    __shared__ TIn tile[LANES][TW + 1];
    __shared__ __attribute__((aligned(32))) double gt[TW * 4];      // the table rows of the current 64-sample step (read at a wave-uniform index: LDS broadcasts)
    const int lane = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * LANES;
    const int nrows = (int)min((int64_t)LANES, nchunks - c0);
    const int64_t base0 = c0 * L;
    const int64_t my_c = c0 + lane;
    const int64_t my_base = base0 + (int64_t)lane * L;
    double s1 = 0, s2 = 0, t1 = 0, t2 = 0;
    double acc = 0.0, pk = 0.0, zz = 0.0, x0c = 0.0, x1c = 0.0, x2c = 0.0, x3c = 0.0;
    TIn v[LANES];
    rows_prefetch<TIn>(v, in, n, base0, L, 0, lane);
    constexpr int KB = 4;                                            // samples per block
    for (int64_t pos = 0; pos < L; pos += TW) {
        rows_commit<TIn, TIn>(tile, v, n, base0, L, pos, lane, nrows);
        if (pos + lane < L) {
            const double4 gv = *reinterpret_cast<const double4 *>(g + 4 * (pos + lane));
            *reinterpret_cast<double4 *>(gt + 4 * lane) = gv;
        }
        __syncthreads();
        if (pos + TW < L) rows_prefetch<TIn>(v, in, n, base0, L, pos + TW, lane);
        if (lane < nrows) {
            const int64_t t0 = my_base + pos;
            const double *gp = gt;
            if (pos + TW <= L && t0 + TW <= n) {
                TIn xa[KB], xb[KB]; double ga[4 * KB], gb[4 * KB];
                auto request = [&](TIn (&xs)[KB], double (&gs)[4 * KB], int j0) {
#pragma unroll
                    for (int u = 0; u < KB; ++u) xs[u] = tile[lane][j0 + u];
#pragma unroll
                    for (int u = 0; u < 4 * KB; ++u) gs[u] = gp[4 * j0 + u];
                };
                auto compute = [&](const TIn (&xs)[KB], const double (&gs)[4 * KB]) {
#pragma unroll
                    for (int u = 0; u < KB; ++u) {
                        const double x0 = (double)xs[u];
                        KW2_STEP(x0)
                        acc = fma(zz, zz, acc); pk = fmax(pk, fabs(x0));
                        x0c = fma(zz, gs[4 * u + 0], x0c); x1c = fma(zz, gs[4 * u + 1], x1c); x2c = fma(zz, gs[4 * u + 2], x2c); x3c = fma(zz, gs[4 * u + 3], x3c);
                    }
                };
                request(xa, ga, 0);
#pragma unroll
                for (int jb = 0; jb < TW; jb += 2 * KB) {
                    request(xb, gb, jb + KB);
                    __builtin_amdgcn_sched_barrier(0);
                    compute(xa, ga);
                    __builtin_amdgcn_sched_barrier(0);
                    if (jb + 2 * KB < TW) request(xa, ga, jb + 2 * KB);
                    __builtin_amdgcn_sched_barrier(0);
                    compute(xb, gb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (int j = 0; j < TW; ++j) {
                    if (pos + j >= L || t0 + j >= n) break;
                    const double x0 = (double)tile[lane][j];
                    KW2_STEP(x0)
                    acc = fma(zz, zz, acc); pk = fmax(pk, fabs(x0));
                    x0c = fma(zz, gp[4 * j + 0], x0c); x1c = fma(zz, gp[4 * j + 1], x1c); x2c = fma(zz, gp[4 * j + 2], x2c); x3c = fma(zz, gp[4 * j + 3], x3c);
                }
            }
        }
        __syncthreads();
    }
    if (lane < nrows) {
        csum[my_c] = acc; cpeak[my_c] = pk;
        zs_out[my_c * 4 + 0] = s1; zs_out[my_c * 4 + 1] = s2; zs_out[my_c * 4 + 2] = t1; zs_out[my_c * 4 + 3] = t2;
        cross[my_c * 4 + 0] = x0c; cross[my_c * 4 + 1] = x1c; cross[my_c * 4 + 2] = x2c; cross[my_c * 4 + 3] = x3c;
    }
}
// csum[c] += 2 s0 . cross[c] + s0' G s0, s0 = the state carried into chunk c; G = sum g_j g_j' over the chunk's length (the last chunk may
// be shorter: Gt), upper triangle row-major (k_kw_fixblocks)
struct KwGram { double f[10], t[10]; };
#undef KW2_STEP

// Per-block sums / peaks from the per-chunk partials, in chunk order (what the host loop of jt_kweight_finish did over 360 k chunks of an
// hour at 192 kHz, 0.4 ms on the critical path between Pass 3 and Pass 4, behind a 5.8 MB copy): one thread per 100 ms block, the chunks
// past the last full block fold into the trailing partial block.
__global__ void __launch_bounds__(64)
k_kw_blocks(const double *__restrict__ csum, const double *__restrict__ cpeak, int64_t nchunks, int m, int64_t nfull, double *__restrict__ out)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b > nfull) return;
    const int64_t c0 = b * m, c1 = b == nfull ? nchunks : min(nchunks, c0 + m);
    double sum = 0.0, pk = 0.0;
    for (int64_t c = c0; c < c1; ++c) { sum += csum[c]; pk = fmax(pk, cpeak[c]); }
    out[b] = sum; out[nfull + 1 + b] = pk;
}

// k_kw_fix and k_kw_blocks as one launch, one thread per 100 ms block: the state carried into the block's first chunk from the series,
// then chunk by chunk s <- F^L s + e_c (the series' own recurrence; fpow[0..15] = F^L), the carried-state terms added to each chunk's
// zero-state energy, the block's sum and peak formed in chunk order.  Two launches and a pass over csum fewer per K-weighting job (nine
// jobs per file).
__global__ void __launch_bounds__(64)
k_kw_fixblocks(const double *__restrict__ zs_in, const double *__restrict__ fpow, int nterms, const double *__restrict__ cross, KwGram G,
               const double *__restrict__ csum, const double *__restrict__ cpeak, int64_t nchunks, int m, int64_t nfull, double *__restrict__ out)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b > nfull) return;
    const int64_t c0 = b * m, c1 = b == nfull ? nchunks : min(nchunks, c0 + m);
    double s[4] = {0, 0, 0, 0};
    for (int q = 0; q < nterms; ++q) {
        const int64_t cc = c0 - 1 - q;
        if (cc < 0) break;
        const double e0 = zs_in[cc * 4 + 0], e1 = zs_in[cc * 4 + 1], e2 = zs_in[cc * 4 + 2], e3 = zs_in[cc * 4 + 3];
        if (q == 0) { s[0] += e0; s[1] += e1; s[2] += e2; s[3] += e3; }
        else {
            const double *M = fpow + (size_t)(q - 1) * 16;
            s[0] += M[0] * e0 + M[1] * e1 + M[2] * e2 + M[3] * e3;
            s[1] += M[4] * e0 + M[5] * e1 + M[6] * e2 + M[7] * e3;
            s[2] += M[8] * e0 + M[9] * e1 + M[10] * e2 + M[11] * e3;
            s[3] += M[12] * e0 + M[13] * e1 + M[14] * e2 + M[15] * e3;
        }
    }
    double sum = 0.0, pk = 0.0;
    for (int64_t c = c0; c < c1; ++c) {
        const double *Gm = c == nchunks - 1 ? G.t : G.f;
        double lin = 0.0, quad = 0.0;
        for (int a = 0; a < 4; ++a) lin += s[a] * cross[c * 4 + a];
        int u = 0;
        for (int a = 0; a < 4; ++a) for (int bb = a; bb < 4; ++bb, ++u) quad += (a == bb ? 1.0 : 2.0) * Gm[u] * s[a] * s[bb];
        sum += csum[c] + 2.0 * lin + quad;
        pk = fmax(pk, cpeak[c]);
        if (c + 1 < c1) {
            const double e0 = zs_in[c * 4 + 0], e1 = zs_in[c * 4 + 1], e2 = zs_in[c * 4 + 2], e3 = zs_in[c * 4 + 3];
            const double n0 = fpow[0] * s[0] + fpow[1] * s[1] + fpow[2] * s[2] + fpow[3] * s[3] + e0;
            const double n1 = fpow[4] * s[0] + fpow[5] * s[1] + fpow[6] * s[2] + fpow[7] * s[3] + e1;
            const double n2 = fpow[8] * s[0] + fpow[9] * s[1] + fpow[10] * s[2] + fpow[11] * s[3] + e2;
            const double n3 = fpow[12] * s[0] + fpow[13] * s[1] + fpow[14] * s[2] + fpow[15] * s[3] + e3;
            s[0] = n0; s[1] = n1; s[2] = n2; s[3] = n3;
        }
    }
    out[b] = sum; out[nfull + 1 + b] = pk;
}

static void mat4_mul(const double *A, const double *B, double *C)
{
    double t[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int q = 0; q < 4; ++q) s += A[i * 4 + q] * B[q * 4 + j]; t[i * 4 + j] = s; }
    std::memcpy(C, t, sizeof(t));
}

// Per-100ms-block K-weighted energies and sample peaks of a device signal.  enqueue: both kernels + the async copy of the
// per-chunk partials into the pinned arena; finish (after the pass's sync): per-block sums in chunk order.
// chunk length: a divisor of the block, at most ~2400 samples, preferably a multiple of the 64-sample tile
static int kw_chunks_per_block(int blk)
{
    int best = -1;
    for (int q = 1; q <= 32; ++q) if (blk % q == 0 && blk / q <= 2400) { if (best < 0) best = q; if ((blk / q) % TW == 0) { best = q; break; } }
    return best > 0 ? best : 1;
}
void jt_kweight_scratch_sizes(int64_t n, int blk, size_t *dev_doubles, size_t *pin_doubles, int64_t chunk_len)
{
    const int64_t L = chunk_len > 0 ? chunk_len : blk / kw_chunks_per_block(blk), nchunks = (n + L - 1) / L;
    *dev_doubles = 10 * (size_t)nchunks + 64;                                      // chunk partials (the block sums / peaks go straight to the host)
    *pin_doubles = 2 * (size_t)(n / blk + 1) + 64;
}

// The job around a K-weighting sweep: chunk plan, transition-matrix powers, scratch, homogeneous-response table; then `sweep` (k_kw1 over a
// signal in memory -- or a kernel that PRODUCES the signal and never stores it: the fused Pass-3 measurement, k_resample.hip), k_kw_fix,
// k_kw_blocks and the copy of the block sums.  chunk_len > 0 forces the chunk length (a divisor of blk).
static void kweight_job(jt_ctx *h, int64_t n, int rate, int blk, int64_t chunk_len, KwJob *job, hipStream_t st, const KwScratch *ext,
                        const std::function<void(const KwSweep &)> &sweep)
{
    BiquadF64 pre, rlb; jt_kweight_design(rate, &pre, &rlb);
    KwCoef k{pre.b0, pre.b1, pre.b2, pre.a1, pre.a2, rlb.b0, rlb.b1, rlb.b2, rlb.a1, rlb.a2};
    const int m = chunk_len > 0 ? (int)(blk / chunk_len) : kw_chunks_per_block(blk);
    const int64_t L = blk / m;
    JT_REQUIRE(L * m == blk, JT_E_INVAL, "K-weighting: the chunk length must divide the 100 ms block");
    const int64_t nchunks = (n + L - 1) / L;
    // Everything that depends on (rate, L) alone is built once per handle and stays on the device: the homogeneous-response table
    // g[j][k] (the output at step j from the k-th unit state, no input: same statements as KW2_STEP), the powers F^L, F^2L, ... of the
    // homogeneous transition matrix (state s1, s2, t1, t2) until they vanish, the Gram matrix of a full chunk.  (It used to be rebuilt and
    // uploaded by every job: nine small copies per file.)
    jt_ctx::KwTab *T = nullptr;
    for (auto &t : h->kw_tab) if (t.rate == rate && t.L == L) T = &t;
    if (!T) {
        T = &h->kw_tab[h->kw_tab_next++ % 8];
        // an evicted slot's table may still be read by a sweep in flight on another stream of this handle (more than eight (rate, chunk
        // length) pairs on one pooled handle): the old buffer is parked, never overwritten
        T->dev.retire();
        T->rate = rate; T->L = L; T->g.assign((size_t)4 * L + 24 * 16, 0.0);
        for (int u = 0; u < 4; ++u) {
            double s1 = u == 0, s2 = u == 1, t1 = u == 2, t2 = u == 3;
            for (int64_t j = 0; j < L; ++j) {
                const double y = s1;                                   // fma(b0, 0, s1)
                s1 = std::fma(-k.a1, y, s2); s2 = std::fma(-k.a2, y, 0.0);
                const double zz = std::fma(k.c0, y, t1);
                t1 = std::fma(-k.d1, zz, std::fma(k.c1, y, t2)); t2 = std::fma(-k.d2, zz, k.c2 * y);
                T->g[(size_t)4 * j + u] = zz;
            }
        }
        double F[16] = {-k.a1, 1, 0, 0,   -k.a2, 0, 0, 0,   k.c1 - k.d1 * k.c0, 0, -k.d1, 1,   k.c2 - k.d2 * k.c0, 0, -k.d2, 0};
        double FL[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, Bq[16];
        std::memcpy(Bq, F, sizeof(F));
        for (int64_t e = L; e > 0; e >>= 1) { if (e & 1) mat4_mul(FL, Bq, FL); mat4_mul(Bq, Bq, Bq); }
        double cur[16]; std::memcpy(cur, FL, sizeof(cur));
        double *pw = T->g.data() + (size_t)4 * L;
        T->nterms = 1;
        for (int q = 0; q < 24; ++q) {
            double mx = 0; for (double v : cur) mx = std::max(mx, std::fabs(v));
            if (mx < 1e-19) break;
            std::memcpy(pw + 16 * q, cur, sizeof(cur)); T->nterms++;
            mat4_mul(cur, FL, cur);
        }
        std::memset(T->gram, 0, sizeof T->gram);
        for (int64_t j = 0; j < L; ++j) {
            int u = 0;
            for (int a = 0; a < 4; ++a) for (int b = a; b < 4; ++b, ++u) T->gram[u] += T->g[(size_t)4 * j + a] * T->g[(size_t)4 * j + b];
        }
        T->dev.ensure(T->g.size());
        JT_HIP(hipMemcpy(T->dev.p, T->g.data(), sizeof(double) * T->g.size(), hipMemcpyHostToDevice));
    }
    const double *d_pw = T->dev.p + (size_t)4 * L;
    const int nterms = T->nterms;
    // device scratch: zs | csum | cpeak | cross.  Each job gets its own region of d_scr0 (several may be in flight in one pass).
    const size_t need = 4 * (size_t)nchunks + 2 * (size_t)nchunks + 4 * (size_t)nchunks;
    double *base = ext ? ext->dev : h->kw_take(need);
    double *d_zs = base, *d_cs = base + 4 * (size_t)nchunks, *d_cross = d_cs + 2 * (size_t)nchunks;
    KwSweep SW; SW.k = k; SW.L = L; SW.nchunks = nchunks; SW.zs = d_zs; SW.csum = d_cs; SW.cpeak = d_cs + nchunks; SW.cross = d_cross;
    SW.pw = d_pw; SW.nterms = nterms; SW.two_sweeps = false;
#ifdef JT_AB
    SW.two_sweeps = h->opts.kw_two_sweeps;
#endif
    const int64_t nfull = n / blk;
    // block sums / peaks, (nfull + 1) entries each: all that travels to the host -- written by the kernel straight into pinned,
    // device-visible host memory (no copy to queue behind it)
    double *hc = ext ? ext->pin : h->pin.take<double>((size_t)(nfull + 1) * 2);
    double *d_out = hc;
    if (SW.two_sweeps) {
        sweep(SW);
        hipLaunchKernelGGL(k_kw_blocks, dim3((unsigned)((nfull + 1 + 63) / 64)), dim3(64), 0, st, d_cs, d_cs + nchunks, nchunks, m, nfull, d_out);
    } else {
        KwGram G; std::memcpy(G.f, T->gram, sizeof G.f); std::memset(G.t, 0, sizeof G.t);
        const int64_t tail = n - (nchunks - 1) * L;                        // samples of the last chunk (1 .. L)
        if (tail == L) std::memcpy(G.t, G.f, sizeof G.t);
        else for (int64_t j = 0; j < tail; ++j) {
            int u = 0;
            for (int a = 0; a < 4; ++a) for (int b = a; b < 4; ++b, ++u) G.t[u] += T->g[(size_t)4 * j + a] * T->g[(size_t)4 * j + b];
        }
        SW.gtab = T->dev.p; SW.gtab_host = T->g.data(); SW.tail = tail;
        sweep(SW);
        hipLaunchKernelGGL(k_kw_fixblocks, dim3((unsigned)((nfull + 1 + 63) / 64)), dim3(64), 0, st, (const double *)d_zs, d_pw, nterms, (const double *)d_cross, G,
                           (const double *)d_cs, (const double *)(d_cs + nchunks), nchunks, m, nfull, d_out);
    }
    job->hc = hc; job->nchunks = nchunks; job->nfull = nfull; job->m = m;
}

template <typename TIn>
static void kweight_enqueue(jt_ctx *h, const TIn *in, int64_t n, int rate, int blk, KwJob *job, hipStream_t st, const KwScratch *ext = nullptr)
{
    kweight_job(h, n, rate, blk, 0, job, st, ext, [&](const KwSweep &W) {
        const unsigned grid = (unsigned)((W.nchunks + LANES - 1) / LANES);
        if (W.two_sweeps) {
#ifdef JT_AB
            hipLaunchKernelGGL((k_kw<TIn, false>), dim3(grid), dim3(LANES), 0, st, in, n, W.L, W.k, (const double *)nullptr, (const double *)nullptr, 0,
                               W.zs, (double *)nullptr, (double *)nullptr, W.nchunks);
            hipLaunchKernelGGL((k_kw<TIn, true>), dim3(grid), dim3(LANES), 0, st, in, n, W.L, W.k, (const double *)W.zs, W.pw, W.nterms,
                               (double *)nullptr, W.csum, W.cpeak, W.nchunks);
#endif
            return;
        }
        hipLaunchKernelGGL((k_kw1<TIn>), dim3(grid), dim3(LANES), 0, st, in, n, W.L, W.k, W.gtab, W.zs, W.csum, W.cpeak, W.cross, W.nchunks);
    });
}
void jt_kweight_enqueue_f32(jt_ctx *h, const float *in, int64_t n, int rate, int blk, KwJob *job, hipStream_t s, const KwScratch *ext) { kweight_enqueue<float>(h, in, n, rate, blk, job, s, ext); }
void jt_kweight_enqueue_f64(jt_ctx *h, const double *in, int64_t n, int rate, int blk, KwJob *job, hipStream_t s, const KwScratch *ext) { kweight_enqueue<double>(h, in, n, rate, blk, job, s, ext); }
void jt_kweight_enqueue_sweep(jt_ctx *h, int64_t n, int rate, int blk, int64_t chunk_len, KwJob *job, hipStream_t s, const KwScratch *ext,
                              const std::function<void(const KwSweep &)> &sweep)
{
    kweight_job(h, n, rate, blk, chunk_len, job, s, ext, sweep);
}
void jt_kweight_finish(const KwJob *job, std::vector<double> &sums, std::vector<double> &peaks)
{
    const int64_t nfull = job->nfull;
    sums.assign(job->hc, job->hc + (size_t)nfull + 1); peaks.assign(job->hc + (size_t)nfull + 1, job->hc + 2 * ((size_t)nfull + 1));
}

// ------------------------------------------------------------------ agate -> acompressor -> deesser (double)
__device__ inline double hermite_interp(double x, double x0, double x1, double p0, double p1, double m0, double m1)
{
    double width = x1 - x0;
    double t = (x - x0) / width;
    m0 *= width; m1 *= width;
    double t2 = t * t, t3 = t2 * t;
    double ct0 = p0, ct1 = m0;
    double ct2 = -3 * p0 - 2 * m0 + 3 * p1 - m1;
    double ct3 = 2 * p0 + m0 - 2 * p1 + m1;
    return ct3 * t3 + ct2 * t2 + ct1 * t + ct0;
}

struct DynState {
    double g_slope, c_slope;
    double s1, s2, s3, ratioA, ratioB, iirA, iirB;
};

__device__ inline double dyn_step(DynState &st, double x, int64_t idx, const DynParams &p)
{
    double v = x;
    if (p.gate_on) {   // af_agate.c gate(): detection=rms, mode=downward
        double a = fabs(v); a *= a;
        st.g_slope += (a - st.g_slope) * (a > st.g_slope ? p.g_attack : p.g_release);
        double gain = 1.0;
        if (st.g_slope > 0.0 && st.g_slope < p.g_lin_knee_stop) {
            double slope = log(st.g_slope);
            double tratio = (fabs(p.g_ratio - 4294967296.0) < 1.0) ? 1000. : p.g_ratio;
            double g = (slope - p.g_thres) * tratio + p.g_thres;
            if (p.g_knee > 1. && slope > p.g_knee_start)
                g = hermite_interp(slope, p.g_knee_start, p.g_knee_stop,
                                   ((p.g_knee_start - p.g_thres) * tratio + p.g_thres), p.g_knee_stop, tratio, 1.);
            gain = fmax(p.g_range, exp(g - slope));
        }
        v = v * (1.0 * gain * p.g_makeup);
    }
    if (p.comp_on) {   // af_sidechaincompress.c compressor(): detection=rms, mode=downward
        double a = fabs(v); a *= a;
        st.c_slope += (a - st.c_slope) * (a > st.c_slope ? p.c_attack : p.c_release);
        double gain = 1.0;
        if (st.c_slope > 0.0 && st.c_slope > p.c_adj_knee_start) {
            double slope = log(st.c_slope) * 0.5;
            double g, delta;
            if (fabs(p.c_ratio - 4294967296.0) < 1.0) { g = p.c_thres; delta = 0.0; }
            else { g = (slope - p.c_thres) / p.c_ratio + p.c_thres; delta = 1.0 / p.c_ratio; }
            if (p.c_knee > 1.0 && slope < p.c_knee_stop)
                g = hermite_interp(slope, p.c_knee_start, p.c_knee_stop, p.c_knee_start, p.c_ckstop, 1.0, delta);
            gain = exp(g - slope);
        }
        v = v * 1.0 * (gain * p.c_makeup * p.c_mix + (1. - p.c_mix));
    }
    if (p.deess_on) {  // af_deesser.c filter_frame()
        double sample = v;
        st.s3 = st.s2; st.s2 = st.s1; st.s1 = sample;
        double m1 = (st.s1 - st.s2) * ((st.s1 - st.s2) / 1.3);
        double m2 = (st.s2 - st.s3) * ((st.s1 - st.s2) / 1.3);
        double sense = (m1 - m2) * ((m1 - m2) / 1.3);
        double attackspeed = 7.0 + sense * 1024;
        sense = 1.0 + p.d_intensity * p.d_intensity * sense;
        sense = fmin(sense, p.d_intensity);
        double recovery = 1.0 + (0.01 / sense);
        double offset = 1.0 - fabs(sample);
        if (idx & 1) {   // flip toggles every sample starting from 0: odd samples use the A state
            st.iirA = (st.iirA * (1.0 - (offset * p.d_iir))) + (sample * (offset * p.d_iir));
            if (st.ratioA < sense) st.ratioA = ((st.ratioA * attackspeed) + sense) / (attackspeed + 1.0);
            else st.ratioA = 1.0 + ((st.ratioA - 1.0) / recovery);
            st.ratioA = fmin(st.ratioA, p.d_maxdess);
            sample = st.iirA + ((sample - st.iirA) / st.ratioA);
        } else {
            st.iirB = (st.iirB * (1.0 - (offset * p.d_iir))) + (sample * (offset * p.d_iir));
            if (st.ratioB < sense) st.ratioB = ((st.ratioB * attackspeed) + sense) / (attackspeed + 1.0);
            else st.ratioB = 1.0 + ((st.ratioB - 1.0) / recovery);
            st.ratioB = fmin(st.ratioB, p.d_maxdess);
            sample = st.iirB + ((sample - st.iirB) / st.ratioB);
        }
        v = sample;
    }
    return v;
}

// The envelope followers are the only sequential part of agate / acompressor: s += (x^2 - s) * (x^2 > s ? att : rel).
// Stage 1 (k_follow_states): lane-serial over long chunks with a warm-up halo, follower ONLY (a few flops per sample),
//   records the follower state at every SC-sample boundary.
// Stage 2 (k_dyn_apply): one lane per SC-sample sub-chunk (hundreds of thousands of lanes), restarts the follower from
//   the recorded state and evaluates the log/exp gain curve -- the expensive part now runs fully parallel, no halo.
constexpr int SC = 256;

template <typename TIn>
__global__ void __launch_bounds__(64)
k_follow_states(const TIn *__restrict__ in, int64_t n, double att, double rel, int64_t chunk, int64_t halo,
                double *__restrict__ states, int64_t nchunks)
{
    __shared__ TIn tile[LANES][TW + 1];
    const int lane = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * LANES;
    const int nrows = (int)min((int64_t)LANES, nchunks - c0);
    const int64_t base0 = c0 * chunk - halo;
    const int64_t my_base = base0 + (int64_t)lane * chunk;
    double s = 0.0;
    const int64_t total = halo + chunk;            // multiples of SC (hence of TW)
    auto run_tile = [&](int64_t pos) {
        if (lane < nrows) {
            const int64_t t0 = my_base + pos;
            if (pos >= halo && ((pos - halo) % SC) == 0 && t0 < n) states[t0 / SC] = s;     // state BEFORE sample t0
            if (t0 >= 0 && t0 + TW <= n) {
                // the squares of 16 samples first (LDS reads and conversions in flight together), then the dependent chain on
                // registers only: left to the scheduler, every second sample waited for its own LDS read
#pragma unroll 1
                for (int j0 = 0; j0 < TW; j0 += 16) {
                    double a2[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) { double a = (double)tile[lane][j0 + j]; a = fabs(a); a2[j] = a * a; }
#pragma unroll
                    for (int j = 0; j < 16; ++j) s += (a2[j] - s) * (a2[j] > s ? att : rel);
                }
            } else if (t0 + TW > 0 && t0 < n) {
                for (int j = 0; j < TW; ++j) {
                    const int64_t idx = t0 + j;
                    if (idx >= 0 && idx < n) {
                        double a = (double)tile[lane][j]; a = fabs(a); a *= a;
                        s += (a - s) * (a > s ? att : rel);
                    }
                }
            }
        }
    };
    if (rows_vec_aligned(in, base0, chunk)) {
        // 16-byte row loads, two tiles in flight (one wave per CU: nobody else hides the HBM round trip); the first and the last
        // workgroup take this path too, their out-of-range rows simply load nothing
        TIn va[LANES], vb[LANES];
        rows_prefetch_vec<TIn>(va, in, base0, chunk, 0, tile_edge(n, base0, 0), nrows, lane);
        rows_prefetch_vec<TIn>(vb, in, base0, chunk, TW, tile_edge(n, base0, TW), nrows, lane);
        for (int64_t pos = 0; pos < total; pos += 2 * TW) {              // total is a multiple of SC = 4 TW
            rows_commit_vec<TIn, TIn>(tile, va, lane);
            __syncthreads();
            if (pos + 2 * TW < total) rows_prefetch_vec<TIn>(va, in, base0, chunk, pos + 2 * TW, tile_edge(n, base0, pos + 2 * TW), nrows, lane);
            run_tile(pos);
            __syncthreads();
            rows_commit_vec<TIn, TIn>(tile, vb, lane);
            __syncthreads();
            if (pos + 3 * TW < total) rows_prefetch_vec<TIn>(vb, in, base0, chunk, pos + 3 * TW, tile_edge(n, base0, pos + 3 * TW), nrows, lane);
            run_tile(pos + TW);
            __syncthreads();
        }
        return;
    }
    TIn v[LANES];
    rows_prefetch<TIn>(v, in, n, base0, chunk, 0, lane);
    for (int64_t pos = 0; pos < total; pos += TW) {
        rows_commit<TIn, TIn>(tile, v, n, base0, chunk, pos, lane, nrows);
        __syncthreads();
        if (pos + TW < total) rows_prefetch<TIn>(v, in, n, base0, chunk, pos + TW, lane);
        run_tile(pos);
        __syncthreads();
    }
}

// The same follower with the samples streamed straight into LDS (global_load_lds_dwordx4, gfx950): loads in flight cost no
// registers, so a lane can afford LONG chunks (few re-reads of the halo) and still have its HBM round trips covered -- with the
// register-staged tiles above, chunks of a whole halo leave 63 waves that each wait for their own loads.
// One wave per workgroup, lane = chunk.  A unit is 16 load instructions: instruction k moves, for every lane, the 16 bytes at
// byte offset 16 k of that lane's row segment to LDS [slot][k][lane] (the hardware writes lane l at base + 16 l), so a lane
// reads its own data back with conflict-free 16-byte reads.  8 slots of 16 KB; three units (48 instructions, the vmcnt budget)
// are outstanding while one is consumed.  Requires 16-byte aligned rows and 16 bytes of readable slack behind the signal.
template <typename TIn>
__global__ void __launch_bounds__(64)
k_follow_states_lds(const TIn *__restrict__ in, int64_t n, double att, double rel, int64_t chunk, int64_t halo,
                    double *__restrict__ states, int64_t nchunks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fs_ring[];
    constexpr int G = 16 / (int)sizeof(TIn);           // samples per 16-byte group
    constexpr int US = 16 * G;                         // samples per unit (64 f32 / 32 f64)
    constexpr int NS = 8, D = 3;
    const int lane = threadIdx.x;
    const int64_t c = (int64_t)blockIdx.x * LANES + lane;
    const bool active = c < nchunks;
    const int64_t row0 = c * chunk - halo;             // first sample of this lane's run (may be negative)
    const int64_t total = halo + chunk;                // multiple of SC, hence of US
    const int nu = (int)(total / US);
    auto issue = [&](int u) {
        unsigned char *slot = fs_ring + (size_t)(u % NS) * (16 * 1024);
        const int64_t g0 = row0 + (int64_t)u * US;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int64_t idx = g0 + k * G;
            if (!active || idx < 0 || idx >= n) idx = 0;                     // never consumed: any valid address
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + idx),
                                             (__attribute__((address_space(3))) void *)(slot + k * 1024), 16, 0, 0);
        }
    };
    for (int u = 0; u < D && u < nu; ++u) issue(u);
    double s = 0.0;
    for (int u = 0; u < nu; ++u) {
        if (u + D < nu) issue(u + D);
        // unit u is complete once at most the younger units' loads are outstanding (loads retire in order)
        const int left = nu - 1 - u;
        if (left >= D) __builtin_amdgcn_s_waitcnt(0x0F70 | (48 & 15) | ((48 >> 4) << 14));
        else if (left == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (32 & 15) | ((32 >> 4) << 14));
        else if (left == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (16 & 15) | ((16 >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_wave_barrier();
        const int64_t pos = (int64_t)u * US, t0 = row0 + pos;
        if (active) {
            if (pos >= halo && ((pos - halo) % SC) == 0 && t0 < n) states[t0 / SC] = s;     // state BEFORE sample t0
            const unsigned char *slot = fs_ring + (size_t)(u % NS) * (16 * 1024) + 16 * lane;
            typedef TIn vecT __attribute__((ext_vector_type(G)));
            double a2[US];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const vecT v = *reinterpret_cast<const vecT *>(slot + k * 1024);
#pragma unroll
                for (int e = 0; e < G; ++e) { double a = (double)v[e]; a = fabs(a); a2[k * G + e] = a * a; }
            }
            if (t0 >= 0 && t0 + US <= n) {
#pragma unroll
                for (int j = 0; j < US; ++j) s += (a2[j] - s) * (a2[j] > s ? att : rel);
            } else if (t0 + US > 0 && t0 < n) {
#pragma unroll
                for (int j = 0; j < US; ++j) { const int64_t idx = t0 + j; if (idx >= 0 && idx < n) s += (a2[j] - s) * (a2[j] > s ? att : rel); }
            }
        }
        __builtin_amdgcn_wave_barrier();               // the slot is re-targeted D units later, after this unit's reads were issued
    }
}

// The follower as ONE CONSUMER AND THREE PRODUCERS (round 4).  A launch of k_follow_states_lds lasts as long as ONE lane's serial run
// (halo + chunk = 54 k samples) whatever the file's length -- 1.6 ms (f32) / 2.2 ms (f64) for ten minutes and for an hour alike -- for
// two reasons found this round: (1) global_load_lds lands 16 KB per CU every 0.65 us (MI355X_MICROARCH.md, "ldsdma-fill": 25 GB/s per
// CU) and one workgroup per CU was streaming its 843 / 1686 units through it; (2) the one wave spent 32 cycles of issue a sample
// (convert, square, the recurrence's six instructions at four cycles per f64 instruction) in series with the recurrence's dependent
// chain.  Here the workgroup has four waves on four SIMDs.  Waves 1-3 load the rows with ordinary 16-byte vector loads, ROW-COALESCED
// (an instruction covers four rows x 256 contiguous bytes; unit u belongs to wave 1 + u % 3, two units in flight per wave in registers),
// convert and square them and leave |x|^2 as doubles in an LDS ring, transposed so that the consumer lane of a row reads its own
// values with conflict-free 16-byte reads; wave 0 runs nothing but the recurrence, sixteen values at a time from registers.  One
// s_barrier per unit.  Same statements on the same values in the same order: bit-identical states.
template <typename TIn>
__global__ void __launch_bounds__(256)
k_follow_states_pc(const TIn *__restrict__ in, int64_t n, double att, double rel, int64_t chunk, int64_t halo,
                   double *__restrict__ states, int64_t nchunks, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fs_ring[];
    constexpr int G = 16 / (int)sizeof(TIn);           // samples per 16-byte group
    constexpr int US = 16 * G;                         // samples per unit and row (64 f32 / 32 f64): 256 bytes
    constexpr int NP = 3;                              // producers = slots of the squared ring
    constexpr int PL = 1024 + 16;                      // bytes of one plane [64 rows] of double2 (+ 16: the producers' writes spread over the banks)
    constexpr int SLOT = US / 2 * PL;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t c0 = (int64_t)blockIdx.x * LANES;
    const int64_t total = halo + chunk;                // multiple of SC, hence of US
    const int nu = (int)(total / US);
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef TIn vecT __attribute__((ext_vector_type(G)));
    if (wave != 0) {
        const int sub = lane >> 4, seg = lane & 15;    // instruction i covers rows 4 i + sub, this lane the 16 bytes at sample seg * G of the unit
        auto producer = [&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            vecT ra[16], rb[16];
            // first sample of this lane's 16 bytes in unit 0 of each of its sixteen rows (kept in registers: the per-load address is one
            // 64-bit add and a range test instead of a 64-bit multiply); a row past the last chunk never passes the range test
            int64_t base[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t cc = c0 + 4 * i + sub;
                base[i] = cc < nchunks ? cc * chunk - halo + seg * G : n;
            }
            auto load = [&](vecT (&r)[16], int u) {
                const int64_t uo = (int64_t)u * US;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    int64_t idx = base[i] + uo;
                    if (idx < 0 || idx >= n) idx = 0;                                // never consumed: any valid address
                    r[i] = *reinterpret_cast<const vecT *>(in + idx);
                }
            };
            auto convert = [&](const vecT (&r)[16], int u) {
                unsigned char *slot = fs_ring + (size_t)(u % NP) * SLOT;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = 4 * i + sub;
#pragma unroll
                    for (int e = 0; e < G; e += 2) {
                        double a = (double)r[i][e], b = (double)r[i][e + 1];
                        a = fabs(a); b = fabs(b);
                        d2 w; w.x = a * a; w.y = b * b;
                        *reinterpret_cast<d2 *>(slot + (size_t)((seg * G + e) / 2) * PL + 16 * row) = w;
                    }
                }
            };
            if (Q < nu) load(ra, Q);
            if (Q + NP < nu) load(rb, Q + NP);
            for (int u6 = 0; u6 < nu; u6 += 2 * NP) {
#pragma unroll
                for (int d = 0; d < 2 * NP; ++d) {
                    const int u = u6 + d;
                    if (u >= nu) break;
                    if (d % NP == Q && !(dbg & 1)) {
                        // the older of this wave's two units in flight is complete once only the younger one's sixteen loads are outstanding
                        if (u + NP < nu) __builtin_amdgcn_s_waitcnt(0x0F70 | (16 & 15) | ((16 >> 4) << 14));
                        else __builtin_amdgcn_s_waitcnt(0x0F70);
                        if (d < NP) { convert(ra, u); if (u + 2 * NP < nu) load(ra, u + 2 * NP); }
                        else { convert(rb, u); if (u + 2 * NP < nu) load(rb, u + 2 * NP); }
                    }
                    __syncthreads();   // unit u's squares are in slot u % 3; the consumer finished that slot's previous tenant three barriers ago
                }
            }
        };
        if (wave == 1) producer(std::integral_constant<int, 0>{});
        else if (wave == 2) producer(std::integral_constant<int, 1>{});
        else producer(std::integral_constant<int, 2>{});
        return;
    }
    const int64_t c = c0 + lane;
    const bool active = c < nchunks;
    const int64_t row0 = c * chunk - halo;             // first sample of this lane's run (may be negative)
    double s = 0.0;
    for (int u = 0; u < nu; ++u) {
        __syncthreads();
        const int64_t pos = (int64_t)u * US, t0 = row0 + pos;
        if (!active) continue;
        if (pos >= halo && ((pos - halo) % SC) == 0 && t0 < n) states[t0 / SC] = s;     // state BEFORE sample t0
        const unsigned char *src = fs_ring + (size_t)(u % NP) * SLOT + 16 * lane;
        const bool inside = t0 >= 0 && t0 + US <= n;
        if (!inside && !(t0 + US > 0 && t0 < n)) continue;
        if (dbg & 2) continue;
#pragma unroll
        for (int b = 0; b < US; b += 16) {
            double a2[16];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const d2 w = *reinterpret_cast<const d2 *>(src + (size_t)(b / 2 + r) * PL);
                a2[2 * r] = w.x; a2[2 * r + 1] = w.y;
            }
            // s += (a - s) * (a > s ? att : rel), with the select moved behind the multiplication: both products start as soon as the
            // difference exists and the comparison runs beside them, so a step's dependent chain is sub -> mul -> select -> add instead
            // of compare -> select -> mul -> add behind the subtraction.  The same product, the same sum: bit-identical.  (The empty
            // asm keeps the compiler from folding the two products back into one product of the selected coefficient.)
            // Round 6: one product instead of two, the coefficient selected by the SIGN OF THE DIFFERENCE: a > s <=> d > 0 (the difference of
            // two finite doubles is zero only when they are equal and carries the comparison's sign), and d > 0 is a 32-bit signed compare
            // of its high word (> 0: positive and not +0; a positive d whose high word is zero is a subnormal below 2^-1042, which no
            // difference of squared samples against a state that started at zero can be).  A lone wave is ISSUE-bound and an f64
            // instruction costs 9 ticks against 5 for a 32-bit one (profiles/r02_gfx950_op_costs.txt): sub 9 + cmp 5 + 2 x cndmask 10 + mul 9
            // + add 9 = 42 ticks a sample where "both products, then select" was sub + 2 mul + cmp_f64 + 2 cndmask + add = 55.  The same
            // product of the same two doubles, the same sum: bit-identical states (the A/B build's one-wave kernel is held to it).
            auto step = [&](double a) {
                const double d = a - s;
                const double c = __double2hiint(d) > 0 ? att : rel;
                s += d * c;
            };
            if (inside) {
#pragma unroll
                for (int j = 0; j < 16; ++j) step(a2[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) { const int64_t idx = t0 + b + j; if (idx >= 0 && idx < n) step(a2[j]); }
            }
        }
    }
}

// MODE 0: agate (af_agate.c gate()), MODE 1: acompressor (af_sidechaincompress.c compressor()); detection = rms
template <int MODE, typename TIn, typename TOut>
__global__ void __launch_bounds__(64)
k_dyn_apply(const TIn *__restrict__ in, TOut *__restrict__ out, int64_t n, const double *__restrict__ states, DynParams p, int64_t nsub)
{
    __shared__ double tile[LANES][TW + 1];
    const int lane = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * LANES;
    const int nrows = (int)min((int64_t)LANES, nsub - c0);
    const int64_t base0 = c0 * SC;
    double s = lane < nrows ? states[c0 + lane] : 0.0;
    for (int pos = 0; pos < SC; pos += TW) {
        tile_load<TIn, double>(tile, in, n, base0, SC, pos, lane, nrows);
        __syncthreads();
        if (lane < nrows) {
            // (samples past the end of the stream are staged as zeros and never stored: no index test, so that the iterations can
            // be unrolled and their gain curves -- independent of each other -- interleave)
#pragma unroll 4
            for (int j = 0; j < TW; ++j) {
                const double v = tile[lane][j];
                double a = fabs(v); a *= a;
                double gain = 1.0, y;
                if (MODE == 0) {
                    s += (a - s) * (a > s ? p.g_attack : p.g_release);
                    if (s > 0.0 && s < p.g_lin_knee_stop) {
                        double slope = log(s);
                        double tratio = (fabs(p.g_ratio - 4294967296.0) < 1.0) ? 1000. : p.g_ratio;
                        double g = (slope - p.g_thres) * tratio + p.g_thres;
                        if (p.g_knee > 1. && slope > p.g_knee_start)
                            g = hermite_interp(slope, p.g_knee_start, p.g_knee_stop,
                                               ((p.g_knee_start - p.g_thres) * tratio + p.g_thres), p.g_knee_stop, tratio, 1.);
                        gain = fmax(p.g_range, exp(g - slope));
                    }
                    y = v * (1.0 * gain * p.g_makeup);
                } else {
                    s += (a - s) * (a > s ? p.c_attack : p.c_release);
                    if (s > 0.0 && s > p.c_adj_knee_start) {
                        double slope = log(s) * 0.5;
                        double g, delta;
                        if (fabs(p.c_ratio - 4294967296.0) < 1.0) { g = p.c_thres; delta = 0.0; }
                        else { g = (slope - p.c_thres) / p.c_ratio + p.c_thres; delta = 1.0 / p.c_ratio; }
                        if (p.c_knee > 1.0 && slope < p.c_knee_stop)
                            g = hermite_interp(slope, p.c_knee_start, p.c_knee_stop, p.c_knee_start, p.c_ckstop, 1.0, delta);
                        gain = exp(g - slope);
                    }
                    y = v * 1.0 * (gain * p.c_makeup * p.c_mix + (1. - p.c_mix));
                }
                tile[lane][j] = y;
            }
        }
        __syncthreads();
        tile_store<TOut, double>(tile, out, n, base0, SC, pos, lane, nrows, 0, SC);
        __syncthreads();
    }
}

// de-esser alone (lane-serial with halo; no transcendental functions)
template <typename TIn>
__global__ void __launch_bounds__(64)
k_deesser(const TIn *__restrict__ in, float *__restrict__ out, int64_t n, int64_t chunk, int64_t halo, DynParams p, int64_t nchunks)
{
    __shared__ double tile[LANES][TW + 1];
    const int lane = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * LANES;
    const int nrows = (int)min((int64_t)LANES, nchunks - c0);
    const int64_t base0 = c0 * chunk - halo;
    const int64_t my_base = base0 + (int64_t)lane * chunk;
    DynState st; st.g_slope = 0; st.c_slope = 0; st.s1 = st.s2 = st.s3 = 0; st.ratioA = st.ratioB = 1.0; st.iirA = st.iirB = 0;
    DynParams q = p; q.gate_on = 0; q.comp_on = 0; q.deess_on = 1;
    const int64_t total = halo + chunk;
    for (int64_t pos = 0; pos < total; pos += TW) {
        tile_load<TIn, double>(tile, in, n, base0, chunk, pos, lane, nrows);
        __syncthreads();
        if (lane < nrows) {
            for (int j = 0; j < TW; ++j) {
                int64_t idx = my_base + pos + j;
                if (idx < 0 || idx >= n) continue;
                tile[lane][j] = dyn_step(st, tile[lane][j], idx, q);
            }
        }
        __syncthreads();
        tile_store<float, double>(tile, out, n, base0, chunk, pos, lane, nrows, halo, chunk);
        __syncthreads();
    }
}

template <typename TIn>
static void run_follow(const TIn *in, int64_t n, double att, double rel, double *states, hipStream_t s, bool in_has_slack, const JtOpts &o)
{
    double rho = std::min(att, rel);
    int64_t halo = 4096;
    if (rho < 1.0 && rho > 0.0) halo = std::max<int64_t>(halo, (int64_t)std::ceil(18.0 / rho));   // 18 time constants ~ 1.5e-8
    halo = (halo + SC - 1) / SC * SC;
    // chunk = halo / 8: a lane runs halo + chunk samples, so shorter chunks cut the serial length (halo/4 -> +1 ms, halo/8 best,
    // halo/16 no better).  Measured with the chain or the staging compiled out: the dependent chain itself is 27 clk per sample
    // (0.5 ms of each launch); the rest is the 9x re-read of the signal, 6.2 GB (f32) / 12.4 GB (f64) at ~4.5 TB/s -- the file
    // is larger than the Infinity Cache, so the re-reads come from HBM.  Longer chunks (fewer re-reads) leave too few waves to
    // cover the load latency with two tiles in flight and lose more than they save.
    if (in_has_slack && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && !JT_AB_ON(o.follow_tiles)) {
        // LDS-streamed variant: chunks of a quarter halo (5x instead of 9x re-read; 250 waves for an hour of audio).  The vmcnt
        // counter (63) caps a wave at 48 KB in flight, about one HBM round trip of the f32 follower's appetite and half of the f64
        // one's, so longer chunks (fewer waves) start waiting for memory again: halo/2 and halo measured 1-2 ms worse.
        // chunk = halo / div: the run is halo + chunk long, the signal is read 1 + div times.  Files up to ~20 minutes are small enough
        // for 9 x (their traffic stays under a millisecond of HBM time); longer ones keep 5 x
        const int div = JT_AB_ON(o.follow_div > 0) ? o.follow_div : (n <= ((int64_t)1 << 26) ? 8 : 4);
        const int64_t lchunk = std::max<int64_t>(1024, (halo / div + SC - 1) / SC * SC);
        const int64_t lnch = (n + lchunk - 1) / lchunk;
        if (!JT_AB_ON(o.follow_one_wave)) {
            // one consumer + three producers (k_follow_states_pc): three slots of squared doubles
            constexpr int US = 16 * (16 / (int)sizeof(TIn));
            const int smem = 3 * (US / 2) * (1024 + 16);
            auto k = k_follow_states_pc<TIn>;
            JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
            hipLaunchKernelGGL(k, dim3((unsigned)((lnch + LANES - 1) / LANES)), dim3(4 * LANES), smem, s, in, n, att, rel, lchunk, halo, states, lnch, JT_AB_ON(o.follow_dbg) ? o.follow_dbg : 0);
            return;
        }
#ifdef JT_AB
        auto k = k_follow_states_lds<TIn>;
        JT_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        hipLaunchKernelGGL(k, dim3((unsigned)((lnch + LANES - 1) / LANES)), dim3(LANES), 128 * 1024, s, in, n, att, rel, lchunk, halo, states, lnch);
#endif
        return;
    }
    int64_t chunk = std::max<int64_t>(1024, (halo / 8 + SC - 1) / SC * SC);
    int64_t nchunks = (n + chunk - 1) / chunk;
    hipLaunchKernelGGL((k_follow_states<TIn>), dim3((unsigned)((nchunks + LANES - 1) / LANES)), dim3(LANES), 0, s, in, n, att, rel, chunk, halo,
                       states, nchunks);
}

// scratch: states[(n/SC)+2] doubles, tmp64[n] doubles (gate output feeding the compressor), tmp64b[n] when the de-esser runs
void launch_dynamics(const float *in, float *out_f32, double *tmp64, double *tmp64b, double *states, int64_t n, const DynParams &d, hipStream_t s,
                     const JtOpts &o, bool in_has_slack)
{
    if (n <= 0) return;
    const int64_t nsub = (n + SC - 1) / SC;
    const unsigned agrid = (unsigned)((nsub + LANES - 1) / LANES);
    const bool de = d.deess_on != 0;
    // stage outputs: gate -> tmp64 (if anything follows) ; comp -> tmp64b (if deesser follows) else f32 out
    const double *cur64 = nullptr; bool have64 = false;
    if (d.gate_on) {
        run_follow<float>(in, n, d.g_attack, d.g_release, states, s, in_has_slack, o);
        if (d.comp_on || de) { hipLaunchKernelGGL((k_dyn_apply<0, float, double>), dim3(agrid), dim3(LANES), 0, s, in, tmp64, n, states, d, nsub); cur64 = tmp64; have64 = true; }
        else { hipLaunchKernelGGL((k_dyn_apply<0, float, float>), dim3(agrid), dim3(LANES), 0, s, in, out_f32, n, states, d, nsub); return; }
    }
    if (d.comp_on) {
        if (have64) run_follow<double>(cur64, n, d.c_attack, d.c_release, states, s, in_has_slack, o);      // (tmp64 is allocated with the same slack)
        else run_follow<float>(in, n, d.c_attack, d.c_release, states, s, in_has_slack, o);
        if (de) {
            if (have64) hipLaunchKernelGGL((k_dyn_apply<1, double, double>), dim3(agrid), dim3(LANES), 0, s, cur64, tmp64b, n, states, d, nsub);
            else hipLaunchKernelGGL((k_dyn_apply<1, float, double>), dim3(agrid), dim3(LANES), 0, s, in, tmp64b, n, states, d, nsub);
            cur64 = tmp64b; have64 = true;
        } else {
            if (have64) hipLaunchKernelGGL((k_dyn_apply<1, double, float>), dim3(agrid), dim3(LANES), 0, s, cur64, out_f32, n, states, d, nsub);
            else hipLaunchKernelGGL((k_dyn_apply<1, float, float>), dim3(agrid), dim3(LANES), 0, s, in, out_f32, n, states, d, nsub);
            return;
        }
    }
    if (de) {
        // a lane runs halo + chunk samples of ~85 f64 instructions each and one wave nearly saturates a SIMD's f64 issue: the launch
        // lasts as long as one lane's run while fewer than 1024 waves exist.  chunk 8192 (330 waves for an hour): 10.2 ms, 4096: 8.6,
        // 3072: 8.0, 2048 (1320 waves, 9x redundant): 15.0.  The halo is what exactness needs (tools/deess_halo.py: against a halo of 262144,
        // 16384 is bit-identical on speech, gated hiss, an 8 kHz tone and full-scale noise at every intensity; 8192 is off by one f32 ulp on
        // full-scale noise at intensity 0.2, 4096 by 1e-5)
        const int64_t halo = JT_AB_ON(o.deess_halo > 0) ? o.deess_halo : 16384, chunk = JT_AB_ON(o.deess_chunk > 0) ? o.deess_chunk : 3072;
        const int64_t nchunks = (n + chunk - 1) / chunk;
        const unsigned g = (unsigned)((nchunks + LANES - 1) / LANES);
        if (have64) hipLaunchKernelGGL((k_deesser<double>), dim3(g), dim3(LANES), 0, s, cur64, out_f32, n, chunk, halo, d, nchunks);
        else hipLaunchKernelGGL((k_deesser<float>), dim3(g), dim3(LANES), 0, s, in, out_f32, n, chunk, halo, d, nchunks);
    }
}

// ------------------------------------------------------------------ look-ahead limiter (af_alimiter.c)
// The limiter's first sweep, fused: per-256-sample maxima of |in| (clean points, hot segments) and out = in * gain, the
// limiter's output wherever it stays at rest.  One wave per block of 256 samples (four per lane as two 16-byte loads), eight
// blocks per wave, no LDS.
// TO16 (the brickwall at the end of Pass 4, whose output nothing reads but the s16 conversion): the product leaves as what
// k_f64_to_s16(round_via_float = 1) makes of it -- the float and the s16 -- instead of as a double that a further sweep converts
__device__ __forceinline__ void lim_emit16(double v, int16_t *o16, float *o32, int64_t i)
{
    const float f = (float)v;
    o32[i] = f;
    double r = rint((double)f * 32768.0);
    r = r < -32768.0 ? -32768.0 : (r > 32767.0 ? 32767.0 : r);
    o16[i] = (int16_t)r;
}
template <bool TO16>
__global__ void __launch_bounds__(256)
k_absmax_copy_f64(const double *__restrict__ in, double *__restrict__ out, int64_t n, double gain, double *__restrict__ out_max, int64_t nblk,
                  int16_t *__restrict__ o16, float *__restrict__ o32)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    auto put2 = [&](int64_t i, double a, double b) {
        if (!TO16) { *reinterpret_cast<double2 *>(out + i) = double2{a, b}; return; }
        const float fa = (float)a, fb = (float)b;
        *reinterpret_cast<float2 *>(o32 + i) = float2{fa, fb};
        double ra = rint((double)fa * 32768.0), rb = rint((double)fb * 32768.0);
        ra = ra < -32768.0 ? -32768.0 : (ra > 32767.0 ? 32767.0 : ra); rb = rb < -32768.0 ? -32768.0 : (rb > 32767.0 ? 32767.0 : rb);
        *reinterpret_cast<short2 *>(o16 + i) = short2{(short)(int16_t)ra, (short)(int16_t)rb};
    };
    if ((w * 8 + 8) * 256 <= n) {
        // a wave's eight blocks, all sixteen loads in flight before the first is used (one block at a time left a wave with two loads in
        // flight and the sweep at 3.3 TB/s)
        const int64_t base0 = w * 8 * 256;
        double2 v[16];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[2 * q] = *reinterpret_cast<const double2 *>(in + base0 + q * 256 + 2 * lane);
            v[2 * q + 1] = *reinterpret_cast<const double2 *>(in + base0 + q * 256 + 128 + 2 * lane);
        }
        double mq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            mq[q] = fmax(fmax(fabs(v[2 * q].x), fabs(v[2 * q].y)), fmax(fabs(v[2 * q + 1].x), fabs(v[2 * q + 1].y)));
            put2(base0 + q * 256 + 2 * lane, v[2 * q].x * gain, v[2 * q].y * gain);
            put2(base0 + q * 256 + 128 + 2 * lane, v[2 * q + 1].x * gain, v[2 * q + 1].y * gain);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) mq[q] = fmax(mq[q], __shfl_down(mq[q], off, 64));
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) out_max[w * 8 + q] = mq[q];
        }
        return;
    }
    for (int q = 0; q < 8; ++q) {
        const int64_t b = w * 8 + q;
        if (b >= nblk) return;
        const int64_t base = b * 256;
        double m = 0.0;
        if (base + 256 <= n) {
            const double2 v0 = *reinterpret_cast<const double2 *>(in + base + 2 * lane);
            const double2 v1 = *reinterpret_cast<const double2 *>(in + base + 128 + 2 * lane);
            m = fmax(fmax(fabs(v0.x), fabs(v0.y)), fmax(fabs(v1.x), fabs(v1.y)));
            put2(base + 2 * lane, v0.x * gain, v0.y * gain);
            put2(base + 128 + 2 * lane, v1.x * gain, v1.y * gain);
        } else {
            for (int64_t i = base + lane; i < n; i += 64) {
                const double v = in[i]; m = fmax(m, fabs(v));
                if (TO16) lim_emit16(v * gain, o16, o32, i); else out[i] = v * gain;
            }
        }
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
        if (lane == 0) out_max[b] = m;
    }
}
// The same sweep fed by the s16 Pass-2 output (the limiter prefix of Pass 3 / Pass 4: volume -> alimiter): the s16 -> dbl conversion of
// k_s16_to_f64 (same arithmetic, incl. the float-precision volume stage) is done here, `out` = converted * gain; `conv` (may be null)
// receives the converted signal for a limiter kernel that reads doubles -- k_limiter_wave converts the few samples of its hot segments
// from the s16 source itself, so the prefix no longer writes 8 bytes per sample that 1.5 % of the samples are read back from.
__global__ void __launch_bounds__(256)
k_absmax_conv_s16(const int16_t *__restrict__ in, double *__restrict__ conv, double *__restrict__ out, int64_t n, double vol, int vol_in_float,
                  double gain, double *__restrict__ out_max, int64_t nblk)
{
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    auto cv = [&](int16_t x) -> double {
        if (vol_in_float) { float v = (float)x * (1.0f / 32768.0f); v = v * (float)vol; return (double)v; }
        return (double)x * (1.0 / 32768.0) * vol;
    };
    if ((w * 8 + 8) * 256 <= n) {
        // (as k_absmax_copy_f64: the wave's eight loads first)
        // Round 6: a lane takes samples 2 l, 2 l + 1 of each HALF of a 256-sample block (as k_absmax_copy_f64 does), so that every store
        // instruction of the wave covers 1024 CONTIGUOUS bytes; with four consecutive samples per lane each of the two 16-byte stores
        // touched every other 16 bytes of 2 KB -- half-written lines twice over, and the sweep sat at 85 % stalled on its stores
        const int64_t base0 = w * 8 * 256;
        short2 xa[8], xb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            xa[q] = *reinterpret_cast<const short2 *>(in + base0 + q * 256 + 2 * lane);
            xb[q] = *reinterpret_cast<const short2 *>(in + base0 + q * 256 + 128 + 2 * lane);
        }
        double mq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double v0 = cv(xa[q].x), v1 = cv(xa[q].y), v2 = cv(xb[q].x), v3 = cv(xb[q].y);
            mq[q] = fmax(fmax(fabs(v0), fabs(v1)), fmax(fabs(v2), fabs(v3)));
            const int64_t o = base0 + q * 256 + 2 * lane;
            if (conv) {
                *reinterpret_cast<double2 *>(conv + o) = double2{v0, v1};
                *reinterpret_cast<double2 *>(conv + o + 128) = double2{v2, v3};
            }
            *reinterpret_cast<double2 *>(out + o) = double2{v0 * gain, v1 * gain};
            *reinterpret_cast<double2 *>(out + o + 128) = double2{v2 * gain, v3 * gain};
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) mq[q] = fmax(mq[q], __shfl_down(mq[q], off, 64));
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) out_max[w * 8 + q] = mq[q];
        }
        return;
    }
    for (int q = 0; q < 8; ++q) {
        const int64_t b = w * 8 + q;
        if (b >= nblk) return;
        const int64_t base = b * 256;
        double m = 0.0;
        if (base + 256 <= n) {
            const short4 x = *reinterpret_cast<const short4 *>(in + base + 4 * lane);           // (base is a multiple of 256: 8-byte aligned)
            const double v0 = cv(x.x), v1 = cv(x.y), v2 = cv(x.z), v3 = cv(x.w);
            m = fmax(fmax(fabs(v0), fabs(v1)), fmax(fabs(v2), fabs(v3)));
            if (conv) {
                *reinterpret_cast<double2 *>(conv + base + 4 * lane) = double2{v0, v1};
                *reinterpret_cast<double2 *>(conv + base + 4 * lane + 2) = double2{v2, v3};
            }
            *reinterpret_cast<double2 *>(out + base + 4 * lane) = double2{v0 * gain, v1 * gain};
            *reinterpret_cast<double2 *>(out + base + 4 * lane + 2) = double2{v2 * gain, v3 * gain};
        } else {
            for (int64_t i = base + lane; i < n; i += 64) { const double v = cv(in[i]); m = fmax(m, fabs(v)); if (conv) conv[i] = v; out[i] = v * gain; }
        }
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
        if (lane == 0) out_max[b] = m;
    }
}
void launch_absmax_conv_s16(const int16_t *in, double *conv, double *out, int64_t n, double vol, int vol_in_float, double gain, double *out_max,
                            int64_t nblk, hipStream_t s)
{
    if (nblk <= 0) return;
    hipLaunchKernelGGL(k_absmax_conv_s16, dim3((unsigned)((nblk + 31) / 32)), dim3(256), 0, s, in, conv, out, n, vol, vol_in_float, gain, out_max, nblk);
}
void launch_absmax_copy_f64(const double *in, double *out, int64_t n, double gain, double *out_max, int64_t nblk, hipStream_t s, const LimOut16 *o16)
{
    if (nblk <= 0) return;
    if (o16) hipLaunchKernelGGL(k_absmax_copy_f64<true>, dim3((unsigned)((nblk + 31) / 32)), dim3(256), 0, s, in, out, n, gain, out_max, nblk, o16->s16, o16->f32);
    else hipLaunchKernelGGL(k_absmax_copy_f64<false>, dim3((unsigned)((nblk + 31) / 32)), dim3(256), 0, s, in, out, n, gain, out_max, nblk, (int16_t *)nullptr, (float *)nullptr);
}

// Segment starts ("clean points") chosen on the device.  A block b is clean when the `need` blocks before it are all at or
// under the limit (after in_gain): no peak is pending, att has recovered to 1 and delta is 0 there, so a lane can start the
// state machine from its initial state.  One candidate per `target`-block stride: cand[k] = first clean block in
// [k*target, (k+1)*target), or -1.  cand[0] = 0 (the stream start is clean by definition).
__global__ void k_lim_bounds(const double *__restrict__ mx, int64_t nblk, int need, int target, double g, double limit,
                             int64_t *__restrict__ cand, int64_t ntargets)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ntargets) return;
    if (k == 0) { cand[0] = 0; return; }
    const int64_t t0 = k * target;
    // consecutive quiet blocks immediately before t0 (capped at need; never counts past block 0, as the sequential counter)
    // (the same counts as the walk `for (b = t0 - 1; ...) { if hot break; quiet++; }`, with the loads of a group of eight issued together:
    // written with the break, every load waited for the one before it -- 27 dependent L2 round trips per thread, 0.2-0.6 ms per launch)
    int64_t quiet = need;
    for (int i0 = ((need + 7) & ~7) - 8; i0 >= 0; i0 -= 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int64_t b = t0 - 1 - (i0 + u); v[u] = (i0 + u < need && b >= 0) ? mx[b] : 0.0; }
#pragma unroll
        for (int u = 7; u >= 0; --u) { const int64_t b = t0 - 1 - (i0 + u); if (i0 + u < need && (b < 0 || v[u] * g > limit)) quiet = i0 + u; }
    }
    int64_t found = -1;
    {
        // target <= 8 blocks ahead
        double v[8]; const int64_t lim = min(t0 + (int64_t)target, nblk);
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (u < target && t0 + u < lim) ? mx[t0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (found < 0 && u < target && t0 + u < lim) {
                if (quiet >= need) found = t0 + u;
                else if (v[u] * g > limit) quiet = 0;
                else quiet++;
            }
        }
    }
    cand[k] = found;
}

// One lane per segment [lo, hi) between consecutive clean points; every segment starts at a provably clean limiter state
// (att = 1, delta = 0, empty peak list), so the sequential FFmpeg state machine is reproduced exactly.
// The ring buffer of af_alimiter.c is replaced by absolute sample indices: buffer[pos'] of the ring that
// holds sample a is in[a]; nextpos[] stores absolute indices; ring distances become index differences.
// The two sample streams a lane walks (the sample entering the look-ahead buffer and the one leaving it) are staged
// through [64][65] LDS tiles with row-coalesced loads; outputs leave through the same tile.
__global__ void __launch_bounds__(64)
k_limiter_f64(const double *__restrict__ in, double *__restrict__ out, int64_t n, int sr, double limit, int B,
              double release, double asc_coeff, const int64_t *__restrict__ cand, int64_t ntargets, int blk, double in_gain,
              double *__restrict__ sdelta, int64_t *__restrict__ spos, const double *__restrict__ block_max, int64_t nblk, double *__restrict__ slp)
{
    __shared__ double tx[LANES][TW + 1];
    __shared__ double te[LANES][TW + 1];
    __shared__ int64_t s_lo[LANES], s_hi[LANES], s_len[LANES];
    __shared__ int act[LANES];
    const int lane = threadIdx.x;
    const int64_t c = (int64_t)blockIdx.x * LANES + lane;
    int64_t lo = -1, hi = -1;
    if (c < ntargets && cand[c] >= 0) {
        lo = cand[c] * blk;
        hi = n;
        for (int64_t q = c + 1; q < ntargets; ++q) { const int64_t v = cand[q]; if (v >= 0) { hi = v * blk; break; } }
        if (lo >= n) { lo = -1; hi = -1; }
    }
    if (lo >= 0) {
        // A segment starts at rest; if nothing it pushes (its own samples and the B-1 it looks ahead into the next segment) is above
        // the limit it stays at rest throughout: att == 1, every output is in * in_gain, which k_absmax_copy_f64 has already written to `out`.
        // Only the segments with a block above the limit run the state machine (and overwrite their range).
        const double g = fabs(in_gain);
        const int64_t b0 = lo / blk, b1 = min(nblk - 1, (hi + B - 2) / blk);
        bool hot = false;
        for (int64_t b = b0; b <= b1; ++b) hot |= block_max[b] * g > limit;
        if (!hot) { lo = -1; hi = -1; }
    }
    s_lo[lane] = lo; s_hi[lane] = hi;
    const bool active = lo >= 0;
    // iterations i = lo .. hi + B - 2: iteration i pushes in[i] (zero past the end) and emits sample i-(B-1)
    const int64_t my_len = active ? (hi - lo) + B - 1 : 0;
    int64_t max_len = my_len;
    for (int o = 32; o > 0; o >>= 1) max_len = max(max_len, __shfl_xor(max_len, o, 64));
    if (max_len == 0) return;                 // (one wave per workgroup: uniform) every segment of this workgroup stays at rest
    // only the rows that run the state machine are staged (a workgroup usually has one or two of its 64 candidates hot: staging all
    // 64 rows of every tile made this kernel read ~50 times what it needed)
    s_len[lane] = my_len;
    const unsigned long long hot_mask = __ballot(active);
    const int nact = __popcll(hot_mask);
    if (active) act[__popcll(hot_mask & ((1ull << lane) - 1ull))] = lane;
    __syncthreads();
    double *nextdelta = sdelta + (size_t)c * B;
    int64_t *nextpos = spos + (size_t)c * B;
    // limit / |pending peak| beside every list entry: the filter recomputes that quotient from its sample buffer for every entry it
    // compares a new peak with (a division per entry and sample above the limit); it is the same quotient, of the same operands, that
    // was formed when the entry's sample was the new peak
    double *nextlp = slp + (size_t)c * B;
    if (active) nextpos[0] = -1;      // list entries are always written (with their -1 terminator) before they are read
    double att = 1.0, delta = 0.0, asc = 0.0; int asc_c = 0;
    int nextiter = 0, nextlen = 0;
    // ring indices: x % B for 0 <= x < 2 B (nextiter < B, nextlen <= B) without the integer division a run-time B costs per use
    auto wrapB = [&](int x) -> int { return x >= B ? x - B : x; };
    for (int64_t pos = 0; pos < max_len; pos += TW) {
        // stage: row r, column = lane
        for (int a0 = 0; a0 < nact; a0 += 8) {
            double vx[8], ve[8]; int rr[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                rr[q] = a0 + q < nact ? act[a0 + q] : -1;
                if (rr[q] >= 0 && pos >= s_len[rr[q]]) rr[q] = -1;            // that segment is finished
                vx[q] = 0.0; ve[q] = 0.0;
                if (rr[q] < 0) continue;                                       // (uniform)
                const int64_t rl = s_lo[rr[q]];
                const int64_t ix = rl + pos + lane, ie = ix - (B - 1);
                const int64_t cx = ix < 0 ? 0 : (ix >= n ? n - 1 : ix), ce = ie < 0 ? 0 : (ie >= n ? n - 1 : ie);
                vx[q] = in[cx]; ve[q] = in[ce];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (rr[q] < 0) continue;
                const int64_t rl = s_lo[rr[q]];
                const int64_t ix = rl + pos + lane, ie = ix - (B - 1);
                tx[rr[q]][lane] = ix < n ? vx[q] * in_gain : 0.0;
                te[rr[q]][lane] = (ie >= 0 && ie < n) ? ve[q] * in_gain : 0.0;
            }
        }
        __syncthreads();
        if (active && pos < my_len) {
            const int jn = (int)min((int64_t)TW, my_len - pos);
            // A tile that starts with the limiter at rest (unit gain, no slope, no pending peak, nothing above the limit inside the
            // look-ahead buffer) and pushes no sample above the limit leaves every sample untouched: att stays 1, o = bufv * 1, the
            // clamp is a no-op, so `te` already holds the output.  On programme that has been gain-planned to sit under the
            // ceiling that is almost every tile.
            bool rest = att == 1. && delta == 0. && nextlen == 0 && asc_c == 0;
            if (rest) {
                double mx = 0.0;
                for (int j = 0; j < jn; ++j) mx = fmax(mx, fabs(tx[lane][j]));
                rest = mx <= limit;
            }
            // A tile inside a release ramp -- no peak waiting in the list, nothing above the limit entering or leaving the look-ahead
            // buffer -- runs the same statements with every peak branch dead: att += delta, the clamps, the output (a fifth of the
            // instructions; a peak is followed by >= release * sr such samples)
            bool ramp = false;
            if (!rest && nextlen == 0 && B > 1) {
                double mx = 0.0;
                for (int j = 0; j < jn; ++j) mx = fmax(mx, fmax(fabs(tx[lane][j]), fabs(te[lane][j])));
                ramp = mx <= limit;
            }
            if (ramp) {
                for (int j = 0; j < jn; ++j) {
                    att += delta;
                    double o = te[lane][j] * att;
                    if (att > 1.) { att = 1.; delta = 0.; nextiter = 0; nextlen = 0; nextpos[0] = -1; }
                    if (att <= 0.) { att = 0.0000000000001; delta = (1.0 - att) / (sr * release); }
                    if (att != 1. && (1. - att) < 0.0000000000001) att = 1.;
                    if (delta != 0. && fabs(delta) < 0.00000000000001) delta = 0.;
                    o = fmin(fmax(o, -limit), limit);
                    te[lane][j] = o;
                }
            }
            for (int j = 0; j < ((rest || ramp) ? 0 : jn); ++j) {
                const int64_t i = lo + pos + j;
                const double x = tx[lane][j];
                double peak = fabs(x);
                if (peak > limit) { asc += peak; asc_c++; }
                if (peak > limit) {
                    const double lp = limit / peak;
                    double patt = fmin(lp, 1.);
                    // (rdelta is only stored with a new list entry: its division is left to the two places that make one -- a hot
                    // sample is a handful of f64 divisions on one lane, and they are the kernel's time)
                    auto rdelta_of = [&]() -> double { return (1.0 - patt) / (sr * release); };
                    double dl = (lp - att) / B * 1;
                    bool found = false;
                    if (dl < delta) {
                        delta = dl;
                        nextpos[0] = i; nextpos[B == 1 ? 0 : 1] = -1; nextdelta[0] = rdelta_of(); nextlp[0] = lp;
                        nextlen = 1; nextiter = 0;
                    } else {
                        int k;
                        for (k = nextiter; k < nextiter + nextlen; k++) {
                            int jj = wrapB(k);
                            double pdelta = (lp - nextlp[jj]) / (double)((i - nextpos[jj]) / 1);
                            if (pdelta < nextdelta[jj]) { nextdelta[jj] = pdelta; found = true; break; }
                        }
                        if (found) {
                            nextlen = k - nextiter + 1;
                            const int jn_ = wrapB(nextiter + nextlen);
                            nextpos[jn_] = i;
                            nextdelta[jn_] = rdelta_of();
                            nextlp[jn_] = lp;
                            nextpos[wrapB(jn_ + 1)] = -1;
                            nextlen++;
                        }
                    }
                }
                const int64_t eidx = i - (B - 1);            // sample leaving the look-ahead buffer
                double bufv = te[lane][j];
                if (B == 1) bufv = x;
                peak = fabs(bufv);
                if (peak > limit) { asc -= peak; asc_c--; }  // asc_pos == -1 always (never armed in af_alimiter.c)
                att += delta;
                double o = bufv * att;
                if (nextlen > 0 && nextpos[nextiter] == eidx && eidx >= 0) {
                    // auto_release (asc=1): get_rdelta(..., asc=1)
                    double rd = (1.0 - att) / (sr * release);
                    if (asc_c > 0) {
                        double a_att = limit / (asc_coeff * asc) * (double)asc_c;
                        if (a_att > att) {
                            double d2 = fmax((a_att - att) / (sr * release), rd / 10);
                            if (d2 < rd) rd = d2;
                        }
                    }
                    delta = rd;
                    if (nextlen > 1) {
                        const int j1_ = wrapB(nextiter + 1);
                        int64_t pnext = nextpos[j1_];
                        double pdelta = (nextlp[j1_] - att) / (double)(pnext - eidx);
                        if (pdelta < delta) delta = pdelta;
                    }
                    nextlen -= 1;
                    nextpos[nextiter] = -1;
                    nextiter = wrapB(nextiter + 1);
                }
                if (att > 1.) { att = 1.; delta = 0.; nextiter = 0; nextlen = 0; nextpos[0] = -1; }
                if (att <= 0.) { att = 0.0000000000001; delta = (1.0 - att) / (sr * release); }
                if (att != 1. && (1. - att) < 0.0000000000001) att = 1.;
                if (delta != 0. && fabs(delta) < 0.00000000000001) delta = 0.;
                o = fmin(fmax(o, -limit), limit);
                te[lane][j] = o;
            }
        }
        __syncthreads();
        // coalesced store of the emitted samples: row r column lane is sample lo_r + pos + lane - (B-1)
        for (int a = 0; a < nact; ++a) {
            const int r = act[a];
            if (pos >= s_len[r]) continue;
            const int64_t rl = s_lo[r];
            const int64_t e = rl + pos + lane - (B - 1);
            if (e >= rl && e < s_hi[r]) out[e] = te[r][lane];
        }
        __syncthreads();
    }
}

// The same state machine with ONE HOT SEGMENT PER WAVE AT A TIME (round 4).  k_limiter_f64 gives a lane a segment, and a launch lasts as long
// as its longest segment: plosives closer than the release time chain into one, and a sample above the limit costs ~1000 cycles there --
// for every such sample the filter walks its list of pending peaks until the first entry whose slope the new peak undercuts (a division
// and three loads per entry, one after the other), while 62 of the wave's 64 lanes have no segment to run.  Here the wave takes its hot
// segments one after the other and all lanes work on the current one: the samples of a tile are loaded coalesced (lane = sample) and handed
// round by v_readlane, the scalar state lives replicated in every lane (all branches are wave-uniform), the pending-peak list sits in LDS
// (one list at a time: 24 B bytes), and the walk over it is ONE step -- lane k evaluates entry k, a ballot finds the first hit.  Tiles at
// rest are skipped (the first sweep already wrote them), tiles inside a release ramp run the five statements that are alive there.  The
// same statements on the same values in the same order as the filter: bit-identical output.
__device__ __forceinline__ double lim_bcast(double v, int j)
{
    const long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)u, j), hi = __builtin_amdgcn_readlane((int)(u >> 32), j);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double lim_wave_max(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64)); return v; }

// S16: the signal is the s16 Pass-2 output behind the volume stage (vol, in float precision when vol_in_float): converted as k_absmax_conv_s16 does
template <bool S16>
__global__ void __launch_bounds__(64)
k_limiter_wave(const void *__restrict__ in_v, double *__restrict__ out, int64_t n, int sr, double limit, int B,
               double release, double asc_coeff, const int64_t *__restrict__ cand, int64_t ntargets, int blk, double in_gain,
               const double *__restrict__ block_max, int64_t nblk, unsigned long long *__restrict__ prof, double vol, int vol_in_float,
               int16_t *__restrict__ o16, float *__restrict__ o32)
{
    const double *in = static_cast<const double *>(in_v);
    const int16_t *in16 = static_cast<const int16_t *>(in_v);
    auto sample = [&](int64_t i) -> double {
        if (!S16) return in[i];
        const int16_t x = in16[i];
        if (vol_in_float) { float v = (float)x * (1.0f / 32768.0f); v = v * (float)vol; return (double)v; }
        return (double)x * (1.0 / 32768.0) * vol;
    };
    extern __shared__ __attribute__((aligned(16))) unsigned char lw_smem[];
#ifdef JT_AB
    // phase clocks of the A/B build (option lim_profile): [0..2] wave cycles in rest / ramp / hot tiles, [3..5] their tile counts, [6] hot
    // samples (above the limit), [7] the longest workgroup, [8] segments, [9] list entries tested
    unsigned long long pc[3] = {0, 0, 0}, pn[3] = {0, 0, 0}, phot = 0, ptest = 0; const unsigned long long pt_start = wall_clock64();
#define LW_MARK(i) { const unsigned long long t_ = wall_clock64(); pc[i] += t_ - ptc; pn[i]++; ptc = t_; }
#else
#define LW_MARK(i)
#endif
    int64_t *nextpos = reinterpret_cast<int64_t *>(lw_smem);
    double *nextdelta = reinterpret_cast<double *>(nextpos + B), *nextlp = nextdelta + B;
    __shared__ int64_t s_lo[LANES], s_hi[LANES];
    __shared__ int act[LANES];
    const int lane = threadIdx.x;
    const int64_t c = (int64_t)blockIdx.x * LANES + lane;
    int64_t lo_l = -1, hi_l = -1;
    if (c < ntargets && cand[c] >= 0) {
        lo_l = cand[c] * blk;
        hi_l = n;
        for (int64_t q = c + 1; q < ntargets; ++q) { const int64_t v = cand[q]; if (v >= 0) { hi_l = v * blk; break; } }
        if (lo_l >= n) { lo_l = -1; hi_l = -1; }
    }
    if (lo_l >= 0) {
        // (as k_limiter_f64: a segment in which nothing is above the limit, its look-ahead into the next one included, stays at rest)
        const double g = fabs(in_gain);
        const int64_t b0 = lo_l / blk, b1 = min(nblk - 1, (hi_l + B - 2) / blk);
        bool hot = false;
        for (int64_t b = b0; b <= b1; ++b) hot |= block_max[b] * g > limit;
        if (!hot) { lo_l = -1; hi_l = -1; }
    }
    s_lo[lane] = lo_l; s_hi[lane] = hi_l;
    const unsigned long long hot_mask = __ballot(lo_l >= 0);
    const int nact = __popcll(hot_mask);
    if (nact == 0) return;
    if (lo_l >= 0) act[__popcll(hot_mask & ((1ull << lane) - 1ull))] = lane;
    __syncthreads();
    auto wrapB = [&](int x) -> int { return x >= B ? x - B : x; };
    for (int a = 0; a < nact; ++a) {
        const int r = act[a];
        const int64_t lo = s_lo[r], hi = s_hi[r];
        const int64_t len = (hi - lo) + B - 1;          // iterations i = lo .. hi + B - 2: iteration i pushes in[i] and emits sample i - (B - 1)
        nextpos[0] = -1;
        double att = 1.0, delta = 0.0, asc = 0.0; int asc_c = 0;
        int nextiter = 0, nextlen = 0;
        // the sample entering the look-ahead buffer and the one leaving it, lane = sample of the tile; the next tile's pair is in flight
        // while the current tile is worked on
        auto fetch = [&](int64_t pos, double &vx, double &ve) {
            const int64_t ix = lo + pos + lane, ie = ix - (B - 1);
            const int64_t cx = ix < 0 ? 0 : (ix >= n ? n - 1 : ix), ce = ie < 0 ? 0 : (ie >= n ? n - 1 : ie);
            vx = sample(cx); ve = sample(ce);
        };
        // (four tiles ahead: a ramp tile takes a fifth of a memory round trip since its gains are formed first)
        constexpr int PF = 4;
        double nvx[PF], nve[PF], cvx[PF], cve[PF];
#pragma unroll
        for (int t = 0; t < PF; ++t) { nvx[t] = 0.0; nve[t] = 0.0; if ((int64_t)t * TW < len) fetch((int64_t)t * TW, nvx[t], nve[t]); }
#ifdef JT_AB
        unsigned long long ptc = wall_clock64();
#endif
        for (int64_t pos0 = 0; pos0 < len; pos0 += PF * TW) {
#pragma unroll
        for (int t = 0; t < PF; ++t) { cvx[t] = nvx[t]; cve[t] = nve[t]; }
#pragma unroll
        for (int t = 0; t < PF; ++t) { const int64_t pn = pos0 + (int64_t)(PF + t) * TW; if (pn < len) fetch(pn, nvx[t], nve[t]); }
#pragma unroll
        for (int t = 0; t < PF; ++t) {
            const int64_t pos = pos0 + (int64_t)t * TW;
            if (pos >= len) break;
            const int64_t ix = lo + pos + lane, ie = ix - (B - 1);
            const double xv = ix < n ? cvx[t] * in_gain : 0.0;
            const double ev = (ie >= 0 && ie < n) ? cve[t] * in_gain : 0.0;
            const int jn = (int)min((int64_t)TW, len - pos);
            bool rest = att == 1. && delta == 0. && nextlen == 0 && asc_c == 0;
            if (rest) rest = !__ballot(lane < jn && fabs(xv) > limit);         // (nothing above the limit <=> the maximum is not)
            if (rest) { LW_MARK(0) continue; }          // att stays 1: the outputs are in * in_gain, which the first sweep has written
            double ov = ev;
            // one sample of af_alimiter.c's loop, all lanes in step (every branch is wave-uniform)
            auto full_step = [&](int j) {
                const int64_t i = lo + pos + j;
                const double x = lim_bcast(xv, j);
                double peak = fabs(x);
                if (peak > limit) { asc += peak; asc_c++; }
                if (peak > limit) {
#ifdef JT_AB
                    phot++; ptest += (unsigned long long)nextlen;
#endif
                    const double lp = limit / peak;
                    const double patt = fmin(lp, 1.);
                    auto rdelta_of = [&]() -> double { return (1.0 - patt) / (sr * release); };
                    const double dl = (lp - att) / B * 1;
                    if (dl < delta) {
                        delta = dl;
                        nextpos[0] = i; nextpos[B == 1 ? 0 : 1] = -1; nextdelta[0] = rdelta_of(); nextlp[0] = lp;
                        nextlen = 1; nextiter = 0;
                    } else {
                        // the filter's walk `for k: pdelta = (lp - lp_k) / (i - pos_k); if (pdelta < delta_k) { delta_k = pdelta; break; }`:
                        // no entry changes before the first hit, so the entries can be tested side by side
                        int kf = -1; double pdf = 0.0;
                        for (int base = 0; base < nextlen; base += LANES) {
                            const int kk = base + lane;
                            bool hit = false; double pd = 0.0;
                            if (kk < nextlen) {
                                const int jj = wrapB(nextiter + kk);
                                pd = (lp - nextlp[jj]) / (double)((i - nextpos[jj]) / 1);
                                hit = pd < nextdelta[jj];
                            }
                            const unsigned long long bal = __ballot(hit);
                            if (bal) { const int f = __ffsll((long long)bal) - 1; kf = base + f; pdf = lim_bcast(pd, f); break; }
                        }
                        if (kf >= 0) {
                            nextdelta[wrapB(nextiter + kf)] = pdf;
                            nextlen = kf + 1;
                            const int jn_ = wrapB(nextiter + nextlen);
                            nextpos[jn_] = i;
                            nextdelta[jn_] = rdelta_of();
                            nextlp[jn_] = lp;
                            nextpos[wrapB(jn_ + 1)] = -1;
                            nextlen++;
                        }
                    }
                }
                const int64_t eidx = i - (B - 1);            // sample leaving the look-ahead buffer
                double bufv = lim_bcast(ev, j);
                if (B == 1) bufv = x;
                peak = fabs(bufv);
                if (peak > limit) { asc -= peak; asc_c--; }
                att += delta;
                double o = bufv * att;
                if (nextlen > 0 && nextpos[nextiter] == eidx && eidx >= 0) {
                    double rd = (1.0 - att) / (sr * release);
                    if (asc_c > 0) {
                        double a_att = limit / (asc_coeff * asc) * (double)asc_c;
                        if (a_att > att) {
                            double d2 = fmax((a_att - att) / (sr * release), rd / 10);
                            if (d2 < rd) rd = d2;
                        }
                    }
                    delta = rd;
                    if (nextlen > 1) {
                        const int j1_ = wrapB(nextiter + 1);
                        const int64_t pnext = nextpos[j1_];
                        const double pdelta = (nextlp[j1_] - att) / (double)(pnext - eidx);
                        if (pdelta < delta) delta = pdelta;
                    }
                    nextlen -= 1;
                    nextpos[nextiter] = -1;
                    nextiter = wrapB(nextiter + 1);
                }
                if (att > 1.) { att = 1.; delta = 0.; nextiter = 0; nextlen = 0; nextpos[0] = -1; }
                if (att <= 0.) { att = 0.0000000000001; delta = (1.0 - att) / (sr * release); }
                if (att != 1. && (1. - att) < 0.0000000000001) att = 1.;
                if (delta != 0. && fabs(delta) < 0.00000000000001) delta = 0.;
                o = fmin(fmax(o, -limit), limit);

                if (lane == j) ov = o;
            };
            // EVENTS are the samples at which the filter does more than `att += delta; out = delayed * att`: a sample above the limit
            // entering the look-ahead buffer or leaving it, the pending peak at the head of the list leaving, and a gain that touches one of
            // the per-sample clamps (above 1, at or below 0, within 1e-13 of 1).  A release ramp is thousands of samples without one (with
            // asc a release lasts up to ten times the release time), and run as written its five statements are a dependent chain of ~200
            // cycles a sample: nine tenths of the launch.  So between events the gains att_j = att_(j-1) + delta are formed first, IN ORDER
            // (additions that round as the filter's do), lane j keeping att_j, and every output is its own product; the events themselves go
            // through full_step.
            bool had_event = false;
            int j = 0;
            while (j < jn) {
                const int64_t head = nextlen > 0 ? nextpos[nextiter] : (int64_t)-1;
                const unsigned long long evm = __ballot(lane >= j && lane < jn && (fabs(xv) > limit || fabs(ev) > limit || (head >= 0 && ie == head)));
                int f = evm ? __ffsll((long long)evm) - 1 : jn;
                if (f > j && !(delta != 0. && fabs(delta) < 0.00000000000001)) {
                    double aj = 0.0, run = att;
                    if (j == 0 && f == TW) {
#pragma unroll
                        for (int q = 0; q < TW; ++q) { run += delta; aj = lane == q ? run : aj; }
                    } else {
                        for (int q = j; q < f; ++q) { run += delta; aj = lane == q ? run : aj; }
                    }
                    const bool mine = lane >= j && lane < f;
                    const unsigned long long tm = __ballot(mine && (aj > 1. || aj <= 0. || (aj != 1. && (1. - aj) < 0.0000000000001)));
                    const int f2 = tm ? __ffsll((long long)tm) - 1 : f;      // the first gain that touches a clamp: that sample is an event too
                    if (lane >= j && lane < f2) ov = fmin(fmax(ev * aj, -limit), limit);
                    if (f2 > j) att = lim_bcast(aj, f2 - 1);
                    j = f2;
                    if (f2 == f && f == jn) break;
                    f = f2;
                } else f = j;
                if (j < jn) { full_step(j); ++j; had_event = true; }
            }
            const bool ramp = !had_event; (void)ramp;
            if (lane < jn && ie >= lo && ie < hi) { if (o16) lim_emit16(ov, o16, o32, ie); else out[ie] = ov; }      // (o16: the brickwall, straight to the float / s16 outputs)
#ifdef JT_AB
            if (ramp) LW_MARK(1) else LW_MARK(2)
#endif
        }
        }
    }
#ifdef JT_AB
    if (prof && lane == 0) {
        for (int i = 0; i < 3; ++i) { atomicAdd(&prof[i], pc[i]); atomicAdd(&prof[3 + i], pn[i]); }
        atomicAdd(&prof[6], phot); atomicMax(&prof[7], wall_clock64() - pt_start); atomicAdd(&prof[8], (unsigned long long)nact); atomicAdd(&prof[9], ptest);
    }
#endif
#undef LW_MARK
}

bool jt_limiter_wave_ok(int buffer_size) { return 24 * (size_t)buffer_size <= 96 * 1024; }
void launch_limiter_f64(const double *in, double *out, int64_t n, int sr, double limit, int buffer_size,
                        double release_s, double asc_coeff, const double *block_max, int64_t nblk, int blk, int need, int target,
                        int64_t *cand, int64_t ntargets, double in_gain, double *scratch_delta, int64_t *scratch_pos, hipStream_t s, double *scratch_lp,
                        bool lane_per_segment, bool lim_profile, const LimSrc16 *src16, const LimOut16 *o16)
{
    if (n <= 0) return;
    (void)lim_profile;
    JT_REQUIRE(!o16 || (jt_limiter_wave_ok(buffer_size) && !lane_per_segment), JT_E_INVAL, "limiter: the s16 output needs the wave-per-segment kernel");
    JT_REQUIRE(!src16 || (jt_limiter_wave_ok(buffer_size) && !lane_per_segment), JT_E_INVAL, "limiter: an s16 source needs the wave-per-segment kernel");
    hipLaunchKernelGGL(k_lim_bounds, dim3((unsigned)((ntargets + 255) / 256)), dim3(256), 0, s, block_max, nblk, need, target,
                       std::fabs(in_gain), limit, cand, ntargets);
    // one hot segment per wave at a time, the pending-peak list (24 B bytes) in LDS; look-ahead buffers too long for that keep a lane per segment
    const size_t smem = 24 * (size_t)buffer_size;
    if (jt_limiter_wave_ok(buffer_size) && !lane_per_segment) {
        JT_HIP(hipFuncSetAttribute((const void *)k_limiter_wave<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        unsigned long long *prof = nullptr;
#ifdef JT_AB
        if (lim_profile) { prof = reinterpret_cast<unsigned long long *>(scratch_lp); JT_HIP(hipMemsetAsync(prof, 0, 16 * sizeof(unsigned long long), s)); }
#endif
        if (src16) {
            JT_HIP(hipFuncSetAttribute((const void *)k_limiter_wave<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL(k_limiter_wave<true>, dim3((unsigned)((ntargets + LANES - 1) / LANES)), dim3(LANES), smem, s, (const void *)src16->p, out, n, sr, limit,
                               buffer_size, release_s, asc_coeff, cand, ntargets, blk, in_gain, block_max, nblk, prof, src16->vol, src16->vol_in_float,
                               o16 ? o16->s16 : (int16_t *)nullptr, o16 ? o16->f32 : (float *)nullptr);
        } else
        hipLaunchKernelGGL(k_limiter_wave<false>, dim3((unsigned)((ntargets + LANES - 1) / LANES)), dim3(LANES), smem, s, (const void *)in, out, n, sr, limit,
                           buffer_size, release_s, asc_coeff, cand, ntargets, blk, in_gain, block_max, nblk, prof, 1.0, 0,
                           o16 ? o16->s16 : (int16_t *)nullptr, o16 ? o16->f32 : (float *)nullptr);
#ifdef JT_AB
        if (prof) {
            unsigned long long c[16];
            JT_HIP(hipStreamSynchronize(s)); JT_HIP(hipMemcpy(c, prof, sizeof c, hipMemcpyDeviceToHost));
            fprintf(stderr, "limiter (B = %d, %lld samples): wave cycles (100 MHz clock) rest %llu ramp %llu hot %llu; tiles %llu / %llu / %llu; samples above the limit %llu, "
                            "list entries tested %llu; %llu hot segments; longest workgroup %llu\n", buffer_size, (long long)n, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[9], c[8], c[7]);
        }
#endif
        return;
    }
    hipLaunchKernelGGL(k_limiter_f64, dim3((unsigned)((ntargets + LANES - 1) / LANES)), dim3(LANES), 0, s, in, out, n, sr, limit,
                       buffer_size, release_s, asc_coeff, cand, ntargets, blk, in_gain, scratch_delta, scratch_pos, block_max, nblk, scratch_lp);
}

// ------------------------------------------------------------------ biquad pair, direct form I (f32)
// Band-RMS region graphs (analyser_bands.go:33): "highpass=f=%f:p=2,lowpass=f=%f:p=2" use the default transform
// (di), float state on flt input: out = i2*b2 + i1*b1 + in*b0 + o2*a2 + o1*a1 (a already negated).
__global__ void __launch_bounds__(64)
k_biquad_di_f32(const float *__restrict__ in, float *__restrict__ out, int64_t n, int64_t chunk, int64_t halo,
                BiquadF32 hp, BiquadF32 lp, int64_t nchunks)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    int64_t lo = c * chunk, hi = min(lo + chunk, n);
    int64_t s0 = max((int64_t)0, lo - halo);
    float i1 = 0, i2 = 0, o1 = 0, o2 = 0, j1 = 0, j2 = 0, p1 = 0, p2 = 0;
    for (int64_t k = s0; k < hi; ++k) {
        float x = in[k];
        float y = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(i2, hp.b2), __fmul_rn(i1, hp.b1)), __fmul_rn(x, hp.b0)),
                                      __fmul_rn(o2, hp.a2)), __fmul_rn(o1, hp.a1));
        i2 = i1; i1 = x; o2 = o1; o1 = y;
        float z = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(j2, lp.b2), __fmul_rn(j1, lp.b1)), __fmul_rn(y, lp.b0)),
                                      __fmul_rn(p2, lp.a2)), __fmul_rn(p1, lp.a1));
        j2 = j1; j1 = y; p2 = p1; p1 = z;
        if (k >= lo) out[k] = z;
    }
}
void launch_biquad_di_f32(const float *in, float *out, int64_t n, BiquadF32 hp, BiquadF32 lp, hipStream_t s)
{
    if (n <= 0) return;
    double rmax = std::max(std::sqrt(std::fabs((double)hp.a2)), std::sqrt(std::fabs((double)lp.a2)));
    int64_t halo = 512;
    if (rmax > 0.0 && rmax < 1.0) halo = std::max<int64_t>(512, (int64_t)std::ceil(std::log(1e-10) / std::log(rmax)));
    halo = std::min<int64_t>(halo, 1 << 20);
    int64_t chunk = std::max<int64_t>(1024, halo / 2);
    int64_t nchunks = (n + chunk - 1) / chunk;
    hipLaunchKernelGGL(k_biquad_di_f32, dim3((unsigned)((nchunks + 63) / 64)), dim3(64), 0, s, in, out, n, chunk, halo, hp, lp, nchunks);
}

// ------------------------------------------------------------------ fused band RMS (all bands of one region, one launch)
// grid.y = band.  Each lane filters one chunk (+ warm-up halo) of the region with that band's highpass/lowpass pair
// (direct form I, float state, as k_biquad_di_f32) and accumulates sum(z^2) in double; wave-reduced, one atomic per wave.
// The chunks stream through a [64][TW+1] LDS tile (row-coalesced loads): read straight from global memory, one uncoalesced
// load per sample per lane left the recurrence waiting on memory latency (1.2 ms for the 15 noise bands of a 10 s region).
// MODE: the sample format libavfilter negotiates for the band graph (nothing in it is float-only, so the source's width survives,
// DESIGN.md section 3): 0 = fltp (float sources), 1 = s16p, 2 = s32p.  af_biquads.c BIQUAD_FILTER(s16, int16_t, float, ..., 1) /
// (s32, int32_t, double, ..., 1): float / double coefficients and state on the raw integer values, the recursion fed with the
// unquantised outputs, every stage's output clipped and stored by a C cast (truncation toward zero); af_astats.c squares integer
// samples divided by INT16_MAX / INT32_MAX.
struct BandBiquads { double hp[16][5]; double lp[16][5]; };      // b0 b1 b2 a1 a2 (a1, a2 negated), af_biquads.c's doubles
template <typename F> struct BandCoef { F b0, b1, b2, a1, a2; };
template <typename F> __device__ __forceinline__ F band_di(F i2, F i1, F x, F o2, F o1, const BandCoef<F> &c);
template <> __device__ __forceinline__ float band_di<float>(float i2, float i1, float x, float o2, float o1, const BandCoef<float> &c)
{
    return __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(i2, c.b2), __fmul_rn(i1, c.b1)), __fmul_rn(x, c.b0)), __fmul_rn(o2, c.a2)), __fmul_rn(o1, c.a1));
}
template <> __device__ __forceinline__ double band_di<double>(double i2, double i1, double x, double o2, double o1, const BandCoef<double> &c)
{
    return __dadd_rn(__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(i2, c.b2), __dmul_rn(i1, c.b1)), __dmul_rn(x, c.b0)), __dmul_rn(o2, c.a2)), __dmul_rn(o1, c.a1));
}
template <int MODE> __global__ void __launch_bounds__(64)
k_band_rms(const float *__restrict__ in, int64_t n, int64_t chunk, int64_t halo, BandBiquads bq, double *__restrict__ sums, int64_t nchunks)
{
    typedef typename std::conditional<MODE == 2, double, float>::type F;
    __shared__ float tile[LANES][TW + 1];
    const int band = blockIdx.y, lane = threadIdx.x;
    BandCoef<F> hp, lp;
    hp.b0 = (F)bq.hp[band][0]; hp.b1 = (F)bq.hp[band][1]; hp.b2 = (F)bq.hp[band][2]; hp.a1 = (F)bq.hp[band][3]; hp.a2 = (F)bq.hp[band][4];
    lp.b0 = (F)bq.lp[band][0]; lp.b1 = (F)bq.lp[band][1]; lp.b2 = (F)bq.lp[band][2]; lp.a1 = (F)bq.lp[band][3]; lp.a2 = (F)bq.lp[band][4];
    const F scale = MODE == 1 ? (F)32768.0 : (F)2147483648.0;
    const F qmin = MODE == 1 ? (F)-32768.0 : (F)-2147483648.0, qmax = MODE == 1 ? (F)32767.0 : (F)2147483647.0;
    const double norm = MODE == 1 ? 1.0 / 32767.0 : 1.0 / 2147483647.0;
    const int64_t c0 = (int64_t)blockIdx.x * LANES;
    const int nrows = (int)min((int64_t)LANES, nchunks - c0);
    const int64_t base0 = c0 * chunk - halo;
    const int64_t my_base = base0 + (int64_t)lane * chunk, my_lo = my_base + halo;
    double acc = 0.0;
    F i1 = 0, i2 = 0, o1 = 0, o2 = 0, j1 = 0, j2 = 0, p1 = 0, p2 = 0;
    const int64_t total = halo + chunk;
    float v[LANES];
    rows_prefetch<float>(v, in, n, base0, chunk, 0, lane);
    for (int64_t pos = 0; pos < total; pos += TW) {
        rows_commit<float, float>(tile, v, n, base0, chunk, pos, lane, nrows);
        __syncthreads();
        if (pos + TW < total) rows_prefetch<float>(v, in, n, base0, chunk, pos + TW, lane);     // next tile in flight during this one
        if (lane < nrows) {
#pragma unroll 8
            for (int j = 0; j < TW; ++j) {
                const int64_t k = my_base + pos + j;
                if (k < 0 || k >= n) continue;          // before the region the state is exactly zero; nothing follows its end
                F x = (F)tile[lane][j];
                if (MODE != 0) { x = x * scale; x = x > qmax ? qmax : x; }       // the integer sample itself (exact: a power of two)
                const F y = band_di<F>(i2, i1, x, o2, o1, hp);
                i2 = i1; i1 = x; o2 = o1; o1 = y;
                F y2 = y;
                if (MODE != 0) { y2 = y < qmin ? qmin : (y > qmax ? qmax : y); y2 = MODE == 1 ? (F)truncf((float)y2) : (F)trunc((double)y2); }
                const F z = band_di<F>(j2, j1, y2, p2, p1, lp);
                j2 = j1; j1 = y2; p2 = p1; p1 = z;
                if (k >= my_lo) {
                    if (MODE == 0) acc += (double)z * (double)z;
                    else {
                        F zq = z < qmin ? qmin : (z > qmax ? qmax : z);
                        const double nd = (MODE == 1 ? (double)truncf((float)zq) : trunc((double)zq)) * norm;
                        acc += nd * nd;
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (threadIdx.x == 0 && acc != 0.0) atomicAdd(&sums[band], acc);
}

void launch_band_rms(const float *in, int64_t n, int nbands, const double (*hp)[5], const double (*lp)[5], int mode, double *sums, hipStream_t s)
{
    if (n <= 0 || nbands <= 0) return;
    BandBiquads bq;
    double rmax = 0.0;
    for (int b = 0; b < nbands && b < 16; ++b) {
        for (int k = 0; k < 5; ++k) { bq.hp[b][k] = hp[b][k]; bq.lp[b][k] = lp[b][k]; }
        rmax = std::max(rmax, std::max(std::sqrt(std::fabs(hp[b][4])), std::sqrt(std::fabs(lp[b][4]))));
    }
    int64_t halo = 512;
    if (rmax > 0.0 && rmax < 1.0) halo = std::max<int64_t>(512, (int64_t)std::ceil(std::log(1e-10) / std::log(rmax)));
    halo = std::min<int64_t>(halo, 1 << 20);
    halo = (halo + TW - 1) / TW * TW;
    // the caller waits for this kernel with an otherwise idle GPU: short chunks (each pays the whole halo again) keep the one
    // thing that matters, the longest serial run per lane, close to the halo itself
    const int64_t chunk = 256;
    int64_t nchunks = (n + chunk - 1) / chunk;
    const dim3 grid((unsigned)((nchunks + 63) / 64), (unsigned)nbands);
    if (mode == 1) hipLaunchKernelGGL(k_band_rms<1>, grid, dim3(64), 0, s, in, n, chunk, halo, bq, sums, nchunks);
    else if (mode == 2) hipLaunchKernelGGL(k_band_rms<2>, grid, dim3(64), 0, s, in, n, chunk, halo, bq, sums, nchunks);
    else hipLaunchKernelGGL(k_band_rms<0>, grid, dim3(64), 0, s, in, n, chunk, halo, bq, sums, nchunks);
}
