// k_flac.hip — FLAC encoder for the s16 mono stage outputs, one 4096-sample frame per wavefront (gfx950).
//
// Replaces the reference's output leg: Encoder.WriteFrame -> avcodec flac encoder, s16, compression_level 5, frame_size 4096
// (encoder.go:54-110,145; asetnsamples n=4096 normalise.go:1318-1330).  The contract is the FLAC format (RFC 9639), not
// FFmpeg's byte stream: FLAC is lossless, so "same result" means the decoded PCM, STREAMINFO (rate, channels, depth, total
// samples, MD5) and block size are identical; the choice of predictor and Rice parameters only moves the file size.
//
// Why on the GPU: the Pass-2 and Pass-4 outputs already sit in HBM as s16, frames are independent, and a 60-minute file is
// 38 760 frames — a scalar CPU encoder spends seconds on it, the DSP before it ~0.13 s.  Per frame (one wave, lane l owns
// samples 64l..64l+63 in registers after one LDS transpose):
//   analyse : Welch-windowed autocorrelation (f64, lags 0..8) -> Levinson-Durbin -> 12-bit quantised predictors of every order
//             1..8 (+ order-0) -> for each order the residual, its 16-sample Rice sums and the estimated size at every
//             partition order 0..8 -> best (order, partition order, Rice parameters) -> exact size of that choice.
//             12-bit coefficients keep 16+12+3 <= 32 bits, so every decoder's 32-bit predictor arithmetic is exact.
//   scan    : exclusive prefix sum of the frame sizes (byte offsets), min/max frame size for STREAMINFO.
//   emit    : recompute the chosen residual, per-lane bit lengths -> wave prefix sum -> every lane writes its Rice codes into
//             an LDS bit buffer (ds_or), CRC-16 by per-lane table CRCs combined with GF(2) multiplications by x^(8·len),
//             dword stores to the final byte offset (byte stores on the two unaligned edges).
// All three are integer / byte work bound by instruction issue and LDS, not HBM (2 B/sample read twice, ~1 B/sample written).
#include "jt_internal.h"
#include <hip/hip_runtime.h>

namespace {
namespace fl {
constexpr int BS = 4096, ROW = 65, TILE = 64 * ROW, MAXORD = 8, PREC = 12, WAVES = 4, VERB_LIMIT = 16;
constexpr int T_CONST = 0, T_VERB = 1, T_LPC = 2, T_FIXED0 = 3;

struct Rec {                     // analysis -> emit, one per frame
    int32_t bytes;               // whole frame: header + subframe + CRC-16
    int32_t sub_bits;            // subframe bits (before byte padding)
    uint8_t type, order, shift, porder;
    int16_t coef[MAXORD];
    uint8_t hdr_len, pad[3];
    uint8_t k[256];              // Rice parameter per partition
};
static_assert(sizeof(Rec) == 288, "Rec layout");
struct StreamCodes { int sr_code, sr_extra_bytes, sr_extra_val; };
struct Summary { long long total; int min_frame, max_frame, mismatches, pad; };

__device__ __forceinline__ int utf8_len(unsigned v) { return v < 0x80 ? 1 : v < 0x800 ? 2 : v < 0x10000 ? 3 : v < 0x200000 ? 4 : v < 0x4000000 ? 5 : 6; }
__device__ __forceinline__ int header_len(unsigned frame, int bs, const StreamCodes &st)
{
    return 4 + utf8_len(frame) + (bs == BS ? 0 : (bs <= 256 ? 1 : 2)) + st.sr_extra_bytes + 1;
}
__device__ __forceinline__ unsigned zigzag(int r) { return ((unsigned)r << 1) ^ (unsigned)(r >> 31); }

// lane l <- samples 64l..64l+63 (x) and the 8 before them (h[k] = sample 64l-1-k), through a [64][65] LDS transpose
__device__ __forceinline__ void load_frame(const int16_t *__restrict__ pcm, int64_t base, int bs, int *tile, int lane,
                                           int (&x)[64], int (&h)[8])
{
#pragma unroll 8
    for (int j = 0; j < 64; j++) {
        const int idx = j * 64 + lane;
        tile[idx + j] = idx < bs ? (int)pcm[base + idx] : 0;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 64; c++) x[c] = tile[ROW * lane + c];
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = lane > 0 ? tile[ROW * (lane - 1) + 63 - k] : 0;
    __builtin_amdgcn_wave_barrier();
}

// residual of every sample of the lane for the predictor (q, sh); visit(c, r) with c a compile-time index after unrolling
template <class Visit>
__device__ __forceinline__ void residual_pass(const int (&x)[64], const int (&h)[8], const int (&q)[8], int sh, Visit &&visit)
{
#pragma unroll
    for (int c = 0; c < 64; c++) {
        int pred = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int t = c - 1 - j;
            pred += __mul24(q[j], t >= 0 ? x[t] : h[-t - 1]);
        }
        visit(c, x[c] - (pred >> sh));
    }
}

// FFmpeg-style Rice parameter estimate from a partition's sum of folded residuals
__device__ __forceinline__ void rice_est(unsigned sum, int n, int &k, unsigned &bits)
{
    const unsigned half = (unsigned)n >> 1;
    if (n <= 0) { k = 0; bits = 0; return; }
    if (sum <= half) { k = 0; bits = (unsigned)n + sum; return; }
    const unsigned s2 = sum - half;
    int kk = (31 - __clz(s2)) - (31 - __clz(n));
    if (kk > 0 && ((unsigned long long)n << kk) > s2) kk--;
    kk = kk < 0 ? 0 : (kk > 14 ? 14 : kk);
    k = kk; bits = (unsigned)n * (kk + 1) + (s2 >> kk);
}

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------- analyse
template <bool FULL>
__global__ __launch_bounds__(256) void k_flac_analyse(const int16_t *__restrict__ pcm, int64_t n, int64_t frame0, int64_t frame_end,
                                                      StreamCodes st, Rec *__restrict__ recs)
{
    __shared__ int tiles[WAVES][TILE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t f = frame0 + (int64_t)blockIdx.x * WAVES + wave;
    if (f >= frame_end) return;
    int *tile = tiles[wave];
    const int64_t base = f * BS;
    const int bs = FULL ? BS : (int)(n - base < BS ? n - base : BS);
    int x[64], h[8];
    load_frame(pcm, base, bs, tile, lane, x, h);
    const int nvalid = FULL ? 64 : (bs - 64 * lane < 0 ? 0 : (bs - 64 * lane > 64 ? 64 : bs - 64 * lane));

    Rec *rec = recs + f;
    const int hdr = header_len((unsigned)f, bs, st);
    const int x0 = __builtin_amdgcn_readfirstlane(x[0]);
    int diff = 0;
#pragma unroll
    for (int c = 0; c < 64; c++) diff |= (FULL || c < nvalid) ? (x[c] ^ x0) : 0;
    const bool constant = __ballot(diff != 0) == 0;
    const int verb_bits = 8 + 16 * bs;
    if (constant || bs <= VERB_LIMIT) {
        if (lane == 0) {
            const int sb = constant ? 8 + 16 : verb_bits;
            rec->type = constant ? T_CONST : T_VERB; rec->order = 0; rec->shift = 0; rec->porder = 0; rec->hdr_len = (uint8_t)hdr;
            rec->sub_bits = sb; rec->bytes = hdr + ((sb + 7) >> 3) + 2;
        }
        return;
    }

    // ---- Welch-windowed autocorrelation, lags 0..8 (f64; fixed summation order -> deterministic bytes)
    double ac[MAXORD + 1];
    {
        const double inv = 2.0 / (double)(bs - 1);
        double prev[MAXORD];
#pragma unroll
        for (int k = 0; k < MAXORD; k++) {
            const int i = 64 * lane - 1 - k;
            const double t = (double)i * inv - 1.0;
            prev[k] = i >= 0 ? (double)h[k] * (1.0 - t * t) : 0.0;
        }
#pragma unroll
        for (int k = 0; k <= MAXORD; k++) ac[k] = 0.0;
#pragma unroll
        for (int c = 0; c < 64; c++) {
            const double t = (double)(64 * lane + c) * inv - 1.0;
            const double cur = (double)x[c] * (1.0 - t * t);          // samples past the end of a short frame are 0
            ac[0] += cur * cur;
#pragma unroll
            for (int k = 0; k < MAXORD; k++) ac[k + 1] += cur * prev[k];
#pragma unroll
            for (int k = MAXORD - 1; k > 0; k--) prev[k] = prev[k - 1];
            prev[0] = cur;
        }
#pragma unroll
        for (int k = 0; k <= MAXORD; k++) ac[k] = wave_sum_f64(ac[k]);
    }

    // ---- Levinson-Durbin; quantise the predictor of every order to PREC bits; table in LDS: [order][0..7] coefs, [8] shift
    int *tab = tile;                                   // the tile is free once x/h are in registers
    int nord = 0;
    {
        double a[MAXORD], err = ac[0];
#pragma unroll
        for (int j = 0; j < MAXORD; j++) a[j] = 0.0;
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 9; j++) tab[j] = 0;    // order 0: no prediction
        }
        bool alive = err > 0.0;
#pragma unroll
        for (int i = 0; i < MAXORD; i++) {
            double acc = ac[i + 1];
#pragma unroll
            for (int j = 0; j < i; j++) acc -= a[j] * ac[i - j];
            const double kr = alive ? acc / err : 0.0;
            double an[MAXORD];
#pragma unroll
            for (int j = 0; j < i; j++) an[j] = a[j] - kr * a[i - 1 - j];
#pragma unroll
            for (int j = 0; j < i; j++) a[j] = an[j];
            a[i] = kr;
            err *= 1.0 - kr * kr;
            alive = alive && err > 0.0 && kr == kr && fabs(kr) < 4.0;
            if (alive) nord = i + 1;
            // quantise order i+1
            double cmax = 0.0;
#pragma unroll
            for (int j = 0; j <= i; j++) cmax = fmax(cmax, fabs(a[j]));
            const double qmax = (double)((1 << (PREC - 1)) - 1);
            int sh = 15;
            while (sh > 0 && cmax * (double)(1 << sh) > qmax) sh--;
            double e = 0.0;
#pragma unroll
            for (int j = 0; j <= i; j++) {
                const double v = a[j] * (double)(1 << sh) + e;
                double qv = rint(v);
                qv = qv > qmax ? qmax : (qv < -qmax ? -qmax : qv);
                e = v - qv;
                if (lane == 0) tab[(i + 1) * 9 + j] = (int)qv;
            }
            if (lane == 0) {
#pragma unroll
                for (int j = i + 1; j < MAXORD; j++) tab[(i + 1) * 9 + j] = 0;
                tab[(i + 1) * 9 + 8] = sh;
            }
        }
    }
    nord = __builtin_amdgcn_readfirstlane(nord);
    __builtin_amdgcn_wave_barrier();

    // ---- every order: residual -> Rice sums -> estimated size per partition order; last iteration: exact size of the winner
    unsigned best_est = 0xffffffffu; int bo = 0, bp = 0; unsigned long long bk = 0;
    unsigned exact_bits = 0;
    const int maxp = FULL ? 8 : 0;
    for (int it = 0; it <= nord + 1; it++) {
        const bool fin = it == nord + 1;
        const int o = fin ? bo : it;
        int q[8];
#pragma unroll
        for (int j = 0; j < 8; j++) q[j] = __builtin_amdgcn_readfirstlane(tab[o * 9 + j]);
        const int sh = __builtin_amdgcn_readfirstlane(tab[o * 9 + 8]);
        int k4[4] = {0, 0, 0, 0};
        if (fin) {
            const int kk = (int)((bk >> (24 + 4 * (6 - (bp > 6 ? 6 : bp)))) & 15);
#pragma unroll
            for (int qd = 0; qd < 4; qd++)
                k4[qd] = bp == 8 ? (int)((bk >> (4 * qd)) & 15) : bp == 7 ? (int)((bk >> (16 + 4 * (qd >> 1))) & 15) : kk;
        }
        unsigned q4[4] = {0, 0, 0, 0}, ex = 0;
        residual_pass(x, h, q, sh, [&](int c, int r) __attribute__((always_inline)) {
            bool valid = FULL || c < nvalid;
            if (c < MAXORD) valid = valid && !(lane == 0 && c < o);
            const unsigned u = valid ? zigzag(r) : 0u;
            q4[c >> 4] += u;
            ex += valid ? (u >> k4[c >> 4]) + (unsigned)k4[c >> 4] + 1u : 0u;
        });
        if (fin) { exact_bits = wave_sum_u32(ex); break; }

        const int first = lane == 0 ? o : 0;
        unsigned long long kp = 0;
        unsigned cost[9];
        {   // partition order 8: 4 partitions of 16 per lane
            unsigned b = 0;
#pragma unroll
            for (int qd = 0; qd < 4; qd++) { int k; unsigned bb; rice_est(q4[qd], 16 - (qd == 0 ? first : 0), k, bb); b += bb + 4; kp |= (unsigned long long)k << (4 * qd); }
            cost[8] = b;
        }
        {   // 7: 2 partitions of 32
            unsigned b = 0;
#pragma unroll
            for (int hf = 0; hf < 2; hf++) { int k; unsigned bb; rice_est(q4[2 * hf] + q4[2 * hf + 1], 32 - (hf == 0 ? first : 0), k, bb); b += bb + 4; kp |= (unsigned long long)k << (16 + 4 * hf); }
            cost[7] = b;
        }
        unsigned s = q4[0] + q4[1] + q4[2] + q4[3];
        { int k; unsigned bb; rice_est(s, 64 - first, k, bb); cost[6] = bb + 4; kp |= (unsigned long long)k << 24; }
        if (FULL) {
#pragma unroll
            for (int lvl = 5; lvl >= 0; lvl--) {                   // groups of g = 2^(6-lvl) lanes share a partition
                const int g = 1 << (6 - lvl);
                s += __shfl_xor(s, g >> 1);
                int k; unsigned bb;
                rice_est(s, 64 * g - (lane < g ? o : 0), k, bb);
                kp |= (unsigned long long)k << (24 + 4 * (6 - lvl));
                unsigned cst = bb + 4;
#pragma unroll
                for (int m = g; m < 64; m <<= 1) cst += __shfl_xor(cst, m);
                cost[lvl] = cst;
            }
            cost[8] = wave_sum_u32(cost[8]); cost[7] = wave_sum_u32(cost[7]); cost[6] = wave_sum_u32(cost[6]);
        } else {
            // short last frame: a single partition over the whole block
            s = wave_sum_u32(s);
            int k; unsigned bb; rice_est(s, bs - o, k, bb);
            kp = (unsigned long long)k << (24 + 4 * 6);
            cost[0] = bb + 4;
        }
        const unsigned fixed_bits = 8u + (unsigned)o * 16u + (o > 0 ? 9u + (unsigned)o * PREC : 0u) + 6u;
#pragma unroll
        for (int p = 0; p <= 8; p++) {
            if (p > maxp) continue;
            const unsigned tot = fixed_bits + cost[p];
            if (tot < best_est) { best_est = tot; bo = o; bp = p; bk = kp; }
        }
        bo = __builtin_amdgcn_readfirstlane(bo); bp = __builtin_amdgcn_readfirstlane(bp);
    }

    const unsigned lpc_bits = 8u + (unsigned)bo * 16u + (bo > 0 ? 9u + (unsigned)bo * PREC : 0u) + 6u + 4u * (1u << bp) + exact_bits;
    const bool use_lpc = lpc_bits < (unsigned)verb_bits;
    if (lane == 0) {
        const int sb = use_lpc ? (int)lpc_bits : verb_bits;
        rec->type = use_lpc ? (bo > 0 ? T_LPC : T_FIXED0) : T_VERB;
        rec->order = (uint8_t)bo; rec->shift = (uint8_t)tab[bo * 9 + 8]; rec->porder = (uint8_t)bp; rec->hdr_len = (uint8_t)hdr;
        for (int j = 0; j < MAXORD; j++) rec->coef[j] = (int16_t)tab[bo * 9 + j];
        rec->sub_bits = sb; rec->bytes = hdr + ((sb + 7) >> 3) + 2;
    }
    if (use_lpc) {
        if (bp == 8) { for (int qd = 0; qd < 4; qd++) rec->k[4 * lane + qd] = (uint8_t)((bk >> (4 * qd)) & 15); }
        else if (bp == 7) { for (int hf = 0; hf < 2; hf++) rec->k[2 * lane + hf] = (uint8_t)((bk >> (16 + 4 * hf)) & 15); }
        else {
            const int g = 1 << (6 - bp);
            if ((lane & (g - 1)) == 0) rec->k[lane >> (6 - bp)] = (uint8_t)((bk >> (24 + 4 * (6 - bp))) & 15);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------- scan
__global__ __launch_bounds__(1024) void k_flac_scan(const Rec *__restrict__ recs, int64_t nframes, long long *__restrict__ offs,
                                                    Summary *__restrict__ sum)
{
    __shared__ long long part[1024];
    __shared__ int smin[1024], smax[1024];
    const int t = threadIdx.x;
    const int64_t per = (nframes + 1023) / 1024, lo = per * t, hi = lo + per < nframes ? lo + per : nframes;
    long long acc = 0; int mn = 0x7fffffff, mx = 0;
    for (int64_t i = lo; i < hi; i++) { const int b = recs[i].bytes; acc += b; mn = b < mn ? b : mn; mx = b > mx ? b : mx; }
    part[t] = acc; smin[t] = mn; smax[t] = mx;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const long long v = t >= d ? part[t - d] : 0;
        const int a = t >= d ? smin[t - d] : 0x7fffffff, b = t >= d ? smax[t - d] : 0;
        __syncthreads();
        part[t] += v; smin[t] = a < smin[t] ? a : smin[t]; smax[t] = b > smax[t] ? b : smax[t];
        __syncthreads();
    }
    long long run = part[t] - acc;
    for (int64_t i = lo; i < hi; i++) { offs[i] = run; run += recs[i].bytes; }
    if (t == 1023) { sum->total = part[1023]; sum->min_frame = smin[1023]; sum->max_frame = smax[1023]; sum->mismatches = 0; }
}

// ------------------------------------------------------------------------------------------------------------------- emit
__device__ __forceinline__ void put_bits(unsigned *buf, unsigned pos, unsigned v, int nb)        // nb <= 32, v < 2^nb
{
    const unsigned w = pos >> 5, s = pos & 31;
    const unsigned long long t = (unsigned long long)v << (64 - nb - (int)s);
    const unsigned hi = (unsigned)(t >> 32), lo = (unsigned)t;
    if (hi) atomicOr(&buf[w], hi);
    if (lo) atomicOr(&buf[w + 1], lo);
}
__device__ __forceinline__ unsigned gf_mulmod(unsigned a, unsigned b)        // a*b mod x^16+x^15+x^2+1 over GF(2)
{
    unsigned r = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        r <<= 1;
        if (r & 0x10000) r ^= 0x18005;
        if ((b >> i) & 1) r ^= a;
    }
    return r & 0xffff;
}

template <bool FULL>
__global__ __launch_bounds__(256) void k_flac_emit(const int16_t *__restrict__ pcm, int64_t n, int64_t frame0, int64_t frame_end,
                                                   StreamCodes st, const Rec *__restrict__ recs, const long long *__restrict__ offs,
                                                   uint8_t *__restrict__ out, Summary *__restrict__ sum)
{
    __shared__ int tiles[WAVES][TILE];
    __shared__ unsigned crc_tab[256];
    {
        unsigned c = threadIdx.x << 8;
        for (int k = 0; k < 8; k++) c = (c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1);
        crc_tab[threadIdx.x] = c & 0xffff;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t f = frame0 + (int64_t)blockIdx.x * WAVES + wave;
    if (f >= frame_end) return;
    int *tile = tiles[wave];
    const int64_t base = f * BS;
    const int bs = FULL ? BS : (int)(n - base < BS ? n - base : BS);
    int x[64], h[8];
    load_frame(pcm, base, bs, tile, lane, x, h);
    const int nvalid = FULL ? 64 : (bs - 64 * lane < 0 ? 0 : (bs - 64 * lane > 64 ? 64 : bs - 64 * lane));
    const Rec *rec = recs + f;
    const int type = rec->type, o = rec->order, sh = rec->shift, bp = rec->porder, hdr = rec->hdr_len, sub_bits = rec->sub_bits;
    const long long off = offs[f];
    const int a = (int)(off & 3);
    unsigned *buf = reinterpret_cast<unsigned *>(tile);
    const int fbytes = hdr + ((sub_bits + 7) >> 3);             // frame bytes before the CRC-16
    const int nwords = (a + fbytes + 2 + 3) >> 2;
    for (int w = lane; w < nwords + 1; w += 64) buf[w] = 0;
    __builtin_amdgcn_wave_barrier();

    // ---- frame header (RFC 9639 §9.1) + subframe header, written by lane 0
    unsigned pos = 8u * (unsigned)a;
    if (lane == 0) {
        unsigned char hb[16]; int nb = 0;
        hb[nb++] = 0xff; hb[nb++] = 0xf8;
        hb[nb++] = (unsigned char)(((bs == BS ? 12 : (bs <= 256 ? 6 : 7)) << 4) | st.sr_code);
        hb[nb++] = 0x08;                                      // 1 channel, 16 bits per sample
        const unsigned v = (unsigned)f;
        if (v < 0x80) hb[nb++] = (unsigned char)v;
        else {
            const int len = utf8_len(v);
            hb[nb++] = (unsigned char)(((0xff00 >> len) & 0xff) | (v >> (6 * (len - 1))));
            for (int i = len - 2; i >= 0; i--) hb[nb++] = (unsigned char)(0x80 | ((v >> (6 * i)) & 0x3f));
        }
        if (bs != BS) { if (bs <= 256) hb[nb++] = (unsigned char)(bs - 1); else { hb[nb++] = (unsigned char)((bs - 1) >> 8); hb[nb++] = (unsigned char)(bs - 1); } }
        if (st.sr_extra_bytes == 1) hb[nb++] = (unsigned char)st.sr_extra_val;
        else if (st.sr_extra_bytes == 2) { hb[nb++] = (unsigned char)(st.sr_extra_val >> 8); hb[nb++] = (unsigned char)st.sr_extra_val; }
        unsigned c8 = 0;
        for (int i = 0; i < nb; i++) { c8 ^= hb[i]; for (int k = 0; k < 8; k++) c8 = (c8 & 0x80) ? ((c8 << 1) ^ 0x07) & 0xff : (c8 << 1) & 0xff; }
        hb[nb++] = (unsigned char)c8;
        unsigned p = pos;
        for (int i = 0; i < nb; i++) { put_bits(buf, p, hb[i], 8); p += 8; }
        const unsigned sfh = type == T_CONST ? 0x00u : type == T_VERB ? 0x02u : type == T_FIXED0 ? 0x10u : (unsigned)((0x20 | (o - 1)) << 1);
        put_bits(buf, p, sfh, 8);
    }
    pos += 8u * (unsigned)hdr + 8u;

    if (type == T_CONST) {
        if (lane == 0) put_bits(buf, pos, (unsigned)x[0] & 0xffffu, 16);
    } else if (type == T_VERB) {
#pragma unroll
        for (int c = 0; c < 64; c++)
            if (FULL || c < nvalid) put_bits(buf, pos + 16u * (unsigned)(64 * lane + c), (unsigned)x[c] & 0xffffu, 16);
    } else {
        int q[8];
#pragma unroll
        for (int j = 0; j < 8; j++) q[j] = __builtin_amdgcn_readfirstlane((int)rec->coef[j]);
        if (lane == 0) {
            unsigned p = pos;
            for (int j = 0; j < o; j++) {                      // warm-up samples (lane 0 owns samples 0..63)
                int v = 0;
#pragma unroll
                for (int c = 0; c < MAXORD; c++) v = c == j ? x[c] : v;
                put_bits(buf, p, (unsigned)v & 0xffffu, 16); p += 16;
            }
            if (o > 0) {
                put_bits(buf, p, PREC - 1, 4); p += 4;
                put_bits(buf, p, (unsigned)sh, 5); p += 5;
                for (int j = 0; j < o; j++) {
                    int v = 0;
#pragma unroll
                    for (int c = 0; c < MAXORD; c++) v = c == j ? q[c] : v;
                    put_bits(buf, p, (unsigned)v & ((1u << PREC) - 1), PREC); p += PREC;
                }
            }
            put_bits(buf, p, 0, 2); p += 2;
            put_bits(buf, p, (unsigned)bp, 4);
        }
        pos += 16u * (unsigned)o + (o > 0 ? 9u + (unsigned)o * PREC : 0u) + 6u;

        // Rice parameter of each 16-sample quarter of this lane, and whether a partition (its 4-bit parameter) starts there
        int k4[4]; bool start[4];
        {
            const int bpl = bp > 6 ? 6 : bp, g = 1 << (6 - bpl);
            const int kk = rec->k[lane >> (6 - bpl)];
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                k4[qd] = bp == 8 ? rec->k[4 * lane + qd] : bp == 7 ? rec->k[2 * lane + (qd >> 1)] : kk;
                start[qd] = bp == 8 ? true : bp == 7 ? (qd & 1) == 0 : (qd == 0 && (lane & (g - 1)) == 0);
            }
        }
        unsigned u[64]; unsigned lane_bits = 0;
        residual_pass(x, h, q, sh, [&](int c, int r) __attribute__((always_inline)) {
            bool valid = FULL || c < nvalid;
            if (c < MAXORD) valid = valid && !(lane == 0 && c < o);
            u[c] = valid ? zigzag(r) : 0xffffffffu;
            if ((c & 15) == 0 && start[c >> 4]) lane_bits += 4;
            lane_bits += valid ? (u[c] >> k4[c >> 4]) + (unsigned)k4[c >> 4] + 1u : 0u;
        });
        unsigned scan = lane_bits;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned v = __shfl_up(scan, d); scan += lane >= d ? v : 0u; }
        const unsigned total = __shfl(scan, 63);
        const unsigned expect = (unsigned)sub_bits - (8u + 16u * (unsigned)o + (o > 0 ? 9u + (unsigned)o * PREC : 0u) + 6u);
        if (lane == 0 && total != expect) atomicAdd(&sum->mismatches, 1);
        unsigned p = pos + scan - lane_bits;
#pragma unroll
        for (int c = 0; c < 64; c++) {
            const int k = k4[c >> 4];
            if ((c & 15) == 0 && start[c >> 4]) { put_bits(buf, p, (unsigned)k, 4); p += 4; }
            if (u[c] != 0xffffffffu) {
                p += u[c] >> k;
                put_bits(buf, p, (1u << k) | (u[c] & ((1u << k) - 1)), k + 1);
                p += (unsigned)k + 1u;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- CRC-16 of the frame (leading zero bytes do not change a zero-initialised CRC): lane l covers one chunk of the
    // right-aligned byte range, the chunks are combined by a log-step tree of multiplications by x^(8*chunk)
    {
        const int T = a + fbytes, C = (T + 63) >> 6, lead = 64 * C - T;
        unsigned crc = 0;
        for (int i = 0; i < C; i++) {
            const int b = lane * C + i - lead;
            if (b >= 0) {
                const unsigned byte = (buf[b >> 2] >> (24 - 8 * (b & 3))) & 0xff;
                crc = ((crc << 8) ^ crc_tab[((crc >> 8) ^ byte) & 0xff]) & 0xffff;
            }
        }
        unsigned m = 1;
        for (int i = 0; i < C; i++) m = ((m << 8) ^ crc_tab[(m >> 8) & 0xff]) & 0xffff;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned partner = __shfl_up(crc, d);
            if ((lane & (2 * d - 1)) == 2 * d - 1) crc = gf_mulmod(partner, m) ^ crc;
            m = gf_mulmod(m, m);
        }
        if (lane == 63) put_bits(buf, 8u * (unsigned)T, crc, 16);
    }
    __builtin_amdgcn_wave_barrier();

    // ---- store: buffer byte b <-> stream byte off - a + b
    {
        const int lo = a, hi = a + fbytes + 2;
        uint8_t *dst = out + (off - a);
        for (int w = lane; w < nwords; w += 64) {
            const unsigned v = buf[w];
            if (4 * w >= lo && 4 * w + 4 <= hi) reinterpret_cast<unsigned *>(dst)[w] = __builtin_bswap32(v);
            else
                for (int b = 0; b < 4; b++) if (4 * w + b >= lo && 4 * w + b < hi) dst[4 * w + b] = (uint8_t)(v >> (24 - 8 * b));
        }
    }
}
} // namespace fl
} // namespace

// ---------------------------------------------------------------------------------------------------------------- launcher
size_t jt_flac_rec_bytes(int64_t nframes) { return sizeof(fl::Rec) * (size_t)nframes + 256; }

void launch_flac_analyse(const int16_t *pcm, int64_t n, int sr_code, int sr_extra_bytes, int sr_extra_val, void *recs,
                         long long *offs, void *summary, hipStream_t s)
{
    const int64_t nframes = (n + fl::BS - 1) / fl::BS, nfull = n / fl::BS;
    const fl::StreamCodes st{sr_code, sr_extra_bytes, sr_extra_val};
    if (nfull > 0)
        hipLaunchKernelGGL(fl::k_flac_analyse<true>, dim3((unsigned)((nfull + fl::WAVES - 1) / fl::WAVES)), dim3(256), 0, s,
                           pcm, n, (int64_t)0, nfull, st, (fl::Rec *)recs);
    if (nframes > nfull)
        hipLaunchKernelGGL(fl::k_flac_analyse<false>, dim3(1), dim3(256), 0, s, pcm, n, nfull, nframes, st, (fl::Rec *)recs);
    hipLaunchKernelGGL(fl::k_flac_scan, dim3(1), dim3(1024), 0, s, (const fl::Rec *)recs, nframes, offs, (fl::Summary *)summary);
    JT_HIP(hipGetLastError());
}

void launch_flac_emit(const int16_t *pcm, int64_t n, int sr_code, int sr_extra_bytes, int sr_extra_val, const void *recs,
                      const long long *offs, uint8_t *out, void *summary, hipStream_t s)
{
    const int64_t nframes = (n + fl::BS - 1) / fl::BS, nfull = n / fl::BS;
    const fl::StreamCodes st{sr_code, sr_extra_bytes, sr_extra_val};
    if (nfull > 0)
        hipLaunchKernelGGL(fl::k_flac_emit<true>, dim3((unsigned)((nfull + fl::WAVES - 1) / fl::WAVES)), dim3(256), 0, s,
                           pcm, n, (int64_t)0, nfull, st, (const fl::Rec *)recs, offs, out, (fl::Summary *)summary);
    if (nframes > nfull)
        hipLaunchKernelGGL(fl::k_flac_emit<false>, dim3(1), dim3(256), 0, s, pcm, n, nfull, nframes, st, (const fl::Rec *)recs,
                           offs, out, (fl::Summary *)summary);
    JT_HIP(hipGetLastError());
}
