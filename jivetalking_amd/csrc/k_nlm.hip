// k_nlm.hip — anlmdn (FFmpeg af_anlmdn.c; filters.go:95-100,811-816: anlmdn=s=0.00001:p=0.0060:r=0.0020:m=3) for gfx950.
// The one ALU-bound stage of the path: 2S patch comparisons per sample (S = r*fs = 96 @48 kHz), each a running SSD over a
// 2K+1 = 577-sample patch.
//
// FFmpeg's structure: hops of H = 2K+1 outputs; per hop and per search offset a running patch distance `cache[j]` that is
// seeded exactly at the hop start and then updated with two squared differences per step (a recurrence ALONG TIME for a
// fixed offset); per output, weights exp(-d*sw) (via a 2^20-entry LUT) are accumulated across the offsets.
//
// k_anlmdn_pair (fast path, 2S = 192 or 384 and K % 4 == 0: the 48 kHz and 96 kHz defaults; three or six offsets per lane):
//   ONE WAVE PER PAIR OF HOPS, no barriers.  Lane l owns 3 ADJACENT offsets, so the sample streams each offset consumes
//   (f[j+K], f[j-K-1]) overlap between its offsets and between consecutive steps: per step a lane reads one new value per
//   stream into a 4-slot register ring.  The two hops of the pair ride in the two halves of packed-f32 registers
//   (v_pk_add_f32 / v_pk_mul_f32).  The patch-distance recurrence keeps FFmpeg's exact f32 operation order (no FMA
//   contraction), so the engagement decisions (`w >= smooth`) are the reference's; when no lane of the wave has a distance
//   under the cut the weight stage is skipped and the output is the input sample (what FFmpeg computes: (0 + x) / (0 + 1)).
//   When engaged, the per-output sums over the 2S offsets are a lane-local sum of 3 terms followed by a DPP row/bank
//   reduction over the 64 lanes (deterministic, but not FFmpeg's sequential order: results agree to f32 round-off of the
//   weighted mean).  LDS per wave: the pair's input window, stored interleaved {hop A, hop B} so a packed operand is one
//   8-byte read (11 KB @48 kHz).  HBM traffic: 1 read (+ halo) and 1 write per
//   sample.
// k_anlmdn (generic path, any K/S): one workgroup per hop, one thread per offset, weights transposed through an LDS tile and
//   summed in FFmpeg's sequential offset order.
#include "jt_internal.h"
#include <type_traits>

// ------------------------------------------------------------------ wave-per-hop fast path
#define JT_DPP(v, ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (rmask), 0xf, false))
// wave64 sum, total valid in lane 63 (gfx9 DPP: quad_perm x2, row_shr 4/8, row_bcast 15/31)
__device__ inline float wave_sum63(float v)
{
    v += JT_DPP(v, 0xb1, 0xf);     // quad_perm:[1,0,3,2]
    v += JT_DPP(v, 0x4e, 0xf);     // quad_perm:[2,3,0,1]
    v += JT_DPP(v, 0x114, 0xf);    // row_shr:4
    v += JT_DPP(v, 0x118, 0xf);    // row_shr:8
    v += JT_DPP(v, 0x142, 0xa);    // row_bcast:15 -> rows 1,3
    v += JT_DPP(v, 0x143, 0xc);    // row_bcast:31 -> rows 2,3
    return v;
}

// ------------------------------------------------------------------ wave-per-hop-PAIR fast path (packed f32)
// Same mapping as k_anlmdn_wave, but a wave carries TWO consecutive hops in the two halves of packed-f32 registers
// (v_pk_add_f32 / v_pk_mul_f32: CDNA's full FP32 rate needs packed issue), and the weight stage is skipped for an output
// when no lane of the wave has a patch distance under the smoothing cut (FFmpeg's `if (w >= smooth) continue;` taken for
// all 2S offsets: the output is the input sample).  On speech at the reference's strength (s = 1e-5) that is the common
// case, so the steady-state cost is the patch-distance recurrence alone: 2 sub, 2 mul, 2 add per offset, exact f32 order.
typedef float f2 __attribute__((ext_vector_type(2)));
// stream ring size for NOFF offsets per lane: at least NOFF + 1 slots (one value is fetched ahead) and a divisor of the 8-step unroll
__host__ __device__ constexpr int nlm_ring(int noff) { return noff + 1 <= 4 ? 4 : 8; }

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int NLM_PADF = 4, NLM_PADB = 24;
constexpr bool NLM_SMEM_CENTRE = false;          // centre samples by scalar loads (true) or LDS broadcast reads (false)      // zero/real-sample padding of the LDS window: no index clamps in the loops

// Four consecutive uniform ("centre") samples of both hops starting at tile-relative position x: one scalar x4 load per hop
// from global memory in the interior (wave-uniform address), LDS broadcast reads at the file edges (zero-padded window).
template <bool INTERIOR>
__device__ inline void nlm_centre4(const float *__restrict__ in, int64_t gbase, const f2 *fw, int x, int H, f2 (&c)[4])
{
    if (INTERIOR) {
        const f4u a = *reinterpret_cast<const f4u *>(in + gbase + x);
        const f4u b = *reinterpret_cast<const f4u *>(in + gbase + x + H);
        c[0] = f2{a.x, b.x}; c[1] = f2{a.y, b.y}; c[2] = f2{a.z, b.z}; c[3] = f2{a.w, b.w};
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = fw[x + u];
    }
}

// One step of the patch-distance recurrence + the engagement test for both hops of the pair (v = unrolled position, ring
// slot of offset q is (q + v) % R).  `first` = hop start (no recurrence update).
template <int NOFF, int V, bool FIRST>
__device__ inline void nlm_step(f2 (&cache)[NOFF], f2 (&hi)[nlm_ring(NOFF)], f2 (&lo)[nlm_ring(NOFF)], const f2 cm, const f2 cp, const f2 *pw,
                                int i, int d0, int K, int S, int H, const f2 *fw,
                                float *__restrict__ out, int64_t hs, int64_t n, float sw, float smooth, float lut_scale,
                                float neg_inv_scale_log2e, float dthr, int lane)
{
    constexpr int R = nlm_ring(NOFF);
    if (!FIRST) {
#pragma unroll
        for (int q = 0; q < NOFF; ++q) {
            const f2 a = cm - lo[(q + V) % R];
            const f2 b = cp - hi[(q + V) % R];
            cache[q] = cache[q] + (-(a * a) + b * b);
        }
    }
    // refill the slot offset 0 just released with what offset NOFF-1 reads two steps ahead: f[i+2+d0+NOFF-1 (+K | -K-1)];
    // pw points at f[i0 + d0 + R] of the block (the slot offset 0 just released takes the value R positions ahead) so the
    // offsets below are compile-time constants
    hi[V % R] = pw[V + K];
    lo[V % R] = pw[V - K - 1];
    float dmin = 3.0e38f;
#pragma unroll
    for (int q = 0; q < NOFF; ++q) dmin = fminf(dmin, fminf(cache[q].x, cache[q].y));
    if (__any(dmin < dthr)) {
        // some offset may contribute (or a distance went negative by round-off, which is also < dthr): FFmpeg's clamp
        // `if (distance < 0) cache = distance = 0`, the exact per-offset test, weights, lane-local sums, wave reductions
        // The kernel is VALU-bound, so the stage is written for instruction count: both hops ride in packed registers, FFmpeg's
        // `if (w >= smooth) continue;` is a select (a skipped offset contributes weight 0; P + 0*f and Q + 0 are exact because the
        // sums start at +0 and never become -0), and the four wave sums share one transposed reduction.
        f2 Pxy = f2{0.f, 0.f}, Qxy = f2{0.f, 0.f};
        const f2 sw2 = f2{sw, sw}, ls2 = f2{lut_scale, lut_scale}, ns2 = f2{neg_inv_scale_log2e, neg_inv_scale_log2e};
#pragma unroll
        for (int q = 0; q < NOFF; ++q) {
            cache[q].x = __builtin_amdgcn_fmed3f(cache[q].x, 0.f, 3.0e38f);
            cache[q].y = __builtin_amdgcn_fmed3f(cache[q].y, 0.f, 3.0e38f);
            const int xc = i + d0 + q;
            const f2 w = cache[q] * sw2;
            f2 idx = w * ls2;
            idx.x = truncf(idx.x); idx.y = truncf(idx.y);
            const f2 ex = idx * ns2;                                                     // weight_lut[idx] = expf(-idx / scale)
            f2 wt;
            wt.x = w.x >= smooth ? 0.f : __builtin_amdgcn_exp2f(ex.x);
            wt.y = w.y >= smooth ? 0.f : __builtin_amdgcn_exp2f(ex.y);
            Pxy = Pxy + wt * fw[xc];
            Qxy = Qxy + wt;
        }
        // after the two quad steps lane (4m + j) holds the quad sum of value j (j = 0..3: Px, Qx, Py, Qy), the row steps keep j
        // in place, two cross-row permutes finish: lanes 60..63 = Px, Qx, Py, Qy
        {
            const bool odd = lane & 1, up = lane & 2;
            const float s1 = odd ? Qxy.x : Pxy.x, o1 = odd ? Pxy.x : Qxy.x;
            const float s2 = odd ? Qxy.y : Pxy.y, o2 = odd ? Pxy.y : Qxy.y;
            const float x = s1 + JT_DPP(o1, 0xb1, 0xf);          // quad_perm:[1,0,3,2]
            const float y = s2 + JT_DPP(o2, 0xb1, 0xf);
            const float s3 = up ? y : x, o3 = up ? x : y;
            float z = s3 + JT_DPP(o3, 0x4e, 0xf);                // quad_perm:[2,3,0,1]
            z += JT_DPP(z, 0x114, 0xf);                          // row_shr:4
            z += JT_DPP(z, 0x118, 0xf);                          // row_shr:8: lanes 12..15 of each row = row sums
            z += __shfl_xor(z, 16, 64);
            z += __shfl_xor(z, 32, 64);
            const float pnum = JT_DPP(z, 0x111, 0xf);            // row_shr:1: lane 61 <- Px, lane 63 <- Py
            const bool isA = lane == 61;
            if ((isA || lane == 63) && i - S < H) {
                const int64_t o = hs + (i - S) + (isA ? 0 : H);
                if (o >= 0 && o < n) out[o] = __fadd_rn(pnum, isA ? fw[i].x : fw[i].y) / __fadd_rn(z, 1.f);
            }
        }
    }
}

template <int NOFF, bool INTERIOR>
__device__ inline void nlm_pair_body(const float *__restrict__ in, float *__restrict__ out, int64_t n, int64_t hs, int64_t gwin,
                                     const f2 *fw, int K, int S, int H, float sw, float smooth, float lut_scale,
                                     float neg_inv_scale_log2e, int lane)
{
    // Streams live in a ring of R >= NOFF+1 packed registers (4 for three offsets per lane, 8 for six): slot (v+q)%R is offset q at unrolled step v, and the spare slot
    // already holds the value offset NOFF-1 needs at the NEXT step, so every LDS refill is issued more than a full step before
    // its first use; the wave-uniform centre samples arrive by scalar x4 loads issued 4 steps ahead (ping-pong A/B sets).
    constexpr int R = nlm_ring(NOFF);
    static_assert(8 % R == 0 && R > NOFF, "the loops below advance 8 steps at a time: the ring must realign");
    const int j0 = lane * NOFF;
    const int d0 = j0 - S + (j0 >= S ? 1 : 0);
    const int64_t gbase = gwin + K;                       // global index of f[0] of hop A
    f2 cache[NOFF];
#pragma unroll
    for (int q = 0; q < NOFF; ++q) cache[q] = f2{0.f, 0.f};
    // ---- seed: compute_distance_ssd over k = -K..K (ascending, mul then add); 2K+1 = 4*(K/2) + 1 steps (K even)
    {
        f2 st[R], ca[4], cb[4];
#pragma unroll
        for (int q = 0; q < R; ++q) { const int x = S + d0 + q - K; st[q] = fw[x]; }   // slot NOFF = (offset NOFF-1, k = -K+1)
        nlm_centre4<INTERIOR>(in, gbase, fw, S - K, H, ca);
        const f2 *pw = fw + S + d0 + R - K;                    // refill source of step k: p[k + K] = position (step + R) of the ring
        for (int k0 = -K; k0 < K; k0 += 8) {
            nlm_centre4<INTERIOR>(in, gbase, fw, S + k0 + 4, H, cb);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int q = 0; q < NOFF; ++q) { const f2 dd = ca[u] - st[(q + u) % R]; cache[q] = cache[q] + dd * dd; }
                st[u % R] = pw[u];
            }
            nlm_centre4<INTERIOR>(in, gbase, fw, S + k0 + 8, H, ca);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int q = 0; q < NOFF; ++q) { const f2 dd = cb[u] - st[(q + 4 + u) % R]; cache[q] = cache[q] + dd * dd; }
                st[(4 + u) % R] = pw[4 + u];
            }
            pw += 8;
        }
        // k = K, the (2K+1)-th term: 2K % 8 == 0 puts it at ring position 0 with its centre already in ca[0]
#pragma unroll
        for (int q = 0; q < NOFF; ++q) { const f2 dd = ca[0] - st[q % R]; cache[q] = cache[q] + dd * dd; }
    }
    // ---- main recurrence.  hi[(q+v)%R] = f[i + d0 + q + K], lo[...] = f[i + d0 + q - K - 1] at step i, v = (i - S) % R
    f2 hi[R], lo[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int xh = S + d0 + q + K; hi[q] = fw[xh];
        const int xl = S + d0 + q - K - 1; lo[q] = fw[xl];         // first used at i = S+1 (front padding covers lane 0)
    }
    const float dthr = (smooth / sw) * 1.000002f;          // cache >= dthr  =>  fl(cache*sw) >= smooth (the exact test follows)
    const f2 zero2 = f2{0.f, 0.f};
    const f2 *pw = fw + S + d0 + R;                              // p[v] = f[S + d0 + R + v]
    nlm_step<NOFF, 0, true>(cache, hi, lo, zero2, zero2, pw, S, d0, K, S, H, fw, out, hs, n, sw, smooth, lut_scale,
                            neg_inv_scale_log2e, dthr, lane);
    // steps S+1 .. S+2K in blocks of 8 (2K % 8 == 0): v = 1..8, ring slot (v % R)
    f2 ma[4], pa[4], mb[4], pb[4];
    nlm_centre4<INTERIOR>(in, gbase, fw, S + 1 - K - 1, H, ma);
    nlm_centre4<INTERIOR>(in, gbase, fw, S + 1 + K, H, pa);
    for (int i0 = S + 1; i0 < H + S; i0 += 8) {
        pw = fw + i0 + d0 + R - 1;                                   // so that p[v], v = 1.., is step (i0 + v - 1)'s refill
        nlm_centre4<INTERIOR>(in, gbase, fw, i0 + 4 - K - 1, H, mb);
        nlm_centre4<INTERIOR>(in, gbase, fw, i0 + 4 + K, H, pb);
        nlm_step<NOFF, 1, false>(cache, hi, lo, ma[0], pa[0], pw, i0 + 0, d0, K, S, H, fw, out, hs, n, sw, smooth, lut_scale, neg_inv_scale_log2e, dthr, lane);
        nlm_step<NOFF, 2, false>(cache, hi, lo, ma[1], pa[1], pw, i0 + 1, d0, K, S, H, fw, out, hs, n, sw, smooth, lut_scale, neg_inv_scale_log2e, dthr, lane);
        nlm_step<NOFF, 3, false>(cache, hi, lo, ma[2], pa[2], pw, i0 + 2, d0, K, S, H, fw, out, hs, n, sw, smooth, lut_scale, neg_inv_scale_log2e, dthr, lane);
        nlm_step<NOFF, 4, false>(cache, hi, lo, ma[3], pa[3], pw, i0 + 3, d0, K, S, H, fw, out, hs, n, sw, smooth, lut_scale, neg_inv_scale_log2e, dthr, lane);
        nlm_centre4<INTERIOR>(in, gbase, fw, i0 + 8 - K - 1, H, ma);
        nlm_centre4<INTERIOR>(in, gbase, fw, i0 + 8 + K, H, pa);
        nlm_step<NOFF, 5, false>(cache, hi, lo, mb[0], pb[0], pw, i0 + 4, d0, K, S, H, fw, out, hs, n, sw, smooth, lut_scale, neg_inv_scale_log2e, dthr, lane);
        nlm_step<NOFF, 6, false>(cache, hi, lo, mb[1], pb[1], pw, i0 + 5, d0, K, S, H, fw, out, hs, n, sw, smooth, lut_scale, neg_inv_scale_log2e, dthr, lane);
        nlm_step<NOFF, 7, false>(cache, hi, lo, mb[2], pb[2], pw, i0 + 6, d0, K, S, H, fw, out, hs, n, sw, smooth, lut_scale, neg_inv_scale_log2e, dthr, lane);
        nlm_step<NOFF, 8, false>(cache, hi, lo, mb[3], pb[3], pw, i0 + 7, d0, K, S, H, fw, out, hs, n, sw, smooth, lut_scale, neg_inv_scale_log2e, dthr, lane);
    }
}

template <int NOFF>
__global__ void __launch_bounds__(64)
k_anlmdn_pair(const float *__restrict__ in, float *__restrict__ out, int64_t n, int K, int S, float sw, float smooth,
              float lut_scale, int64_t nhops)
{
    extern __shared__ float smem_nlm[];
    const int H = 2 * K + 1;
    const int NW = H + 2 * (K + S);
    const int NW2 = NW + H;                        // two consecutive hops share one window
    const int NWP = NLM_PADF + NW + NLM_PADB;
    // the window is stored INTERLEAVED: entry x = {f[x], f[x + H]} = the same position of hop A and hop B, so every packed operand
    // of the loops below is one 8-byte LDS read instead of two 4-byte reads and two moves
    f2 *win = reinterpret_cast<f2 *>(smem_nlm);    // [NWP]
    const int lane = threadIdx.x;
    const int64_t hopA = (int64_t)blockIdx.x * 2;
    const int64_t hs = hopA * H - (K + S);         // first output sample of hop A
    const int64_t gwin = hs - (K + S);             // global index of f[-K] of hop A
    for (int w = lane; w < NWP; w += 64) {
        const int64_t k = gwin - NLM_PADF + w, k2 = k + H;
        win[w] = f2{(k >= 0 && k < n) ? in[k] : 0.f, (k2 >= 0 && k2 < n) ? in[k2] : 0.f};
    }
    __syncthreads();
    const f2 *fw = win + NLM_PADF + K;             // fw[i] = {f[i], f[i + H]}, i in [-K, NW-K)
    // default output = input (every offset skipped); outputs with contributing offsets are overwritten from inside the loop.
    // The fence orders the two stores to the same address (they come from different lanes of this wave).
    for (int t = lane; t < 2 * H; t += 64) {
        const int64_t o = hs + t;
        if (o >= 0 && o < n) out[o] = t < H ? fw[t + S].x : fw[t - H + S].y;
    }
    __threadfence();
    const float nisl = -1.4426950408889634f / lut_scale;
    const bool interior = gwin - NLM_PADF >= 0 && gwin + NW2 + NLM_PADB <= n;
    if (interior && NLM_SMEM_CENTRE) nlm_pair_body<NOFF, true>(in, out, n, hs, gwin, fw, K, S, H, sw, smooth, lut_scale, nisl, lane);
    else nlm_pair_body<NOFF, false>(in, out, n, hs, gwin, fw, K, S, H, sw, smooth, lut_scale, nisl, lane);
}

// ------------------------------------------------------------------ hop-pair kernel, three offsets per lane, deferred weights
// k_anlmdn_pair3<NOFF> (2S = 192 / 384: the 48 and 96 kHz defaults).  Same recurrence, mapping and LDS window as k_anlmdn_pair, but the weight
// stage no longer runs inside the step loop.  Measured on speech (tools/nlm_engage.py): 47 % of the steps have a contributing
// offset, and in 97 % of those only the offsets -6..-1, +1..+6 contribute (lanes 30..33: neighbouring shifts of a low-pass
// signal), so a wave-wide weight stage spends 64 lanes on ~9 useful values.  Here:
//   * the step loop only advances the recurrence and keeps, per block of 8 steps, the running minimum of the lane's distances
//     (the 8 x 3 packed distances of the block stay in registers);
//   * a block with a distance under the cut on lanes 30..33 only (and no negative distance) parks those four lanes' 24 packed
//     distances in LDS (12 ds_write_b128 by four lanes);
//   * every 64 steps the wave turns round: lane t owns output t of the block, walks the 12 near offsets in FFmpeg's ascending
//     order (sequential P / Q sums, as af_anlmdn.c does), divides, and the 64 outputs of each hop leave as one coalesced store.
//     Outputs without a contributing offset are the input sample: (0 + x) / (0 + 1);
//   * any other block (a far offset under the cut, or a distance that went negative by round-off, which FFmpeg clamps in
//     place) is replayed from the block's starting distances by the exact per-step path: clamp, wave-wide weights, DPP
//     reduction; its outputs join the same coalesced store.
// Every output is written exactly once (k_anlmdn_pair pre-stores the input and overwrites engaged outputs with scattered
// 4-byte stores from two lanes: WRITE_SIZE 2.4x the output size in profiles/r01_pmc_traffic.json).
#ifndef JT_NLM3_CB
#define JT_NLM3_CB 64
#endif
constexpr int NLM3_CB = JT_NLM3_CB, NLM3_B8 = NLM3_CB / 8;
// NOFF offsets per lane (3: 2S = 192, the 48 kHz default; 6: 2S = 384, 96 kHz).  The near lanes are the ones that own offsets -6 .. +6:
// four lanes of three offsets, or two lanes of six -- twelve parked distances per step either way.
template <int NOFF> struct Nlm3 {
    static constexpr int S = 32 * NOFF, NL = 12 / NOFF, NEAR0 = 32 - NL / 2, R = nlm_ring(NOFF);
    static constexpr unsigned long long NEARMASK = ((1ull << NL) - 1ull) << NEAR0;
};

#ifdef JT_NLM_PROFILE
__device__ unsigned long long g_nlm_prof[8];      // wave clocks: window fill, seed, steps, park, replay, turn-round; [6] blocks replayed, [7] blocks parked
#define NLM_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define NLM_ACC(k, a, b) (nlm_pr[k] += (b) - (a))
#define NLM_PROF_DECL unsigned long long nlm_pr[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define NLM_PROF_FLUSH do { if (threadIdx.x == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_nlm_prof[k_], nlm_pr[k_]); } while (0)
#else
#define NLM_T(var)
#define NLM_ACC(k, a, b)
#define NLM_PROF_DECL
#define NLM_PROF_FLUSH
#endif

template <int NOFF, int V>
__device__ inline void nlm3_step(f2 (&cache)[NOFF], f2 (&hi)[nlm_ring(NOFF)], f2 (&lo)[nlm_ring(NOFF)], const f2 cm, const f2 cp, const f2 *pw, int K,
                                 f2 (&cs)[8][NOFF], float &mn)
{
    constexpr int R = nlm_ring(NOFF);
#pragma unroll
    for (int q = 0; q < NOFF; ++q) {
        const f2 a = cm - lo[(q + V) % R];
        const f2 b = cp - hi[(q + V) % R];
        cache[q] = cache[q] + (-(a * a) + b * b);
    }
    hi[V % R] = pw[V + K];
    lo[V % R] = pw[V - K - 1];
#pragma unroll
    for (int q = 0; q < NOFF; ++q) {
        cs[V - 1][q] = cache[q];
        mn = fminf(fminf(mn, cache[q].x), cache[q].y);
    }
}

// The wave-wide weight stage of one step (both hops): FFmpeg's clamp, exact per-offset test, weights, lane-local sums, DPP
// reduction; lanes 61 / 63 leave the two outputs in dst->x / dst->y (LDS).
template <int NOFF>
__device__ inline void nlm3_dense_stage(f2 (&cache)[NOFF], int i, int d0, const f2 *fw, f2 *dst, float sw, float smooth,
                                        float lut_scale, float nisl, int lane)
{
    f2 Pxy = f2{0.f, 0.f}, Qxy = f2{0.f, 0.f};
    const f2 sw2 = f2{sw, sw}, ls2 = f2{lut_scale, lut_scale}, ns2 = f2{nisl, nisl};
#pragma unroll
    for (int q = 0; q < NOFF; ++q) {
        cache[q].x = __builtin_amdgcn_fmed3f(cache[q].x, 0.f, 3.0e38f);
        cache[q].y = __builtin_amdgcn_fmed3f(cache[q].y, 0.f, 3.0e38f);
        const f2 w = cache[q] * sw2;
        f2 idx = w * ls2;
        idx.x = truncf(idx.x); idx.y = truncf(idx.y);
        const f2 ex = idx * ns2;
        f2 wt;
        wt.x = w.x >= smooth ? 0.f : __builtin_amdgcn_exp2f(ex.x);
        wt.y = w.y >= smooth ? 0.f : __builtin_amdgcn_exp2f(ex.y);
        Pxy = Pxy + wt * fw[i + d0 + q];
        Qxy = Qxy + wt;
    }
    const bool odd = lane & 1, up = lane & 2;
    const float s1 = odd ? Qxy.x : Pxy.x, o1 = odd ? Pxy.x : Qxy.x;
    const float s2 = odd ? Qxy.y : Pxy.y, o2 = odd ? Pxy.y : Qxy.y;
    const float x = s1 + JT_DPP(o1, 0xb1, 0xf);
    const float y = s2 + JT_DPP(o2, 0xb1, 0xf);
    const float s3 = up ? y : x, o3 = up ? x : y;
    float z = s3 + JT_DPP(o3, 0x4e, 0xf);
    z += JT_DPP(z, 0x114, 0xf);
    z += JT_DPP(z, 0x118, 0xf);
    z += __shfl_xor(z, 16, 64);
    z += __shfl_xor(z, 32, 64);
    const float pnum = JT_DPP(z, 0x111, 0xf);            // lane 61 <- Px (z = Qx), lane 63 <- Py (z = Qy)
    if (lane == 61) reinterpret_cast<float *>(dst)[0] = __fadd_rn(pnum, fw[i].x) / __fadd_rn(z, 1.f);
    if (lane == 63) reinterpret_cast<float *>(dst)[1] = __fadd_rn(pnum, fw[i].y) / __fadd_rn(z, 1.f);
}

// Exact per-step replay of `count` steps starting at step index i0 (slot vs0 of the consumer block), reading the window directly.
template <int NOFF>
__device__ inline void nlm3_slow_steps(f2 (&cache)[NOFF], int i0, int count, int d0, int K, const f2 *fw, f2 *dslot, int vs0,
                                       unsigned long long &dmask, float sw, float smooth, float lut_scale, float nisl, float dthr, int lane)
{
    for (int u = 0; u < count; ++u) {
        const int i = i0 + u;
        const f2 cm = fw[i - K - 1], cp = fw[i + K];
#pragma unroll
        for (int q = 0; q < NOFF; ++q) {
            const f2 a = cm - fw[i + d0 + q - K - 1];
            const f2 b = cp - fw[i + d0 + q + K];
            cache[q] = cache[q] + (-(a * a) + b * b);
        }
        float dmin = 3.0e38f;
#pragma unroll
        for (int q = 0; q < NOFF; ++q) dmin = fminf(dmin, fminf(cache[q].x, cache[q].y));
        if (__any(dmin < dthr)) {
            nlm3_dense_stage<NOFF>(cache, i, d0, fw, dslot + vs0 + u, sw, smooth, lut_scale, nisl, lane);
            dmask |= 1ull << (vs0 + u);
        }
    }
}

// Turn-round: lane t finishes output t of the block (v = vb + t) for both hops and the block leaves as two coalesced stores.
template <int NOFF>
__device__ inline void nlm3_consume(const f2 *fw, const f2 *slot, const f2 *dslot, unsigned long long emask, unsigned long long dmask,
                                    int vb, int count, int H, float *__restrict__ out, int64_t hs, int64_t n,
                                    float sw, float smooth, float lut_scale, float nisl, int lane)
{
    constexpr int S_ = Nlm3<NOFF>::S;
    const int v = vb + lane, i = S_ + v;
    f2 o = fw[i];
    if (emask | dmask) {
        if (emask) {
            f2 P = f2{0.f, 0.f}, Q = f2{0.f, 0.f};
            const f2 sw2 = f2{sw, sw}, ls2 = f2{lut_scale, lut_scale}, ns2 = f2{nisl, nisl};
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                constexpr int j0 = Nlm3<NOFF>::NEAR0 * NOFF;
                const int j = j0 + e;
                const int d = j - S_ + (j >= S_ ? 1 : 0);
                const f2 c = slot[e * NLM3_CB + (lane & (NLM3_CB - 1))];
                const f2 w = c * sw2;
                f2 idx = w * ls2;
                idx.x = truncf(idx.x); idx.y = truncf(idx.y);
                const f2 ex = idx * ns2;
                f2 wt;
                wt.x = w.x >= smooth ? 0.f : __builtin_amdgcn_exp2f(ex.x);
                wt.y = w.y >= smooth ? 0.f : __builtin_amdgcn_exp2f(ex.y);
                P = P + wt * fw[i + d];
                Q = Q + wt;
            }
            f2 r;
            r.x = __fadd_rn(P.x, o.x) / __fadd_rn(Q.x, 1.f);
            r.y = __fadd_rn(P.y, o.y) / __fadd_rn(Q.y, 1.f);
            if ((emask >> lane) & 1) o = r;
        }
        if ((dmask >> lane) & 1) o = dslot[lane & (NLM3_CB - 1)];
    }
    if (lane < count) {
        const int64_t oa = hs + v, ob = oa + H;
        if (oa >= 0 && oa < n) out[oa] = o.x;
        if (ob >= 0 && ob < n) out[ob] = o.y;
    }
}

template <int NOFF>
__global__ void __launch_bounds__(64)
k_anlmdn_pair3(const float *__restrict__ in, float *__restrict__ out, int64_t n, int K, int Sreal, float sw, float smooth, float lut_scale)
{
    extern __shared__ float smem_nlm[];
    constexpr int S = Nlm3<NOFF>::S, R = Nlm3<NOFF>::R, NEAR0 = Nlm3<NOFF>::NEAR0, NL = Nlm3<NOFF>::NL;
    const int H = 2 * K + 1;
    const int NW = H + 2 * (K + S);
    const int NWP = NLM_PADF + NW + NLM_PADB;
    f2 *win = reinterpret_cast<f2 *>(smem_nlm);            // [NWP] interleaved {hop A, hop B}
    f2 *slot = win + NWP;                                  // [12][64] parked near-lane distances of the current block
    f2 *dslot = slot + 12 * NLM3_CB;              // [65] outputs of replayed steps (+1: the hop's first step)
    const int lane = threadIdx.x;
    const int64_t hopA = (int64_t)blockIdx.x * 2;
    // S is the lane layout's radius (64 lanes x NOFF offsets = -S..-1, 1..S); the filter's own radius Sreal <= S sets where the hops
    // start, and the offsets beyond it are dummies: their distance is +inf from the seed on, so they never engage and weigh nothing
    const int64_t hs = hopA * H - (K + Sreal);
    const int64_t gwin = hs - (K + S);
    NLM_PROF_DECL;
    NLM_T(t_a);
    constexpr int NLD = NOFF == 3 ? 24 : 44;                // window rows of 64 entries fetched in one batch
    if (gwin - NLM_PADF >= 0 && gwin - NLM_PADF + NWP + H <= n && NWP <= 64 * NLD) {
        // interior tile: every load of the window is issued before the first one is waited for (one memory round trip per wave
        // instead of one per 64 entries); rows past the window re-read its last entry and are not stored
        const float *src = in + (gwin - NLM_PADF);
        float ax[NLD], ay[NLD];
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int w = min(lane + 64 * r, NWP - 1);
            ax[r] = __builtin_nontemporal_load(src + w);
            ay[r] = __builtin_nontemporal_load(src + w + H);
        }
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int w = lane + 64 * r;
            if (w < NWP) win[w] = f2{ax[r], ay[r]};
        }
    } else {
        for (int w = lane; w < NWP; w += 64) {
            const int64_t k = gwin - NLM_PADF + w, k2 = k + H;
            win[w] = f2{(k >= 0 && k < n) ? in[k] : 0.f, (k2 >= 0 && k2 < n) ? in[k2] : 0.f};
        }
    }
    __syncthreads();
    NLM_T(t_b); NLM_ACC(0, t_a, t_b);
    const f2 *fw = win + NLM_PADF + K;
    const float nisl = -1.4426950408889634f / lut_scale;
    const int j0 = lane * NOFF;
    const int d0 = j0 - S + (j0 >= S ? 1 : 0);
    f2 cache[NOFF];
#pragma unroll
    for (int q = 0; q < NOFF; ++q) cache[q] = f2{0.f, 0.f};
    // ---- seed: compute_distance_ssd over k = -K..K (ascending, mul then add).  The stream ring holds 8 positions and the centre
    // samples arrive 8 steps ahead: a seed step is 9 packed instructions, so the 4-slot ring of the step loop (refill two steps
    // ahead) would leave every refill ~70 cycles to land -- less than an LDS round trip
    {
        f2 st[8], ca[8], cb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int x = S + d0 + q - K; st[q] = fw[x]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) ca[u] = fw[S - K + u];
        const f2 *pw = fw + S + d0 + 8 - K;                    // pw[u]: the stream position 8 ahead of step u
        const f2 *pc = fw + S - K + 8;
        const int n16 = (2 * K) / 16;                                        // 2K + 1 terms: n16 rounds of sixteen, then 1 .. 16 more
        for (int it16 = 0; it16 < n16; ++it16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) cb[u] = pc[u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int q = 0; q < NOFF; ++q) { const f2 dd = ca[u] - st[(q + u) % 8]; cache[q] = cache[q] + dd * dd; }
                st[u % 8] = pw[u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) ca[u] = pc[8 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int q = 0; q < NOFF; ++q) { const f2 dd = cb[u] - st[(q + u) % 8]; cache[q] = cache[q] + dd * dd; }
                st[u % 8] = pw[8 + u];
            }
            pw += 16; pc += 16;
        }
        // the remaining 1 .. 16 terms (k ascending as before) straight from the window: one term when 2K is a multiple of 16
#pragma unroll 1
        for (int t = 16 * n16; t <= 2 * K; ++t) {
            const f2 c = fw[S - K + t];
#pragma unroll
            for (int q = 0; q < NOFF; ++q) { const f2 dd = c - fw[S + d0 + q - K + t]; cache[q] = cache[q] + dd * dd; }
        }
    }
    if (Sreal < S) {
        const float inf = __builtin_inff();
#pragma unroll
        for (int q = 0; q < NOFF; ++q) {
            const int j = j0 + q, d = j - S + (j >= S ? 1 : 0);
            if (d < -Sreal || d > Sreal) cache[q] = f2{inf, inf};
        }
    }
    const float dthr = (smooth / sw) * 1.000002f;
    NLM_T(t_c); NLM_ACC(1, t_b, t_c);
    // ---- first step of the hop (no recurrence update): exact path, one output per hop
    {
        float dmin = 3.0e38f;
#pragma unroll
        for (int q = 0; q < NOFF; ++q) dmin = fminf(dmin, fminf(cache[q].x, cache[q].y));
        const bool eng = __any(dmin < dthr);
        if (eng) nlm3_dense_stage<NOFF>(cache, S, d0, fw, dslot + NLM3_CB, sw, smooth, lut_scale, nisl, lane);
        if (lane == 0) {
            const f2 o = eng ? dslot[NLM3_CB] : fw[S];
            if (hs >= 0 && hs < n) out[hs] = o.x;
            if (hs + H >= 0 && hs + H < n) out[hs + H] = o.y;
        }
    }
    // ---- steps v = 1 .. 2K in blocks of 8; hi[(q+v)%4] = f[i + d0 + q + K], lo[...] = f[i + d0 + q - K - 1], v = (i - S) % 4
    f2 hi[R], lo[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const int xh = S + d0 + q + K; hi[q] = fw[xh];
        const int xl = S + d0 + q - K - 1; lo[q] = fw[xl];
    }
    // the FIRST step's refill (slot 0 takes the values R positions ahead)
    hi[0] = fw[S + d0 + R + K];
    lo[0] = fw[S + d0 + R - K - 1];
    f2 ma[4], pa[4], mb[4], pb[4];
    nlm_centre4<false>(in, 0, fw, S + 1 - K - 1, H, ma);
    nlm_centre4<false>(in, 0, fw, S + 1 + K, H, pa);
    unsigned long long emask = 0, dmask = 0;
    const int nb8 = (2 * K) / 8, r8 = 2 * K - 8 * nb8;       // blocks of eight steps, then r8 = 0, 2, 4 or 6 steps by the exact per-step path
    const bool near = lane >= NEAR0 && lane < NEAR0 + NL;
    for (int b8 = 0; b8 < nb8; ++b8) {
        const int i0 = S + 1 + b8 * 8;
        const int it = b8 % NLM3_B8;
        const f2 *pw = fw + i0 + d0 + R - 1;
        f2 cs[8][NOFF], c0[NOFF];
        float mn = 3.0e38f;
        NLM_T(t_0);
#pragma unroll
        for (int q = 0; q < NOFF; ++q) c0[q] = cache[q];
        nlm_centre4<false>(in, 0, fw, i0 + 4 - K - 1, H, mb);
        nlm_centre4<false>(in, 0, fw, i0 + 4 + K, H, pb);
        nlm3_step<NOFF, 1>(cache, hi, lo, ma[0], pa[0], pw, K, cs, mn);
        nlm3_step<NOFF, 2>(cache, hi, lo, ma[1], pa[1], pw, K, cs, mn);
        nlm3_step<NOFF, 3>(cache, hi, lo, ma[2], pa[2], pw, K, cs, mn);
        nlm3_step<NOFF, 4>(cache, hi, lo, ma[3], pa[3], pw, K, cs, mn);
        nlm_centre4<false>(in, 0, fw, i0 + 8 - K - 1, H, ma);
        nlm_centre4<false>(in, 0, fw, i0 + 8 + K, H, pa);
        nlm3_step<NOFF, 5>(cache, hi, lo, mb[0], pb[0], pw, K, cs, mn);
        nlm3_step<NOFF, 6>(cache, hi, lo, mb[1], pb[1], pw, K, cs, mn);
        nlm3_step<NOFF, 7>(cache, hi, lo, mb[2], pb[2], pw, K, cs, mn);
        nlm3_step<NOFF, 8>(cache, hi, lo, mb[3], pb[3], pw, K, cs, mn);
        const unsigned long long bal = __ballot(mn < dthr);
        NLM_T(t_1); NLM_ACC(2, t_0, t_1);
        if (bal) {
            if ((bal & ~Nlm3<NOFF>::NEARMASK) || __any(mn < 0.f)) {
#pragma unroll
                for (int q = 0; q < NOFF; ++q) cache[q] = c0[q];
                nlm3_slow_steps<NOFF>(cache, i0, 8, d0, K, fw, dslot, it * 8, dmask, sw, smooth, lut_scale, nisl, dthr, lane);
                NLM_T(t_2); NLM_ACC(4, t_1, t_2); NLM_ACC(6, 0ull, 1ull);
            } else {
                if (near) {
#pragma unroll
                    for (int q = 0; q < NOFF; ++q) {
#pragma unroll
                        for (int u = 0; u < 8; u += 2)
                            *reinterpret_cast<float4 *>(slot + ((lane - NEAR0) * NOFF + q) * NLM3_CB + it * 8 + u) =
                                make_float4(cs[u][q].x, cs[u][q].y, cs[u + 1][q].x, cs[u + 1][q].y);
                    }
                }
                emask |= 0xFFull << (it * 8);
                NLM_T(t_2); NLM_ACC(3, t_1, t_2); NLM_ACC(7, 0ull, 1ull);
            }
        }
        NLM_T(t_3);
        if (it == NLM3_B8 - 1 || (b8 == nb8 - 1 && r8 == 0)) {
            nlm3_consume<NOFF>(fw, slot, dslot, emask, dmask, 1 + (b8 - it) * 8, (it + 1) * 8, H, out, hs, n, sw, smooth, lut_scale, nisl, lane);
            emask = dmask = 0;
            NLM_T(t_4); NLM_ACC(5, t_3, t_4);
        }
    }
    if (r8) {
        // 2K not a multiple of 8 (44.1 kHz: K = 265): the hop's last r8 steps, and whatever the last turn-round block still holds
        const int it = nb8 % NLM3_B8;
        nlm3_slow_steps<NOFF>(cache, S + 1 + nb8 * 8, r8, d0, K, fw, dslot, it * 8, dmask, sw, smooth, lut_scale, nisl, dthr, lane);
        nlm3_consume<NOFF>(fw, slot, dslot, emask, dmask, 1 + (nb8 - it) * 8, it * 8 + r8, H, out, hs, n, sw, smooth, lut_scale, nisl, lane);
    }
    NLM_PROF_FLUSH;
}

// ------------------------------------------------------------------ generic path (any K, S)
constexpr int NLM_TI = 64;     // outputs per weight tile

__global__ void k_anlmdn(const float *__restrict__ in, float *__restrict__ out, int64_t n, int K, int S,
                         float sw, float smooth, float lut_scale, int64_t nhops)
{
    extern __shared__ float smem[];
    const int H = 2 * K + 1;
    const int NW = H + 2 * (K + S);
    const int S2 = 2 * S;
    float *win = smem;                       // [NW]
    float *wt = smem + ((NW + 3) & ~3);      // [NLM_TI][S2 + 1]
    const int tid = threadIdx.x;             // offset index j in [0, 2S)
    const int64_t hop = (int64_t)blockIdx.x;
    const int64_t hs = hop * H - (K + S);    // first output sample of this hop
    for (int w = tid; w < NW; w += blockDim.x) {
        int64_t k = hs - (K + S) + w;
        win[w] = (k >= 0 && k < n) ? in[k] : 0.f;
    }
    __syncthreads();
    const float *f = win + K;                // f[i], i in [-K, N-K)
    const int dj = tid - S + (tid >= S ? 1 : 0);   // neighbour offset relative to the centre
    float cache = 0.f;
    const int wstride = S2 + 1;
    for (int i0 = S; i0 < H + S; i0 += NLM_TI) {
        const int ti_n = min(NLM_TI, H + S - i0);
        if (tid < S2) {
            for (int ii = 0; ii < ti_n; ++ii) {
                const int i = i0 + ii;
                if (i == S) {
                    float dist = 0.f;
                    const float *f1 = f + i, *f2 = f + i + dj;
                    for (int k = -K; k <= K; ++k) {
                        float dd = __fsub_rn(f1[k], f2[k]);
                        dist = __fadd_rn(dist, __fmul_rn(dd, dd));
                    }
                    cache = dist;
                } else {
                    const int j = i + dj;
                    float a = __fsub_rn(f[i - K - 1], f[j - K - 1]);
                    float b = __fsub_rn(f[i + K], f[j + K]);
                    float t = __fadd_rn(-__fmul_rn(a, a), __fmul_rn(b, b));
                    cache = __fadd_rn(cache, t);
                }
                float distance = cache;
                if (distance < 0.f) cache = distance = 0.f;
                float w = __fmul_rn(distance, sw);
                float weight = 0.f;
                if (!(w >= smooth)) {
                    unsigned idx = (unsigned)__fmul_rn(w, lut_scale);
                    weight = __expf(-(float)idx / lut_scale);
                }
                wt[ii * wstride + tid] = weight;
            }
        }
        __syncthreads();
        if (tid < ti_n) {
            const int i = i0 + tid;
            const float *wr = wt + tid * wstride;
            float P = 0.f, Q = 0.f;
#pragma unroll 8
            for (int j = 0; j < S2; ++j) {
                const float w = wr[j];        // skipped offsets carry weight 0: adding 0*f and 0 leaves P, Q unchanged
                P = __fadd_rn(P, __fmul_rn(w, f[i - S + j + (j >= S ? 1 : 0)]));
                Q = __fadd_rn(Q, w);
            }
            P = __fadd_rn(P, f[i]);
            Q = __fadd_rn(Q, 1.f);
            int64_t o = hs + (i - S);
            if (o >= 0 && o < n) out[o] = P / Q;
        }
        __syncthreads();
    }
}

void launch_anlmdn(const float *in, float *out, int64_t n, int K, int S, float sw, float smooth, float lut_scale, hipStream_t s, const JtOpts &o)
{
    if (n <= 0) return;
    const int H = 2 * K + 1;
    const int NW = H + 2 * (K + S);
    int64_t nhops = (n + (K + S) + H - 1) / H;
    // 64 lanes x 3 adjacent offsets (radius up to 96: the 44.1 / 48 kHz defaults) or x 6 (up to 192: 88.2 / 96 kHz); a radius below the
    // layout's leaves dummy offsets at both ends, any patch length of 8 or more
    const int Sl = S <= 96 ? 96 : (S <= 192 ? 192 : 0);
    if (Sl && K >= 8 && !o.nlm_generic) {
        const int NWl = H + 2 * (K + Sl);
        size_t smem = sizeof(float) * 2 * (size_t)(NLM_PADF + NWl + NLM_PADB);      // interleaved {hop A, hop B} window
        const unsigned grid = (unsigned)((nhops + 1) / 2);
        const bool exact_layout = 2 * S == 2 * Sl;
        if (!(exact_layout && K % 4 == 0 && JT_AB_ON(o.nlm_old))) {               // (nlm_old: the round-1 kernel, JT_AB build only)
            smem += sizeof(float) * 2 * (size_t)(12 * NLM3_CB + NLM3_CB + 1);
#ifdef JT_NLM_PADSMEM
            smem += JT_NLM_PADSMEM;                // (tools/ab_builds.py: waves per CU by LDS, the occupancy sweep)
#endif
            JT_REQUIRE(smem <= 64 * 1024, JT_E_UNSUPPORTED, "anlmdn: window exceeds the wave-per-hop LDS budget");
            if (Sl == 96) hipLaunchKernelGGL((k_anlmdn_pair3<3>), dim3(grid), dim3(64), smem, s, in, out, n, K, S, sw, smooth, lut_scale);
            else hipLaunchKernelGGL((k_anlmdn_pair3<6>), dim3(grid), dim3(64), smem, s, in, out, n, K, S, sw, smooth, lut_scale);
#ifdef JT_NLM_PROFILE
            {
                unsigned long long pr[8], z[8] = {0};
                JT_HIP(hipStreamSynchronize(s));
                JT_HIP(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_nlm_prof), sizeof pr));
                JT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_nlm_prof), z, sizeof z));
                fprintf(stderr, "anlmdn wave clocks per pair (%u pairs): fill %.0f seed %.0f steps %.0f park %.0f replay %.0f turn %.0f; blocks replayed %.2f parked %.2f of %d\n",
                        grid, (double)pr[0] / grid, (double)pr[1] / grid, (double)pr[2] / grid, (double)pr[3] / grid, (double)pr[4] / grid,
                        (double)pr[5] / grid, (double)pr[6] / grid, (double)pr[7] / grid, K / 4);
            }
#endif
        } else {
#ifdef JT_AB
            JT_REQUIRE(smem <= 64 * 1024, JT_E_UNSUPPORTED, "anlmdn: window exceeds the wave-per-hop LDS budget");
            if (2 * S == 192) hipLaunchKernelGGL((k_anlmdn_pair<3>), dim3(grid), dim3(64), smem, s, in, out, n, K, S, sw, smooth, lut_scale, nhops);
            else hipLaunchKernelGGL((k_anlmdn_pair<6>), dim3(grid), dim3(64), smem, s, in, out, n, K, S, sw, smooth, lut_scale, nhops);
#endif
        }
        return;
    }
    int threads = ((2 * S + 63) / 64) * 64;
    if (threads < NLM_TI) threads = NLM_TI;
    JT_REQUIRE(threads <= 1024, JT_E_UNSUPPORTED, "anlmdn: research radius too large for one workgroup");
    size_t smem = sizeof(float) * (((NW + 3) & ~3) + (size_t)NLM_TI * (2 * S + 1));
    JT_REQUIRE(smem <= 160 * 1024, JT_E_UNSUPPORTED, "anlmdn: patch/research window exceeds LDS");
    JT_HIP(hipFuncSetAttribute((const void *)k_anlmdn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(k_anlmdn, dim3((unsigned)nhops), dim3(threads), smem, s, in, out, n, K, S, sw, smooth, lut_scale, nhops);
}
