// k_declick.hip — adeclick (FFmpeg af_adeclick.c) for gfx950: the click/pop repair of Pass 4
// (normalise.go:1306-1311; filters.go:947-962, defaults t=1.7 w=55 o=50 m=s, filters.go:513-521).
//
// Per window of W = rate*w/1000 samples (hop = W*(1-o/100), overlap-save): biased autocorrelation -> Levinson-Durbin AR
// model of order W*2% -> prediction error -> samples above threshold*sigma_e flagged (+ burst fusion) -> the flagged samples
// are replaced by the least-squares AR interpolation, i.e. the solution of a symmetric positive system whose (j,i) entry is
// the AR-coefficient autocorrelation at lag index[j]-index[i] (zero beyond the AR order).
//
// Mapping: ONE WAVE PER WINDOW, windows are independent (131 k of them for an hour at 44.1 kHz), no workgroup barriers.
// Every floating-point sum keeps FFmpeg's order (sequential, multiply then add), so flags and repaired samples are
// bit-identical to the scalar code; parallelism comes from the independent sums:
//   * autocorrelation: lane = lag (49 chains of W terms);   * detection: lane = sample (49-term chains);
//   * LDL^T of the normal matrix, right-looking, restricted to the true band: entries are zero whenever two flagged samples
//     are more than the AR order apart, and no fill-in leaves that profile, so with ~10 % of the samples flagged the band is
//     a handful of rows instead of 48 — the work per pivot is bw*(bw+1)/2 updates, one per lane.  Skipped terms are exact
//     zeros, and the updates reach every entry in FFmpeg's k-ascending order;
//   * forward substitution fused into the factorisation sweep; back substitution walks the (short) band per row.
// LDS per wave: the window's samples (f64), later reused as the 49x49 sliding block of the factorisation.
#include "jt_internal.h"
#include <utility>

namespace dk {
constexpr int MAXW = 4864;        // window samples (110 ms at 44.1 kHz); LDS: 8*MAXW bytes per wave
constexpr int MAXAR = 48;         // AR order the sliding block is laid out for (2 % of a 55 ms window at 44.1 kHz)
constexpr int BS = MAXAR + 1;     // sliding block side
constexpr int NWORD = MAXW / 64;  // 64-sample flag words per window
// the sequential-order kernel (k_adeclick) lays its per-lag arrays out for 64 entries and takes AR orders up to 62 (lane = lag, and the
// lane after the last ring row stores the entering right-hand side): 55 ms windows at 48 kHz are order 52
constexpr int XMAXAR = 64, XAR_LIMIT = 62, XBS = XAR_LIMIT + 1;
}

struct DeclickParams {
    int W, hop, skip, ar, nburst;
    int sa;                       // doubles reserved for the sample buffer / factorisation block of the instance being launched
    int lb;                       // half-window buffer length (light instance)
    int nw;                       // flag words kept per window (fast kernel)
    double threshold, gain;
    int64_t nwindows;
    // method 'a' (overlap-add, af_adeclick.c filter_channel: buf[j] += dst[j] * window_func_lut[j]): every window's W products go to
    // `prod` (wp doubles per window), k_dk_overlap_add sums them per output sample in window order.  method 1 = 's' (overlap-save).
    int method, wp;
    const double *wlut;            // [W] sin(pi i / W) * (1 - overlap) * pi / 2
    double *prod;                  // [nwindows * wp]
};

// Buffers of the split pipeline (front kernel -> solver kernels): per window a fixed slot of `wp` entries.
struct DkSplit {
    int *F;                        // [nwindows] flagged samples of the window
    unsigned short *index;         // [nwindows * wp] their positions, ascending
    double *rhs;                   // [nwindows * wp] right-hand side of the interpolation system
    double *aux;                   // [nwindows * 56] autocorrelation of the AR coefficients (aux[AR + 1] = 0)
    int *list32, *list64;          // windows whose band fits 31 rows / needs up to 48
    unsigned long long *ctl;       // [0] / [1] list lengths, [2] / [3] work counters of the two solver launches, [32 + 16 x] / [160 + 16 x] the
                                   // per-XCD window heads of the two front launches (a 128-byte line each)
    int wp;
    int xcd;                       // 1: the front launches hand windows out per XCD (contiguous eighths of the file)
    double *r;                     // [nwindows * 64] biased autocorrelation r[0 .. AR] (MODE 2 -> k_dk_levinson)
    double *ac;                    // [nwindows * 64] AR polynomial k[0 .. AR], zero padded to 64 taps; ac[w * 64 + 63] = sigma_e
    unsigned *hist;                // [2][DK_HIST] x DK_HSTRIDE: windows per flagged-sample count of the two lists (k_dk_sort_*: longest window first)
};
constexpr int DK_HIST = 520;      // counts 0 .. 513 used (the solvers' capacity is 512 flagged samples)
// one count per 128-byte line: the scatter's 131 k returning atomics land on ~130 hot counts, and sixteen counts to a line made every line a
// queue of ~10 k atomics served one after another (0.108 ms for the launch; a line of its own per count: 0.048)
constexpr int DK_HSTRIDE = 32;

__device__ inline double dk_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// value of `v` in lane `l` (l wave-uniform): two v_readlane, no LDS round trip
__device__ inline double dk_readlane(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// Three capacity levels of one kernel:
//   light  <FCAP 512, BSD 33, HALF, 0>: the window's samples stream through LDS in two overlapping halves (every phase that reads
//          them only looks AR samples around its position), the factorisation ring is 33 x 33 (bands up to 32 rows): 16 KB of LDS
//          per wave, 10 waves per CU.  A window with more than 512 flagged samples or a band wider than 32 rows appends itself to
//          the first overflow list;
//   middle <FCAP 1024, BSD 49, HALF, 1> redoes exactly those windows (list length read on the device) and overflows into the
//          second list, which <FCAP MAXW, BSD 49, full window resident, 2> takes.
// Windows (or list entries) are handed out by a device-side counter, not by a static stride: their cost varies with the number
// of flagged samples.  The kernel is instruction-issue bound (SQ counters in profiles/r01_pmc_issue.txt), so its phases are written
// for instruction count: one division per LDL^T pivot, carried ring positions, bit-parallel flag handling, packed factor rows.
// stats: [0] repaired samples, [1] singular windows, [2] / [3] lengths of the two overflow lists, [12 + LEVEL] work counters.
template <int FCAP, int BSD, bool HALF, int LEVEL>
__global__ void __launch_bounds__(64)
k_adeclick(const double *__restrict__ in, double *__restrict__ out, int64_t n, DeclickParams P, double *scratch,
           size_t scratch_per_wave, unsigned long long *__restrict__ stats, int *__restrict__ heavy)
{
    extern __shared__ unsigned char dk_smem[];
    const int lane = threadIdx.x;
    const int W = P.W, AR = P.ar;
    constexpr int BS = BSD, MAXAR = dk::XMAXAR;
    double *sbuf = reinterpret_cast<double *>(dk_smem);                // window samples [LB] ; later the BS x BS sliding block
    double *rr = sbuf + P.sa;                                           // r[AR+1]
    double *ac = rr + dk::XMAXAR + 2;                                       // acoefficients k[AR+1]
    double *aux = ac + dk::XMAXAR + 2;                                      // aux[AR+1]
    double *lvec = aux + dk::XMAXAR + 2;                                    // pivot column multipliers (bands wider than 10 rows)
    double *ywin = lvec + dk::XMAXAR + 2;                                   // sliding right-hand side / y
    unsigned long long *obits = reinterpret_cast<unsigned long long *>(ywin + dk::XMAXAR + 2);   // [NWORD] detector flags (bit = sample)
    unsigned long long *fbits = obits + dk::NWORD;                                            // [NWORD] flags after fusion / border clearing
    unsigned short *index = reinterpret_cast<unsigned short *>(fbits + dk::NWORD);          // [FCAP]
    unsigned char *bwv = reinterpret_cast<unsigned char *>(index + FCAP);                    // [FCAP] band width per pivot
    // (a, b) of trailing-update pair t >= 64, a | b << 8 (bands wider than 10 rows have more than 64 pairs): unranked once per wave
    unsigned short *pairtab = reinterpret_cast<unsigned short *>(bwv + FCAP) - 64;
    {
        constexpr int NPAIR = BSD * (BSD - 1) / 2;
        for (int t = lane + 64; t < NPAIR; t += 64) {
            int a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            a += ((a + 1) * (a + 2) / 2 <= t); a -= (a * (a + 1) / 2 > t);
            pairtab[t] = (unsigned short)(a | ((t - a * (a + 1) / 2) << 8));
        }
    }
    auto fbit = [&](int pos) -> bool { return (fbits[pos >> 6] >> (pos & 63)) & 1ull; };
    // global scratch of this wave: L[W][MAXAR] | D[W] | y[W] | rhs[W].  Written with plain stores, read back (by other lanes, after
    // a fence) with agent-scope relaxed loads, which bypass the CU's L1 and may be pipelined freely.
    double *gL = scratch + (size_t)blockIdx.x * scratch_per_wave;
    double *gD = gL + (size_t)W * MAXAR, *gY = gD + W, *gV = gY + W;
    unsigned long long repaired = 0, singular = 0;
    int a0, b0;                                                     // pair (row, column) of the trailing update this lane owns first
    {
        a0 = (int)((sqrtf(8.0f * (float)lane + 1.0f) - 1.0f) * 0.5f);
        a0 += ((a0 + 1) * (a0 + 2) / 2 <= lane); a0 -= (a0 * (a0 + 1) / 2 > lane);
        b0 = lane - a0 * (a0 + 1) / 2;
    }
    // half-window geometry: buffer A = samples [0, LB), buffer B = samples [W-LB, W); positions >= SPLIT are served by B
    const int LB = HALF ? P.lb : W;
    const int SB_B = W - LB;                                         // first sample held by buffer B
    const int SPLIT = HALF ? SB_B + AR : W;
#ifdef JT_DK_PROFILE
    unsigned long long tph[8] = {0,0,0,0,0,0,0,0}; unsigned long long tc = wall_clock64();
#define DK_MARK(i) { unsigned long long t_ = wall_clock64(); tph[i] += t_ - tc; tc = t_; }
#else
#define DK_MARK(i)
#endif

    // LEVEL 0: every window, overflow -> list A (heavy[0..], stats[2]); LEVEL 1: list A, overflow -> list B (heavy[nwindows..], stats[3]);
    // LEVEL 2: list B (full capacity, nothing overflows)
    const int *worklist = LEVEL == 1 ? heavy : heavy + P.nwindows;
    const int64_t nwork = LEVEL == 0 ? P.nwindows : (int64_t)stats[1 + LEVEL];
    // windows are handed out by a device-side counter: their cost varies with the number of flagged samples, and waves that start
    // late (another stream's kernel holding LDS on their CU when the grid was placed) simply take fewer of them
    for (;;) {
        unsigned long long wi_ = 0;
        if (lane == 0) wi_ = atomicAdd(&stats[12 + LEVEL], 1ull);
        const int64_t wi = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wi_ >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)wi_));
        if (wi >= nwork) break;
        const int64_t w = LEVEL == 0 ? wi : (int64_t)worklist[wi];
        const int64_t s0 = w * P.hop - P.skip;                          // input position of window sample 0
        const int64_t o0 = w * P.hop;
        int sb = 0;                                                     // window sample held in sbuf[0]
        // window samples (zeros before the stream and past its end), loudnorm's linear gain applied on the way in
        auto load_buf = [&](int base) {
            sb = base;
            for (int j = lane; j < LB; j += 64) {
                const int64_t p = s0 + base + j;
                sbuf[j] = (p >= 0 && p < n) ? __dmul_rn(in[p], P.gain) : 0.0;
            }
        };
        const double *S = sbuf;                                         // S[x - sb] = window sample x
        load_buf(0);
        DK_MARK(0)
        // ---- 2. autocorrelation(src, AR, W, r, 1/W): lane = lag, terms in j-ascending order, in one or two buffer phases
        double acv = 0.0;
        auto autocorr_range = [&](int ja, int jb) {                     // terms j in [ja, jb) of this lane's lag
            if (lane > AR || ja >= jb) return;
            // 16 terms per block: all 32 LDS reads are issued, then the sequential chain consumes them (the other resident waves
            // cover the latency; a hand-rolled double buffer only made the compiler shuffle 32 register pairs per block)
            const double *sj = S - sb;
            int j = ja;
            for (; j + 16 <= jb; j += 16) {
                double xv[16], yv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { xv[u] = sj[j + u]; yv[u] = sj[j + u - lane]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) acv = __dadd_rn(acv, __dmul_rn(xv[u], yv[u]));
            }
            for (; j < jb; ++j) acv = __dadd_rn(acv, __dmul_rn(sj[j], sj[j - lane]));
        };
        {
            const int jsplit = lane > SPLIT ? lane : SPLIT;
            autocorr_range(lane, jsplit < W ? jsplit : W);
            if (HALF) { load_buf(SB_B); autocorr_range(jsplit, W); }
            if (lane <= AR) rr[lane] = __dmul_rn(acv, 1.0 / W);
        }
        DK_MARK(1)
        // ---- 3. Levinson-Durbin (autoregression()): k -> ac[], sigma_e = sqrt(alpha)
        double sigmae;
        {
            // lane j keeps a[j] in a register; a[i-j-1] arrives by a lane permute that only depends on the previous iteration, so the
            // per-iteration chain is: readlane sum -> divide -> one multiply-add
            const double r0 = rr[0], r1 = rr[1];
            const double k0 = -r1 / r0;
            double areg = lane == 0 ? k0 : 0.0;
            double alpha = __dmul_rn(r0, __dsub_rn(1.0, __dmul_rn(k0, k0)));
            for (int i = 1; i < AR; ++i) {
                // epsilon = sum_{j<i} a[j] * r[i-j] (j ascending) + r[i+1]: the products in parallel, the sum as a readlane chain
                const double rrev = rr[lane < i ? i - lane : 0];
                const double arev = __shfl(areg, lane < i ? i - lane - 1 : 0, 64);
                const double prod = lane < i ? __dmul_rn(areg, rrev) : 0.0;
                // the sequential sum reads the products back as wave-uniform LDS broadcasts (one read per term instead of two
                // v_readlane); the trip count is rounded up to 8 with the +0.0 of the lanes >= i, an exact identity because eps
                // is never -0 (it starts at +0 and x + (-x) rounds to +0)
                __builtin_amdgcn_wave_barrier();
                if (lane < MAXAR) lvec[lane] = prod;
                __builtin_amdgcn_wave_barrier();
                double eps = 0.0;
                for (int j = 0; j < i; j += 8) {
                    double t8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t8[u] = lvec[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) eps = __dadd_rn(eps, t8[u]);
                }
                eps = __dadd_rn(eps, rr[i + 1]);
                const double ki = -eps / alpha;
                alpha = __dmul_rn(alpha, __dsub_rn(1.0, __dmul_rn(ki, ki)));
                // k[j] = a[j] + k[i] * a[i-j-1], j = i-1..0 (independent); k[i] = ki; then a[0..i] = k[0..i]
                if (lane < i) areg = __dadd_rn(areg, __dmul_rn(ki, arev));
                else if (lane == i) areg = ki;
            }
            if (lane == 0) ac[0] = 1.0;
            if (lane < AR) ac[lane + 1] = areg;
            sigmae = sqrt(alpha);
        }
        DK_MARK(2)
        bool finite;
        {
            const double v = lane <= AR ? ac[lane] : 0.0;
            finite = !__any(!isfinite(v));
        }
        int F = 0;
        bool to_heavy = false;
        if (finite) {
            // ---- 4. detection[i] = sum_{j=0..AR} ac[j] * src[i-j] (j ascending), click = |detection| > sigmae * threshold
            const double thr = __dmul_rn(sigmae, P.threshold);
            const int nword = (W + 63) >> 6;
            for (int wd = lane; wd < nword; wd += 64) obits[wd] = 0ull;
            auto detect_range = [&](int ia, int ib) {                  // samples i in [ia, ib) against the buffer currently loaded
                const double *sj = S - sb;
                for (int i0 = (ia >> 6) << 6; i0 < ib; i0 += 512) {
                    int ii[8], ic[8]; double dd[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { ii[q] = i0 + 64 * q + lane; ic[q] = ii[q] < ia ? ia : (ii[q] < ib ? ii[q] : ib - 1); dd[q] = 0.0; }
#pragma unroll 7
                    for (int j = 0; j <= AR; ++j) {
                        const double c = ac[j];
#pragma unroll
                        for (int q = 0; q < 8; ++q) dd[q] = __dadd_rn(dd[q], __dmul_rn(c, sj[ic[q] - j]));
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const unsigned long long bal = __ballot(ii[q] >= ia && ii[q] < ib && fabs(dd[q]) > thr);
                        if (lane == 0 && (i0 >> 6) + q < nword && bal) obits[(i0 >> 6) + q] |= bal;
                    }
                }
            };
            if (HALF) { detect_range(SPLIT > AR ? SPLIT : AR, W); load_buf(0); detect_range(AR, SPLIT < W ? SPLIT : W); }   // detection[i < AR] = 0
            else detect_range(AR, W);
            // ---- 5. burst fusion: the gap between two consecutive flagged samples p < q is filled when 1 < q-p <= nburst (the
            //         sequential loop only ever compares the detector's own flags); 6. borders cleared, index list in order
            // bit-parallel, one 64-sample word per lane: sample j is filled when flags sit at j - d1 and j + d2 with d1, d2 >= 1 and
            // d1 + d2 <= nburst (the nearest pair either side is the tightest one, so this is the same test)
            for (int wd0 = 0; wd0 < nword; wd0 += 64) {
                const int wd = wd0 + lane;
                const bool in = wd < nword;
                const unsigned long long cur = in ? obits[wd] : 0ull;
                const unsigned long long prv = (in && wd > 0) ? obits[wd - 1] : 0ull;
                const unsigned long long nxt = (wd + 1 < nword) ? obits[wd + 1] : 0ull;
                unsigned long long fused = cur;
                const int nb = P.nburst < 64 ? P.nburst : 64;
                for (int d1 = 1; d1 < nb; ++d1) {
                    const unsigned long long below = (cur << d1) | (prv >> (64 - d1));            // bit j = flag[j - d1]
                    unsigned long long above = 0ull;
                    for (int d2 = 1; d1 + d2 <= nb; ++d2) above |= (cur >> d2) | (nxt << (64 - d2));   // bit j = flag[j + d2]
                    fused |= below & above;
                }
                // borders: only samples AR <= j < W - AR are repaired
                const int lo = AR - wd * 64, hi = (W - AR) - wd * 64;                                  // keep bits [lo, hi)
                const unsigned long long mlo = lo <= 0 ? ~0ull : (lo >= 64 ? 0ull : (~0ull << lo));
                const unsigned long long mhi = hi >= 64 ? ~0ull : (hi <= 0 ? 0ull : ((1ull << hi) - 1ull));
                unsigned long long fb = in ? (fused & mlo & mhi) : 0ull;
                if (in) fbits[wd] = fb;
                const int cnt = __popcll(fb);
                int incl = cnt;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) { const int o = __shfl_up(incl, dd, 64); incl += lane >= dd ? o : 0; }
                int slot = F + incl - cnt;
                while (fb) {
                    const int bpos = __ffsll((long long)fb) - 1;
                    if (slot < FCAP) index[slot] = (unsigned short)(wd * 64 + bpos);
                    ++slot; fb &= fb - 1ull;
                }
                F += __builtin_amdgcn_readlane(incl, 63);
            }
            F = __builtin_amdgcn_readfirstlane(F);              // wave-uniform by construction: keep every loop over it scalar
            to_heavy = F > FCAP;
            if (!to_heavy) {
                // band width of every pivot: rows k+1 .. k+bw are within AR samples of row k (index[] increases): upper bound by bisection
                int bwmax = 0;
                for (int k = lane; k < F; k += 64) {
                    const int lim = (int)index[k] + AR;
                    int lo = k, hi = min(F - 1, k + MAXAR);              // last row r in [k, hi] with index[r] <= lim
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)index[mid] <= lim) lo = mid; else hi = mid - 1; }
                    bwv[k] = (unsigned char)(lo - k);
                    bwmax = max(bwmax, lo - k);
                }
                to_heavy = __any(bwmax > BS - 1);
            }
        }
        if (to_heavy) {
            // does not fit this instance's LDS layout: hand the window to the full-capacity pass
            if (LEVEL == 0 && lane == 0) heavy[atomicAdd(&stats[2], 1ull)] = (int)w;
            if (LEVEL == 1 && lane == 0) heavy[P.nwindows + (int64_t)atomicAdd(&stats[3], 1ull)] = (int)w;
            continue;
        }
        DK_MARK(3)
        bool ok = true;
        if (F > 0) {
            // ---- 7. aux = autocorrelation(ac, AR, AR+1, ., 1.)
            if (lane <= AR) {
                double value = 0.0;
                for (int j = lane; j <= AR; ++j) value = __dadd_rn(value, __dmul_rn(ac[j], ac[j - lane]));
                aux[lane] = __dmul_rn(value, 1.0);
            } else if (lane == AR + 1) aux[lane] = 0.0;      // entries beyond the band (LDL^T ring fill)
            // ---- 8. right-hand side: vector[e] = -sum_{j=-AR..AR, index[e]-j not flagged} src[index[e]-j] * aux[|j|]
            // The flagged samples (the unknowns) are zeroed in the LDS copy first: their terms then contribute x - (+-0), which is
            // exact because the running value is never -0 (it starts at +0 and x - x rounds to +0), so the inner loop carries no
            // flag test at all.  Nothing reads the samples from LDS after this stage (the output copy reads the input stream).
            auto zero_flagged = [&]() {
                const int nb = HALF ? LB : W;
                for (int e = lane; e < F; e += 64) { const int o = (int)index[e] - sb; if (o >= 0 && o < nb) sbuf[o] = 0.0; }
            };
            auto rhs_range = [&](int ea, int eb) {                     // entries [ea, eb) against the buffer currently loaded
                const double *sj = S - sb;
                for (int e0 = ea; e0 < eb; e0 += 128) {
                    int ie[2]; double val[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) { const int e = e0 + 64 * q + lane; ie[q] = index[e < eb ? e : eb - 1]; val[q] = 0.0; }
                    // terms j = -AR .. AR in order; 8 at a time: the reads are issued before the chain consumes them
                    int j0 = -AR;
                    for (; j0 + 7 <= AR; j0 += 8) {
                        double tv[2][8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int j = j0 + u;
                            const double ax = aux[j < 0 ? -j : j];
#pragma unroll
                            for (int q = 0; q < 2; ++q) tv[q][u] = __dmul_rn(sj[ie[q] - j], ax);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) { val[0] = __dsub_rn(val[0], tv[0][u]); val[1] = __dsub_rn(val[1], tv[1][u]); }
                    }
                    for (; j0 <= AR; ++j0) {
                        const double ax = aux[j0 < 0 ? -j0 : j0];
                        val[0] = __dsub_rn(val[0], __dmul_rn(sj[ie[0] - j0], ax));
                        val[1] = __dsub_rn(val[1], __dmul_rn(sj[ie[1] - j0], ax));
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) { const int e = e0 + 64 * q + lane; if (e < eb) gV[e] = val[q]; }
                }
            };
            if (HALF) {
                // entries whose +-AR neighbourhood lies inside buffer A (currently loaded), then the rest against buffer B
                int ea = 0;
                for (int e0 = 0; e0 < F; e0 += 64) { const int e = e0 + lane; ea += __popcll(__ballot(e < F && (int)index[e] + AR < LB)); }
                zero_flagged();
                rhs_range(0, ea);
                if (ea < F) { load_buf(SB_B); zero_flagged(); rhs_range(ea, F); }
            } else { zero_flagged(); rhs_range(0, F); }
            __threadfence();
            DK_MARK(4)
            // ---- 9. LDL^T (cholesky_decomposition) right-looking inside the band, forward substitution fused.
            // blk[(j % BS) * BS + (i % BS)] holds entry (j, i), i <= j, of the rows currently within reach of the pivot.
            double *blk = sbuf;
            auto entry0 = [&](int j, int i) -> double {           // original matrix entry (j >= i)
                const int dlt = (int)index[j] - (int)index[i];
                return dlt <= AR ? aux[dlt] : 0.0;
            };
            {
                const int nr = F < BS ? F : BS;                    // rows 0 .. nr-1 enter before the first pivot
                for (int t = lane; t < nr * BS; t += 64) {
                    const int j = t / BS, i = t - j * BS;
                    if (i <= j) blk[j * BS + i] = entry0(j, i);
                }
                if (lane < nr) ywin[lane] = dk_ld(&gV[lane]);
            }
            int vbase = BS; double vreg = (vbase + lane < F) ? dk_ld(&gV[vbase + lane]) : 0.0;      // right-hand sides of the rows about to enter
            // Packed factor stream in global scratch: per pivot k the bw multipliers of its column followed by y_k / d_k.
            size_t goff = 0;
            // The loop is issue-bound, so it is written for instruction count: every LDS operand is fetched unconditionally (rows
            // outside the band read values that are never used), the ring positions are carried instead of recomputed with a
            // division by BS, one f64 division serves the multipliers and y/d, and the row that enters the ring and its right-hand
            // side go out in one store.
            int ks = -1;                                       // k % BS
            int rs = 1 + lane; rs -= rs >= BS ? BS : 0;        // (k + 1 + lane) % BS: ring row of this lane's band row
            int ra = 1 + a0, rb = 1 + b0;                      // ring row / column of this lane's trailing pair (a0, b0 <= 10)
            for (int k = 0; k < F; ++k) {
                ks = ks + 1 == BS ? 0 : ks + 1;
                const int bw = bwv[k];
                const double d = blk[ks * BS + ks];
                const double yk = ywin[ks];
                const double colv = blk[rs * BS + ks];
                const double yrow = ywin[rs];
                double *ep = &blk[ra * BS + rb];
                const double eold = *ep;
                const int idx_i = index[k + 1 + lane];             // only lanes < BS use it (then k + 1 + lane <= k + BS < F)
                if (d == 0.0) { ok = false; break; }
                const double qv = (lane == bw ? yk : colv) / d;
                const double l = lane < bw ? qv : 0.0;
                if (lane <= bw) gL[goff + lane] = qv;
                goff += (size_t)bw + 1;
                // forward substitution (k ascending = FFmpeg's j-ascending order for every row)
                if (lane < bw) ywin[rs] = __dsub_rn(yrow, __dmul_rn(l, yk));
                // trailing update: entry (k+1+a, k+1+b), 0 <= b <= a < bw:  -= (d * L_b) * L_a.  Pair t = lane (a0, b0 fixed per lane) takes
                // its two multipliers by lane permutes; bands wider than 10 rows (more than 64 pairs) finish through LDS.
                const int npairs = bw * (bw + 1) / 2;
                {
                    const double la = __shfl(l, a0, 64), lb = __shfl(l, b0, 64);
                    if (lane < npairs) *ep = __dsub_rn(eold, __dmul_rn(__dmul_rn(d, lb), la));
                }
                if (npairs > 64) {
                    if (lane < MAXAR) lvec[lane] = l;
                    for (int t = lane + 64; t < npairs; t += 64) {
                        const int ab = pairtab[t], a = ab & 0xff, b = ab >> 8;
                        int rra = ks + 1 + a, rrb = ks + 1 + b;
                        rra -= rra >= BS ? BS : 0; rrb -= rrb >= BS ? BS : 0;
                        double *e = &blk[rra * BS + rrb];
                        *e = __dsub_rn(*e, __dmul_rn(__dmul_rn(d, lvec[b]), lvec[a]));
                    }
                }
                // row k leaves the ring; row k + BS enters with its original entries against the rows still in reach (columns
                // k+1 .. k+BS, lanes 0 .. BS-1; aux[AR+1] is 0), and lane BS stores its right-hand side
                const int nj = k + BS;
                if (nj < F) {
                    if (nj >= vbase + 64) { vbase += 64; vreg = (vbase + lane < F) ? dk_ld(&gV[vbase + lane]) : 0.0; }
                    const double vn = dk_readlane(vreg, nj - vbase);
                    const int dlt = (int)index[nj] - idx_i;
                    const double ent = aux[dlt < AR + 1 ? dlt : AR + 1];
                    double *dst = lane == BS ? &ywin[ks] : &blk[ks * BS + rs];
                    if (lane <= BS) *dst = lane == BS ? vn : ent;
                }
                rs = rs + 1 == BS ? 0 : rs + 1;
                ra = ra + 1 == BS ? 0 : ra + 1;
                rb = rb + 1 == BS ? 0 : rb + 1;
            }
            DK_MARK(5)
            if (ok) {
                __threadfence();
                // ---- 10. back substitution: out[i] = y[i]/d[i] - sum_{j>i} L[j][i] * out[j] (j ascending; exact zeros beyond the
                // band are skipped).  L[j][i] = gL[i][j-i-1].  Rows in batches of 16: the batch's factors are fetched together.
                double sw = 0.0;                                    // lane t: solution of row (current row + 1 + t)
                for (int ib = F - 1; ib >= 0; ib -= 16) {
                    double qr = 0.0; int ntr = 0, cnt = 0;
                    if (lane < 16 && ib - lane >= 0) { ntr = bwv[ib - lane]; cnt = ntr + 1; }
                    // packed rows end at goff; row ib - q starts at goff - (cnt_0 + .. + cnt_q) and ends with its y/d
                    int pref = cnt;
#pragma unroll
                    for (int dd = 1; dd < 16; dd <<= 1) { const int o = __shfl_up(pref, dd, 64); pref += lane >= dd ? o : 0; }
                    if (cnt) qr = dk_ld(&gL[goff - (size_t)pref + (size_t)ntr]);
                    double Lr[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int nq = __builtin_amdgcn_readlane(ntr, q);
                        const size_t oq = goff - (size_t)__builtin_amdgcn_readlane(pref, q);
                        Lr[q] = lane < nq ? dk_ld(&gL[oq + lane]) : 0.0;
                    }
                    goff -= (size_t)__builtin_amdgcn_readlane(pref, 15);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        // rows below 0 (last batch) run as empty rows: nt = 0, nothing is stored for them
                        const int nt = __builtin_amdgcn_readlane(ntr, q);
                        double v = dk_readlane(qr, q);
                        // terms beyond the band are +0.0 (v - (+0) is exact for every v), read back as LDS broadcasts four at a time
                        const double term = lane < nt ? __dmul_rn(Lr[q], sw) : 0.0;
                        __builtin_amdgcn_wave_barrier();
                        if (lane < MAXAR) lvec[lane] = term;
                        __builtin_amdgcn_wave_barrier();
                        for (int t = 0; t < nt; t += 4) {
                            double t4[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) t4[u] = lvec[t + u];
#pragma unroll
                            for (int u = 0; u < 4; ++u) v = __dsub_rn(v, t4[u]);
                        }
                        // slide the window: lane t takes lane t-1's solution, lane 0 the new one
                        {
                            const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(sw), 0x138, 0xf, 0xf, false);   // wave_shr:1
                            const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(sw), 0x138, 0xf, 0xf, false);
                            sw = lane == 0 ? v : __hiloint2double(hi, lo);
                        }
                    }
                    // lanes 0..15 now hold the solutions of rows ib-15 .. ib: park them in LDS (the factorisation block is dead)
                    if (lane < 16 && ib - 15 + lane >= 0) blk[ib - 15 + lane] = sw;
                }
                // repaired samples of this window's output hop, all at once (overlap-add: of the whole window, weighted)
                for (int e = lane; e < F; e += 64) {
                    const int pos = index[e];
                    if (P.method == 0) P.prod[(size_t)w * P.wp + pos] = __dmul_rn(blk[e], P.wlut[pos]);
                    else if (pos >= P.skip && pos < P.skip + P.hop && o0 + (pos - P.skip) < n) out[o0 + (pos - P.skip)] = blk[e];
                }
                repaired += (lane == 0) ? (unsigned long long)F : 0ull;
            } else {
                singular += (lane == 0) ? 1ull : 0ull;
            }
        }
        DK_MARK(6)
        // ---- 11. overlap-save output of the samples that were not repaired: out[w*hop + j] = src[skip + j]
        const bool rep = F > 0 && ok;
        if (P.method == 0) {
            // overlap-add: dst[pos] * lut[pos] of every window sample that was not repaired
            for (int pos = lane; pos < W; pos += 64) {
                if (rep && fbit(pos)) continue;
                const int64_t p = s0 + pos;
                const double v = (p >= 0 && p < n) ? __dmul_rn(in[p], P.gain) : 0.0;
                P.prod[(size_t)w * P.wp + pos] = __dmul_rn(v, P.wlut[pos]);
            }
            continue;
        }
        for (int j = lane; j < P.hop; j += 64) {
            const int64_t o = o0 + j;
            const int pos = P.skip + j;
            if (o < n && !(rep && fbit(pos))) {
                const int64_t p = s0 + pos;
                out[o] = (p >= 0 && p < n) ? __dmul_rn(in[p], P.gain) : 0.0;
            }
        }
    }
    if (lane == 0 && (repaired || singular)) { atomicAdd(&stats[0], repaired); atomicAdd(&stats[1], singular); }
#ifdef JT_DK_PROFILE
    DK_MARK(7)
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&stats[4 + i], tph[i]);
#endif
}

// =====================================================================================================================
// k_adeclick_fast — the same filter with the floating-point SUMMATION ORDER relaxed (results within ~1e-12 of the sequential
// code instead of bit-identical; the detector's decisions are the same unless |detection| sits within rounding of the threshold --
// tests/test_gpu_ops.py counts the flips against the oracle).  The exact kernel above is issue-bound on its sequential chains
// (profiles/r01_pmc_issue.txt); here every phase is written for instruction count instead:
//   * autocorrelation on the matrix pipe: r[16m + a - b] = sum over 64-sample steps of C_m[a][b], C_m = X X_m^T with
//     v_mfma_f64_16x16x4_f64 -- operand A is 64 consecutive samples (lane l = sample j0 + l), operand B the same stream 16 m samples
//     earlier, m = 0..3: 4 LDS reads + 4 MFMAs per 64 samples instead of 64 x (2 LDS reads + mul + add) per lane-lag;
//   * detection as a register-blocked FIR: a lane owns DQ consecutive outputs, the DQ + AR samples it needs slide through a
//     register ring (fully unrolled, static indices), one LDS read + one broadcast coefficient read per DQ fused multiply-adds;
//   * LDL^T in a diagonal-major ring blk[dg][i & (NC-1)] = entry (i + dg, i): the pivot column (and the pivot itself, lane 0) is ONE
//     LDS read at lane-constant address + scalar offset, the column that enters the ring is written back to the same address, a
//     trailing pair's address advances by one slot per pivot, the multipliers are taken from the pre-division column (d * l_b = the
//     column entry itself) and 1/d is a reciprocal + two Newton steps;
//   * fused multiply-adds everywhere a product feeds a sum.
// Same capacity levels, work counter, overflow lists, packed factor stream and output handling as k_adeclick.
#ifndef JT_DK_LOADBATCH
#define JT_DK_LOADBATCH 8
#endif
// -DJT_DK_NOMFMA_DETECT: the detector FIR as lane FMAs over a rotating register window (the round's first version; A/B builds)
#ifdef JT_DK_NOMFMA_DETECT
constexpr bool kDetectOnLanes = true;
#else
constexpr bool kDetectOnLanes = false;
#endif
namespace dkf { constexpr int DQ = 11; }
typedef double dk_d4 __attribute__((ext_vector_type(4)));

__device__ inline double dk_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __fma_rn(-d, r, 1.0); r = __fma_rn(r, e, r);
    e = __fma_rn(-d, r, 1.0); r = __fma_rn(r, e, r);
    return r;
}

// MODE 0: the whole filter in one kernel (as described above).  MODE 1: the FRONT of the split pipeline -- window load + pass-through
// copy, AR fit, detector, index list, right-hand side -- which leaves F, index[], rhs[] and aux[] of every window with flagged samples
// in global memory and appends the window to the solver list of its band class; k_dk_solve does the LDL^T and the substitutions.
// The front itself runs as two launches with k_dk_levinson between them: MODE 2 = window load + pass-through copy + autocorrelation
// (r[] to global memory), MODE 3 = detector .. right-hand side with the AR polynomial read back.  Levinson-Durbin is 48 strictly
// sequential steps of a few flops: one wave per window spent a third of the front in it with one lane busy; one LANE per window
// (k_dk_levinson) does the whole file's 131 k recursions in well under 0.1 ms.
template <int FCAP, int NC, int ND, bool HALF, int LEVEL, int MODE = 0>
__global__ void __launch_bounds__(64, 3)                  // three waves per SIMD (<= 168 VGPRs): LDS admits ten waves per CU
k_adeclick_fast(const double *__restrict__ in, double *__restrict__ out, int64_t n, DeclickParams P, double *scratch,
                size_t scratch_per_wave, unsigned long long *__restrict__ stats, int *__restrict__ heavy, DkSplit S)
{
    extern __shared__ unsigned char dk_smem[];
    const int lane = threadIdx.x;
    const int W = P.W, AR = P.ar;
    constexpr int MAXAR = dk::MAXAR, CM = NC - 1, BWMAX = ND - 1, DQ = dkf::DQ;
    // ring row stride: one double of padding, so that the lanes of a pivot column (one per diagonal) and the pairs of a trailing
    // update (diagonal a - b, slot k + 1 + b) fall on different LDS banks; with a stride of NC = 32 or 64 doubles every row started on
    // the same bank and a column read was a 32-way conflict (SQ_LDS_BANK_CONFLICT was 41 % of the LDS cycles of the kernel)
    constexpr int NCP = NC == 32 ? 41 : NC + 1;                 // 41 = 9 (mod 32): columns conflict-free, pairs at most two-way
    static_assert((NC & (NC - 1)) == 0 && ND <= NC + 1 && BWMAX < NC, "ring geometry");
    double *sbuf = reinterpret_cast<double *>(dk_smem);                // window samples ; later the ND x NC ring
    double *rr = sbuf + P.sa;                                           // r[AR+1]                       [50]
    double *ac = rr + 50;                                               // k[AR+1], zero padded          [64]
    double *aux = ac + 64;                                              // aux[AR+1], aux[AR+1] = 0      [50]
    double *lvec = aux + 50;                                            // multipliers (wide bands)      [50]
    double *cvec = lvec + 50;                                           // pivot column (wide bands)     [50]
    double *yring = cvec + 50;                                          // y of the rows in the ring     [NC]
    unsigned long long *obits = reinterpret_cast<unsigned long long *>(yring + NC);            // [P.nw]
    unsigned long long *fbits = obits + P.nw;                                                   // [P.nw]
    unsigned short *index = reinterpret_cast<unsigned short *>(fbits + P.nw);                 // [FCAP + NC + ND]
    unsigned char *bwv = reinterpret_cast<unsigned char *>(index + FCAP + NC + ND);            // [FCAP]
    unsigned short *pairtab = reinterpret_cast<unsigned short *>(bwv + FCAP) - 64;
    double *dscr = reinterpret_cast<double *>(obits);                   // r2[64] scratch of the autocorrelation (the flag words are dead then)
    {
        constexpr int NPAIR = ND * (ND - 1) / 2;
        for (int t = lane + 64; t < NPAIR; t += 64) {
            int a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            a += ((a + 1) * (a + 2) / 2 <= t); a -= (a * (a + 1) / 2 > t);
            pairtab[t] = (unsigned short)(a | ((t - a * (a + 1) / 2) << 8));
        }
    }
    double *gL = scratch + (size_t)blockIdx.x * scratch_per_wave;
    double *gD = gL + (size_t)W * MAXAR, *gY = gD + W, *gV = gY + W;
    unsigned long long repaired = 0, singular = 0;
    int a0, b0;
    {
        a0 = (int)((sqrtf(8.0f * (float)lane + 1.0f) - 1.0f) * 0.5f);
        a0 += ((a0 + 1) * (a0 + 2) / 2 <= lane); a0 -= (a0 * (a0 + 1) / 2 > lane);
        b0 = lane - a0 * (a0 + 1) / 2;
    }
    const int LB = HALF ? P.lb : W;
    const int SB_B = W - LB;
    const int SPLIT = HALF ? SB_B + AR : W;
    const int PADE = 64;                                             // zeros kept behind the samples (the last 64-sample step runs past W)
#ifdef JT_DK_PROFILE
    unsigned long long tph[8] = {0,0,0,0,0,0,0,0}; unsigned long long tc = wall_clock64();
#define DKF_MARK(i) { unsigned long long t_ = wall_clock64(); tph[i] += t_ - tc; tc = t_; }
#else
#define DKF_MARK(i)
#endif
    const int *worklist = LEVEL == 1 ? heavy : heavy + P.nwindows;
    const int64_t nwork = LEVEL == 0 ? P.nwindows : (int64_t)stats[1 + LEVEL];
    // The work counter is fetched one window ahead: the atomic's round trip (and the drain of the previous window's stores that a
    // wait on it implies) took 10 % of a window when the wave asked for its next window only after finishing the current one.
    // The two front launches of the split pipeline hand the windows out PER XCD (HW_REG_XCC_ID), each XCD working through a contiguous
    // eighth of the file with a head of its own, on a 128-byte line of its own, and moving on to the next XCD's eighth when its own is
    // done.  Two things come of it.  (1) One head for the whole chip was a bottleneck: 2560 waves at 23-50 us per window are 50-110
    // returning atomics per us on one word, which saturates near 90 (MI355X_MICROARCH.md, "dequeue"); eight heads took 0.5 ms off the
    // autocorrelation launch (1.70 -> 1.17 ms) and 0.2 ms off the detector launch -- with the eight heads in ONE line the launches were
    // 0.2 ms SLOWER than with one head.  (2) Consecutive windows share half their samples, and with one counter neighbours landed on
    // different XCDs (eight private L2s), so every half window came from HBM twice per launch (PMC FETCH_SIZE was exactly two file
    // lengths): 2.4 -> 1.6 GB and 3.7 -> 2.0 GB.  Placement is a speed matter only: every window is still taken exactly once, whatever
    // the register returns.
    constexpr bool XH = LEVEL == 0 && (MODE == 2 || MODE == 3);
    unsigned long long *xheads = nullptr; int xcur = 0;
    if constexpr (XH) if (S.xcd) {
        unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcur = (int)(id & 7u); xheads = S.ctl + (MODE == 2 ? 32 : 160);                  // one 128-byte line per head
    }
    auto xbase = [&](int x) -> int64_t { return nwork * x / 8; };
    auto take = [&]() -> unsigned long long {
        unsigned long long v = 0;
        if (lane == 0) v = atomicAdd(xheads ? &xheads[16 * xcur] : &stats[12 + LEVEL], 1ull);
        return v;
    };
    auto uniform64 = [](unsigned long long v) -> int64_t {
        return (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    // ticket -> window (wave-uniform); a ticket beyond the XCD's eighth sends the wave to the next eighth, at most seven times
    auto resolve = [&](unsigned long long t_) -> int64_t {
        int64_t v = uniform64(t_);
        if (!xheads) return v;
        for (int tries = 0;;) {
            const int64_t b = xbase(xcur);
            if (v < xbase(xcur + 1) - b) return b + v;
            if (++tries == 8) return nwork;
            xcur = (xcur + 1) & 7;
            v = uniform64(take());
        }
    };
    int64_t wi = resolve(take());
    for (;;) {
        if (wi >= nwork) break;
        const unsigned long long wnext_ = take();                   // consumed after the first buffer load below
        const int64_t w = LEVEL == 0 ? wi : (int64_t)worklist[wi];
        const int64_t s0 = w * P.hop - P.skip;
        const int64_t o0 = w * P.hop;
#ifdef JT_DK_SPLIT0
        DKF_MARK(0)
#endif
        int sb = 0;
        // Window positions [elo, ehi) of a load also leave as output (input * gain): the pass-through copy of the hop rides on the
        // first load of each half instead of re-reading the hop from memory after the solve (that copy was 10 % of a window: three
        // more load round trips behind the solver's stores).  Repaired samples overwrite theirs later, behind a fence.
        auto load_buf = [&](int base, int elo = 0, int ehi = 0) {
            sb = base;
            // eight loads in flight per lane before the first LDS store (one load -> one store per trip left every trip waiting for
            // a full memory latency: 21 trips per half window, four half windows per window)
            constexpr int NB = JT_DK_LOADBATCH;
            for (int j0 = lane; j0 < LB + PADE; j0 += 64 * NB) {
                double t[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int j = j0 + 64 * u; const int64_t p = s0 + base + j;
                    t[u] = (j < LB + PADE && base + j < W && p >= 0 && p < n) ? in[p] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int j = j0 + 64 * u;
                    const double g = __dmul_rn(t[u], P.gain);
                    if (j < LB + PADE) sbuf[j] = g;
                    const int pos = base + j;
                    if (pos >= elo && pos < ehi && o0 + (pos - P.skip) < n) out[o0 + (pos - P.skip)] = g;
                }
            }
        };
        const int emid = HALF ? min(LB, P.skip + P.hop) : P.skip + P.hop;
        if (MODE == 3) load_buf(HALF ? SB_B : 0);                   // (the detector starts on the upper half; MODE 2 wrote the pass-through copy)
        else load_buf(0, P.skip, emid);
        wi = resolve(wnext_);                                       // (the loads above were waited for; the atomic precedes them)
#ifdef JT_DK_SPLIT0
        DKF_MARK(7)
#else
        DKF_MARK(0)
#endif
        // ---- 2. autocorrelation on the matrix pipe
        if (MODE != 3) {
            dk_d4 C0 = {0, 0, 0, 0}, C1 = {0, 0, 0, 0}, C2 = {0, 0, 0, 0}, C3 = {0, 0, 0, 0};
            auto mfma_blocks = [&](int kb0, int kb1) {
                const double *sj = sbuf - sb + lane;                       // sj[j0] = sample j0 + lane
                for (int kb = kb0; kb < kb1; ++kb) {
                    const int j0 = kb << 6;
                    const double a = sj[j0];
                    double b1, b2, b3;
                    if (kb == 0) {                                          // samples before the window do not exist: zero terms
                        b1 = lane >= 16 ? sj[-16] : 0.0; b2 = lane >= 32 ? sj[-32] : 0.0; b3 = lane >= 48 ? sj[-48] : 0.0;
                    } else { b1 = sj[j0 - 16]; b2 = sj[j0 - 32]; b3 = sj[j0 - 48]; }
                    C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, C0, 0, 0, 0);
                    C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, C1, 0, 0, 0);
                    C2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, C2, 0, 0, 0);
                    C3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, C3, 0, 0, 0);
                }
            };
            const int nblk = (W + 63) >> 6;
            const int ka = HALF ? (SPLIT + 63) >> 6 : nblk;              // 64*ka <= LB and 64*ka - AR >= SB_B
            mfma_blocks(0, ka);
            if (HALF) { load_buf(SB_B, emid, P.skip + P.hop); mfma_blocks(ka, nblk); }
            // C_m[i][j] (i = 4 * reg + lane / 16, j = lane % 16) belongs to lag 16 m + i - j: every lane adds its sixteen entries into
            // r2[lag] with LDS floating-point atomics (ds_add_f64; one wave, program order, conflicting lanes of an instruction are
            // served in lane order: the sum order is fixed, the result reproducible)
            __builtin_amdgcn_wave_barrier();
            dscr[lane] = 0.0;
            __builtin_amdgcn_wave_barrier();
            const int lag0 = (lane >> 4) - (lane & 15);
            const dk_d4 CC[4] = {C0, C1, C2, C3};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lag = lag0 + 16 * m + 4 * r;
                    if (lag >= 0 && lag < 64) __hip_atomic_fetch_add(&dscr[lag], CC[m][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            __builtin_amdgcn_wave_barrier();
            const double rl = dscr[lane];
            __builtin_amdgcn_wave_barrier();
            if (lane <= AR) rr[lane] = rl * (1.0 / W);
            __builtin_amdgcn_wave_barrier();
            if (MODE == 2) { S.r[(size_t)w * 64 + lane] = lane <= AR ? rl * (1.0 / W) : 0.0; DKF_MARK(1) continue; }
        }
        DKF_MARK(1)
        // ---- 3. Levinson-Durbin (as k_adeclick)
        double sigmae;
        if (MODE == 3) {
            const double v = S.ac[(size_t)w * 64 + lane];
            sigmae = dk_readlane(v, 63);
            ac[lane] = lane == 63 ? 0.0 : v;
            __builtin_amdgcn_wave_barrier();
        } else {
            const double r0 = rr[0], r1 = rr[1];
            const double k0 = -r1 / r0;
            double areg = lane == 0 ? k0 : 0.0;
            double alpha = r0 * (1.0 - k0 * k0);
            for (int i = 1; i < AR; ++i) {
                const double rrev = rr[lane < i ? i - lane : 0];
                const double arev = __shfl(areg, lane < i ? i - lane - 1 : 0, 64);
                const double prod = lane < i ? areg * rrev : 0.0;
                __builtin_amdgcn_wave_barrier();
                if (lane < MAXAR) lvec[lane] = prod;
                __builtin_amdgcn_wave_barrier();
                double eps = 0.0;
                for (int j = 0; j < i; j += 8) {
                    double t8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t8[u] = lvec[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) eps += t8[u];
                }
                eps += rr[i + 1];
                const double ki = -eps / alpha;
                alpha = alpha * (1.0 - ki * ki);
                if (lane < i) areg = __fma_rn(ki, arev, areg);
                else if (lane == i) areg = ki;
            }
            __builtin_amdgcn_wave_barrier();
            const double aup = __shfl_up(areg, 1, 64);
            ac[lane] = lane == 0 ? 1.0 : (lane <= AR ? aup : 0.0);                         // zero padded to 64 taps
            sigmae = sqrt(alpha);
            __builtin_amdgcn_wave_barrier();
        }
        DKF_MARK(2)
        bool finite;
        {
            const double v = lane <= AR ? ac[lane] : 0.0;
            finite = !__any(!isfinite(v));
        }
        int F = 0;
        bool to_heavy = false, wide = false;
        if (finite) {
            // ---- 4. detection: register-blocked FIR, DQ consecutive outputs per lane
            const double thr = sigmae * P.threshold;
            const int nword = (W + 63) >> 6;
            for (int wd = lane; wd < nword + 1; wd += 64) obits[wd] = 0ull;
            __builtin_amdgcn_wave_barrier();
            auto detect_range = [&](int ia, int ib) {
                const double *sj = sbuf - sb;
                const int lo = ia - AR;                                  // lowest sample any real tap reads (held by the current buffer)
                for (int base = ia; base < ib; base += 64 * DQ) {
                    const int c0 = base + DQ * lane;
                    if (c0 < ib) {
                        double acc[DQ], ring[DQ];                          // ring[s mod DQ] = sample c0 + s, s in [-j, DQ - 1 - j] at tap j
#pragma unroll
                        for (int q = 0; q < DQ; ++q) { ring[q] = sj[c0 + q]; acc[q] = 0.0; }
                        for (int jb = 0; jb <= AR; jb += DQ) {
#pragma unroll
                            for (int v = 0; v < DQ; ++v) {
                                if (v > 0 && jb + v > AR) break;            // wave-uniform: the remaining taps are zero
                                const double c = ac[jb + v];
#pragma unroll
                                for (int q = 0; q < DQ; ++q) acc[q] = __fma_rn(c, ring[(q - v + DQ) % DQ], acc[q]);
                                int xi = c0 - (jb + v) - 1; xi = xi < lo ? lo : xi;
                                ring[(DQ - 1 - v) % DQ] = sj[xi];
                            }
                        }
                        unsigned m = 0;
#pragma unroll
                        for (int q = 0; q < DQ; ++q) m |= (fabs(acc[q]) > thr && c0 + q < ib) ? (1u << q) : 0u;
                        if (m) {
                            const int w0 = c0 >> 6, sh = c0 & 63;
                            atomicOr(&obits[w0], (unsigned long long)m << sh);
                            if (sh + DQ > 64 && (m >> (64 - sh))) atomicOr(&obits[w0 + 1], (unsigned long long)(m >> (64 - sh)));
                        }
                    }
                }
            };
            // The same FIR on the matrix pipe.  A range of outputs [ia, ib) is cut into 16 segments of R outputs; output ia + i + R j is
            // element (i, j) of  D = T X,  T[i][u] = a[i - u] (Toeplitz in the AR coefficients, the same 16 x 64 block for every block of
            // 16 rows),  X[u][j] = x[ia + u + R j]:  per 16 rows, sixteen v_mfma_f64_16x16x4_f64 (K = 64 covers the AR + 16 samples a
            // row block reaches back to), operand A from sixteen registers filled once per window, operand B one LDS read per lane and
            // step (R odd: the sixteen segment starts fall on different banks).  256 outputs x 49 taps per 16 instructions instead of
            // 11 outputs per 49 lane-FMAs; the sum order inside an MFMA is the hardware's (this kernel's results are order-relaxed anyway).
            dk_d4 zero4 = {0, 0, 0, 0};
            double areg[16];
            if (!kDetectOnLanes) {
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const int c = (lane & 15) + 48 - 4 * ks - (lane >> 4);
                    areg[ks] = (c >= 0 && c <= AR) ? ac[c] : 0.0;
                }
            }
            auto detect_range_mfma = [&](int ia, int ib) {
                const double *sj = sbuf - sb;
                const int lo = ia - AR;
                const int len = ib - ia;
                const int R = ((len + 15) >> 4) | 1;
                const int lpos = ia - 48 + (lane >> 4) + R * (lane & 15);
                for (int i0 = 0; i0 < R; i0 += 32) {
                    dk_d4 C0 = zero4, C1 = zero4;
                    const bool two = i0 + 16 < R;
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        int p0 = lpos + i0 + 4 * ks; p0 = p0 < lo ? lo : p0;
                        int p1 = lpos + i0 + 16 + 4 * ks; p1 = p1 < lo ? lo : p1;
                        const double b0v = sj[p0];
                        const double b1v = two ? sj[p1] : 0.0;
                        C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[ks], b0v, C0, 0, 0, 0);
                        C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[ks], b1v, C1, 0, 0, 0);
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const dk_d4 C = h ? C1 : C0;
                        const int ib0 = i0 + 16 * h + (lane >> 4);                 // row of register 0; register r is row ib0 + 4 r
                        const int n0 = ia + ib0 + R * (lane & 15);
                        unsigned m = 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) m |= (fabs(C[r]) > thr && ib0 + 4 * r < R && n0 + 4 * r < ib) ? (1u << (4 * r)) : 0u;
                        if (m) {
                            const int w0 = n0 >> 6, sh = n0 & 63;
                            atomicOr(&obits[w0], (unsigned long long)m << sh);
                            if (sh + 13 > 64 && (m >> (64 - sh))) atomicOr(&obits[w0 + 1], (unsigned long long)(m >> (64 - sh)));
                        }
                    }
                }
            };
            if (HALF) {
                // outputs [AR, W) split in two equal ranges; the upper one is served by buffer B (loaded), the lower one by buffer A
                int dsp = AR + (W - AR + 1) / 2;
                dsp = dsp < SPLIT ? SPLIT : dsp;                           // buffer B holds sample dsp - AR onwards
                if (kDetectOnLanes) { detect_range(dsp, W); load_buf(0); detect_range(AR, dsp); }
                else { detect_range_mfma(dsp, W); load_buf(0); detect_range_mfma(AR, dsp); }
            } else if (kDetectOnLanes) detect_range(AR, W);
            else detect_range_mfma(AR, W);
            __builtin_amdgcn_wave_barrier();
            // ---- 5./6. burst fusion, borders, index list (as k_adeclick: integer work)
            for (int wd0 = 0; wd0 < nword; wd0 += 64) {
                const int wd = wd0 + lane;
                const bool inr = wd < nword;
                const unsigned long long cur = inr ? obits[wd] : 0ull;
                const unsigned long long prv = (inr && wd > 0) ? obits[wd - 1] : 0ull;
                const unsigned long long nxt = (wd + 1 < nword) ? obits[wd + 1] : 0ull;
                unsigned long long fused = cur;
                const int nb = P.nburst < 64 ? P.nburst : 64;
                for (int d1 = 1; d1 < nb; ++d1) {
                    const unsigned long long below = (cur << d1) | (prv >> (64 - d1));
                    unsigned long long above = 0ull;
                    for (int d2 = 1; d1 + d2 <= nb; ++d2) above |= (cur >> d2) | (nxt << (64 - d2));
                    fused |= below & above;
                }
                const int lo = AR - wd * 64, hi = (W - AR) - wd * 64;
                const unsigned long long mlo = lo <= 0 ? ~0ull : (lo >= 64 ? 0ull : (~0ull << lo));
                const unsigned long long mhi = hi >= 64 ? ~0ull : (hi <= 0 ? 0ull : ((1ull << hi) - 1ull));
                unsigned long long fb = inr ? (fused & mlo & mhi) : 0ull;
                if (inr) fbits[wd] = fb;
                const int cnt = __popcll(fb);
                int incl = cnt;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) { const int o = __shfl_up(incl, dd, 64); incl += lane >= dd ? o : 0; }
                int slot = F + incl - cnt;
                while (fb) {
                    const int bpos = __ffsll((long long)fb) - 1;
                    if (slot < FCAP) index[slot] = (unsigned short)(wd * 64 + bpos);
                    ++slot; fb &= fb - 1ull;
                }
                F += __builtin_amdgcn_readlane(incl, 63);
            }
            F = __builtin_amdgcn_readfirstlane(F);
            to_heavy = F > FCAP;
            if (!to_heavy) {
                int bwmax = 0;
                for (int k = lane; k < F; k += 64) {
                    const int lim = (int)index[k] + AR;
                    int lo = k, hi = min(F - 1, k + MAXAR);
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)index[mid] <= lim) lo = mid; else hi = mid - 1; }
                    bwv[k] = (unsigned char)(lo - k);
                    bwmax = max(bwmax, lo - k);
                }
#ifdef JT_DK_HIST
                if (LEVEL == 0) {
                    int bm = bwmax;
                    for (int o = 32; o > 0; o >>= 1) bm = max(bm, __shfl_xor(bm, o, 64));
                    if (lane == 0) atomicAdd(&heavy[2 * P.nwindows + min(bm, 63)], 1);
                    for (int k = lane; k < F; k += 64) atomicAdd(&heavy[2 * P.nwindows + 64 + min((int)bwv[k], 63)], 1);
                    if (lane == 0) atomicAdd(&heavy[2 * P.nwindows + 128 + min(F >> 4, 63)], 1);
                }
#endif
                to_heavy = (MODE == 1 || MODE == 3) ? false : __any(bwmax > BWMAX);
                wide = __any(bwmax > 31);
                // rows past the last one are "far away": every ring entry against them is aux[AR + 1] = 0
                for (int k = F + lane; k < F + NC + ND && k < FCAP + NC + ND; k += 64) index[k] = 0xFFFF;
            }
        }
        if (to_heavy) {
            if (LEVEL == 0 && lane == 0) heavy[atomicAdd(&stats[2], 1ull)] = (int)w;
            if (LEVEL == 1 && lane == 0) heavy[P.nwindows + (int64_t)atomicAdd(&stats[3], 1ull)] = (int)w;
            continue;
        }
        DKF_MARK(3)
        bool ok = true;
        if (F > 0) {
            // the window's place in the list of its band class is asked for here and used at the hand-over below: the atomic's round
            // trip hides behind the right-hand side instead of standing at the end of the window
            unsigned long long lslot_ = 0;
            if ((MODE == 1 || MODE == 3) && lane == 0) lslot_ = atomicAdd(&S.ctl[wide ? 1 : 0], 1ull);
            // ---- 7. aux = autocorrelation(ac, AR, AR+1, ., 1.)
            if (lane <= AR) {
                double value = 0.0;
                for (int j = lane; j <= AR; ++j) value = __fma_rn(ac[j], ac[j - lane], value);
                aux[lane] = value;
            } else if (lane == AR + 1) aux[lane] = 0.0;
            __builtin_amdgcn_wave_barrier();
            // ---- 8. right-hand side (flagged samples zeroed in the LDS copy, so the inner loop carries no flag test)
            auto zero_flagged = [&]() {
                const int nb = HALF ? LB : W;
                for (int e = lane; e < F; e += 64) { const int o = (int)index[e] - sb; if (o >= 0 && o < nb) sbuf[o] = 0.0; }
                __builtin_amdgcn_wave_barrier();
            };
            double *gVw = (MODE == 1 || MODE == 3) ? S.rhs + (size_t)w * S.wp : gV;
            auto rhs_range = [&](int ea, int eb) {
                const double *sj = sbuf - sb;
                for (int e0 = ea; e0 < eb; e0 += 128) {
                    int ie[2]; double val[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) { const int e = e0 + 64 * q + lane; ie[q] = index[e < eb ? e : eb - 1]; val[q] = 0.0; }
                    int j0 = -AR;
                    for (; j0 + 7 <= AR; j0 += 8) {
                        double xv[2][8], ax[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int j = j0 + u;
                            ax[u] = aux[j < 0 ? -j : j];
#pragma unroll
                            for (int q = 0; q < 2; ++q) xv[q][u] = sj[ie[q] - j];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) { val[0] = __fma_rn(-xv[0][u], ax[u], val[0]); val[1] = __fma_rn(-xv[1][u], ax[u], val[1]); }
                    }
                    for (; j0 <= AR; ++j0) {
                        const double ax = aux[j0 < 0 ? -j0 : j0];
                        val[0] = __fma_rn(-sj[ie[0] - j0], ax, val[0]);
                        val[1] = __fma_rn(-sj[ie[1] - j0], ax, val[1]);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) { const int e = e0 + 64 * q + lane; if (e < eb) gVw[e] = val[q]; }
                }
            };
            if (HALF) {
                int ea = 0;
                for (int e0 = 0; e0 < F; e0 += 64) { const int e = e0 + lane; ea += __popcll(__ballot(e < F && (int)index[e] + AR < LB)); }
                zero_flagged();
                rhs_range(0, ea);
                if (ea < F) { load_buf(SB_B); zero_flagged(); rhs_range(ea, F); }
            } else { zero_flagged(); rhs_range(0, F); }
            if (MODE == 1 || MODE == 3) {
                // hand the window to the solver of its band class
                for (int e = lane; e < F; e += 64) S.index[(size_t)w * S.wp + e] = index[e];
                if (lane <= AR + 1) S.aux[(size_t)w * 56 + lane] = aux[lane];
                if (lane == 0) {
                    S.F[w] = F;
                    (wide ? S.list64 : S.list32)[lslot_] = (int)w;
                    atomicAdd(&S.hist[(size_t)((wide ? DK_HIST : 0) + min(F, DK_HIST - 1)) * DK_HSTRIDE], 1u);
                }
                DKF_MARK(4)
                continue;
            }
            __threadfence();
            DKF_MARK(4)
            // ---- 9. LDL^T, right-looking inside the band, in the diagonal-major ring; forward substitution fused
            double *blk = sbuf;
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < ND * NC; t += 64) {
                const int dgi = t / NC, i = t & CM;
                const int dlt = (int)index[i + dgi] - (int)index[i];
                blk[dgi * NCP + i] = aux[dlt < AR + 1 ? dlt : AR + 1];
            }
            if (lane < NC) yring[lane] = lane < F ? dk_ld(&gV[lane]) : 0.0;
            // right-hand sides of the rows about to enter the ring: 64 at a time through LDS (ac[] is dead).  A global load inside the
            // pivot loop would put an s_waitcnt vmcnt(0) on every pivot, and that also waits for the factor stores of the same pivot.
            double *vstage = ac;
            int vbase = NC;
            vstage[lane] = (vbase + lane < F) ? dk_ld(&gV[vbase + lane]) : 0.0;
            const int dgl = lane < ND ? lane : ND - 1;                // this lane's diagonal of the pivot column
            double *colp = blk + dgl * NCP;
            double *pairp = blk + (a0 - b0) * NCP;                      // diagonal of this lane's trailing pair
            int a1, b1, a2, b2;                                         // pairs lane + 64 and lane + 128 (bands of 11 .. 18 rows)
            {
                auto unrank = [](int t, int &a, int &b) {
                    a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                    a += ((a + 1) * (a + 2) / 2 <= t); a -= (a * (a + 1) / 2 > t);
                    b = t - a * (a + 1) / 2;
                };
                unrank(lane + 64, a1, b1); unrank(lane + 128, a2, b2);
                if (a1 - b1 >= ND) { a1 = b1 = 0; }                      // (never taken for ND >= 20; keeps the pointers inside the ring)
                if (a2 - b2 >= ND) { a2 = b2 = 0; }
            }
            double *pairp1 = blk + (a1 - b1) * NCP, *pairp2 = blk + (a2 - b2) * NCP;
            // packed factor stream in global scratch: per pivot k the bw multipliers of its column (rows k+1 .. k+bw), then y_k / d_k
            unsigned goff = 0;
            __builtin_amdgcn_wave_barrier();
            int bw = bwv[0];
            for (int k = 0; k < F; ++k) {
                const int ks = k & CM;
                // every LDS read that does not depend on this pivot's arithmetic is issued up front (the chain per pivot is then:
                // column read -> reciprocal -> multipliers -> lane permute -> update -> write, three LDS round trips instead of six)
                const double colv = colp[ks];
                const double yrow = yring[(k + lane) & CM];
                double *ep = pairp + ((k + 1 + b0) & CM);
                const double eold = *ep;
                const int bwn = bwv[k + 1];
                const int nj = k + NC;
                int idn = 0; double vn = 0.0;
                if (nj < F) {
                    if (nj >= vbase + 64) { vbase += 64; vstage[lane] = (vbase + lane < F) ? dk_ld(&gV[vbase + lane]) : 0.0; __builtin_amdgcn_wave_barrier(); }
                    idn = index[nj + dgl];
                    vn = vstage[nj - vbase];
                }
                const double cb = __shfl(colv, b0 + 1, 64);
                const double d = dk_readlane(colv, 0), yk = dk_readlane(yrow, 0);
                if (d == 0.0) { ok = false; break; }
                const int dlt = idn - __builtin_amdgcn_readfirstlane(idn);
                const double ent = aux[dlt < AR + 1 ? dlt : AR + 1];
                const double rinv = dk_rcp(d);
                const double l = colv * rinv;
                const bool inband = (unsigned)(lane - 1) < (unsigned)bw;
                const double la = __shfl(l, a0 + 1, 64);
                if (lane <= bw) gL[goff + (lane ? lane - 1 : bw)] = lane ? l : yk * rinv;
                goff += (unsigned)bw + 1u;
                if (inband) yring[(k + lane) & CM] = __fma_rn(-l, yk, yrow);
                const int npairs = bw * (bw + 1) / 2;
                if (lane < npairs) *ep = __fma_rn(-cb, la, eold);
                if (npairs > 64) {
                    // pairs 64 .. 191 (bands up to 18 rows): this lane's second and third pair, operands by lane permute like the first
                    {
                        const double la1 = __shfl(l, a1 + 1, 64), cb1 = __shfl(colv, b1 + 1, 64);
                        double *e1 = pairp1 + ((k + 1 + b1) & CM);
                        if (lane + 64 < npairs) *e1 = __fma_rn(-cb1, la1, *e1);
                    }
                    if (npairs > 128) {
                        const double la2 = __shfl(l, a2 + 1, 64), cb2 = __shfl(colv, b2 + 1, 64);
                        double *e2 = pairp2 + ((k + 1 + b2) & CM);
                        if (lane + 128 < npairs) *e2 = __fma_rn(-cb2, la2, *e2);
                    }
                    if (npairs > 192) {
                        if (lane < MAXAR + 1) { lvec[lane] = l; cvec[lane] = colv; }
                        __builtin_amdgcn_wave_barrier();
                        for (int t = lane + 192; t < npairs; t += 64) {
                            const int ab = pairtab[t], a = ab & 0xff, b = ab >> 8;
                            double *e = blk + (a - b) * NCP + ((k + 1 + b) & CM);
                            *e = __fma_rn(-cvec[b + 1], lvec[a + 1], *e);
                        }
                    }
                }
                // column k + NC takes the slot of column k (every diagonal), its right-hand side the slot of y_k
                if (nj < F) {
                    if (lane < ND) colp[ks] = ent;
                    if (lane == 0) yring[ks] = vn;
                }
                bw = __builtin_amdgcn_readfirstlane(bwn);
                __builtin_amdgcn_wave_barrier();
            }
            DKF_MARK(5)
            if (ok) {
                __threadfence();
                // ---- 10. back substitution, column form: out[i] = y_i / d_i - A_i, where A_j collects L[i][j] * out[i] from every row i
                // already solved.  Lane t holds A of row i - t; one step = broadcast of lane 0's row, one fused multiply-add, one lane
                // shift (a dot-product form waits for a wave reduction per row).  L[i][i - t] is element t - 1 of pivot (i - t)'s packed
                // segment: every lane walks the segment offsets of "its" pivot k = i - t downwards with the band widths in LDS
                // (seg(k) = seg(k + 1) - (bw_k + 1)); lane 0 fetches y_i / d_i, the last element of row i's own segment.  Rows in
                // batches of 16: their sixteen gathers are in flight together.
                double A = 0.0;
                unsigned segv;
                {
                    const int k1 = F - 1 - lane;                       // the pivot this lane meets first
                    unsigned c = k1 >= 0 ? (unsigned)bwv[k1] + 1u : 0u, inc = c;
#pragma unroll
                    for (int dd = 1; dd < 64; dd <<= 1) { const unsigned o = __shfl_up(inc, dd, 64); inc += lane >= dd ? o : 0u; }
                    segv = goff - (inc - c);                            // = seg(k1) + bw_k1 + 1: the first step subtracts that pivot's segment
                }
                for (int ib = F - 1; ib >= 0; ib -= 16) {
                    double Lr[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int k = ib - q - lane;
                        const int bwk = k >= 0 ? (int)bwv[k] : 0;
                        segv -= (unsigned)bwk + 1u;
                        const bool valid = k >= 0 && ib - q >= 0 && bwk >= lane;      // lane 0: always (its element is y/d)
                        Lr[q] = valid ? dk_ld(&gL[segv + (lane ? (unsigned)lane - 1u : (unsigned)bwk)]) : 0.0;
                    }
                    double xs = 0.0;                                    // lane q: solution of row ib - q
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const double x = dk_readlane(Lr[q], 0) - dk_readlane(A, 0);
                        A = __fma_rn(Lr[q], x, A);
                        {
                            const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(A), 0x130, 0xf, 0xf, true);    // wave_shl:1
                            const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(A), 0x130, 0xf, 0xf, true);
                            A = __hiloint2double(hi, lo);
                        }
                        xs = lane == q ? x : xs;
                    }
                    if (lane < 16 && ib - lane >= 0) blk[ib - lane] = xs;
                }
                __builtin_amdgcn_wave_barrier();
                for (int e = lane; e < F; e += 64) {
                    const int pos = index[e];
                    if (pos >= P.skip && pos < P.skip + P.hop && o0 + (pos - P.skip) < n) out[o0 + (pos - P.skip)] = blk[e];
                }
                repaired += (lane == 0) ? (unsigned long long)F : 0ull;
            } else {
                singular += (lane == 0) ? 1ull : 0ull;
            }
        }
        DKF_MARK(6)
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0 && (repaired || singular)) { atomicAdd(&stats[0], repaired); atomicAdd(&stats[1], singular); }
#ifdef JT_DK_PROFILE
    DKF_MARK(7)
    if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&stats[4 + i], tph[i]);
#endif
}


// Levinson-Durbin for every window of the file, ONE LANE PER WINDOW (autoregression() of af_adeclick.c; the same operations in the
// same order as the in-kernel version above: products a[j] * r[i - j] summed j-ascending, one division per step, the coefficient
// update as a fused multiply-add on the previous step's values).  Order 48 (55 ms windows at 44.1 kHz), fully unrolled: the
// coefficients and r[] live in registers with static indices.  in: r[w * 64 + 0 .. 48]; out: ac[w * 64 + 0 .. 48] = 1, k_1 .. k_48,
// zeros up to 62, sigma_e in slot 63.
__global__ void __launch_bounds__(64, 2)
k_dk_levinson(const double *__restrict__ rin, double *__restrict__ acout, int64_t nwindows)
{
    constexpr int AR = dk::MAXAR;
    const int64_t w = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (w >= nwindows) return;
    double r[AR + 1], a[AR];
    const double *rp = rin + (size_t)w * 64;
#pragma unroll
    for (int i = 0; i <= AR; ++i) r[i] = rp[i];
    const double k0 = -r[1] / r[0];
#pragma unroll
    for (int j = 0; j < AR; ++j) a[j] = 0.0;
    a[0] = k0;
    double alpha = r[0] * (1.0 - k0 * k0);
#pragma unroll
    for (int i = 1; i < AR; ++i) {
        double eps = 0.0;
#pragma unroll
        for (int j = 0; j < i; ++j) eps += a[j] * r[i - j];
        eps += r[i + 1];
        const double ki = -eps / alpha;
        alpha = alpha * (1.0 - ki * ki);
#pragma unroll
        for (int j = 0; j < (i + 1) / 2; ++j) {
            const int m = i - 1 - j;
            const double aj = a[j], am = a[m];
            a[j] = __fma_rn(ki, am, aj);
            if (m != j) a[m] = __fma_rn(ki, aj, am);
        }
        a[i] = ki;
    }
    double *op = acout + (size_t)w * 64;
    op[0] = 1.0;
#pragma unroll
    for (int j = 0; j < AR; ++j) op[j + 1] = a[j];
#pragma unroll
    for (int j = AR + 1; j < 63; ++j) op[j] = 0.0;
    op[63] = sqrt(alpha);
}

// =====================================================================================================================
// k_dk_solve -- the solver of the split pipeline: LDL^T of the banded interpolation matrix, forward and back substitution, repaired
// samples.  The factorisation keeps the sliding trailing block in REGISTERS: a window is served by a group of G lanes, lane t = diagonal
// t, register s = column k0 + s, so entry (i + t, i) of the band lives in lane t.  A pivot then costs
//   * its column is ONE register across the group's lanes (c[p]); the multipliers are that register times 1/d;
//   * the trailing update of entry (k + b + t, k + b) is  c[p + b] -= colv[b] * l[t + b]:  the two operands are a broadcast read and a
//     lane-shifted read of two small LDS vectors the group has just written (static offsets, no address arithmetic), b = 1 .. the
//     widest band of the windows in the wave, in chunks of eight with a wave-uniform exit;
//   * the column that enters the band is one original entry per lane (aux[index[k + BW + 1 + t] - index[k + BW + 1]]);
//   * y slides through one register per lane (wave_shl:1), 1 / d_(k+1) is started as soon as lane 0 has updated its entry.
// Registers are renamed by unrolling eight pivots and moved down by eight columns per block.  With the block out of LDS a window needs
// 7 KB (its index list, right-hand side, band widths and the two vectors): G = 32 packs TWO windows into a wave (bands up to 31 rows:
// 92 % of the bench's windows) and a CU holds twice as many windows as the one-kernel version; G = 64 takes the bands up to 48 rows.
// Same operations on the same values as k_adeclick_fast's ring (multiplier = column entry * 1/d, update = fma(-colv[b], l[a], e),
// 1/d = reciprocal + two Newton steps), so the repaired samples are bit-identical to that kernel's.
__device__ inline double dk_group0(double v, int q, bool two)
{
#ifndef JT_DK_NO_DPP_BCAST
    // two groups of 32 lanes, on the vector pipe: row_newbcast:0 hands every row of 16 lanes its own lane 0 (one 64-bit DPP move), then
    // row_bcast:15 copies lane 15 of rows 0 and 2 -- lane 0's / lane 32's value by now -- over rows 1 and 3.  Three moves and no LDS
    // round trip on a chain that has one broadcast per pivot and one per substitution row (round 6; ds_swizzle with and_mask = 0, two
    // LDS-pipe instructions, was -0.1 ms slower over the two solver launches; through v_readlane it takes four reads, four moves and two selects)
    if (two) {
        long long x = __double_as_longlong(v);
        x = __builtin_amdgcn_update_dpp(x, x, 0x150, 0xf, 0xf, false);
        int lo = (int)x, hi = (int)(x >> 32);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x142, 0xa, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x142, 0xa, 0xf, false);
        return __hiloint2double(hi, lo);
    }
#else
    if (two) return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), 0), __builtin_amdgcn_ds_swizzle(__double2loint(v), 0));
#endif
    const int lo0 = __builtin_amdgcn_readlane(__double2loint(v), 0), hi0 = __builtin_amdgcn_readlane(__double2hiint(v), 0);
    (void)q;
    return __hiloint2double(hi0, lo0);
}
__device__ inline double dk_wave_shl1(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

#define DK_BCH 4
#ifndef DK_SB
#define DK_SB 8        // rows per back-substitution batch (two batches of gathers in flight)
#endif
// LDS reads of the trailing update, written as single ds_read_b64 instructions: the compiler merges neighbouring 8-byte reads into
// ds_read2_b64, which occupies the LDS for 8 cycles where two ds_read_b64 take 2 + 2 (MI355X_MICROARCH.md, LDS table) -- with eleven
// waves per CU each issuing 2 x band-width reads per pivot, the merged form made the solver LDS-bandwidth bound.  The reads are
// volatile asm (program order, behind the stores of the two vectors); dk_lds_wait ties their results to an s_waitcnt so that no use
// can be scheduled above it.
template <int OFF> __device__ __forceinline__ double dk_lds_rd(unsigned a)
{
    double x;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x) : "v"(a), "n"(OFF) : "memory");
    return x;
}
template <int B0, int... U>
__device__ __forceinline__ void dk_round_reads(unsigned cva, unsigned lva, double *cvb, double *lvb, std::integer_sequence<int, U...>)
{
    ((cvb[U] = dk_lds_rd<(B0 + U) * 8>(cva), lvb[U] = dk_lds_rd<(B0 + U) * 8>(lva)), ...);
}
// wait until at most CNT younger LDS operations are outstanding (LDS operations of a wave return in order): the N + N reads of the
// round that is about to be used are complete, the next round's (CNT of them, already issued) may still be in flight
template <int N, int CNT> __device__ __forceinline__ void dk_lds_wait(double *a, double *b)
{
    static_assert(CNT >= 0 && CNT <= 15, "lgkmcnt is four bits");
    if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(CNT));
    else if constexpr (N == 3)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(CNT));
    else if constexpr (N == 2)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(CNT));
    else
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(b[0]) : "n"(CNT));
}
__device__ __forceinline__ unsigned dk_lds_addr(const void *p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
template <typename Fn, int... I>
__device__ __forceinline__ void dk_all_rounds(Fn &&fn, std::integer_sequence<int, I...>)
{
    (void)(fn(std::integral_constant<int, I>{}) && ...);               // left to right, stops at the first round that declines
}
template <int G, int BW, int FC>
__global__ void __launch_bounds__(64, G == 32 ? 3 : 1)
k_dk_solve(double *__restrict__ out, int64_t n, DeclickParams P, DkSplit S, double *scratch, size_t scratch_per_slot,
           unsigned long long *__restrict__ stats)
{
    extern __shared__ unsigned char dk_smem[];
    constexpr int NW = 64 / G, CR = BW + 9;
    constexpr bool TWO = NW == 2, STAGE = G == 32;
    static_assert(G == 32 || G == 64, "group size");
    static_assert(BW < G, "one lane per diagonal");
    constexpr int IDXN = (FC + BW + 16 + G + 7) & ~7;                   // index entries kept per window (sentinels behind the last one)
    constexpr int BWN = (FC + 16 + 7) & ~7, YN = (FC + BW + 16 + 7) & ~7;
    constexpr size_t SLOTB = sizeof(double) * (size_t)(YN + 56 + 6 * G) + 2 * (size_t)IDXN + (size_t)BWN;
    const int lane = threadIdx.x, gl = lane % G, q = lane / G;
    const int AR = P.ar;
    unsigned char *base = dk_smem + (size_t)q * SLOTB;
    double *yv = reinterpret_cast<double *>(base);                      // [YN] right-hand side, later the solution
    double *aux = yv + YN;                                              // [56]
    double *CV = aux + 56;                                              // [2G] pivot column (slot 0: y_k), zeros behind G
    double *LV = CV + 2 * G;                                            // [2G] multipliers, zeros behind G
    double *stg = LV + 2 * G;                                           // [2G] staging ring of the factor stream
    unsigned short *idx = reinterpret_cast<unsigned short *>(stg + 2 * G);  // [IDXN]
    unsigned char *bwv = reinterpret_cast<unsigned char *>(idx + IDXN);     // [BWN]
    CV[G + gl] = 0.0; LV[G + gl] = 0.0;
    // per-column base registers of the trailing update's LDS reads (see the pivot's rounds): opaque to the optimiser
    typedef const __attribute__((address_space(3))) double *dk_ldsp;
    dk_ldsp cvp[DK_BCH], lvp[DK_BCH];
#pragma unroll
    for (int u = 0; u < DK_BCH; ++u) {
        cvp[u] = (dk_ldsp)(CV + 1 + u); lvp[u] = (dk_ldsp)(LV + gl + 1 + u);
        asm volatile("" : "+v"(cvp[u]), "+v"(lvp[u]));
    }
    double *gL = scratch + ((size_t)blockIdx.x * NW + q) * scratch_per_slot;
    const int *list = G == 32 ? S.list32 : S.list64;
    const int64_t nwork = (int64_t)S.ctl[G == 32 ? 0 : 1];
    unsigned long long repaired = 0, singular = 0;
    auto uniform64 = [](unsigned long long v) -> int64_t {
        return (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    auto take = [&]() -> unsigned long long { unsigned long long v = 0; if (lane == 0) v = atomicAdd(&S.ctl[G == 32 ? 2 : 3], (unsigned long long)NW); return v; };
#ifdef JT_DK_PROFILE
    unsigned long long tph[4] = {0, 0, 0, 0}; unsigned long long tc = wall_clock64();
#define DKS_MARK(i) { unsigned long long t_ = wall_clock64(); tph[i] += t_ - tc; tc = t_; }
    // sub-phases of a pivot (shader clocks; every mark first waits for the LDS operations in flight, so a phase owns the round trips it started; stores to memory are not waited for)
    unsigned long long tps[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tcs = 0;
#define DKS_SUB0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tcs = __builtin_amdgcn_s_memtime(); }
#define DKS_SUB(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tps[i] += t_ - tcs; tcs = t_; }
#else
#define DKS_MARK(i)
#define DKS_SUB0()
#define DKS_SUB(i)
#endif
    int64_t wi = uniform64(take());
    for (;;) {
        if (wi >= nwork) break;
        const unsigned long long wnext_ = take();
        const bool have = wi + q < nwork;
        const int64_t w = have ? (int64_t)list[wi + q] : 0;
        const int F = have ? S.F[w] : 0;                               // uniform within the group
        const int Fmax = TWO ? max(__builtin_amdgcn_readlane(F, 0), __builtin_amdgcn_readlane(F, 32)) : __builtin_amdgcn_readlane(F, 0);
        const int64_t o0 = w * P.hop;
        // ---- prologue: index list (sentinels behind it), right-hand side, aux
        {
            const unsigned short *gi = S.index + (size_t)w * S.wp;
            const double *gr = S.rhs + (size_t)w * S.wp;
            const int ni = min(IDXN, ((Fmax + BW + 16 + G) + 7) & ~7), ny = min(YN, Fmax + BW + 16);
            // every load of a trip is issued before the first LDS store (one load -> one store per trip left each of the ~20 trips of a
            // window of 230 flagged samples waiting for a memory round trip of its own: 20 us of a wave's 165 us per pair of windows)
            constexpr int PB = 8;
            double ta[(56 + G - 1) / G];
#pragma unroll
            for (int u = 0; u < (56 + G - 1) / G; ++u) { const int e = gl + u * G; ta[u] = (have && e <= AR + 1 && e < 56) ? S.aux[(size_t)w * 56 + e] : 0.0; }
            for (int e0 = gl; e0 < max(ni, ny); e0 += PB * G) {
                unsigned short ti[PB]; double ty[PB];
#pragma unroll
                for (int u = 0; u < PB; ++u) { const int e = e0 + u * G; ti[u] = e < F ? gi[e] : (unsigned short)0xFFFF; ty[u] = e < F ? gr[e] : 0.0; }
#pragma unroll
                for (int u = 0; u < PB; ++u) { const int e = e0 + u * G; if (e < ni) idx[e] = ti[u]; if (e < ny) yv[e] = ty[u]; }
            }
#pragma unroll
            for (int u = 0; u < (56 + G - 1) / G; ++u) { const int e = gl + u * G; if (e < 56) aux[e] = ta[u]; }
        }
        wi = uniform64(wnext_);
        __builtin_amdgcn_wave_barrier();
        // band width of every pivot (as k_adeclick_fast); zero behind the last one
        {
            const int nb = min(BWN, Fmax + 16);
            for (int k = gl; k < nb; k += G) {
                int bwk = 0;
                if (k < F) {
                    const int lim = (int)idx[k] + AR;
                    int lo = k, hi = min(F - 1, k + dk::MAXAR);
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)idx[mid] <= lim) lo = mid; else hi = mid - 1; }
                    bwk = lo - k;
                }
                bwv[k] = (unsigned char)bwk;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- LDL^T with the forward substitution fused
        double c[CR];
#pragma unroll
        for (int s = 0; s < CR; ++s) {
            if (s <= BW) { const int dlt = (int)idx[s + gl] - (int)idx[s]; c[s] = aux[dlt < AR + 1 ? dlt : AR + 1]; }
            else c[s] = 0.0;
        }
        DKS_MARK(0)
        double yr = yv[gl];                                              // y of row k + gl
        double rinv_n = dk_rcp(c[0]);
        bool okw = true;
        unsigned goff = 0, gfl = 0;                                      // next stream position / first position not yet in global memory
        int bwk = bwv[0];
        for (int k0 = 0; k0 < Fmax; k0 += 8) {
            // one pivot; p (the register of its column) and the update's column offsets are compile-time constants
            auto pivot = [&](auto pc) {
                constexpr int p = decltype(pc)::value;
                const int k = k0 + p;
                const bool live = k < F;
                DKS_SUB0()
                const double colv = c[p];
                const double rinv = dk_group0(rinv_n, q, TWO);
                if (live && rinv != rinv) okw = false;                     // d == 0: the reciprocal's Newton steps turn +-inf into NaN
                const double l = colv * rinv;
                CV[gl] = gl ? colv : yr;
                LV[gl] = l;
                __builtin_amdgcn_wave_barrier();
                DKS_SUB(0)
                const int bwn = bwv[k + 1];
                const int i0 = idx[k + BW + 1], i1 = idx[k + BW + 1 + gl];
                const double ynew = yv[k + BW + 1];
                const double yk = CV[0];
                const int dlt = i1 - i0;
                const double ent = aux[dlt < AR + 1 ? dlt : AR + 1];
                DKS_SUB(1)
                // trailing update, DK_BCH columns per round, wave-uniform exit behind the widest band of the windows in the wave.
                // SOFTWARE-PIPELINED (round 5): a round's 2 x DK_BCH LDS reads are issued one round AHEAD of its FMAs, into the other of
                // two register buffers, and the wait before a round's FMAs leaves the next round's reads in flight (lgkmcnt counts the
                // younger operations; LDS returns in order).  Issued and awaited back to back, every round exposed a full LDS round trip
                // (~100 cycles) in front of 32 cycles of FMAs, seven times per pivot, with three waves per SIMD to cover it (round 4's
                // counters: 32 % active, 57 % waiting).  Same operands, same FMAs, same order: bit-identical.
                // The reads are ordinary LDS loads (the compiler's own s_waitcnt pass then knows what is in flight: an asm read whose
                // result is awaited a round later is invisible to it, and a register move or spill of the still-empty result register
                // silently reads garbage).  What the asm reads of round 4 were for -- keeping neighbouring 8-byte reads from being merged
                // into ds_read2_b64, 8 LDS cycles instead of 2 + 2 -- is done by giving every column of a round its own, opaque base
                // register (cvp[u], lvp[u]: set once per wave): reads from different base registers are never merged.
                // sched_barrier keeps the issue order: next round's reads, this round's FMAs.
                const int bmax = TWO ? max(__builtin_amdgcn_readlane(bwk, 0), __builtin_amdgcn_readlane(bwk, 32)) : __builtin_amdgcn_readlane(bwk, 0);
                double cvb[2][DK_BCH], lvb[2][DK_BCH];
                if (1 <= bmax) {
#pragma unroll
                    for (int u = 0; u < DK_BCH; ++u) if (1 + u <= BW) { cvb[0][u] = cvp[u][0]; lvb[0][u] = lvp[u][0]; }
                    __builtin_amdgcn_sched_barrier(0);
                }
                auto round = [&](auto bc) -> bool {
                    constexpr int R = decltype(bc)::value;
                    constexpr int b0 = 1 + DK_BCH * R;
                    if (b0 > bmax) return false;
                    constexpr int b0n = b0 + DK_BCH;
                    bool next = false;
                    if constexpr (b0n <= BW) {
                        next = b0n <= bmax;
                        if (next) {
#pragma unroll
                            for (int u = 0; u < DK_BCH; ++u) if (b0n + u <= BW) { cvb[(R + 1) & 1][u] = cvp[u][DK_BCH * (R + 1)]; lvb[(R + 1) & 1][u] = lvp[u][DK_BCH * (R + 1)]; }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < DK_BCH; ++u) if (b0 + u <= BW) {
                        c[p + b0 + u] = __fma_rn(-cvb[R & 1][u], lvb[R & 1][u], c[p + b0 + u]);
                        // pin the result here: the optimiser otherwise sinks every round's FMAs below the reads of all later rounds
                        // (their results are only used after the last round) and spills the operands it then holds for them
                        asm volatile("" : "+v"(c[p + b0 + u]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    return next;
                };
                dk_all_rounds(round, std::make_integer_sequence<int, (BW + DK_BCH - 1) / DK_BCH>{});
                DKS_SUB(2)
                // 1 / d of the next pivot: entry (k + 1, k + 1) is lane 0's c[p + 1]
                rinv_n = dk_rcp(c[p + 1]);
                asm volatile("" : "+v"(rinv_n));
                DKS_SUB(3)
                // factor stream: the bw multipliers of the column, then y_k / d_k (lane 0), in one store
                // (a global store per pivot cost a quarter of the factorisation: the stream is staged in a 2 G-entry LDS ring per window and
                //  leaves G values at a time, one coalesced store per three or four pivots)
                //  (G = 64: a pivot of the wide class appends 30-49 values, a flush every other pivot buys nothing: direct store)
                if (STAGE) {
                    if (live) {
                        if (gl <= bwk) stg[(goff + (gl ? (unsigned)gl - 1u : (unsigned)bwk)) & (2u * G - 1u)] = gl ? l : yk * rinv;
                        goff += (unsigned)bwk + 1u;
                    }
                    if (__any(goff - gfl >= (unsigned)G)) {
                        __builtin_amdgcn_wave_barrier();
                        if (goff - gfl >= (unsigned)G) { gL[gfl + (unsigned)gl] = stg[(gfl + (unsigned)gl) & (2u * G - 1u)]; gfl += (unsigned)G; }
                        __builtin_amdgcn_wave_barrier();
                    }
                } else if (live) {
                    if (gl <= bwk) gL[goff + (gl ? (unsigned)gl - 1u : (unsigned)bwk)] = gl ? l : yk * rinv;
                    goff += (unsigned)bwk + 1u;
                }
                DKS_SUB(4)
                // forward substitution and the slide of y
                const double yupd = __fma_rn(-l, yk, yr);
                yr = dk_wave_shl1(yupd);
                if (gl == BW) yr = ynew;
                c[p + BW + 1] = ent;
                bwk = bwn;
                __builtin_amdgcn_wave_barrier();
                DKS_SUB(5)
            };
            pivot(std::integral_constant<int, 0>{}); pivot(std::integral_constant<int, 1>{}); pivot(std::integral_constant<int, 2>{});
            pivot(std::integral_constant<int, 3>{}); pivot(std::integral_constant<int, 4>{}); pivot(std::integral_constant<int, 5>{});
            pivot(std::integral_constant<int, 6>{}); pivot(std::integral_constant<int, 7>{});
#pragma unroll
            for (int s = 0; s <= BW; ++s) c[s] = c[s + 8];
        }
        __builtin_amdgcn_wave_barrier();
        if (STAGE && (unsigned)gl < goff - gfl) gL[gfl + (unsigned)gl] = stg[(gfl + (unsigned)gl) & (2u * G - 1u)];      // what is left in the ring
        DKS_MARK(1)
        // ---- back substitution, column form (as k_adeclick_fast): lane t of the group holds the sum collected for row i - t
        {
            __threadfence();
            double A = 0.0;
            unsigned segv;
            {
                const int k1 = F - 1 - gl;
                const unsigned cc = k1 >= 0 ? (unsigned)bwv[k1] + 1u : 0u; unsigned inc = cc;
#pragma unroll
                for (int dd = 1; dd < G; dd <<= 1) { const unsigned o = __shfl_up(inc, dd, G); inc += gl >= dd ? o : 0u; }
                segv = goff - (inc - cc);
            }
            // rows ib .. ib - DK_SB + 1 of the longer window per batch (DK_SB rows); this group's window is Fmax - F rows shorter: its row index is shifted.
            // The sixteen gathers of the NEXT batch are in flight while the current one is solved (their addresses only need bwv[]).
            auto gather = [&](int ib, double (&Lr)[DK_SB]) {
#pragma unroll
                for (int r = 0; r < DK_SB; ++r) {
                    const int i = ib - r - (Fmax - F);                      // this window's row (negative: nothing to do any more)
                    const int k = i - gl;
                    const bool act = i >= 0 && k >= 0;
                    const int bwk2 = act ? (int)bwv[k] : 0;
                    if (act) segv -= (unsigned)bwk2 + 1u;
                    const bool valid = act && bwk2 >= gl;
                    Lr[r] = valid ? dk_ld(&gL[segv + (gl ? (unsigned)gl - 1u : (unsigned)bwk2)]) : 0.0;
                }
            };
            auto solve16 = [&](int ib, double (&Lr)[DK_SB]) {
                double xs = 0.0;
#pragma unroll
                for (int r = 0; r < DK_SB; ++r) {
                    const double x = dk_group0(Lr[r] - A, q, TWO);               // lane 0 of the group: y_i / d_i minus what the rows below left
                    A = __fma_rn(Lr[r], x, A);
                    A = dk_wave_shl1(A);
                    if (gl == G - 1) A = 0.0;                               // (the lane behind the group's last one belongs to the other window)
                    xs = gl == r ? x : xs;
                }
                const int i = ib - gl - (Fmax - F);
                if (gl < DK_SB && i >= 0) yv[i] = xs;
            };
            double La[DK_SB], Lb[DK_SB];
            gather(Fmax - 1, La);
            for (int ib = Fmax - 1; ib >= 0; ib -= 2 * DK_SB) {
                if (ib - DK_SB >= 0) gather(ib - DK_SB, Lb);
                solve16(ib, La);
                if (ib - DK_SB < 0) break;
                if (ib - 2 * DK_SB >= 0) gather(ib - 2 * DK_SB, La);
                solve16(ib - DK_SB, Lb);
            }
        }
        __builtin_amdgcn_wave_barrier();
        DKS_MARK(2)
        if (okw) {
            for (int e = gl; e < F; e += G) {
                const int pos = idx[e];
                if (pos >= P.skip && pos < P.skip + P.hop && o0 + (pos - P.skip) < n) out[o0 + (pos - P.skip)] = yv[e];
            }
            repaired += gl == 0 ? (unsigned long long)F : 0ull;
        } else {
            singular += (gl == 0 && have) ? 1ull : 0ull;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (gl == 0 && (repaired || singular)) { atomicAdd(&stats[0], repaired); atomicAdd(&stats[1], singular); }
#ifdef JT_DK_PROFILE
    DKS_MARK(3)
    if (lane == 0 && G == 32) for (int i = 0; i < 4; ++i) atomicAdd(&S.ctl[4 + i], tph[i]);
    if (lane == 0) for (int i = 0; i < 6; ++i) atomicAdd(&S.ctl[(G == 32 ? 8 : 16) + i], tps[i]);
#endif
}

// k_dk_sort_scan / _scatter -- the two solver lists, longest window first (counting sort by the window's flagged samples F: the front
// kernel counts the windows per F as it appends them, one workgroup per list turns the counts into first places, one thread per window
// takes a place).
// A wave of the <32> solver serves two windows and runs for the LONGER one; a launch ends with its last window.  The front kernels append
// the windows in the order their waves finish, file order more or less: speech alternates between windows with a few dozen flagged
// samples and windows with ~300, so neighbours in the list paired 166 flagged samples on average with a partner of 229 (the bench file:
// sum of max(F) over the pairs = 1.38 x half the sum of F), and the <48> launch's last windows were as likely long as short.  Sorted,
// a pair's windows differ by a sample or two and the longest windows start first.  A window's result does not depend on its partner
// or its place in the list (the partner only adds update rounds on structural zeros), so the order inside one F is left to the atomics.
__global__ void __launch_bounds__(1024)
k_dk_sort_scan(DkSplit S)
{
    // hist[f] <- number of windows of the list with MORE flagged samples than f: the first place of the count's windows in the sorted list
    __shared__ unsigned a[2][1024];
    unsigned *h = S.hist + (size_t)blockIdx.x * DK_HIST * DK_HSTRIDE;
    const int t = threadIdx.x, f = DK_HIST - 1 - t;                   // t ascends as the count descends
    unsigned v = t < DK_HIST ? h[(size_t)f * DK_HSTRIDE] : 0u;
    const unsigned own = v;
    int cur = 0;
    a[0][t] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        v = a[cur][t] + (t >= d ? a[cur][t - d] : 0u);
        cur ^= 1;
        a[cur][t] = v;
        __syncthreads();
    }
    if (t < DK_HIST) h[(size_t)f * DK_HSTRIDE] = v - own;
}
__global__ void __launch_bounds__(256)
k_dk_sort_scatter(DkSplit S, int64_t nwindows, int *__restrict__ sorted)
{
    const int cls = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)S.ctl[cls]) return;
    const int w = (cls ? S.list64 : S.list32)[i];
    sorted[(size_t)cls * nwindows + atomicAdd(&S.hist[(size_t)(cls * DK_HIST + min(max(S.F[w], 0), DK_HIST - 1)) * DK_HSTRIDE], 1u)] = w;
}

// host side --------------------------------------------------------------------------------------------------------
// Overlap-add of method 'a': af_adeclick.c adds every window's W weighted samples into a buffer that starts at zero and hands out the
// first hop of it, so output sample p = k * hop + j is ((0 + c[k - i_max]) + ...) + c[k], c[k - i] = window (k - i)'s product at offset
// j + i * hop, the windows in the order they were processed (one add per window, unfused).
__global__ void __launch_bounds__(256)
k_dk_overlap_add(const double *__restrict__ prod, double *__restrict__ out, int64_t n, int hop, int W, int wp)
{
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int64_t k = p / hop; const int j = (int)(p - k * hop);
    double acc = 0.0;
    for (int i = (W - 1 - j) / hop; i >= 0; --i) {
        if (k - i < 0) continue;
        acc = __dadd_rn(acc, prod[(size_t)(k - i) * wp + j + i * hop]);
    }
    out[p] = acc;
}

bool jt_adeclick_supported(int sample_rate, double window_ms, double overlap_pct, double ar_pct, int method, std::string *why)
{
    int W = (int)(sample_rate * window_ms / 1000.); if (W < 100) W = 100;
    int ar = (int)(W * ar_pct / 100.); if (ar < 1) ar = 1;
    (void)overlap_pct;
    if (method != 0 && method != 1) { if (why) *why = "adeclick: method must be 0 (overlap-add, m=a) or 1 (overlap-save, m=s)"; return false; }
    if (W > dk::MAXW) { if (why) *why = "adeclick: window too long for this build"; return false; }
    if (ar > dk::XAR_LIMIT) { if (why) *why = "adeclick: AR order above 62 is not built"; return false; }
    return true;
}

void launch_adeclick(jt_ctx *h, const double *in, double *out, int64_t n, int sample_rate, double threshold, double window_ms,
                     double overlap_pct, double ar_pct, double burst, double gain, unsigned long long *d_stats, hipStream_t s, int method)
{
    // control words (per-XCD work heads, list lengths: [0, 288)) and the stage's statistics ([288, 304)) share one buffer and ONE fill;
    // d_stats == nullptr: the statistics live there (jt_adeclick_stats)
    constexpr size_t CTLN = 304 + (size_t)DK_HIST * DK_HSTRIDE;          // (+ 2 x DK_HIST 32-bit counts of the list sort, a 128-byte line each)
    h->declick_ctl.ensure(CTLN);
    JT_HIP(hipMemsetAsync(h->declick_ctl.p, 0, CTLN * sizeof(unsigned long long), s));
    if (!d_stats) d_stats = h->declick_ctl.p + 288;
    DeclickParams P;
    P.W = (int)(sample_rate * window_ms / 1000.); if (P.W < 100) P.W = 100;
    P.ar = (int)(P.W * ar_pct / 100.); if (P.ar < 1) P.ar = 1;
    P.nburst = (int)(P.W * burst / 1000.);
    P.hop = (int)(P.W * (1. - (overlap_pct / 100.))); if (P.hop < 1) P.hop = 1;
    P.skip = method ? (P.W - P.hop) / 2 : 0;
    P.threshold = threshold; P.gain = gain;
    P.nwindows = (n + P.hop - 1) / P.hop;
    P.method = method; P.wp = (P.W + 7) & ~7; P.wlut = nullptr; P.prod = nullptr;
    if (method == 0) {
        // overlap-add is not the reference's configuration (filters.go:513-521 sets m=s): it takes the sequential-order kernel, which has
        // two output sites, and pays a product buffer of W doubles per window
        std::vector<double> lut((size_t)P.W);
        for (int i = 0; i < P.W; ++i) lut[(size_t)i] = std::sin(M_PI * i / P.W) * (1. - (overlap_pct / 100.)) * M_PI_2;
        h->declick_wlut.ensure((size_t)P.W); h->declick_prod.ensure((size_t)P.nwindows * P.wp);
        JT_HIP(hipMemcpyAsync(h->declick_wlut.p, lut.data(), sizeof(double) * (size_t)P.W, hipMemcpyHostToDevice, s));
        JT_HIP(hipStreamSynchronize(s));                       // (pageable source about to go out of scope)
        P.wlut = h->declick_wlut.p; P.prod = h->declick_prod.p;
    }
    constexpr int LIGHT = 512, LBS = 33, MID = 1024;
    // half-window buffer: every +-AR neighbourhood must fit one of the two halves (lb >= (W + 2 AR) / 2), multiple of 8
    P.lb = ((P.W + 2 * P.ar + 1) / 2 + 16 + 7) & ~7;
    const bool half_ok = P.lb < P.W && P.ar < P.lb / 4;
    auto sa_for = [&](bool half, int bs) { return ((half ? std::max(P.lb, bs * bs) : std::max(P.W, bs * bs)) + 1) & ~1; };
    auto smem_for = [&](int sa, int fcap, int bs) {
        return sizeof(double) * (size_t)(sa + 5 * (dk::XMAXAR + 2)) + sizeof(unsigned long long) * 2 * dk::NWORD + (size_t)fcap * 3 + 16 + (size_t)bs * (bs - 1);
    };
    // the bit-exact sequential-order kernel: parity tests / A-B, m=a, and AR orders above the 48 the fast kernels are laid out for
    // (its overflow levels then carry a 63 x 63 block instead of 49 x 49)
    const bool wide_ar = P.ar > dk::MAXAR;
    const JtOpts &O = h->opts;
    const bool exact = O.adeclick_exact || method == 0 || wide_ar;
    const int bsx = wide_ar ? dk::XBS : dk::BS;
    const int sa0 = sa_for(half_ok, LBS), sa1 = sa_for(half_ok, bsx), sa2 = sa_for(false, bsx);
    const size_t sm0 = smem_for(sa0, LIGHT, LBS), sm1 = smem_for(sa1, MID, bsx), sm2 = smem_for(sa2, dk::MAXW, bsx);
    JT_REQUIRE(sm2 <= 64 * 1024 || !exact, JT_E_UNSUPPORTED, "adeclick: window does not fit the per-wave LDS budget");
    const size_t per_wave = (size_t)P.W * (exact ? dk::XMAXAR : dk::MAXAR) + 3 * (size_t)P.W;
    h->declick_heavy.ensure(2 * (size_t)P.nwindows + 192);
    if (JT_AB_ON(O.dk_profile)) JT_HIP(hipMemsetAsync(h->declick_heavy.p + 2 * P.nwindows, 0, 192 * sizeof(int), s));
#define DK_LAUNCH(KERN, GRID, SMEM, PP) do { auto k_ = KERN; \
        JT_HIP(hipFuncSetAttribute((const void *)k_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SMEM))); \
        hipLaunchKernelGGL(k_, dim3((unsigned)(GRID)), dim3(64), (SMEM), s, in, out, n, PP, h->declick_scr.p, per_wave, d_stats, h->declick_heavy.p); } while (0)
#define DKF_LAUNCH(KERN, GRID, SMEM, PP) do { auto k_ = KERN; \
        JT_HIP(hipFuncSetAttribute((const void *)k_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SMEM))); \
        hipLaunchKernelGGL(k_, dim3((unsigned)(GRID)), dim3(64), (SMEM), s, in, out, n, PP, h->declick_scr.p, per_wave, d_stats, h->declick_heavy.p, SP); } while (0)
    if (!exact) {
        // fast kernel: ring NC columns x ND diagonals (bands up to ND - 1 rows); flag words sized by the window
        P.nw = ((P.W + 63) >> 6) + 2;
        constexpr int NC0 = 32, ND0 = 32, NC1 = 64, ND1 = dk::BS;
        auto sa_fast = [&](bool half, int ring) { return ((half ? std::max(P.lb + 64, ring) : std::max(P.W + 64, ring)) + 1) & ~1; };
        auto smem_fast = [&](int sa, int fcap, int nc, int nd) {
            return sizeof(double) * (size_t)(sa + 50 + 64 + 50 + 50 + 50 + nc) + sizeof(unsigned long long) * 2 * (size_t)P.nw +
                   2 * (size_t)(fcap + nc + nd) + (size_t)fcap + 2 * (size_t)std::max(0, nd * (nd - 1) / 2 - 64) + 16;
        };
        const int fa0 = sa_fast(half_ok, 41 * ND0), fa1 = sa_fast(half_ok, (NC1 + 1) * ND1), fa2 = sa_fast(false, (NC1 + 1) * ND1);
        const size_t fm0 = smem_fast(fa0, LIGHT, NC0, ND0), fm1 = smem_fast(fa1, MID, NC1, ND1), fm2 = smem_fast(fa2, dk::MAXW, NC1, ND1);
        JT_REQUIRE(fm2 <= 64 * 1024, JT_E_UNSUPPORTED, "adeclick: window does not fit the per-wave LDS budget");
        int v0 = (int)std::min<size_t>(12, (160 * 1024) / fm0);
        const int v1 = (int)std::min<size_t>(8, (160 * 1024) / fm1), v2 = (int)std::min<size_t>(8, (160 * 1024) / fm2);
        if (JT_AB_ON(O.dk_waves > 0)) v0 = std::max(1, std::min(v0, O.dk_waves));
        const int64_t f0 = std::min<int64_t>(P.nwindows, (int64_t)256 * v0), f1 = std::min<int64_t>(P.nwindows, (int64_t)256 * v1),
                      f2 = std::min<int64_t>(P.nwindows, (int64_t)256 * v2);
        if (JT_AB_ON(O.adeclick_fused)) h->declick_scr.ensure(per_wave * (size_t)std::max(f0, std::max(f1, f2)));
        DeclickParams Q0 = P, Q1 = P, Q2 = P; Q0.sa = fa0; Q1.sa = fa1; Q2.sa = fa2;
        DkSplit SP{};
        // Split pipeline (default): the front kernel leaves every window's index list / right-hand side in global memory, two solver
        // launches (bands up to 31 rows: two windows per wave; up to 48: one) factor and substitute with the trailing block in registers.
        // Option adeclick_fused (JT_AB build) keeps everything in the one kernel of round 2 (A/B, and the reference the split is tested against).
        const bool fused = JT_AB_ON(O.adeclick_fused);
        if (!fused) {
            constexpr int FCS = LIGHT, G32 = 32, BW32 = 31, G64 = 64, BW64 = 48;
            const int wp = (P.W + 63) & ~63;
            h->declick_F.ensure((size_t)P.nwindows); h->declick_lists.ensure(4 * (size_t)P.nwindows);      // two lists as appended, two sorted
            h->declick_idx.ensure((size_t)P.nwindows * wp); h->declick_rhs.ensure((size_t)P.nwindows * wp);
            h->declick_aux.ensure((size_t)P.nwindows * 56);
            // (r[] and the AR polynomial share one buffer: 64 doubles each per window)
            const bool lev_split = P.ar == dk::MAXAR && !JT_AB_ON(O.dk_levinson_in_kernel);
            if (lev_split) h->declick_r.ensure(2 * (size_t)P.nwindows * 64);
            SP.F = h->declick_F.p; SP.index = h->declick_idx.p; SP.rhs = h->declick_rhs.p; SP.aux = h->declick_aux.p;
            SP.list32 = h->declick_lists.p; SP.list64 = h->declick_lists.p + P.nwindows; SP.ctl = h->declick_ctl.p; SP.wp = wp;
            SP.hist = reinterpret_cast<unsigned *>(h->declick_ctl.p + 304);
            SP.xcd = JT_AB_ON(O.dk_no_xcd) ? 0 : 1;
            SP.r = lev_split ? h->declick_r.p : nullptr; SP.ac = lev_split ? h->declick_r.p + (size_t)P.nwindows * 64 : nullptr;
            auto slot_bytes = [&](int G, int BW) {
                const int idxn = (FCS + BW + 16 + G + 7) & ~7, bwn = (FCS + 16 + 7) & ~7, yn = (FCS + BW + 16 + 7) & ~7;
                return sizeof(double) * (size_t)(yn + 56 + 6 * G) + 2 * (size_t)idxn + (size_t)bwn;
            };
            const size_t sm32 = 2 * slot_bytes(G32, BW32) + 16, sm64 = slot_bytes(G64, BW64) + 16;
            const int w32 = (int)std::min<size_t>(16, (160 * 1024) / sm32), w64 = (int)std::min<size_t>(8, (160 * 1024) / sm64);
            const int64_t g32 = std::min<int64_t>((P.nwindows + 1) / 2, (int64_t)256 * w32), g64 = std::min<int64_t>(P.nwindows, (int64_t)256 * w64);
            const size_t slot32 = (size_t)FCS * (BW32 + 1), slot64 = (size_t)FCS * (BW64 + 1);
            // the level-1 / level-2 kernels below (windows with more than LIGHT flagged samples) index the same scratch per wave
            h->declick_scr.ensure(std::max(per_wave * (size_t)std::max(f1, f2), slot32 * 2 * (size_t)g32 + slot64 * (size_t)g64));
            if (lev_split) {
                // autocorrelation (+ the pass-through copy) -> Levinson-Durbin, one lane per window -> detector .. right-hand side; the two
                // front launches hand their windows out with the same counter, which is reset in between
                if (half_ok) DKF_LAUNCH((k_adeclick_fast<LIGHT, NC0, ND0, true, 0, 2>), f0, fm0, Q0);
                else DKF_LAUNCH((k_adeclick_fast<LIGHT, NC0, ND0, false, 0, 2>), f0, fm0, Q0);
                hipLaunchKernelGGL(k_dk_levinson, dim3((unsigned)((P.nwindows + 63) / 64)), dim3(64), 0, s, SP.r, SP.ac, P.nwindows);
                if (!SP.xcd) JT_HIP(hipMemsetAsync(d_stats + 12, 0, sizeof(unsigned long long), s));      // (the per-XCD heads of the two launches are separate words)
                if (half_ok) DKF_LAUNCH((k_adeclick_fast<LIGHT, NC0, ND0, true, 0, 3>), f0, fm0, Q0);
                else DKF_LAUNCH((k_adeclick_fast<LIGHT, NC0, ND0, false, 0, 3>), f0, fm0, Q0);
            } else if (half_ok) DKF_LAUNCH((k_adeclick_fast<LIGHT, NC0, ND0, true, 0, 1>), f0, fm0, Q0);
            else DKF_LAUNCH((k_adeclick_fast<LIGHT, NC0, ND0, false, 0, 1>), f0, fm0, Q0);
            if (!O.dk_unsorted) {
                hipLaunchKernelGGL(k_dk_sort_scan, dim3(2), dim3(1024), 0, s, SP);
                hipLaunchKernelGGL(k_dk_sort_scatter, dim3((unsigned)((P.nwindows + 255) / 256), 2), dim3(256), 0, s, SP, (int64_t)P.nwindows,
                                   h->declick_lists.p + 2 * (size_t)P.nwindows);
                SP.list32 = h->declick_lists.p + 2 * (size_t)P.nwindows; SP.list64 = h->declick_lists.p + 3 * (size_t)P.nwindows;
            }
            {
                auto k32 = k_dk_solve<G32, BW32, FCS>; auto k64 = k_dk_solve<G64, BW64, FCS>;
                JT_HIP(hipFuncSetAttribute((const void *)k32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm32));
                JT_HIP(hipFuncSetAttribute((const void *)k64, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm64));
                // the two solver launches are independent (disjoint windows, disjoint scratch): the wide-band one runs beside the other
                // on a stream of its own
                // (dk_stream and dk_ev[] belong to the handle: jt_open_ex creates them -- an alias of the main stream for a one-stream handle)
                if (!h->dk_stream || !h->dk_ev[0] || !h->dk_ev[1]) throw JtError{JT_E_STATE, "adeclick: the handle has no solver stream"};
                if (JT_AB_ON(O.dk_serial)) {
                    hipLaunchKernelGGL(k64, dim3((unsigned)g64), dim3(64), sm64, s, out, n, P, SP, h->declick_scr.p + slot32 * 2 * (size_t)g32, slot64, d_stats);
                    hipLaunchKernelGGL(k32, dim3((unsigned)g32), dim3(64), sm32, s, out, n, P, SP, h->declick_scr.p, slot32, d_stats);
                } else {
                JT_HIP(hipEventRecord(h->dk_ev[0], s));
                JT_HIP(hipStreamWaitEvent(h->dk_stream, h->dk_ev[0], 0));
                hipLaunchKernelGGL(k64, dim3((unsigned)g64), dim3(64), sm64, h->dk_stream, out, n, P, SP, h->declick_scr.p + slot32 * 2 * (size_t)g32, slot64, d_stats);
                JT_HIP(hipEventRecord(h->dk_ev[1], h->dk_stream));
                hipLaunchKernelGGL(k32, dim3((unsigned)g32), dim3(64), sm32, s, out, n, P, SP, h->declick_scr.p, slot32, d_stats);
                // the wide-band launch ends last (5.5 against 4.3 ms on the bench file); a queue takes ~70-140 us to notice another queue's event
                // on this part, a spinning host thread ~25 (jt_pass2's head) -- and this thread has nothing to queue that could start
                // before the solvers are done.  Handles that poll with sleeps (pools) and option dk_device_join keep the wait in the queue.
                if (h->blocking || h->dk_stream == s || O.dk_device_join) JT_HIP(hipStreamWaitEvent(s, h->dk_ev[1], 0));
                else JT_HIP(jt_event_wait(h, h->dk_ev[1]));
                }
            }
            if (half_ok) DKF_LAUNCH((k_adeclick_fast<MID, NC1, ND1, true, 1>), f1, fm1, Q1);
            else DKF_LAUNCH((k_adeclick_fast<MID, NC1, ND1, false, 1>), f1, fm1, Q1);
            DKF_LAUNCH((k_adeclick_fast<dk::MAXW, NC1, ND1, false, 2>), f2, fm2, Q2);
            if (JT_AB_ON(O.dk_profile)) {
                unsigned long long c[24];
                JT_HIP(hipStreamSynchronize(s));
                JT_HIP(hipMemcpy(c, h->declick_ctl.p, sizeof c, hipMemcpyDeviceToHost));
                fprintf(stderr, "adeclick split: %llu windows in the 31-row class, %llu in the 48-row class; solver<32> clocks (prologue, factorisation, "
                                "back substitution, output): %llu %llu %llu %llu\n", c[0], c[1], c[4], c[5], c[6], c[7]);
                // how the lists pair and queue the windows: a wave of the <32> solver runs for the LONGER of its two windows, a launch ends
                // with its last window
                fprintf(stderr, "  pivot sub-phases, shader clocks summed over waves (broadcast + multiplier + column store, scalar LDS reads, update rounds, "
                                "reciprocal, stream staging, forward substitution): <32> %llu %llu %llu %llu %llu %llu  <48> %llu %llu %llu %llu %llu %llu\n",
                        c[8], c[9], c[10], c[11], c[12], c[13], c[16], c[17], c[18], c[19], c[20], c[21]);
                std::vector<int> hF((size_t)P.nwindows), hl((size_t)2 * P.nwindows);      // (the lists the solvers ran on: sorted unless dk_unsorted)
                JT_HIP(hipMemcpy(hF.data(), h->declick_F.p, sizeof(int) * hF.size(), hipMemcpyDeviceToHost));
                JT_HIP(hipMemcpy(hl.data(), SP.list32, sizeof(int) * hl.size(), hipMemcpyDeviceToHost));
                for (int cls = 0; cls < 2; ++cls) {
                    const int *l = hl.data() + (cls ? P.nwindows : 0); const int64_t m = (int64_t)c[cls];
                    int64_t sum = 0, summax = 0, hist[9] = {0}; int fmx = 0;
                    for (int64_t i = 0; i < m; ++i) { const int f = hF[l[i]]; sum += f; fmx = std::max(fmx, f); ++hist[std::min(8, f / 64)]; }
                    for (int64_t i = 0; i < m; i += 2) summax += std::max(hF[l[i]], i + 1 < m ? hF[l[i + 1]] : 0);
                    fprintf(stderr, "  class %d: %lld windows, F sum %lld (mean %.1f, max %d), sum over list pairs of max(F) %lld (x%.3f of half the sum); F/64 histogram:",
                            cls ? 48 : 31, (long long)m, (long long)sum, m ? (double)sum / m : 0., fmx, (long long)summax, sum ? 2.0 * summax / sum : 0.);
                    for (int b = 0; b < 9; ++b) fprintf(stderr, " %lld", (long long)hist[b]);
                    fprintf(stderr, "\n");
                }
            }
            return;
        }
#ifdef JT_AB
        if (half_ok) {
            DKF_LAUNCH((k_adeclick_fast<LIGHT, NC0, ND0, true, 0>), f0, fm0, Q0);
            DKF_LAUNCH((k_adeclick_fast<MID, NC1, ND1, true, 1>), f1, fm1, Q1);
        } else {
            DKF_LAUNCH((k_adeclick_fast<LIGHT, NC0, ND0, false, 0>), f0, fm0, Q0);
            DKF_LAUNCH((k_adeclick_fast<MID, NC1, ND1, false, 1>), f1, fm1, Q1);
        }
        DKF_LAUNCH((k_adeclick_fast<dk::MAXW, NC1, ND1, false, 2>), f2, fm2, Q2);
#endif
        return;
    }
    // resident waves per CU = what LDS admits (10 at the 44.1 kHz defaults).  With the windows handed out dynamically the time
    // falls monotonically with residency on the 60-min workload: 6 -> 33.8 ms, 7 -> 30.7, 8 -> 28.5, 9 -> 27.6
    int w0 = (int)std::min<size_t>(12, (160 * 1024) / sm0);
    const int w1 = (int)std::min<size_t>(8, (160 * 1024) / sm1), w2 = (int)std::min<size_t>(8, (160 * 1024) / sm2);
    if (JT_AB_ON(O.dk_waves > 0)) w0 = std::max(1, std::min(w0, O.dk_waves));      // occupancy experiments
    const int64_t g0 = std::min<int64_t>(P.nwindows, (int64_t)256 * w0), g1 = std::min<int64_t>(P.nwindows, (int64_t)256 * w1),
                  g2 = std::min<int64_t>(P.nwindows, (int64_t)256 * w2);
    h->declick_scr.ensure(per_wave * (size_t)std::max(g0, std::max(g1, g2)));
    DeclickParams P0 = P, P1 = P, P2 = P; P0.sa = sa0; P1.sa = sa1; P2.sa = sa2;
    // pass 0: 512 flags / 33 x 33 block per window; pass 1: the windows that overflowed that (bands wider than 32 rows: 7 % of the
    // bench signal), 1024 flags / 49 x 49; pass 2: anything denser still, full capacity.  List lengths are read on the device.
    if (wide_ar) {
        if (half_ok) { DK_LAUNCH((k_adeclick<LIGHT, LBS, true, 0>), g0, sm0, P0); DK_LAUNCH((k_adeclick<MID, dk::XBS, true, 1>), g1, sm1, P1); }
        else { DK_LAUNCH((k_adeclick<LIGHT, LBS, false, 0>), g0, sm0, P0); DK_LAUNCH((k_adeclick<MID, dk::XBS, false, 1>), g1, sm1, P1); }
        DK_LAUNCH((k_adeclick<dk::MAXW, dk::XBS, false, 2>), g2, sm2, P2);
    } else {
    if (half_ok) {
        DK_LAUNCH((k_adeclick<LIGHT, LBS, true, 0>), g0, sm0, P0);
        DK_LAUNCH((k_adeclick<MID, dk::BS, true, 1>), g1, sm1, P1);
    } else {
        DK_LAUNCH((k_adeclick<LIGHT, LBS, false, 0>), g0, sm0, P0);
        DK_LAUNCH((k_adeclick<MID, dk::BS, false, 1>), g1, sm1, P1);
    }
    DK_LAUNCH((k_adeclick<dk::MAXW, dk::BS, false, 2>), g2, sm2, P2);
    }
    if (method == 0) hipLaunchKernelGGL(k_dk_overlap_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, P.prod, out, n, P.hop, P.W, P.wp);
#undef DK_LAUNCH
#undef DKF_LAUNCH
}
