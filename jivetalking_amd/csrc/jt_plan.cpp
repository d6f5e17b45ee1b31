// jt_plan.cpp — host-side design of the constant tables the kernels consume (filter coefficients, resampler
// phase banks, afftdn band tables) and the O(#blocks) finishing steps of the R128 measurements (gating,
// histograms, percentiles) and astats (ordered merge of per-chunk summaries).  Scalar double arithmetic; none
// of this touches sample data.
#include "jt_internal.h"
#include <algorithm>
#include <cfloat>
#include <thread>
#include <exception>
#include <deque>
#include <condition_variable>
#include <mutex>
#include <functional>

// ---------------------------------------------------------------- biquads (af_biquads.c config_filter)
void jt_biquad_design(int type, double freq, double q, int sr, double b[3], double a[3], int normalize)
{
    const double w0 = 2 * M_PI * freq / sr;
    const double alpha = std::sin(w0) / (2 * q);
    const double cw = std::cos(w0);
    a[0] = 1 + alpha; a[1] = -2 * cw; a[2] = 1 - alpha;
    if (type == 0) { b[0] = (1 + cw) / 2; b[1] = -(1 + cw); b[2] = (1 + cw) / 2; }
    else           { b[0] = (1 - cw) / 2; b[1] = 1 - cw;    b[2] = (1 - cw) / 2; }
    const double a0 = a[0];
    a[1] /= a0; a[2] /= a0; b[0] /= a0; b[1] /= a0; b[2] /= a0; a[0] = 1.0;
    if (normalize && std::fabs(b[0] + b[1] + b[2]) > 1e-6) {
        const double factor = (a[0] + a[1] + a[2]) / (b[0] + b[1] + b[2]);
        b[0] *= factor; b[1] *= factor; b[2] *= factor;
    }
}

// f_ebur128.c config_audio_input(): "reversed parametrization of PRE and RLB from 48 kHz"
void jt_kweight_design(int sr, BiquadF64 *pre, BiquadF64 *rlb)
{
    double f0 = 1681.974450955533, G = 3.999843853973347, Q = 0.7071752369554196;
    double K = std::tan(M_PI * f0 / (double)sr);
    double Vh = std::pow(10.0, G / 20.0);
    double Vb = std::pow(Vh, 0.4996667741545416);
    double a0 = 1.0 + K / Q + K * K;
    pre->b0 = (Vh + Vb * K / Q + K * K) / a0;
    pre->b1 = 2.0 * (K * K - Vh) / a0;
    pre->b2 = (Vh - Vb * K / Q + K * K) / a0;
    pre->a1 = 2.0 * (K * K - 1.0) / a0;
    pre->a2 = (1.0 - K / Q + K * K) / a0;
    f0 = 38.13547087602444; Q = 0.5003270373238773;
    K = std::tan(M_PI * f0 / (double)sr);
    rlb->b0 = 1.0; rlb->b1 = -2.0; rlb->b2 = 1.0;
    rlb->a1 = 2.0 * (K * K - 1.0) / (1.0 + K / Q + K * K);
    rlb->a2 = (1.0 - K / Q + K * K) / (1.0 + K / Q + K * K);
}

// ---------------------------------------------------------------- swresample phase bank (resample.c build_filter)
static double bessel_i0(double x)
{
    double v = 1, last = 0, t = 1;
    x = x * x / 4;
    for (int i = 1; v != last; i++) { last = v; t *= x / ((double)i * i); v += t; }
    return v;
}

void jt_swr_plan(SwrPlanHost *p, int in_rate, int out_rate)
{
    const int filter_size = 32; const double cutoff = 0.97, beta = 9.0;
    const double factor = std::min(out_rate * cutoff / in_rate, 1.0);
    int phase_count = 1 << 10;
    int L = std::max((int)std::ceil(filter_size / factor), 1);
    if (L > 1) L = (L + 1) & ~1;
    int64_t a = out_rate, b = in_rate;
    while (b) { int64_t t = a % b; a = b; b = t; }
    const int64_t g = a;
    const int64_t pc_exact = out_rate / g;
    // exact_rational (swresample's default): when the exact phase count fits the 1024-entry bank, every output sits on a bank phase.
    // Otherwise (22050 / 11025 -> 192000 need 1280 / 2560 phases; odd rates more) resample.c keeps 1024 phases and steps
    //   index += dst_incr_div; frac += dst_incr_mod; carry at src_incr        (swri_resample, linear = 0)
    // so output m uses bank row floor(m * in_rate * 1024 / out_rate) mod 1024 -- the row is NOT interpolated.  That sequence is
    // periodic in pc_exact outputs, so it is restated here as an exact polyphase plan of pc_exact phases whose row for exact
    // phase q is the 1024-grid row floor(q * 1024 / pc_exact): the kernels keep their integer phase stepping.
    const bool exact = pc_exact <= phase_count;
    if (exact) phase_count = (int)pc_exact;
    else {
        JT_REQUIRE(pc_exact <= 65536, JT_E_UNSUPPORTED, "resampler: the rate pair needs more than 65536 exact phases");
        phase_count = (int)pc_exact;
    }
    p->phase_count = phase_count; p->filter_length = L; p->center = (L - 1) / 2;
    // per output sample the phase index advances by in_rate*phase_count/out_rate
    p->step = exact ? (int64_t)(in_rate / g) * (phase_count / (out_rate / g)) : (int64_t)(in_rate / g);
    p->bank.assign((size_t)phase_count * L, 0.0);
    for (int ph = 0; ph < phase_count; ++ph) {
        double norm = 0.0;
        double *tab = &p->bank[(size_t)ph * L];
        const double frac = exact ? (double)ph / phase_count : (double)(((int64_t)ph * 1024) / pc_exact) / 1024.0;
        for (int i = 0; i < L; ++i) {
            const double x = M_PI * ((double)(i - p->center) - frac) * factor;
            double y = (x == 0) ? 1.0 : std::sin(x) / x;
            const double w = 2.0 * x / (factor * L * M_PI);
            y *= bessel_i0(beta * std::sqrt(std::max(1 - w * w, 0.0)));
            tab[i] = y; norm += y;
        }
        for (int i = 0; i < L; ++i) tab[i] /= norm;
    }
}

// ---------------------------------------------------------------- small fork-join helper for the O(blocks) host finishing
// fn(lo, hi, part) over [0, n) in contiguous parts; one part when n is small.  The per-block work (2-3 log10 each) is the
// only host arithmetic that scales with the file, so long files spread it over a few cores.  The workers are a process-wide
// pool created on first use (a pass calls this several times; spawning threads each time cost more than the work itself on
// 10-minute files).  Callers may be concurrent (one host thread per file in flight): tasks of different calls share the queue
// and every call waits on its own counter.
namespace {
struct JtPool {
    std::mutex mu; std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::vector<std::thread> workers;
    bool stop = false;
    explicit JtPool(unsigned n)
    {
        for (unsigned i = 0; i < n; ++i)
            workers.emplace_back([this] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [this] { return stop || !q.empty(); });
                        if (stop && q.empty()) return;
                        job = std::move(q.front()); q.pop_front();
                    }
                    job();
                }
            });
    }
    ~JtPool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : workers) t.join();
    }
};
JtPool &jt_pool()
{
    static JtPool pool(std::min(7u, std::max(1u, std::thread::hardware_concurrency()) - 1u));
    return pool;
}
} // namespace

void jt_parallel_for(int64_t n, const std::function<void(int64_t, int64_t, int)> &fn, int *nparts_out)
{
    int nt = n >= 8192 ? (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (nparts_out) *nparts_out = nt;
    if (nt <= 1) { fn(0, n, 0); return; }
    JtPool &pool = jt_pool();
    const int64_t per = (n + nt - 1) / nt;
    struct Wait { std::mutex mu; std::condition_variable cv; int left; std::exception_ptr err; } w;
    w.left = nt - 1;
    {
        std::lock_guard<std::mutex> lk(pool.mu);
        for (int k = 1; k < nt; ++k)
            pool.q.emplace_back([&, k] {
                try { fn(std::min(n, k * per), std::min(n, (k + 1) * per), k); }
                catch (...) { std::lock_guard<std::mutex> g(w.mu); if (!w.err) w.err = std::current_exception(); }
                std::lock_guard<std::mutex> g(w.mu);
                if (--w.left == 0) w.cv.notify_one();
            });
    }
    pool.cv.notify_all();
    std::exception_ptr mine;
    try { fn(0, std::min(n, per), 0); } catch (...) { mine = std::current_exception(); }
    { std::unique_lock<std::mutex> lk(w.mu); w.cv.wait(lk, [&] { return w.left == 0; }); }
    if (mine) std::rethrow_exception(mine);
    if (w.err) std::rethrow_exception(w.err);
}
constexpr int JT_MAX_PARTS = 8;

// ---------------------------------------------------------------- f_ebur128.c gating / LRA on block energies
namespace {
constexpr int ABS_THRES = -70, ABS_UP_THRES = 10, HIST_GRAIN = 100;
constexpr int HIST_SIZE = (ABS_UP_THRES - ABS_THRES) * HIST_GRAIN + 1;
inline double LOUDNESS(double e) { return -0.691 + 10 * std::log10(e); }
inline double ENERGY(double l) { return std::pow(10., (l + 0.691) / 10.); }
inline int HIST_POS(double l) { return (int)((l - ABS_THRES) * HIST_GRAIN); }
inline int clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
}

void jt_r128_finish(const double *bs, int64_t nblocks, int blk, int sr, bool dualmono, R128Series *o, bool integrated_only)
{
    (void)blk;
    const double pan_law = -3.01029995663978;
    const int bins400 = sr * 4 / 10, bins3000 = sr * 3;
    // scratch that keeps its storage from call to call, per host thread.  The names used below are LOCAL references to the calling
    // thread's instances: a thread_local named inside a jt_parallel_for body would be the pool worker's own (empty) object.
    static thread_local std::vector<unsigned> tl_h400, tl_h3000;
    static thread_local std::vector<double> tl_pw400, tl_pw3000;
    std::vector<unsigned> &h400 = tl_h400, &h3000 = tl_h3000;
    std::vector<double> &pw400 = tl_pw400, &pw3000 = tl_pw3000;
    h400.assign(HIST_SIZE, 0); h3000.assign(HIST_SIZE, 0);
    double kept400 = 0, kept3000 = 0; int64_t nk400 = 0, nk3000 = 0;
    double rel400 = 0;
    // integrated_only (the limiter-plan hook of Pass 2, which waits for nothing else): the 3 s windows, their logarithms and the
    // loudness range are left out; `integrated` and `rel_threshold` come out of the same statements as in the full run
    o->M.assign(nblocks, 0.0); o->S.assign(integrated_only ? 0 : nblocks, 0.0);
    // per block: window powers and momentary / short-term loudness (the log10s) in parallel; the powers and the un-panned
    // loudness values are kept for the serial gating walk
    pw400.resize((size_t)nblocks); pw3000.resize((size_t)(integrated_only ? 0 : nblocks));
    jt_parallel_for(nblocks, [&](int64_t lo, int64_t hi, int) {
        for (int64_t k = lo; k < hi; ++k) {
            // window sums recomputed from the block energies each step (no add/sub drift)
            double p400 = 1e-12, p3000 = 1e-12;
            if (k >= 3)  { const double w400 = bs[k] + bs[k - 1] + bs[k - 2] + bs[k - 3]; p400 += w400; p400 /= bins400; }
            pw400[(size_t)k] = p400;
            o->M[k] = LOUDNESS(p400);
            if (integrated_only) continue;
            if (k >= 29) { double w = 0; for (int q = 29; q >= 0; --q) w += bs[k - q]; p3000 += w; p3000 /= bins3000; }
            pw3000[(size_t)k] = p3000;
            o->S[k] = LOUDNESS(p3000);
        }
    });
    for (int64_t k = 0; k < nblocks; ++k) {
        const double l400 = o->M[k];
        if (l400 >= ABS_THRES) {
            h400[clipi(HIST_POS(l400), 0, HIST_SIZE - 1)]++;
            kept400 += pw400[(size_t)k]; nk400++;
        }
        if (integrated_only) continue;
        const double l3000 = o->S[k];
        if (l3000 >= ABS_THRES) {
            h3000[clipi(HIST_POS(l3000), 0, HIST_SIZE - 1)]++;
            kept3000 += pw3000[(size_t)k]; nk3000++;
        }
        if (dualmono) { o->M[k] = l400 - pan_law; o->S[k] = l3000 - pan_law; }
    }
    // relative threshold as of the last gated block (f_ebur128.c recomputes it per gated block; only the final value is read)
    if (nk400) { double rt = kept400 / nk400; if (!rt) rt = 1e-12; rel400 = LOUDNESS(rt) + (-10); }
    // integrated loudness = value after the last gated block
    o->integrated = ABS_THRES; o->lra = 0; o->lra_low = 0; o->lra_high = 0; o->rel_threshold = rel400;
    if (nk400) {
        int gate = clipi(HIST_POS(rel400), 0, HIST_SIZE - 1);
        double isum = 0; uint64_t nint = 0;
        // (an empty bin adds 0 * e = +0.0 to a non-negative sum, i.e. nothing: only occupied bins pay for their pow() -- a few hundred of
        //  the 7501; the walk sits on the limiter plan's critical path inside Pass 2)
        for (int i = gate; i < HIST_SIZE; ++i) if (h400[i]) { nint += h400[i]; isum += h400[i] * ENERGY(i / (double)HIST_GRAIN + ABS_THRES); }
        if (nint) { o->integrated = LOUDNESS(isum / nint); if (dualmono) o->integrated -= pan_law; }
    }
    if (nk3000) {
        double rt = kept3000 / nk3000; if (!rt) rt = 1e-12;
        int gate = clipi(HIST_POS(LOUDNESS(rt) + (-20)), 0, HIST_SIZE - 1);
        uint64_t np = 0;
        for (int i = gate; i < HIST_SIZE; ++i) np += h3000[i];
        if (np) {
            uint64_t nn = 0, npow = (uint64_t)(10 * np * 0.01 + 0.5);
            for (int i = gate; i < HIST_SIZE; ++i) { nn += h3000[i]; if (nn >= npow) { o->lra_low = i / (double)HIST_GRAIN + ABS_THRES; break; } }
            nn = np; npow = (uint64_t)(95 * np * 0.01 + 0.5);
            for (int i = HIST_SIZE - 1; i >= 0; --i) { nn -= std::min<uint64_t>(nn, h3000[i]); if (nn < npow) { o->lra_high = i / (double)HIST_GRAIN + ABS_THRES; break; } }
            o->lra = o->lra_high - o->lra_low;
        }
    }
}

// ---------------------------------------------------------------- libavfilter/ebur128.c (af_loudnorm.c statistics)
namespace {
double hist_e[1000], hist_b[1001]; std::once_flag hist_once;
// find_histogram_index of libavfilter/ebur128.c: the bisection over the bin boundaries
size_t hist_index_bisect(double e)
{
    size_t lo = 0, hi = 1000, mid;
    do { mid = (lo + hi) / 2; if (e >= hist_b[mid]) lo = mid; else hi = mid; } while (hi - lo != 1);
    return lo;
}
// The same bin without the ten unpredictable branches (round 6: binning an hour's 36 000 gating blocks was 0.19 ms of host time between
// Pass 3's last kernel and Pass 4's first, on eight threads).  The bins are 0.1 dB = 2.3 % of energy wide; the top 20 bits of a double
// (sign, exponent, eight mantissa bits) cut the axis into steps of 0.39 %, so the bin of a step's LOWER edge (tabulated with the bisection
// above) is the answer or one below it: one table read, one or two comparisons against the boundaries themselves.  The answer satisfies
// hist_b[j] <= e < hist_b[j + 1] (j = 999: no upper bound), which is what the bisection returns for every e >= hist_b[0].
constexpr int HK_LO = (1023 - 24) << 8, HK_HI = (1023 + 11) << 8;     // keys of 2^-24 (below hist_b[0] = 1.17e-7) .. 2^11 (above hist_b[1000] = 1.17e3)
unsigned short hist_key[HK_HI - HK_LO + 1];
void hist_init()        // (a pool's workers arrive here together)
{
    std::call_once(hist_once, [] {
        hist_b[0] = std::pow(10.0, (-70.0 + 0.691) / 10.0);
        for (int i = 0; i < 1000; ++i) hist_e[i] = std::pow(10.0, ((double)i / 10.0 - 69.95 + 0.691) / 10.0);
        for (int i = 1; i < 1001; ++i) hist_b[i] = std::pow(10.0, ((double)i / 10.0 - 70.0 + 0.691) / 10.0);
        for (int k = HK_LO; k <= HK_HI; ++k) {
            const uint64_t u = (uint64_t)k << 44; double e; std::memcpy(&e, &u, sizeof e);
            hist_key[k - HK_LO] = (unsigned short)(e >= hist_b[0] ? hist_index_bisect(e) : 0);
        }
    });
}
size_t hist_index(double e)          // e >= hist_b[0], as at every call site
{
    uint64_t u; std::memcpy(&u, &e, sizeof u);
    const int64_t k = (int64_t)(u >> 44);
    if (k < HK_LO || k > HK_HI) return hist_index_bisect(e);          // (above 2^12, NaN, negative: not a block energy; the reference's answer anyway)
    size_t j = hist_key[k - HK_LO];
    while (j < 999 && e >= hist_b[j + 1]) ++j;
    return j;
}
double e2l(double e) { return 10 * (std::log(e) / std::log(10.0)) - 0.691; }
}

// test entry (include/jt_host.h): energies whose bin by the table differs from the bisection's
extern "C" int64_t jt_host_hist_index_check(const double *e, int64_t n)
{
    hist_init();
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) if (e[i] >= hist_b[0] && hist_index(e[i]) != hist_index_bisect(e[i])) ++bad;
    // and every boundary itself with its two neighbours
    for (int j = 0; j <= 1000; ++j)
        for (double x : {std::nextafter(hist_b[j], 0.0), hist_b[j], std::nextafter(hist_b[j], 1e300)})
            if (x >= hist_b[0] && hist_index(x) != hist_index_bisect(x)) ++bad;
    return bad;
}

void jt_loudnorm_finish(const double *bs, int64_t nblocks, int64_t s100, bool dual_mono, double scale_energy,
                        double *out_i, double *out_lra, double *out_thresh)
{
    hist_init();
    static thread_local std::vector<unsigned long> tl_bh, tl_sh, tl_pbh, tl_psh;      // (local references: see jt_r128_finish)
    std::vector<unsigned long> &bh = tl_bh, &sh = tl_sh, &pbh = tl_pbh, &psh = tl_psh;
    bh.assign(1000, 0); sh.assign(1000, 0);
    const double ch = dual_mono ? 2.0 : 1.0;
    // gating blocks: 400 ms every 100 ms; short-term blocks: 3 s, first at 3 s then every 1 s.  Histogram counts are
    // order-independent, so the blocks are binned in parallel parts and the counts added.
    pbh.assign((size_t)JT_MAX_PARTS * 1000, 0); psh.assign((size_t)JT_MAX_PARTS * 1000, 0);
    int nparts = 1;
    jt_parallel_for(nblocks, [&](int64_t lo, int64_t hi, int part) {
        unsigned long *b = &pbh[(size_t)part * 1000], *sm = &psh[(size_t)part * 1000];
        for (int64_t k = std::max<int64_t>(lo, 3); k < hi; ++k) {
            double sum = (bs[k - 3] + bs[k - 2] + bs[k - 1] + bs[k]) * scale_energy * ch / (double)(s100 * 4);
            if (sum >= hist_b[0]) ++b[hist_index(sum)];
            if (k >= 29 && (k - 29) % 10 == 0) {
                double st = 0; for (int q = 29; q >= 0; --q) st += bs[k - q];
                st = st * scale_energy * ch / (double)(s100 * 30);
                if (st >= hist_b[0]) ++sm[hist_index(st)];
            }
        }
    }, &nparts);
    for (int pt = 0; pt < nparts; ++pt) for (int j = 0; j < 1000; ++j) { bh[j] += pbh[(size_t)pt * 1000 + j]; sh[j] += psh[(size_t)pt * 1000 + j]; }
    double rel = 0; long above = 0;
    for (int j = 0; j < 1000; ++j) { rel += bh[j] * hist_e[j]; above += bh[j]; }
    if (above) { rel /= (double)above; rel *= 0.1; }
    *out_thresh = above ? e2l(rel) : -70.0;
    if (!above) *out_i = -HUGE_VAL;
    else {
        size_t st;
        if (rel < hist_b[0]) st = 0; else { st = hist_index(rel); if (rel > hist_e[st]) ++st; }
        double g = 0; long cnt = 0;
        for (size_t j = st; j < 1000; ++j) { g += bh[j] * hist_e[j]; cnt += bh[j]; }
        *out_i = cnt ? e2l(g / cnt) : -HUGE_VAL;
    }
    size_t sz = 0; double pw = 0;
    for (int j = 0; j < 1000; ++j) { sz += sh[j]; pw += sh[j] * hist_e[j]; }
    *out_lra = 0.0;
    if (sz) {
        pw /= sz;
        double integ = 0.01 * pw; size_t index;
        if (integ < hist_b[0]) index = 0; else { index = hist_index(integ); if (integ > hist_e[index]) ++index; }
        sz = 0; for (size_t j = index; j < 1000; ++j) sz += sh[j];
        if (sz) {
            size_t pl = (size_t)((sz - 1) * 0.1 + 0.5), ph = (size_t)((sz - 1) * 0.95 + 0.5), j = index;
            sz = 0;
            while (sz <= pl) sz += sh[j++];
            double l_en = hist_e[j - 1];
            while (sz <= ph) sz += sh[j++];
            double h_en = hist_e[j - 1];
            *out_lra = e2l(h_en) - e2l(l_en);
        }
    }
}

// Input-meter series for loudnorm's dynamic mode: after INNER frame k (k = 0 .. n_inner-1; frame k completes 100 ms block 30 + k
// when it is a full frame) the short-term loudness (last 3 s), the gated integrated loudness and the relative threshold of everything
// metered so far -- ff_ebur128_loudness_shortterm / _global / relative_threshold of r128_in (libavfilter/ebur128.c, histogram
// gating).  out[3k .. 3k+2].  A trailing partial frame completes no block: its integrated / threshold values repeat the previous
// frame's and its short-term entry is filled in by the caller from a shifted measurement.
void jt_loudnorm_series(const double *bs, int64_t nfull, int64_t s100, bool dual_mono, int64_t n_inner, double *out)
{
    hist_init();
    std::vector<unsigned long> bh(1000, 0);
    unsigned long long occ[16] = {0};
    const double ch = dual_mono ? 2.0 : 1.0;
    double rel_sum = 0; long above = 0;
    auto add_block = [&](int64_t k) {          // the 400 ms gating block that ends with 100 ms block k
        if (k < 3 || k >= nfull) return;
        const double sum = (bs[k - 3] + bs[k - 2] + bs[k - 1] + bs[k]) * ch / (double)(s100 * 4);
        if (sum >= hist_b[0]) { const size_t j = hist_index(sum); ++bh[j]; occ[j >> 6] |= 1ull << (j & 63); rel_sum += hist_e[j]; ++above; }
    };
    // the sums below walk the histogram in bin order; an empty bin adds +0.0 to a sum that is never negative, i.e. nothing, so only the
    // occupied bins are visited (a bitmap of them, ascending): 1000 dependent additions twice per frame were 5 ms of host time for ten
    // minutes -- a seventh of a dynamic-mode file's job once the limiter was out of the way
    auto for_bins_from = [&](size_t from, auto &&f) {
        for (size_t w = from >> 6; w < 16; ++w) {
            unsigned long long m = occ[w];
            if (w == (from >> 6)) m &= ~0ull << (from & 63);
            while (m) { const size_t j = 64 * w + (size_t)__builtin_ctzll(m); m &= m - 1ull; f(j); }
        }
    };
    for (int64_t k = 0; k < 30; ++k) add_block(k);
    double g_prev = -HUGE_VAL, rt_prev = -70.0;
    for (int64_t f = 0; f < n_inner; ++f) {
        const int64_t k = 30 + f;
        if (k < nfull) {
            add_block(k);
            double st = 0; for (int q = 29; q >= 0; --q) st += bs[k - q];
            st = st * ch / (double)(s100 * 30);
            out[3 * f] = st <= 0.0 ? -HUGE_VAL : e2l(st);
            // the histogram sums are rebuilt in bin order, as ff_ebur128_relative_threshold / loudness_global walk them
            double rel = 0; long cnt = 0;
            for_bins_from(0, [&](size_t j) { rel += bh[j] * hist_e[j]; cnt += (long)bh[j]; });
            if (!cnt) { g_prev = -HUGE_VAL; rt_prev = -70.0; }
            else {
                rel = rel / (double)cnt * 0.1;
                rt_prev = e2l(rel);
                size_t stx;
                if (rel < hist_b[0]) stx = 0; else { stx = hist_index(rel); if (rel > hist_e[stx]) ++stx; }
                double g = 0; long c2 = 0;
                for_bins_from(stx, [&](size_t j) { g += bh[j] * hist_e[j]; c2 += (long)bh[j]; });
                g_prev = c2 ? e2l(g / (double)c2) : -HUGE_VAL;
            }
        } else out[3 * f] = -HUGE_VAL;         // (partial frame: the caller overwrites this)
        out[3 * f + 1] = g_prev; out[3 * f + 2] = rt_prev;
    }
    (void)rel_sum; (void)above;
}
// ebur128_init_filter(): the two K-weighting biquads multiplied out into the 4th-order section the library runs
void jt_kweight_coeffs5(int sr, double b[5], double a[5])
{
    BiquadF64 pre, rlb; jt_kweight_design(sr, &pre, &rlb);
    const double pb[3] = {pre.b0, pre.b1, pre.b2}, pa[3] = {1.0, pre.a1, pre.a2}, rb[3] = {rlb.b0, rlb.b1, rlb.b2}, ra[3] = {1.0, rlb.a1, rlb.a2};
    b[0] = pb[0] * rb[0];
    b[1] = pb[0] * rb[1] + pb[1] * rb[0];
    b[2] = pb[0] * rb[2] + pb[1] * rb[1] + pb[2] * rb[0];
    b[3] = pb[1] * rb[2] + pb[2] * rb[1];
    b[4] = pb[2] * rb[2];
    a[0] = pa[0] * ra[0];
    a[1] = pa[0] * ra[1] + pa[1] * ra[0];
    a[2] = pa[0] * ra[2] + pa[1] * ra[1] + pa[2] * ra[0];
    a[3] = pa[1] * ra[2] + pa[2] * ra[1];
    a[4] = pa[2] * ra[2];
}

// ---------------------------------------------------------------- dynamics parameter derivation
void jt_dyn_design(const jt_filter_params *p, int sr, DynParams *d)
{
    std::memset(d, 0, sizeof(*d));
    d->gate_on = p->gate_enabled; d->comp_on = p->comp_enabled;
    d->deess_on = p->deess_enabled && p->deess_i > 0;
    if (d->gate_on) {   // agate_config_input, detection=rms
        double lin_threshold = p->gate_threshold * p->gate_threshold;
        double ks = std::sqrt(p->gate_knee);
        d->g_attack = std::min(1., 1. / (p->gate_attack_ms * sr / 4000.));
        d->g_release = std::min(1., 1. / (p->gate_release_ms * sr / 4000.));
        d->g_lin_knee_stop = lin_threshold * ks;
        double lks = lin_threshold / ks;
        d->g_thres = std::log(lin_threshold);
        d->g_knee_start = std::log(lks);
        d->g_knee_stop = std::log(d->g_lin_knee_stop);
        d->g_ratio = p->gate_ratio; d->g_knee = p->gate_knee; d->g_range = p->gate_range; d->g_makeup = p->gate_makeup;
    }
    if (d->comp_on) {   // compressor_config_output
        d->c_thres = std::log(p->comp_threshold);
        double lks = p->comp_threshold / std::sqrt(p->comp_knee), lke = p->comp_threshold * std::sqrt(p->comp_knee);
        d->c_adj_knee_start = lks * lks;
        d->c_knee_start = std::log(lks); d->c_knee_stop = std::log(lke);
        d->c_ckstop = (d->c_knee_stop - d->c_thres) / p->comp_ratio + d->c_thres;
        d->c_attack = std::min(1., 1. / (p->comp_attack_ms * sr / 4000.));
        d->c_release = std::min(1., 1. / (p->comp_release_ms * sr / 4000.));
        d->c_ratio = p->comp_ratio; d->c_knee = p->comp_knee; d->c_makeup = p->comp_makeup; d->c_mix = p->comp_mix;
    }
    if (d->deess_on) {
        double overallscale = sr < 44100 ? 44100.0 / sr : sr / 44100.0;
        d->d_intensity = std::pow(p->deess_i, 5) * (8192 / overallscale);
        d->d_maxdess = 1.0 / std::pow(10.0, ((p->deess_m - 1.0) * 48.0) / 20);
        d->d_iir = std::pow(p->deess_f, 2) / overallscale;
    }
}

// ---------------------------------------------------------------- afftdn tables (af_afftdn.c config_input / set_parameters)
namespace {
const int kBandCentre[15] = {80, 125, 195, 290, 440, 660, 1000, 1500, 2250, 3350, 5000, 7500, 11200, 16000, 24000};
double bark(double x) { double d = x / 7500.0; return 13.0 * std::atan(7.6E-4 * x) + 3.5 * std::atan(d * d); }
void lu_factor(double *m, int n)
{
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) {
            double d = m[j + i * n] / m[i + i * n];
            m[j + i * n] = d;
            for (int k = i + 1; k < n; k++) m[j + k * n] -= d * m[i + k * n];
        }
}
void lu_solve(const double *m, double *v, int n)
{
    for (int i = 0; i < n - 1; i++) for (int j = i + 1; j < n; j++) v[j] -= m[j + i * n] * v[i];
    v[n - 1] /= m[n * n - 1];
    for (int i = n - 2; i >= 0; i--) {
        double d = v[i];
        for (int j = i + 1; j < n; j++) d -= m[i + j * n] * v[j];
        v[i] = d / m[i + i * n];
    }
}
}

void jt_afftdn_plan(AfftdnPlanHost *pl, int sr, double nr, double nf, const double *band_noise_in)
{
    const double Cc = M_LN10 * 0.1;
    pl->sr = sr; pl->A = sr / 80; pl->W = 3 * pl->A;
    { int v = pl->W, bits = 0; while (v) { ++bits; v >>= 1; } pl->L = 1 << bits; }
    pl->bins = pl->L / 2 + 1;
    const int bins = pl->bins, L = pl->L;
    double bn[15];
    for (int i = 0; i < 15; ++i) bn[i] = band_noise_in ? std::min(std::max((double)(float)band_noise_in[i], -24.), 24.) : 0.0;
    { double mean = 0; for (double v : bn) mean += v; mean /= 15; for (double &v : bn) v -= mean; }
    // polynomial extrapolation of the profile above the last band (process_get_band_noise, band >= 15)
    double ma[25], mb[75];
    for (int j = 0; j < 5; j++) for (int k = 0; k < 5; k++) { ma[j + k * 5] = 0; for (int m = 0; m < 15; m++) ma[j + k * 5] += std::pow(m, j + k); }
    lu_factor(ma, 5);
    { int i = 0; for (int j = 0; j < 5; j++) for (int k = 0; k < 15; k++) mb[i++] = std::pow(k, j); }
    auto band_noise = [&](int band) -> double {
        if (band < 15) return bn[band];
        double vb[5]; int i = 0;
        for (int j = 0; j < 5; j++) { double s = 0; for (int k = 0; k < 15; k++) s += mb[i++] * bn[k]; vb[j] = s; }
        lu_solve(ma, vb, 5);
        double f = (0.5 * sr) / kBandCentre[14];
        f = 15.0 + std::log(f / 1.5) / std::log(1.5);
        double sum = 0, prod = 1; for (int j = 0; j < 5; j++) { sum += prod * vb[j]; prod *= f; }
        return sum;
    };
    pl->bin2band.resize(bins);
    const double sdiv = 1.25;
    for (int i = 0; i < bins; ++i) pl->bin2band[i] = (int)std::lrint(sdiv * bark(((double)i * sr) / L));
    pl->nbands = pl->bin2band[bins - 1] + 1;
    const int nb = pl->nbands;
    // spread function with excitation normalisation
    pl->spread.assign((size_t)nb * nb, 0.0);
    std::vector<double> be(nb, 0.0), pbe(nb, 0.0);
    { double p1 = std::pow(0.1, 2.5 / sdiv), p2 = std::pow(0.1, 1.0 / sdiv); int j = 0;
      for (int m = 0; m < nb; m++) for (int n2 = 0; n2 < nb; n2++) pl->spread[j++] = n2 < m ? std::pow(p2, m - n2) : (n2 > m ? std::pow(p1, n2 - m) : 1.0); }
    for (int m = 0; m < bins; ++m) be[pl->bin2band[m]] += 1.0;
    { int j = 0; for (int m = 0; m < nb; m++) for (int n2 = 0; n2 < nb; n2++) pbe[m] += pl->spread[j++] * be[n2]; }
    { double mn = std::pow(0.1, 2.5), mx = std::pow(0.1, 1.0);
      for (int i = 0; i < nb; i++) {
          be[i] = (i < std::lrint(12.0 * sdiv)) ? std::pow(0.1, 1.45 + 0.1 * i / sdiv) : std::pow(0.1, 2.5 - 0.2 * (i / sdiv - 14.0));
          be[i] = std::min(std::max(be[i], mn), mx);
      }
      int j = 0; for (int i = 0; i < nb; i++) for (int k = 0; k < nb; k++) pl->spread[j++] *= be[i] / pbe[i]; }
    pl->alpha.assign(nb, 0.0); pl->beta.assign(nb, 0.0);
    { int j = 0; const double sar = pl->A / (double)sr;
      for (int i = 0; i < bins; i++) if ((i == L / 2) || (pl->bin2band[i] > j)) {
          double d6 = (i - 1) * (double)sr / L, d7 = std::fmin(0.008 + 2.2 / d6, 0.03);
          pl->alpha[j] = std::exp(-sar / d7); pl->beta[j] = 1.0 - pl->alpha[j]; j = pl->bin2band[i];
      } }
    pl->window.resize(pl->W);
    double sum = 0; const double wscale = std::sqrt(8.0 / (9.0 * L));
    for (int i = 0; i < pl->W; i++) { double d = std::sin(i * M_PI / pl->W); d *= wscale * d; pl->window[i] = d; sum += d * d; }
    const double floor = (double)(1LL << 48) * std::exp(-23.025558369790467) * (0.5 * sum);
    const double max_var = floor * std::exp((100.0 + nf) * Cc);
    pl->max_gain = std::exp(nr * (0.5 * Cc));
    const double gain_scale = 1.0 / (pl->max_gain * pl->max_gain);
    // set_band_parameters(): log-linear interpolation of the band profile across bins
    std::vector<double> rel(bins, 1.0);
    { double d2 = 1, d5 = 0, bnz = band_noise(0); int i = 0, j = 0, k = 0;
      for (int m = j; m < bins; m++) {
          if (m == j) { i = j; d5 = bnz; j = (k >= 15) ? bins : (int)((double)L * kBandCentre[k] / sr); d2 = j - i; bnz = band_noise(k); k++; }
          double d3 = (j - m) / d2, d4 = (m - i) / d2;
          rel[m] = std::exp((d5 * d3 + bnz * d4) * Cc);
      } }
    pl->abs_var.resize(bins); pl->min_abs_var.resize(bins);
    for (int i = 0; i < bins; i++) { pl->abs_var[i] = std::fmax(max_var * rel[i], 1.0); pl->min_abs_var[i] = gain_scale * pl->abs_var[i]; }
    pl->rel_var.assign(rel.begin(), rel.begin() + bins); pl->floor = floor; pl->noise_floor = nf;      // (noise tracking, tn=1)
}

// ---------------------------------------------------------------- MD5 (RFC 1321) for the FLAC STREAMINFO signature
// One dependent chain per stream by construction (Merkle-Damgard), so this stays on a host core next to the GPU encode.
namespace {
#define JT_MD5_STEP(f, a, b, c, d, x, t, s) do { (a) += f((b), (c), (d)) + (x) + (t); (a) = ((a) << (s)) | ((a) >> (32 - (s))); (a) += (b); } while (0)
#define JT_F1(x, y, z) ((z) ^ ((x) & ((y) ^ (z))))
#define JT_F2(x, y, z) JT_F1(z, x, y)
// round 2 with G(b, c, d) = (b & d) | (c & ~d) taken apart: the two terms have no bit in common, so they may be ADDED, and (c & ~d) does
// not involve b -- the value the previous step has just produced -- so it joins the sum off the chain: b -> and -> add -> rotate -> add
// instead of b -> xor -> and -> xor -> add -> rotate -> add (the chain is all a single stream's MD5 is: +7 % measured)
#define JT_MD5_STEP2(a, b, c, d, x, t, s) do { (a) += ((c) & ~(d)) + (x) + (t); (a) += ((b) & (d)); (a) = ((a) << (s)) | ((a) >> (32 - (s))); (a) += (b); } while (0)
#define JT_F3(x, y, z) ((x) ^ ((y) ^ (z)))
#define JT_F4(x, y, z) ((y) ^ ((x) | ~(z)))
inline void md5_blocks(uint32_t st[4], const unsigned char *p, size_t nblocks)
{
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3];
    for (; nblocks; nblocks--, p += 64) {
        uint32_t w[16];
        memcpy(w, p, 64);                                // little-endian host (x86-64)
        const uint32_t sa = a, sb = b, sc = c, sd = d;
        JT_MD5_STEP(JT_F1, a, b, c, d, w[0], 0xd76aa478, 7);  JT_MD5_STEP(JT_F1, d, a, b, c, w[1], 0xe8c7b756, 12);
        JT_MD5_STEP(JT_F1, c, d, a, b, w[2], 0x242070db, 17); JT_MD5_STEP(JT_F1, b, c, d, a, w[3], 0xc1bdceee, 22);
        JT_MD5_STEP(JT_F1, a, b, c, d, w[4], 0xf57c0faf, 7);  JT_MD5_STEP(JT_F1, d, a, b, c, w[5], 0x4787c62a, 12);
        JT_MD5_STEP(JT_F1, c, d, a, b, w[6], 0xa8304613, 17); JT_MD5_STEP(JT_F1, b, c, d, a, w[7], 0xfd469501, 22);
        JT_MD5_STEP(JT_F1, a, b, c, d, w[8], 0x698098d8, 7);  JT_MD5_STEP(JT_F1, d, a, b, c, w[9], 0x8b44f7af, 12);
        JT_MD5_STEP(JT_F1, c, d, a, b, w[10], 0xffff5bb1, 17); JT_MD5_STEP(JT_F1, b, c, d, a, w[11], 0x895cd7be, 22);
        JT_MD5_STEP(JT_F1, a, b, c, d, w[12], 0x6b901122, 7); JT_MD5_STEP(JT_F1, d, a, b, c, w[13], 0xfd987193, 12);
        JT_MD5_STEP(JT_F1, c, d, a, b, w[14], 0xa679438e, 17); JT_MD5_STEP(JT_F1, b, c, d, a, w[15], 0x49b40821, 22);
        JT_MD5_STEP2(a, b, c, d, w[1], 0xf61e2562, 5);  JT_MD5_STEP2(d, a, b, c, w[6], 0xc040b340, 9);
        JT_MD5_STEP2(c, d, a, b, w[11], 0x265e5a51, 14); JT_MD5_STEP2(b, c, d, a, w[0], 0xe9b6c7aa, 20);
        JT_MD5_STEP2(a, b, c, d, w[5], 0xd62f105d, 5);  JT_MD5_STEP2(d, a, b, c, w[10], 0x02441453, 9);
        JT_MD5_STEP2(c, d, a, b, w[15], 0xd8a1e681, 14); JT_MD5_STEP2(b, c, d, a, w[4], 0xe7d3fbc8, 20);
        JT_MD5_STEP2(a, b, c, d, w[9], 0x21e1cde6, 5);  JT_MD5_STEP2(d, a, b, c, w[14], 0xc33707d6, 9);
        JT_MD5_STEP2(c, d, a, b, w[3], 0xf4d50d87, 14); JT_MD5_STEP2(b, c, d, a, w[8], 0x455a14ed, 20);
        JT_MD5_STEP2(a, b, c, d, w[13], 0xa9e3e905, 5); JT_MD5_STEP2(d, a, b, c, w[2], 0xfcefa3f8, 9);
        JT_MD5_STEP2(c, d, a, b, w[7], 0x676f02d9, 14); JT_MD5_STEP2(b, c, d, a, w[12], 0x8d2a4c8a, 20);
        JT_MD5_STEP(JT_F3, a, b, c, d, w[5], 0xfffa3942, 4);  JT_MD5_STEP(JT_F3, d, a, b, c, w[8], 0x8771f681, 11);
        JT_MD5_STEP(JT_F3, c, d, a, b, w[11], 0x6d9d6122, 16); JT_MD5_STEP(JT_F3, b, c, d, a, w[14], 0xfde5380c, 23);
        JT_MD5_STEP(JT_F3, a, b, c, d, w[1], 0xa4beea44, 4);  JT_MD5_STEP(JT_F3, d, a, b, c, w[4], 0x4bdecfa9, 11);
        JT_MD5_STEP(JT_F3, c, d, a, b, w[7], 0xf6bb4b60, 16); JT_MD5_STEP(JT_F3, b, c, d, a, w[10], 0xbebfbc70, 23);
        JT_MD5_STEP(JT_F3, a, b, c, d, w[13], 0x289b7ec6, 4); JT_MD5_STEP(JT_F3, d, a, b, c, w[0], 0xeaa127fa, 11);
        JT_MD5_STEP(JT_F3, c, d, a, b, w[3], 0xd4ef3085, 16); JT_MD5_STEP(JT_F3, b, c, d, a, w[6], 0x04881d05, 23);
        JT_MD5_STEP(JT_F3, a, b, c, d, w[9], 0xd9d4d039, 4);  JT_MD5_STEP(JT_F3, d, a, b, c, w[12], 0xe6db99e5, 11);
        JT_MD5_STEP(JT_F3, c, d, a, b, w[15], 0x1fa27cf8, 16); JT_MD5_STEP(JT_F3, b, c, d, a, w[2], 0xc4ac5665, 23);
        JT_MD5_STEP(JT_F4, a, b, c, d, w[0], 0xf4292244, 6);  JT_MD5_STEP(JT_F4, d, a, b, c, w[7], 0x432aff97, 10);
        JT_MD5_STEP(JT_F4, c, d, a, b, w[14], 0xab9423a7, 15); JT_MD5_STEP(JT_F4, b, c, d, a, w[5], 0xfc93a039, 21);
        JT_MD5_STEP(JT_F4, a, b, c, d, w[12], 0x655b59c3, 6); JT_MD5_STEP(JT_F4, d, a, b, c, w[3], 0x8f0ccc92, 10);
        JT_MD5_STEP(JT_F4, c, d, a, b, w[10], 0xffeff47d, 15); JT_MD5_STEP(JT_F4, b, c, d, a, w[1], 0x85845dd1, 21);
        JT_MD5_STEP(JT_F4, a, b, c, d, w[8], 0x6fa87e4f, 6);  JT_MD5_STEP(JT_F4, d, a, b, c, w[15], 0xfe2ce6e0, 10);
        JT_MD5_STEP(JT_F4, c, d, a, b, w[6], 0xa3014314, 15); JT_MD5_STEP(JT_F4, b, c, d, a, w[13], 0x4e0811a1, 21);
        JT_MD5_STEP(JT_F4, a, b, c, d, w[4], 0xf7537e82, 6);  JT_MD5_STEP(JT_F4, d, a, b, c, w[11], 0xbd3af235, 10);
        JT_MD5_STEP(JT_F4, c, d, a, b, w[2], 0x2ad7d2bb, 15); JT_MD5_STEP(JT_F4, b, c, d, a, w[9], 0xeb86d391, 21);
        a += sa; b += sb; c += sc; d += sd;
    }
    st[0] = a; st[1] = b; st[2] = c; st[3] = d;
}
} // namespace

void jt_md5(const void *data, size_t len, uint8_t out[16])
{
    uint32_t st[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
    const unsigned char *p = static_cast<const unsigned char *>(data);
    const size_t full = len / 64;
    md5_blocks(st, p, full);
    unsigned char tail[128] = {0};
    const size_t rem = len - full * 64;
    memcpy(tail, p + full * 64, rem);
    tail[rem] = 0x80;
    const size_t tl = rem < 56 ? 64 : 128;
    const uint64_t bits = (uint64_t)len * 8;
    memcpy(tail + tl - 8, &bits, 8);
    md5_blocks(st, tail, tl / 64);
    memcpy(out, st, 16);
}
