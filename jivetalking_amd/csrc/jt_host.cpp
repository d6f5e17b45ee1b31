// jt_host.cpp — C++ restatement of the reference's per-file control logic (Go, internal/processor) for the
// four-pass path: interval construction, noise-floor seed, voice-activity detection, speech election, adaptive
// tuning, filter-spec strings, limiter/loudnorm planning and the ProcessAudio orchestration over the GPU engine.
// Scalar double arithmetic in the same order as the Go code so results are bit-identical; durations are Go
// time.Duration nanoseconds (int64).  See include/jt_host.h for the file:line map.
#include "jt_internal.h"
#include <unistd.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <cerrno>
#include "../../include/jt_host.h"
#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <functional>
#include <chrono>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <sched.h>
#include <pthread.h>
#include <cctype>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
namespace {

constexpr int64_t kSecond = 1000000000LL, kMs = 1000000LL;
constexpr int64_t kHop = 250 * kMs;                 // analysisIntervalHop (analyser_vad.go:16)

std::string sfmt(const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    return std::string(buf);
}
double qfmt_slow(const char *fmt, double v)
{
    if (!std::isfinite(v)) return v;
    char buf[64]; snprintf(buf, sizeof(buf), fmt, v); return std::strtod(buf, nullptr);
}
// Value a C "%.Nf" / "%g" print-and-reparse would yield, without going through printf on the hot path (hundreds of
// thousands of metadata values per file).  Decimal scaling + round-to-nearest-even; falls back to printf when the scaled
// value sits within a few ulp of a rounding tie, so the result always equals the printf round trip.
double qdec(double v, int decimals)
{
    if (!std::isfinite(v)) return v;
    static const double p10[] = {1, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9};
    const double sc = p10[decimals];
    const double t = v * sc;
    if (std::fabs(t) > 4e15) return v;
    const double r = std::nearbyint(t);
    const double frac = std::fabs(t - r);
    if (std::fabs(frac - 0.5) < 1e-6) { char f[8]; snprintf(f, sizeof(f), "%%.%df", decimals); return qfmt_slow(f, v); }
    return r / sc;
}
double qsig6(double v)      // "%g": 6 significant digits
{
    if (!std::isfinite(v) || v == 0.0) return v;
    // decimal exponent from the binary one (no log10/pow on the hot path); a mis-estimate at a power-of-ten boundary leaves
    // r outside [1e5, 1e6) and takes the printf path, so the result never depends on the estimate being exact
    static const double p10[] = {1, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18};
    static const double n10[] = {1, 1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9, 1e-10, 1e-11, 1e-12, 1e-13};
    const double av = std::fabs(v);
    int be; std::frexp(av, &be);                       // av in [2^(be-1), 2^be)
    int e = (int)std::floor((be - 1) * 0.30102999566398120);
    if (e >= -13 && e < 6) {
        const double up = e + 1 >= 0 ? p10[e + 1] : n10[-(e + 1)];
        if (av >= up) ++e;
    } else return qfmt_slow("%g", v);
    const int dec = 5 - e;
    if (dec < 0 || dec > 18) return qfmt_slow("%g", v);
    const double sc = p10[dec];
    const double t = v * sc;
    const double r = std::nearbyint(t);
    if (std::fabs(std::fabs(t - r) - 0.5) < 1e-6 || std::fabs(r) >= 1e6 || std::fabs(r) < 1e5) return qfmt_slow("%g", v);
    return r / sc;
}
double qfmt(const char *fmt, double v)
{
    if (!std::strcmp(fmt, "%.3f")) return qdec(v, 3);
    if (!std::strcmp(fmt, "%f")) return qdec(v, 6);
    if (!std::strcmp(fmt, "%.2f")) return qdec(v, 2);
    if (!std::strcmp(fmt, "%g")) return qsig6(v);
    return qfmt_slow(fmt, v);
}
// Go's %g / strconv 'g' with shortest round-trip precision
std::string go_g(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "+Inf" : "-Inf";
    char buf[64];
    for (int p = 1; p <= 17; ++p) {
        snprintf(buf, sizeof(buf), "%.*e", p - 1, v);
        if (std::strtod(buf, nullptr) == v) {
            int exp10 = 0; const char *e = std::strchr(buf, 'e'); if (e) exp10 = std::atoi(e + 1);
            if (exp10 < -4 || exp10 >= 21) {
                // Go prints e-notation with at least two exponent digits
                std::string m(buf, e - buf);
                return m + sfmt("e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
            }
            int decimals = std::max(0, p - 1 - exp10);
            snprintf(buf, sizeof(buf), "%.*f", decimals, v);
            return std::string(buf);
        }
    }
    snprintf(buf, sizeof(buf), "%.17g", v); return std::string(buf);
}

double DbToLinear(double db) { return std::pow(10.0, db / 20.0); }
double LinearToDb(double lin) { return lin <= 0 ? -120.0 : 20.0 * std::log10(lin); }
double linearRatioToDB(double r) { return r <= 0 ? -120.0 : 20 * std::log10(r); }
double linearSampleToDBFS(double s)
{
    double a = std::fabs(s);
    if (a <= 0) return -120.0;
    if (a > 1.0) a /= 32768.0;
    if (a > 1.0) a = 1.0;
    return 20 * std::log10(a);
}
bool isFiniteD(double v) { return !std::isnan(v) && !std::isinf(v); }
double sanitizeFloat(double v, double d) { return isFiniteD(v) ? v : d; }

// ------------------------------------------------------------------ spectral helpers
void spec_add(jt_spectral &a, const jt_spectral &b) { double *pa = &a.mean; const double *pb = &b.mean; for (int i = 0; i < 13; ++i) pa[i] += pb[i]; }
jt_spectral spec_avg(const jt_spectral &a, double n) { jt_spectral o; double *po = &o.mean; const double *pa = &a.mean; for (int i = 0; i < 13; ++i) po[i] = pa[i] / n; return o; }
jt_spectral spec_zero() { jt_spectral z; std::memset(&z, 0, sizeof(z)); return z; }
jt_spectral spec_quant(const jt_spectral &a) { jt_spectral o; double *po = &o.mean; const double *pa = &a.mean; for (int i = 0; i < 13; ++i) po[i] = qfmt("%g", pa[i]); return o; }

// ------------------------------------------------------------------ VAD (analyser_vad.go)
constexpr double vadLevelFloorDB = -115.0;
constexpr double speechCentroidMin = 200.0, speechCentroidMax = 6000.0, speechMinimumNoiseMarginDB = 2.0, speechEntropyMax = 0.70;
bool isFlooredLevel(double l) { return std::isinf(l) || std::isnan(l) || l <= vadLevelFloorDB; }
int intervalsForDuration(int64_t d, int64_t hop) { return hop <= 0 ? 0 : (int)((d + hop / 2) / hop); }
double level_of(const jt_interval &s) { return s.momentary_lufs; }     // axisMomentaryLUFS

struct Hist { std::vector<int> bins; double binWidth = 0, minLevel = 0, maxLevel = 0; int count = 0;
              double centre(int i) const { return minLevel + ((double)i + 0.5) * binWidth; } };

Hist buildLevelHistogram(const std::vector<jt_interval> &iv, double bw)
{
    Hist h; if (bw <= 0) return h;
    std::vector<double> levels; double mn = INFINITY, mx = -INFINITY;
    for (auto &s : iv) { double l = level_of(s); if (isFlooredLevel(l)) continue; levels.push_back(l); mn = std::min(mn, l); mx = std::max(mx, l); }
    if (levels.empty()) return h;
    int bc = (int)((mx - mn) / bw) + 1;
    h.bins.assign(bc, 0); h.binWidth = bw; h.minLevel = mn; h.maxLevel = mx;
    for (double l : levels) { int idx = (int)((l - mn) / bw); if (idx >= bc) idx = bc - 1; h.bins[idx]++; h.count++; }
    return h;
}
std::vector<double> vadLevels(const std::vector<jt_interval> &iv)
{
    std::vector<double> l; for (auto &s : iv) { double v = level_of(s); if (!isFlooredLevel(v)) l.push_back(v); }
    std::sort(l.begin(), l.end()); return l;
}
double percentileOfSorted(const std::vector<double> &s, double pct)
{
    if (s.empty()) return 0;
    pct = std::max(0.0, std::min(100.0, pct));
    int idx = (int)(pct / 100 * (double)(s.size() - 1));
    return s[idx];
}
double otsuSplit(const Hist &h)
{
    if (h.count == 0 || h.bins.size() < 2) return (h.minLevel + h.maxLevel) / 2;
    double total = (double)h.count, sumAll = 0;
    for (size_t i = 0; i < h.bins.size(); ++i) sumAll += h.centre((int)i) * (double)h.bins[i];
    double wB = 0, sB = 0, best = 0; int bestIdx = -1;
    for (size_t i = 0; i + 1 < h.bins.size(); ++i) {
        wB += (double)h.bins[i]; sB += h.centre((int)i) * (double)h.bins[i];
        double wF = total - wB;
        if (wB == 0 || wF == 0) continue;
        double mB = sB / wB, mF = (sumAll - sB) / wF, diff = mB - mF, var = wB * wF * diff * diff;
        if (var > best) { best = var; bestIdx = (int)i; }
    }
    if (bestIdx < 0) return (h.minLevel + h.maxLevel) / 2;
    return h.minLevel + (double)(bestIdx + 1) * h.binWidth;
}
double percentileFloor(const std::vector<double> &levels, double seed)
{ return std::max(percentileOfSorted(levels, 10.0), seed + speechMinimumNoiseMarginDB); }
double clampSplit(double split, double nf, double p75)
{ double lower = nf + speechMinimumNoiseMarginDB; if (p75 < lower) return lower; return std::max(lower, std::min(p75, split)); }
bool passesSpectralVeto(const jt_interval &s)
{ return s.spectral.centroid >= speechCentroidMin && s.spectral.centroid <= speechCentroidMax && s.spectral.entropy < speechEntropyMax; }
bool isSpeechInterval(const jt_interval &s, double split) { return level_of(s) >= split && passesSpectralVeto(s); }
double hysteresisMargin(const Hist &h, double split)
{
    double w = 0, c = 0;
    for (size_t i = 0; i < h.bins.size(); ++i) { double ce = h.centre((int)i); if (ce >= split) { w += ce * (double)h.bins[i]; c += (double)h.bins[i]; } }
    double upper = c == 0 ? split : w / c;
    double dist = upper - split;
    return dist <= 0 ? 1.0 : dist * 0.25;
}
int gapToleranceIntervals(const std::vector<bool> &flags, int64_t hop)
{
    int floor_ = intervalsForDuration(2 * kSecond, hop), ceil_ = intervalsForDuration(10 * kSecond, hop);
    int first = -1, last = -1;
    for (size_t i = 0; i < flags.size(); ++i) if (flags[i]) { if (first < 0) first = (int)i; last = (int)i; }
    if (first < 0) return floor_;
    std::vector<double> gaps; int gl = 0;
    for (int i = first; i <= last; ++i) { if (flags[i]) { if (gl > 0) gaps.push_back((double)gl); gl = 0; continue; } gl++; }
    if (gaps.empty()) return floor_;
    std::sort(gaps.begin(), gaps.end());
    int p75 = (int)std::round(percentileOfSorted(gaps, 75));
    return std::max(floor_, std::min(ceil_, p75));
}
std::vector<jt_region> buildSpeechRuns(const std::vector<jt_interval> &iv, double split, double margin, int tol, int64_t hop)
{
    std::vector<jt_region> runs;
    int minIntervals = intervalsForDuration(10 * kSecond, hop);
    if ((int)iv.size() < minIntervals || minIntervals <= 0) return runs;
    double high = split + margin, low = split - margin;
    int64_t runStart = 0; int runSpeech = 0, lastSpeechIdx = 0, pendingGap = 0; bool inRun = false;
    auto flush = [&](int endIdx) {
        if (inRun && runSpeech >= minIntervals) { int64_t e = iv[endIdx].timestamp_ns + hop; runs.push_back(jt_region{runStart, e, e - runStart}); }
        inRun = false; runSpeech = 0; pendingGap = 0;
    };
    for (size_t i = 0; i < iv.size(); ++i) {
        const jt_interval &s = iv[i];
        double level = level_of(s); bool veto = passesSpectralVeto(s); bool isSpeech = level >= split && veto;
        if (!inRun) { if (level >= high && veto) { runStart = s.timestamp_ns; runSpeech = 1; lastSpeechIdx = (int)i; pendingGap = 0; inRun = true; } continue; }
        if (isSpeech) { runSpeech++; lastSpeechIdx = (int)i; pendingGap = 0; continue; }
        if (level >= split && !veto) { flush(lastSpeechIdx); continue; }
        if (level < low) { pendingGap++; if (pendingGap > tol) flush(lastSpeechIdx); }
    }
    flush(lastSpeechIdx);
    return runs;
}

// ------------------------------------------------------------------ candidates (analyser_candidates_shared.go)
constexpr int64_t goldenIntervalSize = 250 * kMs;
std::vector<jt_interval> getIntervalsInRange(const std::vector<jt_interval> &iv, int64_t start, int64_t end)
{
    std::vector<jt_interval> r;
    if (iv.empty()) return r;
    size_t lo = 0, hi = iv.size();
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (iv[mid].timestamp_ns < start) lo = mid + 1; else hi = mid; }
    for (size_t i = lo; i < iv.size(); ++i) { if (iv[i].timestamp_ns >= end) break; r.push_back(iv[i]); }
    return r;
}
struct Acc { double rmsSum = 0, peakMax = -120, tpMax = -120, spMax = -120, mSum = 0, sSum = 0; jt_spectral spec = spec_zero(); };
Acc accumulateIntervalMetrics(const std::vector<jt_interval> &r)
{
    Acc a;
    for (auto &s : r) {
        a.rmsSum += s.rms_level; if (s.peak_level > a.peakMax) a.peakMax = s.peak_level;
        spec_add(a.spec, s.spectral); a.mSum += s.momentary_lufs; a.sSum += s.shortterm_lufs;
        if (s.true_peak > a.tpMax) a.tpMax = s.true_peak;
        if (s.sample_peak > a.spMax) a.spMax = s.sample_peak;
    }
    return a;
}
double calculateRolloffScore(double r)
{
    if (r >= 4000.0 && r <= 8000.0) return 1.0;
    if (r >= 2500.0 && r < 4000.0) return 0.5 + 0.5 * (r - 2500.0) / (4000.0 - 2500.0);
    if (r > 8000.0 && r <= 10000.0) return 0.5 + 0.5 * (10000.0 - r) / (10000.0 - 8000.0);
    return 0.0;
}
double calculateFluxScore(double f)
{
    if (f <= 0.004) return 1.0;
    if (f <= 0.010) return 1.0 - (f - 0.004) / (0.010 - 0.004) * 0.3;
    if (f <= 0.020) return 0.7 - (f - 0.010) / (0.020 - 0.010) * 0.3;
    if (f <= 0.030) return 0.4 - (f - 0.020) / (0.030 - 0.020) * 0.2;
    return 0.2;
}
// Window scorer over structure-of-arrays views (identical arithmetic and summation order to the per-interval form;
// the sliding-window refinement evaluates O(N*W) windows, so the fields are laid out contiguously once).
struct ScoreSoA { std::vector<double> kurt, flat, cent, rms, roll, flux; };
double scoreSpeechWindowSoA(const ScoreSoA &a, size_t off, size_t cnt)
{
    if (!cnt) return 0;
    const double *K = a.kurt.data() + off, *F = a.flat.data() + off, *Cc = a.cent.data() + off, *R = a.rms.data() + off,
                 *Ro = a.roll.data() + off, *Fl = a.flux.data() + off;
    double n = (double)cnt, kS = 0, fS = 0, cS = 0, rS = 0, roS = 0, flS = 0;
    for (size_t i = 0; i < cnt; ++i) { kS += K[i]; fS += F[i]; cS += Cc[i]; rS += R[i]; roS += Ro[i]; flS += Fl[i]; }
    double aK = kS / n, aF = fS / n, aC = cS / n, aR = rS / n, aRo = roS / n, aFl = flS / n;
    double kv = 0; for (size_t i = 0; i < cnt; ++i) { double d = K[i] - aK; kv += d * d; }
    double kVar = kv / n;
    int voiced = 0; for (size_t i = 0; i < cnt; ++i) if (K[i] > 4.5) voiced++;
    double voicingScore = std::max(0.0, std::min(((double)voiced / n) / 0.6, 1.0));
    double kurtosisScore = std::max(0.0, std::min(aK / 7.5, 1.0));
    double flatnessScore = std::max(0.0, std::min(1.0 - aF, 1.0));
    double centroidScore = 0.0;
    if (aC >= speechCentroidMin && aC <= speechCentroidMax) {
        double mid = (speechCentroidMin + speechCentroidMax) / 2, hw = (speechCentroidMax - speechCentroidMin) / 2;
        centroidScore = 1.0 - (std::fabs(aC - mid) / hw) * 0.5;
    }
    double consistencyScore = std::max(0.0, std::min(1.0 - (kVar / 100.0), 1.0));
    double rmsScore = 0.0; if (aR > -30.0) rmsScore = std::max(0.0, std::min((aR - (-30.0)) / 18.0, 1.0));
    return kurtosisScore * 0.15 + flatnessScore * 0.10 + centroidScore * 0.10 + consistencyScore * 0.10 + rmsScore * 0.10 +
           voicingScore * 0.15 + calculateRolloffScore(aRo) * 0.15 + calculateFluxScore(aFl) * 0.15;
}

struct Refined { int64_t start, end, duration; bool ok; };
Refined refineToSubregion(int64_t rs, int64_t re, int64_t rd, const std::vector<jt_interval> &iv, int64_t winDur, int64_t winMin,
                          bool speech_scorer /* false: lowest mean RMS (room tone) */)
{
    Refined out{rs, re, rd, false};
    if (rd <= winDur) return out;
    std::vector<jt_interval> c = getIntervalsInRange(iv, rs, re);
    if (c.empty()) return out;
    int w = (int)(winDur / goldenIntervalSize), mn = (int)(winMin / goldenIntervalSize);
    if ((int)c.size() < mn) return out;
    if ((int)c.size() < w) w = (int)c.size();
    ScoreSoA a;
    const size_t nc = c.size();
    a.rms.resize(nc);
    for (size_t i = 0; i < nc; ++i) a.rms[i] = c[i].rms_level;
    if (speech_scorer) {
        a.kurt.resize(nc); a.flat.resize(nc); a.cent.resize(nc); a.roll.resize(nc); a.flux.resize(nc);
        for (size_t i = 0; i < nc; ++i) { a.kurt[i] = c[i].spectral.kurtosis; a.flat[i] = c[i].spectral.flatness; a.cent[i] = c[i].spectral.centroid;
                                          a.roll[i] = c[i].spectral.rolloff; a.flux[i] = c[i].spectral.flux; }
    }
    auto score = [&](size_t off) -> double {
        if (speech_scorer) return scoreSpeechWindowSoA(a, off, (size_t)w);
        double sum = 0; const double *R = a.rms.data() + off; for (int i = 0; i < w; ++i) sum += R[i]; return sum / (double)w;   // scoreIntervalWindow
    };
    int bestIdx = 0; double best = score(0);
    for (int st = 1; st <= (int)nc - w; ++st) {
        double sc = score((size_t)st);
        bool better = speech_scorer ? (sc > best) : (sc < best);
        if (better) { best = sc; bestIdx = st; }
    }
    int64_t st = c[bestIdx].timestamp_ns, du = (int64_t)w * goldenIntervalSize;
    return Refined{st, st + du, du, true};
}

double levelVariance(const std::vector<jt_interval> &r)
{
    double n = (double)r.size(); if (n == 0) return 0;
    double s = 0; for (auto &x : r) s += level_of(x); double mean = s / n;
    double v = 0; for (auto &x : r) { double d = level_of(x) - mean; v += d * d; }
    return v / n;
}
bool measureSpeechCandidate(const jt_region &reg, const std::vector<jt_interval> &iv, jt_speech_candidate *out)
{
    std::vector<jt_interval> r = getIntervalsInRange(iv, reg.start_ns, reg.end_ns);
    if (r.empty()) return false;
    Acc a = accumulateIntervalMetrics(r);
    double n = (double)r.size(), avgRMS = a.rmsSum / n;
    int voiced = 0; for (auto &x : r) if (x.spectral.kurtosis > 4.5) voiced++;
    std::memset(out, 0, sizeof(*out));
    out->region = reg;
    out->sample.rms_level = avgRMS; out->sample.peak_level = a.peakMax; out->sample.crest_factor = a.peakMax - avgRMS;
    out->sample.spectral = spec_avg(a.spec, n);
    out->sample.momentary_lufs = a.mSum / n; out->sample.shortterm_lufs = a.sSum / n;
    out->sample.true_peak = a.tpMax; out->sample.sample_peak = a.spMax;
    out->voicing_density = (double)voiced / n;
    return true;
}
double groundedSNRScore(double snr)
{
    if (snr <= 0) return 0.0;
    if (snr < 20.0) return 0.5 * (snr / 20.0);
    if (snr >= 40.0) return 1.0;
    return 0.5 + 0.5 * (snr - 20.0) / (40.0 - 20.0);
}
double scoreSpeechCandidateGrounded(const jt_speech_candidate &m, double noiseFloorDB, double levelVar)
{
    double snr = groundedSNRScore(m.sample.rms_level - noiseFloorDB);
    const int64_t adequacy = 30 * kSecond;
    double dur = m.region.duration_ns >= adequacy ? 1.0
               : std::max(0.0, std::min(((double)m.region.duration_ns / 1e9) / ((double)adequacy / 1e9), 1.0));
    double tie = std::max(0.0, std::min(1.0 - (levelVar / 25.0), 1.0)) * 0.02;
    return snr * 0.6 + dur * 0.4 + tie;
}

struct BestResult { bool has = false; jt_region best{}; std::vector<jt_speech_candidate> cands; };
BestResult findBestSpeechRegion(const std::vector<jt_region> &regions, const std::vector<jt_interval> &iv, bool hasNoise, double noiseFloor)
{
    BestResult res;
    if (regions.empty()) return res;
    double nf = hasNoise ? noiseFloor : -INFINITY;
    int bestI = -1; double bestScore = 0; bool hasFb = false; jt_region fb{}; double fbScore = 0;
    for (size_t i = 0; i < regions.size(); ++i) {
        jt_speech_candidate m;
        if (!measureSpeechCandidate(regions[i], iv, &m)) continue;
        std::vector<jt_interval> r = getIntervalsInRange(iv, regions[i].start_ns, regions[i].end_ns);
        double score = scoreSpeechCandidateGrounded(m, nf, levelVariance(r));
        m.score = score;
        res.cands.push_back(m);
        if (!hasFb || score > fbScore) { fb = m.region; fbScore = score; hasFb = true; }
        if (score >= 0.3 && (bestI < 0 || score > bestScore)) { bestI = (int)i; bestScore = score; }
    }
    bool have = false; jt_region best{};
    if (bestI >= 0) { best = regions[bestI]; have = true; } else if (hasFb) { best = fb; have = true; }
    const int64_t gw = 60 * kSecond, gmin = 30 * kSecond;
    if (have && best.duration_ns > gw) {
        jt_region orig = best;
        Refined rf = refineToSubregion(best.start_ns, best.end_ns, best.duration_ns, iv, gw, gmin, true);
        jt_region refined = rf.ok ? jt_region{rf.start, rf.end, rf.duration} : best;
        bool wasRefined = refined.start_ns != orig.start_ns || refined.duration_ns != orig.duration_ns;
        if (wasRefined) {
            jt_speech_candidate rm;
            if (measureSpeechCandidate(refined, iv, &rm)) {
                std::vector<jt_interval> ri = getIntervalsInRange(iv, refined.start_ns, refined.end_ns);
                rm.score = scoreSpeechCandidateGrounded(rm, nf, levelVariance(ri));
                rm.was_refined = 1; rm.original_start_ns = orig.start_ns; rm.original_duration_ns = orig.duration_ns;
                for (auto &c : res.cands) if (c.region.start_ns == orig.start_ns) { c = rm; break; }
                best = refined;
            }
        }
    }
    res.has = have; res.best = best;
    return res;
}

bool pickLowClusterRegion(const std::vector<jt_interval> &iv, double split, int64_t hop, jt_region *out)
{
    bool haveBest = false; jt_region best{}; int64_t runStart = 0; bool inRun = false;
    auto closeRun = [&](int endIdx) {
        if (!inRun) return;
        int64_t e = iv[endIdx].timestamp_ns + hop; jt_region r{runStart, e, e - runStart};
        if (!haveBest || r.duration_ns > best.duration_ns) { best = r; haveBest = true; }
        inRun = false;
    };
    for (size_t i = 0; i < iv.size(); ++i) {
        bool below = level_of(iv[i]) < split;
        if (below) { if (!inRun) { runStart = iv[i].timestamp_ns; inRun = true; } continue; }
        if (inRun) closeRun((int)i - 1);
    }
    if (inRun) closeRun((int)iv.size() - 1);
    if (!haveBest) return false;
    Refined rf = refineToSubregion(best.start_ns, best.end_ns, best.duration_ns, iv, 10 * kSecond, 8 * kSecond, false);
    *out = rf.ok ? jt_region{rf.start, rf.end, rf.duration} : best;
    return true;
}

// ------------------------------------------------------------------ noise seed (analyser_noise_seed.go)
double roomToneScore(const jt_interval &s, double levelP50, double fluxP50)
{
    double amp = 1.0;
    if (s.momentary_lufs > levelP50) { amp = 1.0 - (s.momentary_lufs - levelP50) / 6.0; if (amp < 0) amp = 0; }
    double fl = 1.0;
    if (fluxP50 > 0 && s.spectral.flux > fluxP50) { double ratio = s.spectral.flux / fluxP50; if (ratio > 1) fl = 1.0 / ratio; }
    return 0.6 * amp + 0.4 * fl;
}
bool nan_less(double a, double b) { if (std::isnan(a)) return !std::isnan(b); if (std::isnan(b)) return false; return a < b; }   // Go slices.Sort: NaN first
int cmpd(double a, double b) { if (nan_less(a, b)) return -1; if (nan_less(b, a)) return 1; return 0; }
bool estimateNoiseFloorAndThreshold(const std::vector<jt_interval> &iv, double *nf, double *thr)
{
    if (iv.size() < 10) return false;
    std::vector<double> levels(iv.size()), flux(iv.size());
    for (size_t i = 0; i < iv.size(); ++i) { levels[i] = iv[i].momentary_lufs; flux[i] = iv[i].spectral.flux; }
    // element at index n/2 of the NaN-first sorted slice (computeSilenceMedians sorts; only the median is read): selection, not a sort
    auto median_nan_first = [](std::vector<double> &v) {
        const size_t k = v.size() / 2;
        auto mid = std::partition(v.begin(), v.end(), [](double x) { return std::isnan(x); });
        const size_t nn = (size_t)(mid - v.begin());
        if (k < nn) return v[k];
        std::nth_element(mid, v.begin() + k, v.end());
        return v[k];
    };
    double levelP50 = median_nan_first(levels), fluxP50 = median_nan_first(flux);
    struct SI { int idx; double level, score; };
    std::vector<SI> sc(iv.size());
    bool any_nan = false;
    for (size_t i = 0; i < iv.size(); ++i) {
        sc[i] = SI{(int)i, iv[i].momentary_lufs, roomToneScore(iv[i], levelP50, fluxP50)};
        any_nan |= std::isnan(sc[i].level) || std::isnan(sc[i].score);
    }
    // top fifth by (score desc, level asc, index asc): a strict total order, so selecting the first cc elements gives the same
    // set as the reference's full sort (analyser_noise_seed.go:150-175)
    size_t cc0 = sc.size() / 5; cc0 = std::max<size_t>(cc0, 8); cc0 = std::min(cc0, sc.size());
    if (cc0 < sc.size()) {
        if (any_nan)
            std::nth_element(sc.begin(), sc.begin() + cc0, sc.end(), [](const SI &a, const SI &b) {
                int c = cmpd(b.score, a.score); if (c) return c < 0;
                c = cmpd(a.level, b.level); if (c) return c < 0;
                return a.idx < b.idx; });
        else
            std::nth_element(sc.begin(), sc.begin() + cc0, sc.end(), [](const SI &a, const SI &b) {
                if (a.score != b.score) return a.score > b.score;
                if (a.level != b.level) return a.level < b.level;
                return a.idx < b.idx; });
    }
    size_t cc = sc.size() / 5; cc = std::max<size_t>(cc, 8); cc = std::min(cc, sc.size());
    double mx = -120.0; bool seen = false;
    for (size_t i = 0; i < cc; ++i) { double l = sc[i].level; if (isFlooredLevel(l)) continue; if (!seen || l > mx) { mx = l; seen = true; } }
    if (!seen) return false;
    *nf = mx; *thr = mx + 1.0;
    return true;
}
double calculateAdaptiveSilenceThreshold(double nf) { double t = nf + 6.0; if (t < -70.0) t = -70.0; if (t > -35.0) t = -35.0; return t; }

}  // namespace

// =====================================================================================================
// intervals
// =====================================================================================================
extern "C" int64_t jt_host_build_intervals(int sr, int64_t n_samples, int frame_samples, int channels,
                                           const double *fss, const double *fpk, int64_t n_frames,
                                           const jt_frame_meta *meta, int64_t n_meta, int quantize, jt_interval *out, int64_t cap)
{
    return jt_host_build_intervals_v(sr, n_samples, frame_samples, nullptr, channels, fss, fpk, n_frames, meta, n_meta, quantize, out, cap);
}

extern "C" int64_t jt_host_build_intervals_v(int sr, int64_t n_samples, int frame_samples, const int32_t *frame_lens, int channels,
                                             const double *fss, const double *fpk, int64_t n_frames,
                                             const jt_frame_meta *meta, int64_t n_meta, int quantize, jt_interval *out, int64_t cap)
{
    struct IA { int frameCount = 0; double rawSS = 0; int64_t rawN = 0; double rawPk = 0; jt_spectral spec = spec_zero(); bool specFound = false;
                double mSum = 0, sSum = 0, tpMax = 0, spMax = 0; };
    IA acc; int64_t nout = 0; int64_t intervalStart = 0; int64_t processed = 0;
    const int blk = sr / 10;
    int64_t meta_next = 0;
    auto finalize = [&](int64_t ts) {
        if (nout >= cap) { nout++; return; }
        jt_interval s; std::memset(&s, 0, sizeof(s));
        s.timestamp_ns = ts;
        s.peak_level = acc.rawPk > 0 ? 20.0 * std::log10(acc.rawPk) : -120.0;
        s.true_peak = acc.tpMax; s.sample_peak = acc.spMax;
        if (acc.rawN > 0) { double rms = std::sqrt(acc.rawSS / (double)acc.rawN); s.rms_level = rms < 0.00001 ? -120.0 : 20.0 * std::log10(rms); }
        else s.rms_level = -120.0;
        if (acc.frameCount > 0) {
            double n = (double)acc.frameCount;
            s.spectral = spec_avg(acc.spec, n); s.spectral_found = acc.specFound ? 1 : 0;
            s.momentary_lufs = acc.mSum / n; s.shortterm_lufs = acc.sSum / n;
        }
        out[nout++] = s;
    };
    auto reset = [&]() { acc = IA(); acc.tpMax = -120.0; acc.spMax = -120.0; };
    // The print-format quantisation of the metadata (17 values per 100 ms frame: "%.3f" ebur128, "%g" aspectralstats) is the only
    // O(file) host work on this path; it is elementwise, so long files split it across a few threads before the serial walk.
    std::vector<jt_frame_meta> qmeta;
    if (quantize && n_meta > 0) {
        const double tq0 = now_ms();
        qmeta.resize((size_t)n_meta);
        auto quant_range = [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) {
                jt_frame_meta q = meta[i];
                if (!std::isnan(q.momentary)) {
                    q.momentary = qdec(q.momentary, 3); q.shortterm = qdec(q.shortterm, 3);
                    q.true_peak = qdec(q.true_peak, 3); q.sample_peak = qdec(q.sample_peak, 3);
                }
                q.spectral = spec_quant(q.spectral);
                qmeta[(size_t)i] = q;
            }
        };
        int nt = 1;
        jt_parallel_for(n_meta, [&](int64_t lo, int64_t hi, int) { quant_range(lo, hi); }, &nt);
        meta = qmeta.data();
        if (jt_host_timing().load(std::memory_order_relaxed)) fprintf(stderr, "quant prepass %.3f ms (%d parts)\n", now_ms() - tq0, nt);
    }
    auto add_meta = [&](const jt_frame_meta &m) {
        bool has = !std::isnan(m.momentary);
        double tp = 0, sp = 0, M = 0, S = 0;
        if (has) {
            M = m.momentary; S = m.shortterm;
            tp = linearRatioToDB(m.true_peak);
            sp = linearRatioToDB(m.sample_peak);
        }
        if (acc.frameCount == 0 || tp > acc.tpMax) acc.tpMax = tp;
        if (acc.frameCount == 0 || sp > acc.spMax) acc.spMax = sp;
        spec_add(acc.spec, m.spectral); acc.specFound = true;
        acc.mSum += M; acc.sSum += S; acc.frameCount++;
    };
    for (int64_t f = 0; f < n_frames; ++f) {
        // inputFrame.NbSamples() (analyser.go:592): the frame's own length
        int64_t nb = frame_lens ? (int64_t)frame_lens[f] : std::min<int64_t>(frame_samples, n_samples - f * (int64_t)frame_samples);
        int64_t t = (int64_t)((double)processed / (double)sr * 1e9);
        processed += nb;
        acc.rawSS += fss[f]; acc.rawN += nb * channels; if (fpk[f] > acc.rawPk) acc.rawPk = fpk[f];
        if (t - intervalStart >= kHop) { finalize(intervalStart); intervalStart = t; reset(); }
        // output frames released by this push: ebur128 has seen floor(processed/1024)*1024 samples (aspectralstats hop buffering)
        int64_t seen = (processed / 1024) * 1024;
        while (meta_next < n_meta && (meta_next + 1) * (int64_t)blk <= seen) add_meta(meta[meta_next++]);
    }
    while (meta_next < n_meta) add_meta(meta[meta_next++]);     // EOF flush
    if (acc.rawN > 0) finalize(intervalStart);
    return nout;
}

// =====================================================================================================
// measurements + VAD
// =====================================================================================================
extern "C" void jt_host_afftdn_band_edges(int index, double *lo, double *hi)
{
    static const double c[15] = {80, 125, 195, 290, 440, 660, 1000, 1500, 2250, 3350, 5000, 7500, 11200, 16000, 24000};
    const int last = 14;
    if (index <= 0) *lo = c[0] / std::sqrt(c[1] / c[0]); else *lo = std::sqrt(c[index - 1] * c[index]);
    if (index >= last) *hi = c[last] * std::sqrt(c[last] / c[last - 1]); else *hi = std::sqrt(c[index] * c[index + 1]);
}

static void detect_vad(const std::vector<jt_interval> &iv, double seed, jt_measurements *m)
{
    // detectVoiceActivity (analyser_vad.go:728-783)
    const int64_t hop = kHop;
    Hist h = buildLevelHistogram(iv, 1.0);
    std::vector<double> levels = vadLevels(iv);
    double p75 = percentileOfSorted(levels, 75);
    double split = clampSplit(otsuSplit(h), seed, p75);
    double floor_ = percentileFloor(levels, seed);
    const double tv0 = now_ms();
    std::vector<bool> flags(iv.size());
    for (size_t i = 0; i < iv.size(); ++i) flags[i] = isSpeechInterval(iv[i], split);
    double margin = hysteresisMargin(h, split);
    int tol = gapToleranceIntervals(flags, hop);
    std::vector<jt_region> runs = buildSpeechRuns(iv, split, margin, tol, hop);
    m->vad_split = split; m->vad_margin = margin; m->vad_gap_tol = tol;
    m->n_speech_regions = (int)std::min<size_t>(runs.size(), JT_MAX_REGIONS);
    for (int i = 0; i < m->n_speech_regions; ++i) m->speech_regions[i] = runs[i];

    jt_region noiseRegion{};
    bool haveNoiseRegion = pickLowClusterRegion(iv, split, hop, &noiseRegion);
    bool haveProfile = false;
    if (haveNoiseRegion) {
        std::vector<jt_interval> r = getIntervalsInRange(iv, noiseRegion.start_ns, noiseRegion.start_ns + noiseRegion.duration_ns);
        if (!r.empty()) {
            Acc ac = accumulateIntervalMetrics(r);
            double n = (double)r.size(), avgRMS = ac.rmsSum / n;
            jt_spectral avg = spec_avg(ac.spec, n);
            jt_noise_profile &np = m->noise_profile;
            np.start_ns = noiseRegion.start_ns; np.duration_ns = noiseRegion.duration_ns;
            np.measured_noise_floor = floor_;           // overwritten with the percentile floor (analyser_vad.go:751)
            np.peak_level = ac.peakMax; np.crest_factor = ac.peakMax - avgRMS; np.entropy = avg.entropy; np.spectral = avg;
            np.warning = noiseRegion.duration_ns < 8 * kSecond ? 1 : (noiseRegion.duration_ns > 18 * kSecond ? 2 : 0);
            haveProfile = true; m->has_noise_profile = 1;
            // setVADRoomToneSample
            m->has_room_tone_sample = 1;
            m->room_tone_sample.rms_level = avgRMS; m->room_tone_sample.peak_level = ac.peakMax; m->room_tone_sample.crest_factor = ac.peakMax - avgRMS;
            m->room_tone_sample.spectral = avg; m->room_tone_sample.momentary_lufs = ac.mSum / n; m->room_tone_sample.shortterm_lufs = ac.sSum / n;
            m->room_tone_sample.true_peak = ac.tpMax; m->room_tone_sample.sample_peak = ac.spMax;
        }
    }
    const double tv1 = now_ms();
    BestResult br = findBestSpeechRegion(runs, iv, haveProfile, floor_);
    if (jt_host_timing().load(std::memory_order_relaxed)) fprintf(stderr, "vad: runs+noise %.3f ms, best speech %.3f ms\n", tv1 - tv0, now_ms() - tv1);
    m->n_candidates = (int)std::min<size_t>(br.cands.size(), JT_MAX_REGIONS);
    for (int i = 0; i < m->n_candidates; ++i) m->candidates[i] = br.cands[i];
    bool elected = false; jt_region electedRegion{};
    if (br.has) for (auto &c : br.cands) if (c.region.start_ns == br.best.start_ns) { m->speech_profile = c; m->has_speech_profile = 1; elected = true; electedRegion = c.region; break; }
    // deriveGateStatistics
    {
        std::vector<double> voiced, noise;
        for (auto &s : iv) { double l = level_of(s); if (isFlooredLevel(l)) continue; if (l < split) noise.push_back(l); }
        if (elected) { std::vector<jt_interval> r = getIntervalsInRange(iv, electedRegion.start_ns, electedRegion.end_ns);
                       for (auto &s : r) if (isSpeechInterval(s, split)) voiced.push_back(level_of(s)); }
        std::sort(voiced.begin(), voiced.end()); std::sort(noise.begin(), noise.end());
        m->voiced_low_percentile = percentileOfSorted(voiced, 10.0);
        m->noise_high_percentile = percentileOfSorted(noise, 95.0);
        m->gate_separation_db = m->voiced_low_percentile - m->noise_high_percentile;
    }
    m->floor = floor_; m->floor_source = 3;
    { double counted = 0, fl = 0; for (auto &s : iv) { double l = level_of(s); counted++; if (std::isnan(l) || l <= vadLevelFloorDB) fl++; }
      m->floored_fraction = counted == 0 ? 0 : fl / counted; }
    m->voice_activated = m->floored_fraction >= 0.20 ? 1 : 0;
}

extern "C" int jt_host_detect(const jt_analysis *p1, const jt_interval *ivp, int64_t n_iv, double duration_s, double target_i,
                              int quantize, jt_measurements *m)
{
    if (!p1 || !m || (n_iv > 0 && !ivp)) return JT_E_INVAL;
    std::memset(m, 0, sizeof(*m));
    std::vector<jt_interval> iv(ivp, ivp + n_iv);
    auto q3 = [&](double v) { return quantize ? qfmt("%.3f", v) : v; };
    auto q6 = [&](double v) { return quantize ? qfmt("%f", v) : v; };
    m->duration_s = duration_s;
    // buildInputMeasurements (analyser.go:364-406)
    double nf = 0, thr = 0;
    const double td0 = now_ms();
    if (!estimateNoiseFloorAndThreshold(iv, &nf, &thr)) { nf = vadLevelFloorDB; thr = calculateAdaptiveSilenceThreshold(vadLevelFloorDB); }
    m->floor_prescan = nf; m->room_tone_detect_level = thr;
    m->input_i = q3(p1->r128.integrated);
    m->input_tp = linearRatioToDB(q3(p1->r128.true_peak));
    m->input_lra = q3(p1->r128.lra);
    m->input_thresh = m->input_i - 10.0;
    m->target_offset = target_i - m->input_i;
    m->momentary = q3(p1->r128.momentary); m->shortterm = q3(p1->r128.shortterm);
    m->sample_peak = linearRatioToDB(q3(p1->r128.sample_peak));
    m->spectral = p1->spectral_mean;
    // assignAstatsMeasurements with the metadata conversions (analyser_metrics.go:535-605)
    const jt_astats &a = p1->astats;
    jt_astats d = a;
    d.dc_offset = q6(a.dc_offset); d.min_difference = q6(a.min_difference); d.max_difference = q6(a.max_difference);
    d.mean_difference = q6(a.mean_difference); d.rms_difference = q6(a.rms_difference);
    d.peak_level = q6(a.peak_level); d.rms_level = q6(a.rms_level); d.rms_peak = q6(a.rms_peak); d.rms_trough = q6(a.rms_trough);
    d.crest_factor = linearRatioToDB(q6(a.crest_factor)); d.flat_factor = q6(a.flat_factor);
    d.min_level = linearSampleToDBFS(q6(a.min_level)); d.max_level = linearSampleToDBFS(q6(a.max_level));
    d.noise_floor = q6(a.noise_floor); d.entropy = q6(a.entropy); d.dynamic_range = q6(a.dynamic_range);
    d.zero_crossings_rate = q6(a.zero_crossings_rate);
    m->dynamics = d; m->floor_astats = d.noise_floor;
    // assignInputNoiseFloor (analyser.go:488-511)
    if (d.rms_trough != 0 && !(std::isinf(d.rms_trough) && d.rms_trough < 0)) { m->floor = d.rms_trough; m->floor_source = 0; }
    else if (d.rms_level != 0 && !(std::isinf(d.rms_level) && d.rms_level < 0)) { m->floor = d.rms_level - 15.0; m->floor_source = 1; }
    else { double off = m->input_i > -20.0 ? 18.0 : (m->input_i > -30.0 ? 12.0 : 8.0); m->floor = m->input_thresh - off; m->floor_source = 2; }
    m->floor = std::max(-90.0, std::min(-30.0, m->floor));

    const double td1 = now_ms();
    detect_vad(iv, m->floor_prescan, m);
    if (jt_host_timing().load(std::memory_order_relaxed)) fprintf(stderr, "detect: seed %.3f ms, vad %.3f ms\n", td1 - td0, now_ms() - td1);
    return JT_OK;
}

extern "C" void jt_host_finish_measurements(jt_measurements *m)
{
    if (m->dynamics.rms_level != 0 && m->floor != 0) {
        m->reduction_headroom = std::max(0.0, std::min(60.0, m->dynamics.rms_level - m->floor));
        return;
    }
    m->reduction_headroom = m->input_i > -20.0 ? 40.0 : (m->input_i > -30.0 ? 25.0 : 15.0);
}

// =====================================================================================================
// configuration, adaptation, filter-spec strings
// =====================================================================================================
extern "C" void jt_host_default_config(jt_host_config *c)
{
    std::memset(c, 0, sizeof(*c));
    c->downmix_enabled = 1; c->analysis_enabled = 1;
    c->resample_enabled = 1; c->resample_rate = 44100; c->resample_frame = 4096;
    c->rumble_hp = jt_biquad_cfg{1, 80.0, 2, 0.707, 1.0, 1};
    c->bandlimit_lp = jt_biquad_cfg{1, 20500.0, 2, 0.707, 1.0, 1};
    c->nr_enabled = 1; c->nr_strength = 0.00001; c->nr_patch_s = 0.0060; c->nr_research_s = 0.0020; c->nr_smooth = 3.0;
    c->afftdn_enabled = 1; c->afftdn_nr = 12; c->afftdn_custom = 0; c->afftdn_track_noise = 1; c->afftdn_noise_floor = 0;
    c->gate_enabled = 1; c->gate_threshold = 0.01; c->gate_ratio = 2.0; c->gate_attack = 5.0; c->gate_release = 200.0;
    c->gate_range = DbToLinear(-14.0); c->gate_knee = 3.0; c->gate_makeup = 1.0; c->gate_detection_set = 1;
    c->comp_enabled = 1; c->comp_threshold_db = -18; c->comp_ratio = 3.0; c->comp_attack = 10; c->comp_release = 200;
    c->comp_makeup_db = 0; c->comp_knee = 4.0; c->comp_mix = 1.0;
    c->deess_enabled = 1; c->deess_intensity = 0.0; c->deess_amount = 0.50; c->deess_frequency = 0.80;
    c->adeclick_enabled = 1; c->adeclick_threshold = 1.7; c->adeclick_window = 55.0; c->adeclick_overlap = 50.0; c->adeclick_method_s = 1;
    c->loudnorm_enabled = 1; c->target_i = -16.0; c->target_tp = -1.0; c->target_lra = 20.0; c->dual_mono = 1; c->linear = 1;
}

static std::string buildAfftdnBandNoise(const double *bands, int n)
{
    if (n == 0) return "";
    double sum = 0; int fin = 0;
    for (int i = 0; i < n; ++i) if (isFiniteD(bands[i])) { sum += bands[i]; fin++; }
    if (!fin) return "";
    double mean = sum / (double)fin;
    std::string out;
    for (int i = 0; i < n; ++i) {
        if (i) out += "|";
        if (!isFiniteD(bands[i])) { out += sfmt("%.1f", 0.0); continue; }
        double shape = std::max(-24.0, std::min(24.0, bands[i] - mean));
        out += sfmt("%.1f", shape);
    }
    return out;
}

extern "C" void jt_host_adapt(const jt_host_config *base, const jt_measurements *m, jt_host_config *c, jt_adaptive_diag *dg)
{
    *c = *base;
    jt_adaptive_diag d; std::memset(&d, 0, sizeof(d));
    // tuneBandlimitLowPass
    c->bandlimit_lp.enabled = 1; c->bandlimit_lp.frequency = 20500.0; c->bandlimit_lp.poles = 2; c->bandlimit_lp.mix = 1.0;
    // tuneNoiseReduction
    if (m->voice_activated) { c->afftdn_enabled = 0; d.afftdn_enabled = 0; d.afftdn_disabled_voice_activated = 1; }
    else {
        d.afftdn_enabled = c->afftdn_enabled;
        if (m->floor != 0) {
            double fl = std::max(-80.0, std::min(-20.0, m->floor));
            c->afftdn_noise_floor = fl; c->afftdn_track_noise = 0; d.afftdn_noise_floor_db = fl;
            c->afftdn_custom = 0;
            bool custom = m->has_noise_profile && m->noise_profile.bands_measured && !(m->gate_separation_db < 12.0) &&
                          m->noise_profile.spectral.flatness >= 0.45;
            if (custom) {
                std::string bn = buildAfftdnBandNoise(m->noise_profile.band_noise, m->noise_profile.band_noise_n);
                if (!bn.empty()) { c->afftdn_custom = 1; snprintf(c->afftdn_band_noise, sizeof(c->afftdn_band_noise), "%s", bn.c_str()); }
            }
            d.afftdn_custom = c->afftdn_custom;
        }
    }
    // tuneSpeechGate
    {
        double crest = 15.0, peak = 0;
        if (m->has_noise_profile) { crest = m->noise_profile.crest_factor; peak = m->noise_profile.peak_level; }
        double lufsGap = c->target_i - m->input_i; if (lufsGap < 0) lufsGap = 0;
        c->gate_ratio = m->input_lra > 15.0 ? 1.5 : 2.0;
        bool narrow = false;
        if (m->has_speech_profile) {
            double thrDB = m->voiced_low_percentile - 6.0;
            narrow = m->gate_separation_db < (6.0 + 6.0);
            thrDB = std::max(-80.0, std::min(thrDB, -25.0));
            c->gate_threshold = DbToLinear(thrDB);
            d.gate_narrow_gap = narrow; d.gate_quiet_speech_estimate = m->voiced_low_percentile; d.gate_separation = m->gate_separation_db;
            d.gate_threshold_unclamped = m->voiced_low_percentile - 6.0;
            d.gate_speech_headroom = m->voiced_low_percentile - LinearToDb(c->gate_threshold);
        } else {
            double thrDB;
            bool usePeak = crest > 20.0 && peak != 0 && lufsGap < 25.0;
            if (usePeak) thrDB = peak + 3.0;
            else { double minGap = 12.0 / (1.0 - 1.0 / c->gate_ratio); thrDB = std::max(m->floor + minGap, -40.0); }
            thrDB = std::max(-80.0, std::min(thrDB, -25.0));
            c->gate_threshold = DbToLinear(thrDB);
        }
        c->gate_attack = 5.0; c->gate_release = 200.0;
        double depth = narrow ? 8.0 : 14.0;
        c->gate_range = DbToLinear(-depth); d.gate_depth_db = depth;
        c->gate_knee = 3.0; c->gate_detection_set = 1;
    }
    // tuneDeesser
    if (!m->has_speech_profile || !m->speech_profile.bands_measured) c->deess_intensity = 0.0;
    else {
        double ex = m->speech_profile.sib_band_rms - m->speech_profile.body_band_rms;
        if (ex < -6.0) c->deess_intensity = 0.0;
        else if (ex < -3.0) c->deess_intensity = ((ex - (-6.0)) / (-3.0 - (-6.0))) * 0.6;
        else if (ex < 0.0) c->deess_intensity = 0.6 + ((ex - (-3.0)) / (0.0 - (-3.0))) * (0.85 - 0.6);
        else c->deess_intensity = 0.85;
    }
    // tuneLevellingCompressor
    c->comp_ratio = 3.0; c->comp_attack = 10.0; c->comp_release = 200.0; c->comp_knee = 4.0; c->comp_mix = 1.0; c->comp_makeup_db = 0.0;
    {
        bool set = false; double thr = 0;
        if (m->has_speech_profile) {
            double eff = m->speech_profile.sample.rms_level, full = m->dynamics.rms_level;
            if (full < 0 && !(std::isinf(full) && full < 0)) eff = std::max(eff, full);
            thr = eff + 9.0; set = true;
        } else if (std::isnan(m->dynamics.peak_level) || std::isinf(m->dynamics.peak_level)) c->comp_threshold_db = -18.0;
        else { thr = m->dynamics.peak_level - 20.0; set = true; }
        if (set) c->comp_threshold_db = std::max(-45.0, std::min(thr, -6.0));
    }
    // sanitizeConfig
    c->rumble_hp.frequency = sanitizeFloat(c->rumble_hp.frequency, 80.0); c->rumble_hp.width = sanitizeFloat(c->rumble_hp.width, 0.707); c->rumble_hp.mix = sanitizeFloat(c->rumble_hp.mix, 1.0);
    c->bandlimit_lp.frequency = sanitizeFloat(c->bandlimit_lp.frequency, 20500.0); c->bandlimit_lp.width = sanitizeFloat(c->bandlimit_lp.width, 0.707); c->bandlimit_lp.mix = sanitizeFloat(c->bandlimit_lp.mix, 1.0);
    c->nr_strength = sanitizeFloat(c->nr_strength, 0.00001); c->nr_patch_s = sanitizeFloat(c->nr_patch_s, 0.0060);
    c->nr_research_s = sanitizeFloat(c->nr_research_s, 0.0020); c->nr_smooth = sanitizeFloat(c->nr_smooth, 3.0);
    c->afftdn_nr = sanitizeFloat(c->afftdn_nr, 12); c->afftdn_noise_floor = sanitizeFloat(c->afftdn_noise_floor, 0);
    if (c->afftdn_custom && c->afftdn_band_noise[0] == 0) c->afftdn_custom = 0;
    if (!isFiniteD(c->gate_threshold) || c->gate_threshold <= 0) c->gate_threshold = 0.01;
    c->gate_ratio = sanitizeFloat(c->gate_ratio, 2.0); c->gate_attack = sanitizeFloat(c->gate_attack, 5.0); c->gate_release = sanitizeFloat(c->gate_release, 200.0);
    c->gate_range = sanitizeFloat(c->gate_range, DbToLinear(-14.0)); c->gate_knee = sanitizeFloat(c->gate_knee, 3.0); c->gate_makeup = sanitizeFloat(c->gate_makeup, 1.0);
    c->comp_ratio = sanitizeFloat(c->comp_ratio, 3.0); c->comp_threshold_db = sanitizeFloat(c->comp_threshold_db, -18.0);
    c->comp_attack = sanitizeFloat(c->comp_attack, 10); c->comp_release = sanitizeFloat(c->comp_release, 200); c->comp_makeup_db = sanitizeFloat(c->comp_makeup_db, 0);
    c->comp_knee = sanitizeFloat(c->comp_knee, 4.0); c->comp_mix = sanitizeFloat(c->comp_mix, 1.0);
    c->deess_intensity = sanitizeFloat(c->deess_intensity, 0.0); c->deess_amount = sanitizeFloat(c->deess_amount, 0.50); c->deess_frequency = sanitizeFloat(c->deess_frequency, 0.80);
    if (dg) *dg = d;
}

namespace {
std::string buildBiquadFilter(const jt_biquad_cfg &b, const char *kw)
{
    if (!b.enabled) return "";
    int poles = b.poles < 1 ? 2 : b.poles; double width = b.width <= 0 ? 0.707 : b.width;
    std::string s = sfmt("%s=f=%.0f:poles=%d:width_type=q:width=%.3f:normalize=1", kw, b.frequency, poles, width);
    if (b.transform_tdii) s += ":a=tdii";
    if (b.mix > 0 && b.mix < 1.0) s += sfmt(":m=%.2f", b.mix);
    return s;
}
std::string buildAfftdn(const jt_host_config &c)
{
    if (!c.afftdn_enabled) return "";
    int tn = c.afftdn_track_noise ? 1 : 0; std::string s;
    if (c.afftdn_custom && c.afftdn_band_noise[0]) s = "afftdn=nr=" + go_g(c.afftdn_nr) + ":nt=custom:bn=" + c.afftdn_band_noise + sfmt(":tn=%d", tn);
    else s = "afftdn=nr=" + go_g(c.afftdn_nr) + ":nt=" + (c.afftdn_custom ? "custom" : "w") + sfmt(":tn=%d", tn);
    if (c.afftdn_noise_floor < 0) s += ":nf=" + go_g(c.afftdn_noise_floor);
    return s;
}
std::string analysisSpec(const jt_host_config &c)
{
    if (!c.analysis_enabled) return "";
    return sfmt("astats=metadata=1:measure_perchannel=all,aspectralstats=win_size=2048:win_func=hann:measure=all,"
                "ebur128=metadata=1:peak=sample+true:dualmono=true:target=%.0f", c.target_i);
}
std::string outputFormatSpec(const jt_host_config &c)
{ return sfmt("aformat=sample_rates=%d:channel_layouts=mono:sample_fmts=s16,asetnsamples=n=%d", c.resample_rate, c.resample_frame); }
void join(std::string &acc, const std::string &part) { if (part.empty()) return; if (!acc.empty()) acc += ","; acc += part; }
std::string preLimiterPrefix(double preGain, double ceiling, bool needed)
{
    if (!needed) return "";
    std::string s;
    if (preGain > 0) join(s, sfmt("volume=%.1fdB", preGain));
    join(s, sfmt("alimiter=limit=%.6f:attack=5:release=100:level_in=1:level_out=1:level=0:latency=1:asc=1:asc_level=0.8", DbToLinear(ceiling)));
    return s;
}
}  // namespace

extern "C" int jt_host_filter_spec(const jt_host_config *c, int pass, char *buf, int cap)
{
    std::string s;
    if (c->downmix_enabled) join(s, "aformat=channel_layouts=mono");
    if (pass == 2) {
        join(s, buildBiquadFilter(c->rumble_hp, "highpass"));
        join(s, buildBiquadFilter(c->bandlimit_lp, "lowpass"));
        if (c->nr_enabled) {
            std::string nr = sfmt("anlmdn=s=%.5f:p=%.4f:r=%.4f:m=%.0f", c->nr_strength, c->nr_patch_s, c->nr_research_s, c->nr_smooth);
            join(nr, buildAfftdn(*c));
            join(s, nr);
        }
        if (c->gate_enabled)
            join(s, sfmt("agate=threshold=%.6f:ratio=%.1f:attack=%.2f:release=%.0f:range=%.4f:knee=%.1f:detection=%s:makeup=%.1f",
                         c->gate_threshold, c->gate_ratio, c->gate_attack, c->gate_release, c->gate_range, c->gate_knee, "rms", c->gate_makeup));
        if (c->comp_enabled)
            join(s, sfmt("acompressor=threshold=%.6f:ratio=%.1f:attack=%.0f:release=%.0f:makeup=%.2f:knee=%.1f:detection=rms:mix=%.2f",
                         DbToLinear(c->comp_threshold_db), c->comp_ratio, c->comp_attack, c->comp_release, DbToLinear(c->comp_makeup_db), c->comp_knee, c->comp_mix));
        if (c->deess_enabled && c->deess_intensity > 0)
            join(s, sfmt("deesser=i=%.2f:m=%.2f:f=%.2f", c->deess_intensity, c->deess_amount, c->deess_frequency));
    }
    join(s, analysisSpec(*c));
    if (pass == 2 && c->resample_enabled) join(s, outputFormatSpec(*c));
    if (buf && cap > 0) snprintf(buf, (size_t)cap, "%s", s.c_str());
    return (int)s.size();
}

extern "C" void jt_host_filter_params(const jt_host_config *c, jt_filter_params *p)
{
    std::memset(p, 0, sizeof(*p));
    p->hp_enabled = c->rumble_hp.enabled; p->hp_freq = qfmt("%.0f", c->rumble_hp.frequency); p->hp_q = qfmt("%.3f", c->rumble_hp.width <= 0 ? 0.707 : c->rumble_hp.width);
    p->lp_enabled = c->bandlimit_lp.enabled; p->lp_freq = qfmt("%.0f", c->bandlimit_lp.frequency); p->lp_q = qfmt("%.3f", c->bandlimit_lp.width <= 0 ? 0.707 : c->bandlimit_lp.width);
    p->nlm_enabled = c->nr_enabled; p->nlm_strength = qfmt("%.5f", c->nr_strength); p->nlm_patch_s = qfmt("%.4f", c->nr_patch_s);
    p->nlm_research_s = qfmt("%.4f", c->nr_research_s); p->nlm_smooth = qfmt("%.0f", c->nr_smooth);
    p->fft_enabled = c->nr_enabled && c->afftdn_enabled; p->fft_nr = std::strtod(go_g(c->afftdn_nr).c_str(), nullptr);
    p->fft_nf = c->afftdn_noise_floor < 0 ? std::strtod(go_g(c->afftdn_noise_floor).c_str(), nullptr) : 0.0;
    p->fft_track_noise = c->afftdn_track_noise;
    p->fft_custom = (c->afftdn_custom && c->afftdn_band_noise[0]) ? 1 : 0;
    if (p->fft_custom) {
        const char *s = c->afftdn_band_noise; int i = 0;
        while (*s && i < 15) { char *e; p->fft_band_noise[i++] = std::strtod(s, &e); s = e; while (*s == '|' || *s == ' ') ++s; }
    }
    p->gate_enabled = c->gate_enabled; p->gate_threshold = qfmt("%.6f", c->gate_threshold); p->gate_ratio = qfmt("%.1f", c->gate_ratio);
    p->gate_attack_ms = qfmt("%.2f", c->gate_attack); p->gate_release_ms = qfmt("%.0f", c->gate_release); p->gate_range = qfmt("%.4f", c->gate_range);
    p->gate_knee = qfmt("%.1f", c->gate_knee); p->gate_makeup = qfmt("%.1f", c->gate_makeup);
    p->comp_enabled = c->comp_enabled; p->comp_threshold = qfmt("%.6f", DbToLinear(c->comp_threshold_db)); p->comp_ratio = qfmt("%.1f", c->comp_ratio);
    p->comp_attack_ms = qfmt("%.0f", c->comp_attack); p->comp_release_ms = qfmt("%.0f", c->comp_release);
    p->comp_makeup = qfmt("%.2f", DbToLinear(c->comp_makeup_db)); p->comp_knee = qfmt("%.1f", c->comp_knee); p->comp_mix = qfmt("%.2f", c->comp_mix);
    p->deess_enabled = (c->deess_enabled && c->deess_intensity > 0) ? 1 : 0;
    p->deess_i = qfmt("%.2f", c->deess_intensity); p->deess_m = qfmt("%.2f", c->deess_amount); p->deess_f = qfmt("%.2f", c->deess_frequency);
    p->out_rate = c->resample_rate; p->out_frame_samples = c->resample_frame;
}

// =====================================================================================================
// normalisation planning (normalise.go)
// =====================================================================================================
extern "C" void jt_host_calculate_limiter_ceiling(double mi, double mtp, double ti, double ttp, double *ceiling, int *needed, int *clamped)
{
    double gain = ti - mi, proj = mtp + gain;
    *ceiling = 0; *needed = 0; *clamped = 0;
    if (proj <= ttp) return;
    double c = ttp - gain;
    if (c < -24.0) { c = -24.0; *clamped = 1; }
    *ceiling = c; *needed = 1;
}
extern "C" void jt_host_calculate_pre_gain(double mi, double ti, double ttp, double *pre, double *red)
{
    double gain = ti - mi, ideal = ttp - gain;
    *pre = 0; *red = 0;
    if (ideal >= -24.0) return;
    *pre = -24.0 - ideal;
    double postI = mi + *pre, ng = ti - postI;
    *red = ttp - ng;
}
extern "C" void jt_host_plan_limiter(double oi, double otp, const jt_host_config *cfg, jt_limiter_decision *o, jt_limiter_plan *plan)
{
    std::memset(o, 0, sizeof(*o));
    double ceiling; int needed, clamped; jt_host_calculate_limiter_ceiling(oi, otp, cfg->target_i, cfg->target_tp, &ceiling, &needed, &clamped);
    double pre, red; jt_host_calculate_pre_gain(oi, cfg->target_i, cfg->target_tp, &pre, &red);
    if (clamped) ceiling = red;
    o->pre_gain_db = pre; o->ceiling_db = ceiling; o->needed = needed; o->clamped = clamped; o->gain_db = cfg->target_i - oi; o->filtered_tp = otp;
    std::string pf = preLimiterPrefix(pre, ceiling, needed != 0);
    snprintf(o->pass3_prefix, sizeof(o->pass3_prefix), "%s", pf.c_str());
    if (plan) {
        plan->needed = needed; plan->pre_gain_db = (needed && pre > 0) ? qfmt("%.1f", pre) : 0.0;
        plan->limit = needed ? qfmt("%.6f", DbToLinear(ceiling)) : 1.0;
    }
}
extern "C" double jt_host_loudnorm_internal_target_tp(double ti, double mtp, double mi) { return mtp + (ti - mi) + 0.1 + 0.2; }
extern "C" void jt_host_calculate_linear_mode_target(double mi, double mtp, double di, double ttp, double *eff, double *off, int *lin)
{
    double maxLinear = ttp - mtp + mi - 0.1;
    if (di <= maxLinear) { *eff = di; *off = di - mi; *lin = 1; return; }
    *eff = maxLinear; *off = maxLinear - mi; *lin = 0;
}
extern "C" int jt_host_pass4_spec(const jt_host_config *cfg, const jt_loudnorm_stats *ms, double offset, const jt_limiter_decision *lim,
                                  int source_rate, const char *stats_path, char *buf, int cap, jt_loudnorm_apply *ap)
{
    // AdeclickConfig.Method is one of five spellings (filters.go:240-246); any other code would print no m= option and run overlap-add silently
    if (cfg->adeclick_enabled && (cfg->adeclick_method_s < 0 || cfg->adeclick_method_s > 4)) return JT_E_INVAL;
    double internalTP = jt_host_loudnorm_internal_target_tp(cfg->target_i, ms->input_tp, ms->input_i);
    double emittedTP = std::max(-9.0, std::min(internalTP, 0.0));
    double brickDB = cfg->target_tp - 0.9;
    std::string s;
    if (lim) join(s, preLimiterPrefix(lim->pre_gain_db, lim->ceiling_db, lim->needed != 0));
    std::string ln = sfmt("loudnorm=I=%.2f:TP=%.2f:LRA=%.1f:measured_I=%.2f:measured_TP=%.2f:measured_LRA=%.2f:measured_thresh=%.2f:offset=%.2f:dual_mono=%s:linear=%s:print_format=json",
                          cfg->target_i, emittedTP, cfg->target_lra, ms->input_i, ms->input_tp, ms->input_lra, ms->input_thresh, offset,
                          cfg->dual_mono ? "true" : "false", cfg->linear ? "true" : "false");
    if (stats_path && stats_path[0]) ln += std::string(":stats_file=") + stats_path;
    join(s, ln);
    if (source_rate > 0) join(s, sfmt("aresample=%d", source_rate));
    if (cfg->adeclick_enabled) {
        std::string a = sfmt("adeclick=t=%.1f:w=%.0f:o=%.0f", cfg->adeclick_threshold, cfg->adeclick_window, cfg->adeclick_overlap);
        // AdeclickConfig.Method goes into the spec verbatim (filters.go:958-960): "" -> nothing (af_adeclick.c defaults to m=a); af_adeclick's
        // option table names each method twice, "s" / "save" and "a" / "add"
        if (cfg->adeclick_method_s == 1) a += ":m=s";
        else if (cfg->adeclick_method_s == 2) a += ":m=a";
        else if (cfg->adeclick_method_s == 3) a += ":m=save";
        else if (cfg->adeclick_method_s == 4) a += ":m=add";
        join(s, a);
    }
    join(s, sfmt("alimiter=limit=%.6f:attack=1:release=50:level_in=1:level_out=1:level=0:latency=1:asc=1:asc_level=0.8", DbToLinear(brickDB)));
    join(s, "astats=metadata=1:measure_perchannel=all");
    join(s, "aspectralstats=win_size=2048:win_func=hann:measure=all");
    join(s, "ebur128=metadata=1:peak=sample+true:dualmono=true");
    join(s, outputFormatSpec(*cfg));
    if (buf && cap > 0) snprintf(buf, (size_t)cap, "%s", s.c_str());
    if (ap) {
        std::memset(ap, 0, sizeof(*ap));
        ap->target_i = qfmt("%.2f", cfg->target_i); ap->target_tp = qfmt("%.2f", emittedTP); ap->target_lra = qfmt("%.1f", cfg->target_lra);
        ap->measured_i = qfmt("%.2f", ms->input_i); ap->measured_tp = qfmt("%.2f", ms->input_tp);
        ap->measured_lra = qfmt("%.2f", ms->input_lra); ap->measured_thresh = qfmt("%.2f", ms->input_thresh);
        ap->offset = qfmt("%.2f", offset);
        ap->adeclick_enabled = cfg->adeclick_enabled; ap->adeclick_threshold = qfmt("%.1f", cfg->adeclick_threshold);
        ap->adeclick_window_ms = qfmt("%.0f", cfg->adeclick_window); ap->adeclick_overlap_pct = qfmt("%.0f", cfg->adeclick_overlap);
        ap->adeclick_method = (cfg->adeclick_method_s == 1 || cfg->adeclick_method_s == 3) ? 1 : 0;
        ap->brickwall_limit = qfmt("%.6f", DbToLinear(brickDB));
    }
    return (int)s.size();
}

// =====================================================================================================
// orchestration (processor.go:29-216)
// =====================================================================================================
static double secs_of(int64_t ns) { return qfmt("%f", (double)ns / 1e9); }    // regions travel through "%f"-formatted filter options

// Replay of the reference's intra-pass progress ticks (every 100th decoder frame: analyser.go:602-618, processor.go:320-335,
// normalise.go:292-301,1108-1117; 17 band ticks: analyser_band_runner.go:47-88).  A pass is one launch sequence here, so the ticks
// of a pass are emitted together when it completes, with the values the reference would have sent: progress from the frame count,
// Level = calculateFrameLevel of the frame at that position.
struct TickSink {
    jt_progress_fn cb = nullptr; void *user = nullptr; bool ticks = false; double duration = 0;
    void tick(int pass, const char *name, double progress, double level, const jt_measurements *ms = nullptr) const
    {
        if (!cb || !ticks) return;
        jt_progress_update u; std::memset(&u, 0, sizeof(u));
        u.pass = pass; u.pass_name = name; u.progress = progress; u.level = level; u.duration = duration; u.measurements = ms;
        cb(user, &u);
    }
};
static double level_from_sumsq(double sumsq, double count)
{
    if (count <= 0) return -70.0;
    const double rms = std::sqrt(sumsq / count);
    if (rms < 0.00001) return -70.0;
    return std::max(-70.0, std::min(0.0, 20.0 * std::log10(rms)));
}
static double bandPhaseProgress(int completed, int total)       // analyser_band_runner.go:72-84
{
    if (total <= 0) return 0.95;
    if (completed < 0) completed = 0;
    const double p = 0.95 + (1.0 - 0.95) * ((double)completed / (double)total);
    return p > 1.0 ? 1.0 : p;
}

static int analyse_core(jt_ctx *h, const jt_host_config *base, int frame_samples, jt_process_result *out,
                        jt_host_config *eff_out, bool pass2_follows, const TickSink *tk = nullptr)
{
    int rc;
    jt_analysis a1; std::memset(&a1, 0, sizeof(a1));
    const int64_t n = h->n; const int sr = h->sr;
    // frame_samples == 0: the input's own decoder-frame cadence (what Reader.ReadFrame would deliver for this file: jt_load_audio
    // recorded it), frame by frame when the stream's frames differ in length; a positive value is the caller's
    const int32_t *flens = (frame_samples == 0 && !h->dec_frame_lens.empty()) ? h->dec_frame_lens.data() : nullptr;
    const int fs_arg = frame_samples;                         // (jt_pass1 resolves 0 the same way)
    if (frame_samples < 0) { h->err = "frame_samples must be 0 (the input's own cadence) or positive"; return JT_E_INVAL; }
    if (frame_samples == 0) frame_samples = h->dec_frame_samples;
    if (frame_samples <= 0) return JT_E_INVAL;
    const int64_t nfr = flens ? (int64_t)h->dec_frame_lens.size() : (n + frame_samples - 1) / frame_samples;
    const int64_t cap_meta = n / (sr / 10) + 2;
    // the handle's own scratch, grown and never cleared (jt_pass1 writes nfr sums / peaks and a1.n_frames_meta records, nothing else is
    // read): an hour's frame records are 4.9 MB, and a fresh zeroed vector per call was 0.3 ms of page faults before Pass 1's first launch
    std::vector<double> &fss = h->host_fss, &fpk = h->host_fpk;
    std::vector<jt_frame_meta> &meta = h->host_meta;
    if (fss.size() < (size_t)nfr) { fss.resize((size_t)nfr); fpk.resize((size_t)nfr); }
    if (meta.size() < (size_t)cap_meta) meta.resize((size_t)cap_meta);
    if (pass2_follows) {
        // AdaptConfig takes the rumble filter, the band-limit filter and anlmdn from the base configuration, never from the
        // measurements (jt_host_adapt: tuneBandlimitLowPass is a constant, the other two are not tuned): the GPU can start them beside
        // the Pass-1 analysis (jt_pass1 queues them right behind its own kernels) and keep going while this thread builds intervals,
        // runs the VAD and waits for the band measurements.  jt_pass2 checks the parameters.
        jt_measurements none; std::memset(&none, 0, sizeof(none));
        jt_host_config guess; jt_adaptive_diag dg;
        jt_host_adapt(base, &none, &guess, &dg);
        jt_filter_params fp; jt_host_filter_params(&guess, &fp);
        (void)jt_pass2_prefetch_after_pass1(h, &fp);          // best effort: Pass 2 runs from the input if this did not start
    }
    double t0 = now_ms();
    if ((rc = jt_pass1(h, fs_arg, &a1, fss.data(), fpk.data(), nfr, meta.data(), cap_meta)) != JT_OK) return rc;
    out->pass_ms[0] = h->timers.pass1_ms;
    out->stage_ms[0] = now_ms() - t0; t0 = now_ms();
    if (tk && tk->ticks) {
        const double est = ((double)n / sr) * sr / 4096.0;                    // estimatedTotalFrames (analyser.go:560-566)
        for (int64_t fc = 0; fc < nfr && est > 0; fc += 100) {
            const int64_t cnt = flens ? (int64_t)flens[fc] * h->channels
                                      : std::min<int64_t>((int64_t)frame_samples * h->channels, (n - fc * frame_samples) * h->channels);
            tk->tick(1, "Analysing", std::min(0.95, (double)fc / est * 0.95), level_from_sumsq(fss[(size_t)fc], (double)cnt));
        }
    }
    std::vector<jt_interval> iv((size_t)(n / (sr / 5) + 16));
    int64_t niv = jt_host_build_intervals_v(sr, n, frame_samples, flens, h->channels, fss.data(), fpk.data(), nfr, meta.data(), a1.n_frames_meta, 1,
                                            iv.data(), (int64_t)iv.size());
    if (niv > (int64_t)iv.size()) niv = (int64_t)iv.size();
    h->last_intervals.assign(iv.begin(), iv.begin() + niv);
    jt_measurements &m = out->input;
    if ((rc = jt_host_detect(&a1, iv.data(), niv, (double)n / sr, base->target_i, 1, &m)) != JT_OK) return rc;
    out->stage_ms[1] = now_ms() - t0; t0 = now_ms();
    // measureSpeechBands (analyser_bands.go:115-166)
    if (m.has_speech_profile && m.speech_profile.region.duration_ns > 0) {
        const double lo[2] = {1000.0, 6000.0}, hi[2] = {3000.0, 9000.0}; double db[2]; int ok[2];
        rc = jt_band_rms(h, secs_of(m.speech_profile.region.start_ns), secs_of(m.speech_profile.region.duration_ns), lo, hi, 2, db, ok);
        if (rc == JT_OK) {
            if (ok[0]) m.speech_profile.body_band_rms = qfmt("%f", db[0]);
            if (ok[1]) m.speech_profile.sib_band_rms = qfmt("%f", db[1]);
            m.speech_profile.bands_measured = (ok[0] && ok[1]) ? 1 : 0;
        }
    }
    if (tk) for (int b = 1; b <= 2; ++b) tk->tick(1, "Analysing frequency bands", bandPhaseProgress(b, 17), 0.0);
    // measureNoiseBands (analyser_noise_bands.go:65-119)
    if (m.has_noise_profile && m.noise_profile.duration_ns > 0) {
        double lo[15], hi[15], db[15]; int ok[15];
        for (int i = 0; i < 15; ++i) jt_host_afftdn_band_edges(i, &lo[i], &hi[i]);
        rc = jt_band_rms(h, secs_of(m.noise_profile.start_ns), secs_of(m.noise_profile.duration_ns), lo, hi, 15, db, ok);
        if (rc == JT_OK) {
            int finite = 0;
            for (int i = 0; i < 15; ++i) { m.noise_profile.band_noise[i] = ok[i] ? qfmt("%f", db[i]) : 0.0; if (ok[i] && isFiniteD(db[i])) finite++; }
            m.noise_profile.band_noise_n = 15; m.noise_profile.bands_measured = finite >= 10 ? 1 : 0;
        }
    }
    if (tk) for (int b = 3; b <= 17; ++b) tk->tick(1, "Analysing frequency bands", bandPhaseProgress(b, 17), 0.0);
    out->stage_ms[2] = now_ms() - t0; t0 = now_ms();
    jt_host_finish_measurements(&m);
    jt_host_adapt(base, &m, eff_out, &out->diag);
    out->stage_ms[3] = now_ms() - t0;
    out->effective = *eff_out;
    out->input_lufs = m.input_i; out->input_tp_db = m.input_tp;
    return JT_OK;
}

extern "C" int jt_analyse_only(jt_ctx *h, const jt_host_config *base, int frame_samples, jt_process_result *out)
{
    if (!h || !base || !out) return JT_E_INVAL;
    std::memset(out, 0, sizeof(*out));
    jt_host_config eff;
    return analyse_core(h, base, frame_samples, out, &eff, false);
}

// generateLUFSOutputPath / lufsFilenameValue (processor.go:379-388): filepath.Dir / Base / Ext semantics for slash-separated paths
extern "C" int jt_host_lufs_filename_value(double output_lufs) { return (int)std::round(std::fabs(output_lufs)); }
extern "C" int jt_host_output_path(const char *input_path, int lufs_value, char *out, int cap)
{
    if (!input_path || !out || cap <= 0) return -1;
    std::string p(input_path);
    while (p.size() > 1 && p.back() == '/') p.pop_back();                      // filepath.Base strips trailing separators
    const size_t sl = p.find_last_of('/');
    std::string dir = sl == std::string::npos ? "." : (sl == 0 ? "/" : p.substr(0, sl));
    std::string base = sl == std::string::npos ? p : p.substr(sl + 1);
    const size_t dot = base.find_last_of('.');                                 // filepath.Ext: from the last '.' of the last element
    if (dot != std::string::npos) base = base.substr(0, dot);
    std::string joined = dir == "." ? "" : (dir == "/" ? "/" : dir + "/");     // filepath.Join cleans "./x" to "x"
    std::string r = joined + sfmt("%s-LUFS-%d-processed.flac", base.c_str(), lufs_value);
    if ((int)r.size() + 1 > cap) return -1;
    std::memcpy(out, r.c_str(), r.size() + 1);
    return (int)r.size();
}

static int process_audio_impl(jt_ctx *h, const jt_host_config *base, int frame_samples, jt_progress_fn cb, void *user, jt_process_result *out, bool ticks)
{
    if (!h || !base || !out) return JT_E_INVAL;
    if (base->adeclick_enabled && (base->adeclick_method_s < 0 || base->adeclick_method_s > 4)) { h->err = "adeclick_method_s must be 0..4"; return JT_E_INVAL; }
    TickSink tk; tk.cb = cb; tk.user = user; tk.ticks = ticks && cb; tk.duration = h->sr > 0 ? (double)h->n / h->sr : 0.0;
    // calculateFrameLevel of the frames of a stage output, for the ticks of Passes 2-4
    auto stage_levels = [&](int stage, std::vector<double> *lv) {
        lv->clear();
        if (!tk.ticks) return;
        int64_t nf = 0;
        if (jt_output_frame_levels(h, stage, 4096, nullptr, 0, &nf) != JT_OK || nf <= 0) return;
        lv->resize((size_t)nf);
        if (jt_output_frame_levels(h, stage, 4096, lv->data(), nf, &nf) != JT_OK) lv->clear();
    };
    std::memset(out, 0, sizeof(*out));
    auto emit = [&](int pass, const char *name, double progress, double duration, const jt_measurements *ms, const jt_host_config *cfg,
                    const jt_adaptive_diag *dg, const jt_limiter_decision *lim) {
        if (!cb) return;
        jt_progress_update u; std::memset(&u, 0, sizeof(u));
        u.pass = pass; u.pass_name = name; u.progress = progress; u.duration = duration; u.measurements = ms; u.config = cfg; u.diag = dg;
        if (lim) { u.has_limiter = 1; u.limiter_enabled = lim->needed; u.limiter_ceiling = lim->ceiling_db; }
        cb(user, &u);
    };
    int rc; jt_host_config eff;
    emit(1, "Analysing", 0.0, 0.0, nullptr, nullptr, nullptr, nullptr);
    if ((rc = analyse_core(h, base, frame_samples, out, &eff, true, &tk)) != JT_OK) return rc;
    jt_measurements &m = out->input;
    const double dur = m.duration_s;
    emit(1, "Analysing", 1.0, dur, &m, nullptr, nullptr, nullptr);
    emit(2, "Processing", 0.0, dur, &m, &eff, &out->diag, nullptr);
    // Pass 2
    jt_host_filter_spec(&eff, 2, out->pass2_spec, (int)sizeof(out->pass2_spec));
    jt_filter_params fp; jt_host_filter_params(&eff, &fp);
    // MeasureOutputRegions' two ranges are known from Pass 1: announce them so that Pass 2 / Pass 4 measure them in their own tails
    double reg_st[2] = {0, 0}, reg_du[2] = {0, 0};
    if (m.has_noise_profile && m.noise_profile.duration_ns > 0) { reg_st[0] = secs_of(m.noise_profile.start_ns); reg_du[0] = secs_of(m.noise_profile.duration_ns); }
    if (m.has_speech_profile && m.speech_profile.region.duration_ns > 0) { reg_st[1] = secs_of(m.speech_profile.region.start_ns); reg_du[1] = secs_of(m.speech_profile.region.duration_ns); }
    const bool have_regions = reg_du[0] > 0 || reg_du[1] > 0;
    if (have_regions) jt_region_prefetch(h, 2, reg_st, reg_du);
    double t0 = now_ms();
    // planLimiterForLoudnorm needs only Pass 2's integrated loudness and true peak (normalise.go:373-561): hand jt_pass2 the planner so
    // that a plan with the limiter prefix starts Pass 3's measurement while the rest of Pass 2's analysis is still running.  The values
    // go through the same "%.3f" print quantisation as below, so jt_pass3 is asked for exactly the plan that was started.
    struct PlanCtx { const jt_host_config *eff; } pctx{&eff};
    if (eff.loudnorm_enabled)
        jt_pass3_plan_hook(h, [](void *u, double integrated, double tp_lin, jt_limiter_plan *plan) -> int {
            const jt_host_config *e = static_cast<PlanCtx *>(u)->eff;
            jt_limiter_decision dec;
            jt_host_plan_limiter(qfmt("%.3f", integrated), linearRatioToDB(qfmt("%.3f", tp_lin)), e, &dec, plan);
            return JT_OK;
        }, &pctx);
    rc = jt_pass2(h, &fp, &out->filtered);
    jt_pass3_plan_hook(h, nullptr, nullptr);
    if (rc != JT_OK) return rc;
    out->pass_ms[1] = h->timers.pass2_ms;
    out->stage_ms[4] = now_ms() - t0; t0 = now_ms();
    std::vector<double> lv2;
    stage_levels(2, &lv2);
    if (tk.ticks && !lv2.empty()) {
        // every 100th INPUT frame; Level = the filtered frame the sink had delivered by then (processor.go:320-338)
        const int fsz = frame_samples > 0 ? frame_samples : h->dec_frame_samples;
        const int64_t nfr_in = (frame_samples == 0 && !h->dec_frame_lens.empty()) ? (int64_t)h->dec_frame_lens.size() : (h->n + fsz - 1) / fsz;
        const double est = ((double)h->n / h->sr) * h->sr / 4096.0;
        for (int64_t fc = 100; fc <= nfr_in && est > 0; fc += 100) {
            const size_t oi = (size_t)std::min<int64_t>((int64_t)lv2.size() - 1, fc * (int64_t)lv2.size() / std::max<int64_t>(1, nfr_in));
            tk.tick(2, "Processing", std::min(1.0, (double)fc / est), lv2[oi], &m);
        }
    }
    emit(2, "Processing", 1.0, dur, &m, nullptr, nullptr, nullptr);
    auto measure_regions = [&](int stage, jt_region_sample *rt, jt_region_sample *sp) {
        if (!have_regions) return;
        jt_region_sample o[2];
        if (jt_region_measure_pair(h, stage, reg_st, reg_du, o) == JT_OK) { if (reg_du[0] > 0) *rt = o[0]; if (reg_du[1] > 0) *sp = o[1]; }
    };
    measure_regions(2, &out->filtered_room_tone, &out->filtered_speech);
    out->stage_ms[5] = now_ms() - t0; t0 = now_ms();
    // Pass 3/4 (ApplyNormalisation, normalise.go:806-922)
    if (!eff.loudnorm_enabled) return JT_OK;
    const double outI = qfmt("%.3f", out->filtered.r128.integrated);
    const double outTP = linearRatioToDB(qfmt("%.3f", out->filtered.r128.true_peak));
    jt_limiter_plan plan; jt_host_plan_limiter(outI, outTP, &eff, &out->limiter, &plan);
    jt_loudnorm_stats m3;
    out->stage_ms[6] = now_ms() - t0; t0 = now_ms();
    emit(3, "Measuring", 0.0, dur, nullptr, nullptr, nullptr, nullptr);
    if ((rc = jt_pass3(h, &plan, eff.target_i, eff.target_tp, eff.target_lra, &m3)) != JT_OK) return rc;
    out->pass_ms[2] = h->timers.pass3_ms;
    out->stage_ms[7] = now_ms() - t0; t0 = now_ms();
    if (tk.ticks && !lv2.empty()) {
        // Pass 3 reads the Pass-2 output frame by frame (normalise.go:285-301): progress by samples, capped at 0.99
        const int64_t total = h->m_p2;
        for (int64_t fc = 100; fc <= (int64_t)lv2.size(); fc += 100)
            tk.tick(3, "Measuring", std::min(0.99, (double)std::min<int64_t>(fc * 4096, total) / (double)total), lv2[(size_t)fc - 1]);
    }
    m3.input_i = qfmt("%.2f", m3.input_i); m3.input_tp = qfmt("%.2f", m3.input_tp);
    m3.input_lra = qfmt("%.2f", m3.input_lra); m3.input_thresh = qfmt("%.2f", m3.input_thresh);
    out->measure = m3;
    if ((std::isinf(m3.input_i) && m3.input_i < 0) || m3.input_i < -70.0) {
        h->err = sfmt("cannot normalise silent audio (measured %.1f LUFS)", m3.input_i);
        return JT_E_SILENT;
    }
    double effI, offs; int lin;
    jt_host_calculate_linear_mode_target(m3.input_i, m3.input_tp, eff.target_i,
                                         jt_host_loudnorm_internal_target_tp(eff.target_i, m3.input_tp, m3.input_i), &effI, &offs, &lin);
    const double offset = effI - m3.input_i;
    out->effective_target_i = effI; out->offset = offset; out->linear_possible = lin;
    jt_host_config effcfg = eff; effcfg.target_i = effI;
    jt_loudnorm_apply ap;
    if (jt_host_pass4_spec(&effcfg, &m3, offset, &out->limiter, h->out_rate, nullptr, out->pass4_spec, (int)sizeof(out->pass4_spec), &ap) < 0) { h->err = "adeclick_method_s must be 0..4"; return JT_E_INVAL; }
    emit(3, "Measuring", 1.0, dur, nullptr, nullptr, nullptr, nullptr);
    emit(4, "Normalising", 0.0, dur, nullptr, nullptr, nullptr, &out->limiter);
    if (have_regions) jt_region_prefetch(h, 4, reg_st, reg_du);
    if ((rc = jt_pass4(h, &plan, &ap, &out->final_, &out->loudnorm)) != JT_OK) return rc;
    if (tk.ticks) {
        std::vector<double> lv4;
        stage_levels(4, &lv4);
        const int64_t total = h->m_p4;
        for (int64_t fc = 100; fc <= (int64_t)lv4.size(); fc += 100)
            tk.tick(4, "Normalising", std::min(0.99, (double)std::min<int64_t>(fc * 4096, total) / (double)total), lv4[(size_t)fc - 1]);
    }
    emit(4, "Normalising", 1.0, dur, nullptr, nullptr, nullptr, nullptr);
    out->pass_ms[3] = h->timers.pass4_ms;
    out->stage_ms[8] = now_ms() - t0; t0 = now_ms();
    measure_regions(4, &out->final_room_tone, &out->final_speech);
    out->stage_ms[9] = now_ms() - t0;
    out->has_region_samples = 1;
    out->output_lufs = qfmt("%.3f", out->final_.r128.integrated);
    out->output_tp_db = linearRatioToDB(qfmt("%.3f", out->final_.r128.true_peak));
    out->within_target = std::fabs(out->output_lufs - effI) <= 0.5 ? 1 : 0;
    return JT_OK;
}

extern "C" int jt_process_audio_cb(jt_ctx *h, const jt_host_config *base, int frame_samples, jt_progress_fn cb, void *user, jt_process_result *out)
{
    return process_audio_impl(h, base, frame_samples, cb, user, out, false);
}
extern "C" int jt_process_audio_ticks(jt_ctx *h, const jt_host_config *base, int frame_samples, jt_progress_fn cb, void *user, jt_process_result *out)
{
    return process_audio_impl(h, base, frame_samples, cb, user, out, true);
}
extern "C" int jt_process_audio(jt_ctx *h, const jt_host_config *base, int frame_samples, jt_process_result *out)
{
    return process_audio_impl(h, base, frame_samples, nullptr, nullptr, out, false);
}

extern "C" int64_t jt_host_last_intervals(const jt_ctx *h, jt_interval *out, int64_t cap)
{
    if (!h) return 0;
    const int64_t n = (int64_t)h->last_intervals.size();
    if (out) for (int64_t i = 0; i < n && i < cap; ++i) out[i] = h->last_intervals[(size_t)i];
    return n;
}

extern "C" int64_t jt_host_sizeof(int which)
{
    switch (which) {
    case 0: return sizeof(jt_interval);
    case 1: return sizeof(jt_measurements);
    case 2: return sizeof(jt_host_config);
    case 3: return sizeof(jt_process_result);
    case 4: return sizeof(jt_speech_candidate);
    case 5: return sizeof(jt_noise_profile);
    case 6: return sizeof(jt_limiter_decision);
    case 7: return sizeof(jt_adaptive_diag);
    case 8: return sizeof(jt_filter_params);
    case 9: return sizeof(jt_loudnorm_apply);
    case 10: return sizeof(jt_analysis);
    case 11: return sizeof(jt_region_sample);
    case 12: return sizeof(jt_flac_info);
    case 13: return sizeof(jt_audio_meta);
    case 14: return sizeof(jt_file_result);
    case 15: return sizeof(jt_timers);
    }
    return -1;
}

// ---- granular VAD helpers, exported so the parity tests can mirror the reference's own table tests
// (analyser_vad_test.go) one function at a time
extern "C" int jt_host_vad_detect(const jt_interval *ivp, int64_t n, double seed, jt_measurements *m)
{
    if (!m || (n > 0 && !ivp)) return JT_E_INVAL;
    std::memset(m, 0, sizeof(*m));
    std::vector<jt_interval> iv(ivp, ivp + n);
    detect_vad(iv, seed, m);
    return JT_OK;
}
extern "C" void jt_host_vad_split(const jt_interval *ivp, int64_t n, double seed, double *otsu_raw, double *split, double *floor_, double *margin, int *tol)
{
    std::vector<jt_interval> iv(ivp, ivp + n);
    Hist h = buildLevelHistogram(iv, 1.0);
    std::vector<double> levels = vadLevels(iv);
    double o = otsuSplit(h), sp = clampSplit(o, seed, percentileOfSorted(levels, 75));
    if (otsu_raw) *otsu_raw = o;
    if (split) *split = sp;
    if (floor_) *floor_ = percentileFloor(levels, seed);
    if (margin) *margin = hysteresisMargin(h, sp);
    if (tol) { std::vector<bool> f(iv.size()); for (size_t i = 0; i < iv.size(); ++i) f[i] = isSpeechInterval(iv[i], sp); *tol = gapToleranceIntervals(f, kHop); }
}
extern "C" int jt_host_vad_speech_runs(const jt_interval *ivp, int64_t n, double split, double margin, int tol, jt_region *out, int cap)
{
    std::vector<jt_interval> iv(ivp, ivp + n);
    std::vector<jt_region> r = buildSpeechRuns(iv, split, margin, tol, kHop);
    for (int i = 0; i < (int)r.size() && i < cap; ++i) out[i] = r[i];
    return (int)r.size();
}
extern "C" int jt_host_vad_gap_tolerance(const int *flags, int64_t n)
{
    std::vector<bool> f((size_t)n); for (int64_t i = 0; i < n; ++i) f[(size_t)i] = flags[i] != 0;
    return gapToleranceIntervals(f, kHop);
}
extern "C" void jt_host_vad_gate_stats(const jt_interval *ivp, int64_t n, double split, const jt_region *region,
                                       double *voiced_low, double *noise_high, double *separation)
{
    std::vector<jt_interval> iv(ivp, ivp + n);
    std::vector<double> voiced, noise;
    for (auto &s : iv) { double l = level_of(s); if (isFlooredLevel(l)) continue; if (l < split) noise.push_back(l); }
    if (region) { std::vector<jt_interval> r = getIntervalsInRange(iv, region->start_ns, region->end_ns);
                  for (auto &s : r) if (isSpeechInterval(s, split)) voiced.push_back(level_of(s)); }
    std::sort(voiced.begin(), voiced.end()); std::sort(noise.begin(), noise.end());
    double v = percentileOfSorted(voiced, 10.0), nh = percentileOfSorted(noise, 95.0);
    *voiced_low = v; *noise_high = nh; *separation = v - nh;
}
extern "C" int jt_host_vad_noise_seed(const jt_interval *ivp, int64_t n, double *nf, double *thr)
{
    std::vector<jt_interval> iv(ivp, ivp + n);
    double a = 0, b = 0; bool ok = estimateNoiseFloorAndThreshold(iv, &a, &b);
    *nf = a; *thr = b; return ok ? 1 : 0;
}
extern "C" int jt_host_vad_pick_low_cluster(const jt_interval *ivp, int64_t n, double split, jt_region *out)
{
    std::vector<jt_interval> iv(ivp, ivp + n);
    return pickLowClusterRegion(iv, split, kHop, out) ? 1 : 0;
}
extern "C" double jt_host_vad_floored_fraction(const jt_interval *ivp, int64_t n)
{
    double counted = 0, fl = 0;
    for (int64_t i = 0; i < n; ++i) { double l = ivp[i].momentary_lufs; counted++; if (std::isnan(l) || l <= vadLevelFloorDB) fl++; }
    return counted == 0 ? 0 : fl / counted;
}

// ---- speech election steps (analyser_candidates_speech.go / _shared.go), exported for the reference's table tests
extern "C" double jt_host_score_speech_candidate(double rms_level, int64_t duration_ns, double noise_floor_db, double level_var)
{
    jt_speech_candidate m; std::memset(&m, 0, sizeof m);
    m.sample.rms_level = rms_level; m.region.duration_ns = duration_ns;
    return scoreSpeechCandidateGrounded(m, noise_floor_db, level_var);
}
extern "C" double jt_host_level_variance(const jt_interval *ivp, int64_t n, int axis)
{
    if (n <= 0 || !ivp) return 0.0;
    std::vector<jt_interval> r(ivp, ivp + n);
    if (axis == 0) return levelVariance(r);
    // axisRMS (analyser_vad.go:55-56,79): the same population variance over the per-interval RMS level
    double s = 0; for (auto &x : r) s += x.rms_level; const double mean = s / (double)n;
    double v = 0; for (auto &x : r) { const double d = x.rms_level - mean; v += d * d; }
    return v / (double)n;
}
extern "C" int jt_host_find_best_speech_region(const jt_region *regions, int n_regions, const jt_interval *ivp, int64_t n_iv,
                                               int has_noise_profile, double noise_floor_db, jt_region *best, jt_speech_candidate *cands, int cap)
{
    std::vector<jt_region> rg(regions, regions + (n_regions > 0 ? n_regions : 0));
    std::vector<jt_interval> iv(ivp, ivp + (n_iv > 0 ? n_iv : 0));
    BestResult r = findBestSpeechRegion(rg, iv, has_noise_profile != 0, noise_floor_db);
    for (int i = 0; i < (int)r.cands.size() && i < cap; ++i) cands[i] = r.cands[(size_t)i];
    if (!r.has) return -1;
    if (best) *best = r.best;
    return (int)r.cands.size();
}
// getIntervalsInRange (analyser_candidates_shared.go): intervals with start <= timestamp < end; returns the count, copies at most cap
extern "C" int64_t jt_host_intervals_in_range(const jt_interval *ivp, int64_t n, int64_t start_ns, int64_t end_ns, jt_interval *out, int64_t cap)
{
    std::vector<jt_interval> iv(ivp, ivp + (n > 0 ? n : 0));
    std::vector<jt_interval> r = getIntervalsInRange(iv, start_ns, end_ns);
    for (size_t i = 0; i < r.size() && (int64_t)i < cap; ++i) out[i] = r[i];
    return (int64_t)r.size();
}
// scoreIntervalWindow (room-tone refinement: mean RMS level, 0 for an empty window) and scoreSpeechIntervalWindow (golden speech window)
extern "C" double jt_host_score_interval_window(const jt_interval *iv, int64_t n)
{
    if (n <= 0) return 0.0;
    double s = 0; for (int64_t i = 0; i < n; ++i) s += iv[i].rms_level;
    return s / (double)n;
}
extern "C" double jt_host_score_speech_interval_window(const jt_interval *iv, int64_t n)
{
    if (n <= 0) return 0.0;
    ScoreSoA a; const size_t nc = (size_t)n;
    a.kurt.resize(nc); a.flat.resize(nc); a.cent.resize(nc); a.rms.resize(nc); a.roll.resize(nc); a.flux.resize(nc);
    for (size_t i = 0; i < nc; ++i) { a.kurt[i] = iv[i].spectral.kurtosis; a.flat[i] = iv[i].spectral.flatness; a.cent[i] = iv[i].spectral.centroid;
                                      a.rms[i] = iv[i].rms_level; a.roll[i] = iv[i].spectral.rolloff; a.flux[i] = iv[i].spectral.flux; }
    return scoreSpeechWindowSoA(a, 0, nc);
}
// measureSpeechCandidateFromIntervals: 0 when no interval lies in the region (the reference returns nil), 1 otherwise
extern "C" int jt_host_measure_speech_candidate(const jt_region *region, const jt_interval *ivp, int64_t n, jt_speech_candidate *out)
{
    std::vector<jt_interval> iv(ivp, ivp + (n > 0 ? n : 0));
    return measureSpeechCandidate(*region, iv, out) ? 1 : 0;
}
// refineToGoldenSpeechSubregion: the 60 s window of a longer region that scores best (first wins ties); *out = the candidate itself
// when it is not longer than 60 s, has fewer than 30 s of intervals, or none.  Returns 1 when a sub-region was chosen, 0 when unchanged.
extern "C" int jt_host_refine_golden_speech(const jt_region *cand, const jt_interval *ivp, int64_t n, jt_region *out)
{
    std::vector<jt_interval> iv(ivp, ivp + (n > 0 ? n : 0));
    Refined r = refineToSubregion(cand->start_ns, cand->end_ns, cand->duration_ns, iv, 60 * kSecond, 30 * kSecond, true);
    out->start_ns = r.start; out->end_ns = r.end; out->duration_ns = r.duration;
    return r.ok ? 1 : 0;
}
// calculateFrameLevel (encoder.go:235-257): the VU level the progress callback carries -- 20 log10(rms) of one s16 frame, clamped
// to [-70, 0]; an empty or all-zero frame reads -70.
extern "C" double jt_host_frame_level_s16(const int16_t *pcm, int n)
{
    if (!pcm || n <= 0) return -70.0;
    double sum = 0.0;
    for (int i = 0; i < n; ++i) { const double v = (double)pcm[i] / 32768.0; sum += v * v; }
    const double rms = std::sqrt(sum / (double)n);
    if (rms < 0.00001) return -70.0;                       // "Floor near-silence before the log" (encoder.go:246-249)
    double db = 20.0 * std::log10(rms);
    if (db < -70.0) db = -70.0;
    if (db > 0.0) db = 0.0;
    return db;
}

// ---------------------------------------------------------------- file in, file out (processor.go:78-330)
// fault injection for the no-residue tests (the reference injects the same three failures through its package-level seams
// processorCreateSiblingTempPath / processorRename and a failing encoder: processor_test.go:552-627, normalise_test.go:573-820)
static std::atomic<int> g_fault_create_temp{0}, g_fault_write{0}, g_fault_rename{0};
extern "C" void jt_host_test_inject_fault(int create_temp, int write, int rename_)
{
    g_fault_create_temp.store(create_temp); g_fault_write.store(write); g_fault_rename.store(rename_);
}
// A file's job in two halves.  file_front: read, decode, four passes, encode -- everything that needs the handle.  file_tail: the
// STREAMINFO MD5 (one dependent chain: 64 ms of a host core per ten minutes of audio), the sibling temp file, the rename -- nothing
// that needs the GPU or the handle's device buffers, only the two pinned buffers of the handle's current I/O set.  jt_process_file
// runs both on the calling thread; a handle pool gives the tail to a finisher thread and starts the handle's next file in the other
// I/O set (the MD5 used to hold the handle: 64 of a ten-minute file's 80 ms).
namespace {
double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// The output's temp file, created and its blocks reserved WHILE THE PASSES RUN (files whose output will be 32 MB or more): reserving
// 150 MB of page cache is 15 ms of one thread's zero-filling, three quarters of what writing an hour's FLAC took, and the host has 40 ms
// with nothing to do.  The size is an estimate (0.48 of the s16 PCM; FLAC of speech: 0.42); the tail reserves what is missing and cuts
// off what is left over.  Same sibling ".processing-*.tmp.flac" name, same no-residue rule: discard() on every path that does not publish.
struct PreTemp {
    std::thread th; int fd = -1; std::string path; size_t reserved = 0; bool started = false;
    uint8_t *map = nullptr;      // the reserved blocks mapped AND populated (MAP_POPULATE: one batched fault-in instead of 37 000 traps in the copy)
    void start(const std::string &input_path, size_t est) {
        started = true;
        th = std::thread([this, input_path, est] {
            const size_t sl = input_path.find_last_of('/');
            std::string tmpl = (sl == std::string::npos ? std::string() : input_path.substr(0, sl + 1)) + ".processing-XXXXXX.tmp.flac";
            std::vector<char> tmp(tmpl.begin(), tmpl.end()); tmp.push_back(0);
            const int f = mkstemps(tmp.data(), 9);
            if (f < 0) return;
            path = tmp.data(); fd = f;
            if (posix_fallocate(f, 0, (off_t)est) == 0) {
                reserved = est;
                void *m = mmap(nullptr, est, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, f, 0);
                if (m != MAP_FAILED) map = static_cast<uint8_t *>(m);
            }
        });
    }
    void wait() { if (th.joinable()) th.join(); }
    void unmap() { if (map) { munmap(map, reserved); map = nullptr; } }
    void discard() { wait(); unmap(); if (fd >= 0) { close(fd); unlink(path.c_str()); fd = -1; } }
    ~PreTemp() { discard(); }
};
struct FileTail {
    std::shared_ptr<PreTemp> pre;
    uint8_t *image = nullptr; int64_t len = 0;            // the finished .flac image (pinned, STREAMINFO signature still zero when pcm != null)
    const int16_t *pcm = nullptr; size_t n_pcm = 0;       // what the signature covers (pinned); null = no signature wanted / already in place
    std::string input_path, final_path;
    const std::atomic<int> *cancelled = nullptr;          // the handle's flag when the tail runs inside the handle's own job
};
// One file's bytes in or out.  A single read() / write() of an hour's FLAC (150 MB) is a single core's memcpy into or out of the page
// cache -- 14 ms to read and 30 ms to write on tmpfs, a third of the job's wall time with the passes at 41 ms --
// so images above 32 MB move as up to eight slices, each on a thread of its own: pread at its own offset; for the output a shared
// mapping of the (pre-sized) file that the threads fill with memcpy -- parallel pwrite()s serialise on the inode's write lock (measured:
// 51 ms instead of 30), page faults of a mapping do not.  (A pool's ten-minute files are below the threshold: its workers and
// finishers are the parallelism there.)  Any failure of the mapped path falls back to plain write().
bool io_slices(int fd, uint8_t *buf, size_t len, bool writing, size_t reserved = 0)
{
    auto run = [&](size_t off, size_t end) {
        while (off < end) {
            const size_t want = std::min<size_t>(end - off, (size_t)1 << 30);
            const ssize_t r = writing ? pwrite(fd, buf + off, want, (off_t)off) : pread(fd, buf + off, want, (off_t)off);
            if (r < 0) { if (errno == EINTR) continue; return false; }
            if (r == 0) return false;                          // (the file shrank under us / the device is full)
            off += (size_t)r;
        }
        return true;
    };
    const size_t slice_min = (size_t)32 << 20;
    const size_t parts = std::min<size_t>(8, len / slice_min);
    if (parts < 2) return run(0, len);
    const size_t step = ((len + parts - 1) / parts + 4095) & ~(size_t)4095;
    uint8_t *map = nullptr;
    if (writing) {
        // posix_fallocate reserves the blocks up front: a full device fails here, not as SIGBUS inside the copy.  (Sized sparse and
        // populated per slice with madvise(MADV_POPULATE_WRITE), tmpfs allocated the pages under contention: 38-48 ms against 20.)
        if (reserved < len && posix_fallocate(fd, (off_t)reserved, (off_t)(len - reserved)) != 0) return run(0, len);
        void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) return run(0, len);
        map = static_cast<uint8_t *>(m);
    }
    std::atomic<int> bad{0};
    auto part = [&](size_t k) {
        const size_t lo = std::min(len, k * step), hi = std::min(len, (k + 1) * step);
        if (hi <= lo) return;
        if (map) memcpy(map + lo, buf + lo, hi - lo);
        else if (!run(lo, hi)) bad.store(1);
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < parts; ++k) th.emplace_back(part, k);
    part(0);
    for (auto &t : th) t.join();
    if (map && munmap(map, len) != 0) bad.store(1);
    return bad.load() == 0;
}
int file_front(jt_ctx *h, const char *input_path, const jt_host_config *base, int frame_samples, int flac_flags,
               jt_progress_fn cb, void *user, jt_process_result *out, FileTail *tail, double io_ms[4])
{
    auto fail = [&](int code, const std::string &msg) { h->err = msg; return code; };
    double t = wall_ms();
    // the job starts here: a jt_cancel() from now on (during the read, the decode, any pass, the encode, the write) ends it
    if (!h->hold_cancel) h->cancelled.store(0);            // (inside a jt_begin_job bracket the caller has cleared it already)
    struct Hold { jt_ctx *h; bool was; Hold(jt_ctx *c) : h(c), was(c->hold_cancel) { h->hold_cancel = true; } ~Hold() { h->hold_cancel = was; } } hold(h);
    // the file image goes straight into pinned memory (the I/O set's FLAC staging arena: nothing of this set is in flight),
    // so the upload runs at the full PCIe rate instead of through HIP's pageable bounce buffers
    uint8_t *image = nullptr; size_t image_len = 0;
    {
        const int fd = open(input_path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) return fail(JT_E_INVAL, std::string("failed to open input file: ") + input_path);
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return fail(JT_E_INVAL, std::string("failed to open input file: ") + input_path); }
        if (st.st_size <= 0) { close(fd); return fail(JT_E_INVAL, std::string("empty input file: ") + input_path); }
        image_len = (size_t)st.st_size;
        try {
            if (hipSetDevice(h->device) != hipSuccess) throw JtError{JT_E_HIP, "hipSetDevice failed"};
            h->pin_flac().begin(image_len + 64);
            image = h->pin_flac().take<uint8_t>(image_len);
        } catch (const JtError &e) { close(fd); return fail(e.code, e.msg); }
        const bool ok = io_slices(fd, image, image_len, false);
        close(fd);
        if (!ok) return fail(JT_E_INVAL, std::string("short read on input file: ") + input_path);
    }
    if (io_ms) io_ms[0] = wall_ms() - t;
    t = wall_ms();
    if (h->cancelled.load()) return fail(JT_E_CANCELLED, "cancelled");
    jt_audio_meta meta;
    int rc = jt_load_audio(h, image, (int64_t)image_len, &meta);
    if (rc != JT_OK) return rc;
    if (io_ms) io_ms[1] = wall_ms() - t;
    {
        const size_t est = (size_t)((double)meta.frames * 44100.0 / (double)std::max(1, meta.sample_rate) * 2.0 * 0.48) + ((size_t)1 << 20);
        if (est >= ((size_t)jt_early_temp_min_kb().load() << 10) && !g_fault_create_temp) { tail->pre = std::make_shared<PreTemp>(); tail->pre->start(input_path, est); }
    }
    rc = process_audio_impl(h, base, frame_samples, cb, user, out, (flac_flags & JT_FILE_PROGRESS_TICKS) != 0);
    if (rc != JT_OK) { tail->pre.reset(); return rc; }
    t = wall_ms();
    const uint8_t *data = nullptr; int64_t len = 0; jt_flac_info info;
    rc = jt_flac_encode_file(h, 4, (flac_flags & JT_FLAC_MD5) ? JT_FLAC_MD5_DEFER : 0, &data, &len, &info);
    if (rc != JT_OK) { tail->pre.reset(); return rc; }
    if (io_ms) io_ms[2] = wall_ms() - t;
    char path[4096];
    if (jt_host_output_path(input_path, jt_host_lufs_filename_value(out->output_lufs), path, (int)sizeof path) < 0)
        { tail->pre.reset(); return fail(JT_E_INVAL, "output path too long"); }
    tail->image = const_cast<uint8_t *>(data); tail->len = len;          // (the handle's own pinned arena)
    tail->pcm = h->flac_deferred.pcm; tail->n_pcm = h->flac_deferred.n;
    tail->input_path = input_path; tail->final_path = path;
    return JT_OK;
}
// one page cache for every reader of the file?  (tmpfs, ext*, xfs, btrfs, f2fs, overlayfs on those, ramfs, zfs); anything else -- NFS, CIFS /
// SMB, FUSE, 9p, Lustre, Ceph ... -- is treated as shared with other hosts
bool jt_fd_is_local(int fd)
{
    struct statfs sf;
    if (fstatfs(fd, &sf) != 0) return false;
    switch ((unsigned long)sf.f_type) {
    case 0x01021994ul /* tmpfs */: case 0xEF53ul /* ext2/3/4 */: case 0x58465342ul /* xfs */: case 0x9123683Eul /* btrfs */:
    case 0xF2F52010ul /* f2fs */: case 0x794C7630ul /* overlayfs */: case 0x858458F6ul /* ramfs */: case 0x2FC12FC1ul /* zfs */:
        return true;
    default: return false;
    }
}
// munmap off the caller's thread: one lazily started reaper per process, joined by its static destructor (process exit / dlclose)
void jt_unmap_later(uint8_t *m, size_t n)
{
    struct Reaper {
        std::mutex mu; std::condition_variable cv; std::vector<std::pair<uint8_t *, size_t>> q; bool stop = false, started = false; std::thread th;
        void run() {
            std::unique_lock<std::mutex> l(mu);
            for (;;) {
                cv.wait(l, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                auto jobs = std::move(q); q.clear();
                l.unlock();
                for (auto &j : jobs) munmap(j.first, j.second);
                l.lock();
            }
        }
        ~Reaper() { { std::lock_guard<std::mutex> l(mu); stop = true; cv.notify_all(); } if (th.joinable()) th.join(); }
    };
    static Reaper R;
    {
        std::lock_guard<std::mutex> l(R.mu);
        if (!R.started) {
            try { R.th = std::thread([] { R.run(); }); R.started = true; }
            catch (const std::system_error &) { /* no thread to be had: unmap here */ }
        }
        if (R.started) { R.q.emplace_back(m, n); R.cv.notify_all(); return; }
    }
    munmap(m, n);
}
// returns JT_OK or the code + message jt_process_file reports; ms[0] = the MD5, ms[1] = write + rename
int file_tail(const FileTail &t, std::string *err, double ms[2])
{
    double t0 = wall_ms();
    if (t.pcm) {
        // STREAMINFO: "fLaC" + 4 header bytes + 18 bytes of block sizes / rate / length, then the 16-byte signature (RFC 9639 section 8.2)
        uint8_t md5[16];
        jt_md5(t.pcm, sizeof(int16_t) * t.n_pcm, md5);
        memcpy(t.image + 26, md5, 16);
    }
    if (ms) ms[0] = wall_ms() - t0;
    t0 = wall_ms();
    // createSiblingTempPath(inputPath, "processing") + publishOutput (file_write.go:13-53, processor.go:126-135,206-213): the
    // image goes to a hidden ".processing-*.tmp.flac" beside the input and is renamed over the final name only when complete;
    // on any failure the temp file is removed and the final name is never touched.
    std::string dir(t.final_path);
    const size_t sl = dir.find_last_of('/');
    dir = sl == std::string::npos ? "" : dir.substr(0, sl + 1);
    std::string tmpl = dir + ".processing-XXXXXX.tmp.flac";
    std::vector<char> tmp(tmpl.begin(), tmpl.end()); tmp.push_back(0);
    int fd = -1; size_t reserved = 0;
    if (t.pre && !g_fault_create_temp) {
        // the temp file the front created while the passes ran (its blocks reserved already): taken over here
        t.pre->wait();
        if (t.pre->fd >= 0) { fd = t.pre->fd; reserved = t.pre->reserved; tmp.assign(t.pre->path.begin(), t.pre->path.end()); tmp.push_back(0); t.pre->fd = -1; }
    } else if (t.pre) t.pre->discard();
    bool copied = false;
    if (fd >= 0 && t.pre && t.pre->map && (size_t)t.len <= reserved && !g_fault_write) {
        // the mapping is there and populated: eight threads copy, nothing else is left of the write
        const size_t len = (size_t)t.len, parts = std::max<size_t>(1, std::min<size_t>(8, len / ((size_t)16 << 20)));
        const size_t step = ((len + parts - 1) / parts + 4095) & ~(size_t)4095;
        std::vector<std::thread> th;
        auto part = [&](size_t k) { const size_t lo = std::min(len, k * step), hi = std::min(len, (k + 1) * step); if (hi > lo) memcpy(t.pre->map + lo, t.image + lo, hi - lo); };
        for (size_t k = 1; k < parts; ++k) th.emplace_back(part, k);
        part(0);
        for (auto &x : th) x.join();
        copied = true;
    }
    const double tw1 = wall_ms();
    // tearing the mapping down (43 000 dirty shared pages: 9-11 ms) is nobody's business but this process's: it happens on a thread of
    // its own AFTER the file has been published (beside the truncate it fought it for the address-space lock: 8-12 ms instead of 4)
    // (the library's reaper thread, not a detached one: nothing of this library runs after it is unloaded, and a thread that cannot be
    //  created means an inline munmap instead of std::terminate in a destructor: ADVICE r5)
    struct Unmap { uint8_t *m = nullptr; size_t n = 0; ~Unmap() { if (m) jt_unmap_later(m, n); } } unmap_later;
    if (t.pre && t.pre->map && copied) { unmap_later.m = t.pre->map; unmap_later.n = t.pre->reserved; t.pre->map = nullptr; }
    else if (t.pre) t.pre->unmap();
    const double tw2 = wall_ms();
    if (fd < 0) fd = g_fault_create_temp ? -1 : mkstemps(tmp.data(), 9);
    if (fd < 0) { *err = std::string("failed to create temporary output next to ") + t.input_path; return JT_E_INVAL; }
    // A shared mapping's dirty pages reach a NETWORK filesystem at msync / munmap, not at close(): published by rename with the mapping
    // still live, another client could read stale or zero blocks.  Local filesystems (one page cache) keep the fast path; anywhere else
    // the written range is flushed synchronously before the file gets its final name.
    if (copied && unmap_later.m && !jt_fd_is_local(fd) && msync(unmap_later.m, (size_t)t.len, MS_SYNC) != 0) copied = false;
    bool okw = !g_fault_write && (copied || io_slices(fd, t.image, (size_t)t.len, true, reserved));
    if (okw && reserved > (size_t)t.len && ftruncate(fd, (off_t)t.len) != 0) okw = false;      // (the estimate's surplus)
    if (fchmod(fd, 0644) != 0) { /* keep mkstemp's 0600: not fatal */ }
    if (close(fd) != 0) okw = false;
    if (!okw) { unlink(tmp.data()); *err = std::string("failed to write temporary output ") + tmp.data(); return JT_E_INVAL; }
    if (t.cancelled && t.cancelled->load()) { unlink(tmp.data()); *err = "cancelled"; return JT_E_CANCELLED; }
    if (g_fault_rename || rename(tmp.data(), t.final_path.c_str()) != 0) { unlink(tmp.data()); *err = std::string("failed to publish output to ") + t.final_path; return JT_E_INVAL; }
    if (ms) ms[1] = wall_ms() - t0;
    if (jt_host_timing().load(std::memory_order_relaxed)) fprintf(stderr, "file tail: wait + copy %.2f ms, unmap %.2f, truncate + close + rename %.2f\n", tw1 - t0, tw2 - tw1, wall_ms() - tw2);
    return JT_OK;
}
} // namespace

extern "C" int jt_process_file(jt_ctx *h, const char *input_path, const jt_host_config *base, int frame_samples, int flac_flags,
                               jt_progress_fn cb, void *user, jt_process_result *out, char *output_path, int cap, double io_ms[4])
{
    if (!h || !input_path || !out) return JT_E_INVAL;
    FileTail tail;
    int rc = file_front(h, input_path, base, frame_samples, flac_flags, cb, user, out, &tail, io_ms);
    if (rc != JT_OK) return rc;
    tail.cancelled = &h->cancelled;
    double ms[2] = {0, 0}; std::string err;
    rc = file_tail(tail, &err, ms);
    if (rc != JT_OK) { h->err = err; return rc; }
    if (io_ms) { io_ms[2] += ms[0]; io_ms[3] = ms[1]; }       // (the signature counts as part of the encode, as before)
    if (output_path && cap > 0) { strncpy(output_path, tail.final_path.c_str(), (size_t)cap - 1); output_path[cap - 1] = 0; }
    return JT_OK;
}

// ---------------------------------------------------------------- several files in flight on one or more GPUs (pool.go:122-228)
// One shared queue, (devices x in_flight) workers: a worker is a host thread with a handle of its own on its device and takes
// the next unclaimed file when it finishes one -- the reference's runBoundedPool semantics (a bounded number of ProcessAudio calls
// at a time, a file's failure never stops the others), with the GPU as the bounded resource.  Files are handed out longest first
// when their sizes are known (the on-disk size is the proxy), so the tail of the batch is made of short files.
// A pool owns its handles: they are opened once (in parallel: hipMalloc serialises a worker's first-file allocations anyway) and serve
// any number of batches -- what a long-running host wants (the Go shim pools handles the same way), and what keeps 0.1 s of start-up
// out of every sub-second batch.  A device that cannot be opened gets no worker; its worker tries the devices nobody serves yet.

// ---------------------------------------------------------------- host topology
// A pool's worker thread, its finisher jobs and the pinned, file-sized I/O sets they touch first belong on the NUMA node the GPU hangs
// off: on a two-socket host (2 x EPYC 9575F on the GPU box, profiles/r05_gpubox_probe.txt) eight pools' threads otherwise land wherever
// the scheduler puts them and half of the pinned traffic crosses the socket link.  The node comes from the device's PCI address
// (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<addr>/numa_node -> /sys/devices/system/node/node<k>/cpulist), intersected with the CPUs
// the process is allowed (a cpuset / taskset is respected); a host without that information (one node, a VM: numa_node = -1) is left alone.
// Pinned memory follows the threads: hipHostMalloc's pages are faulted in by the thread that calls it (first touch, default policy).
// Option pool_numa (process-wide, default on).
namespace {
struct DeviceCpus { bool ok = false; int node = -1; cpu_set_t set; };
const DeviceCpus &device_cpus(int device)
{
    static std::mutex m; static std::map<int, DeviceCpus> cache;
    std::lock_guard<std::mutex> l(m);
    auto it = cache.find(device);
    if (it != cache.end()) return it->second;
    DeviceCpus d; CPU_ZERO(&d.set);
    char bus[64] = {0};
    int count = 0;
    // (a device index the runtime does not know is not asked about: a failed HIP call leaves its error as the calling thread's "last
    //  error", and the next launch check on that thread would report it)
    const bool known = hipGetDeviceCount(&count) == hipSuccess && device >= 0 && device < count;
    if (!known) (void)hipGetLastError();
    if (known && hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); bus[0] = 0; }
    if (bus[0]) {
        for (char *c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
        char path[256]; snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
        int node = -1;
        if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
        if (node >= 0) {
            snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
            if (FILE *f = fopen(path, "r")) {
                char buf[4096] = {0};
                if (fgets(buf, sizeof buf, f)) {
                    int n = 0;
                    for (char *p = buf; *p && *p != '\n';) {                    // "0-63,128-191"
                        char *e; long a = strtol(p, &e, 10); if (e == p) break; long b = a;
                        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
                        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, &d.set); ++n; }
                        p = *e == ',' ? e + 1 : e;
                    }
                    cpu_set_t allowed;
                    if (n > 0 && sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
                        CPU_AND(&d.set, &d.set, &allowed);
                        d.ok = CPU_COUNT(&d.set) > 0; d.node = node;
                    }
                }
                fclose(f);
            }
        }
    }
    return cache.emplace(device, d).first->second;
}
// binds the calling thread for the lifetime of the object (a pool's batch runs its first worker on the CALLER's thread: restored afterwards)
struct NodeBinding {
    bool bound = false; cpu_set_t old;
    explicit NodeBinding(int device) {
        if (!jt_pool_numa().load()) return;
        const DeviceCpus &d = device_cpus(device);
        if (!d.ok || pthread_getaffinity_np(pthread_self(), sizeof old, &old) != 0) return;
        bound = pthread_setaffinity_np(pthread_self(), sizeof d.set, &d.set) == 0;
    }
    ~NodeBinding() { if (bound) (void)pthread_setaffinity_np(pthread_self(), sizeof old, &old); }
};
} // namespace
extern "C" int jt_host_device_numa_node(int device, int *n_cpus)
{
    const DeviceCpus &d = device_cpus(device);
    if (n_cpus) *n_cpus = d.ok ? CPU_COUNT(&d.set) : 0;
    return d.ok ? d.node : -1;
}

struct jt_handle_pool {
    std::vector<int> devices;
    int in_flight = 1;
    int n_streams = 0, open_flags = 0;      // what the handles were opened with (a handle that fails is replaced by one of the same kind)
    struct Worker { jt_ctx *h; int device; };
    std::vector<Worker> workers;
    int open_rc = JT_OK, open_dev = -1;
    std::mutex busy;                        // one batch at a time
    // finisher threads: the tails of finished files (STREAMINFO MD5, temp file, rename; file_tail) -- as many as there are handles
    std::mutex fm; std::condition_variable fcv; std::deque<std::function<void()>> fq; bool fstop = false;
    std::vector<std::thread> finishers;
    // where the last batch's time went, summed over its files (ms): waiting for a free I/O set, read, decode, passes, encode on the
    // handle's thread; waiting for a finisher, MD5, write + rename on a finisher thread
    std::mutex sm; double stats[JT_POOL_STATS] = {0};
    void add_stats(const double *v) { std::lock_guard<std::mutex> l(sm); for (int i = 0; i < JT_POOL_STATS; ++i) stats[i] += v[i]; }
    void finisher_loop() {
        for (;;) {
            std::function<void()> job;
            { std::unique_lock<std::mutex> l(fm); fcv.wait(l, [&] { return fstop || !fq.empty(); }); if (fq.empty()) return; job = std::move(fq.front()); fq.pop_front(); }
            job();
        }
    }
    void finish_async(std::function<void()> job) { { std::lock_guard<std::mutex> l(fm); fq.push_back(std::move(job)); } fcv.notify_one(); }
};

extern "C" int jt_handle_pool_open(const int *devices, int n_devices, int in_flight_per_device, int max_workers, jt_handle_pool **out)
{
    if (!out) return JT_E_INVAL;
    *out = nullptr;
    if (!devices || n_devices < 1 || in_flight_per_device < 1) return JT_E_INVAL;
    jt_handle_pool *P = new jt_handle_pool();
    P->devices.assign(devices, devices + n_devices); P->in_flight = in_flight_per_device;
    const int want = n_devices * in_flight_per_device;
    const int workers = max_workers > 0 ? std::min(want, max_workers) : want;
    std::atomic<int> spare{workers};                 // devices[workers..] have no worker of their own (fewer workers than devices)
    std::atomic<int> open_rc{JT_OK}, open_dev{-1};
    std::mutex m;
    // three or more handles on a device: one stream each and sleeping host waits (jt_open_ex; include/jtgpu.h says why)
    const bool shared = in_flight_per_device >= 3;
    const int ns = shared ? jt_pool_streams().load() : 0, of = shared && jt_pool_blocking().load() ? JT_OPEN_BLOCKING_SYNC : 0;
    auto open_one = [&](int device) {
        NodeBinding nb(device);                  // (the handle's first pinned arenas are allocated here)
        jt_ctx *h = nullptr;
        int orc = jt_open_ex(device, ns, of, &h);
        while ((orc != JT_OK || !h) && workers < n_devices) {
            const int k = spare.fetch_add(1);
            if (k >= n_devices) break;
            int ok = JT_OK;
            if (open_rc.compare_exchange_strong(ok, orc != JT_OK ? orc : JT_E_NOGPU)) open_dev.store(device);
            device = devices[k]; h = nullptr; orc = jt_open_ex(device, ns, of, &h);
        }
        if (orc != JT_OK || !h) {
            int ok = JT_OK;
            if (open_rc.compare_exchange_strong(ok, orc != JT_OK ? orc : JT_E_NOGPU)) open_dev.store(device);
            return;
        }
        std::lock_guard<std::mutex> g(m);
        P->workers.push_back({h, device});
    };
    std::vector<std::thread> th;
    // worker w serves device w % n_devices: with fewer workers than slots every device still gets one before any gets two
    for (int w = 1; w < workers; ++w) th.emplace_back(open_one, devices[w % n_devices]);
    open_one(devices[0]);
    for (auto &t : th) t.join();
    P->open_rc = open_rc.load(); P->open_dev = open_dev.load(); P->n_streams = ns; P->open_flags = of;
    for (size_t i = 0; i < P->workers.size(); ++i) P->finishers.emplace_back([P] { P->finisher_loop(); });
    // (deterministic worker order: by device, so that a report of "which worker served what" reads the same run to run)
    std::stable_sort(P->workers.begin(), P->workers.end(), [](const jt_handle_pool::Worker &a, const jt_handle_pool::Worker &b) { return a.device < b.device; });
    *out = P;
    return JT_OK;
}

extern "C" int jt_handle_pool_workers(const jt_handle_pool *P, int *devices_out, int cap)
{
    if (!P) return JT_E_INVAL;
    for (int i = 0; i < (int)P->workers.size() && i < cap && devices_out; ++i) devices_out[i] = P->workers[(size_t)i].device;
    return (int)P->workers.size();
}

extern "C" int jt_handle_pool_stats(jt_handle_pool *P, double *out, int cap)
{
    if (!P || !out || cap < 0) return JT_E_INVAL;
    std::lock_guard<std::mutex> l(P->sm);
    for (int i = 0; i < JT_POOL_STATS && i < cap; ++i) out[i] = P->stats[i];
    return JT_POOL_STATS;
}

extern "C" void jt_handle_pool_close(jt_handle_pool *P)
{
    if (!P) return;
    {
        std::lock_guard<std::mutex> g(P->busy);          // a batch still running ends first; nobody may start one on a pool being closed
        { std::lock_guard<std::mutex> l(P->fm); P->fstop = true; }
        P->fcv.notify_all();
        for (auto &t : P->finishers) t.join();
        std::vector<std::thread> th;
        for (auto &w : P->workers) th.emplace_back([h = w.h] { jt_close(h); });
        for (auto &t : th) t.join();
        P->workers.clear();
    }
    delete P;
}

extern "C" int jt_handle_pool_process_files(jt_handle_pool *P, const char *const *paths, int n_files, const jt_host_config *base, int frame_samples, int flac_flags,
                                     jt_file_result *results, int *device_of_file)
{
    if (!P || !paths || !results || n_files < 0) return JT_E_INVAL;
    std::lock_guard<std::mutex> busy(P->busy);
    { std::lock_guard<std::mutex> l(P->sm); for (double &v : P->stats) v = 0; }
    for (int i = 0; i < n_files; ++i) { std::memset(&results[i], 0, sizeof results[i]); results[i].rc = JT_E_STATE; if (device_of_file) device_of_file[i] = -1; }
    if (n_files == 0) return 0;
    if (P->workers.empty()) {
        // no worker has a device: every file fails with the reason the first jt_open gave (pool.go:122-153 - a result per file)
        const int rc = P->open_rc != JT_OK ? P->open_rc : JT_E_NOGPU;
        for (int i = 0; i < n_files; ++i) {
            results[i].rc = rc;
            snprintf(results[i].error, sizeof results[i].error, "jt_open(%d) failed (%d): no worker has a device", P->open_dev, rc);
        }
        return n_files;
    }
    // longest first (LPT): order[] is the queue
    std::vector<int> order((size_t)n_files); std::vector<long long> size((size_t)n_files, 0);
    for (int i = 0; i < n_files; ++i) { order[(size_t)i] = i; struct stat st; if (paths[i] && stat(paths[i], &st) == 0) size[(size_t)i] = (long long)st.st_size; }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return size[(size_t)a] > size[(size_t)b]; });
    std::atomic<int> next{0}, failed{0};
    // a handle that ends a file with anything but a verdict about that FILE (bad input, unsupported, silent, cancelled) is not
    // trusted with another: it is closed and replaced, and a worker whose replacement cannot be opened stops taking files (a handle
    // that fails in microseconds would otherwise claim most of the queue)
    auto file_verdict = [](int rc) { return rc == JT_OK || rc == JT_E_CANCELLED || rc == JT_E_INVAL || rc == JT_E_UNSUPPORTED || rc == JT_E_SILENT; };
    auto worker = [&](jt_handle_pool::Worker *W) {
        NodeBinding nb(W->device);               // this thread, and through first touch the I/O sets it pins, on the GPU's NUMA node
        // two I/O sets per handle: a file's tail runs on a finisher thread from set k while the handle fills set 1 - k
        struct Gate { std::mutex m; std::condition_variable cv; bool busy = false;
                      void wait_free() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !busy; }); }
                      // (notify UNDER the lock: the gates live on this worker's stack, and a worker that saw !busy between an unlock and
                      //  a later notify could leave the frame while the finisher was still about to touch the condition variable: ADVICE r5)
                      void release() { std::lock_guard<std::mutex> l(m); busy = false; cv.notify_all(); } } gate[2];
        // set by a finisher job that met a HIP failure (its event query / device selection): the handle's stream may have faulted, and
        // the worker only learns of front errors by itself -- it checks this before it takes the next file, drains both gates and
        // replaces the handle exactly as it does for a front failure
        auto suspect = std::make_shared<std::atomic<int>>(0);
        int set = 0;
        while (W->h) {
            if (suspect->exchange(0)) {
                gate[0].wait_free(); gate[1].wait_free();
                jt_close(W->h); W->h = nullptr;
                if (jt_open_ex(W->device, P->n_streams, P->open_flags, &W->h) != JT_OK) { W->h = nullptr; break; }
            }
            const int q = next.fetch_add(1);
            if (q >= n_files) break;
            const int i = order[(size_t)q];
            jt_file_result &r = results[i];
            if (device_of_file) device_of_file[i] = W->device;
            const double t0 = wall_ms();
            if (!paths[i]) { r.rc = JT_E_INVAL; snprintf(r.error, sizeof r.error, "null path"); r.wall_ms = wall_ms() - t0; failed.fetch_add(1); continue; }
            gate[set].wait_free();
            double st[JT_POOL_STATS] = {0}, io[4] = {0, 0, 0, 0};
            st[0] = wall_ms() - t0;
            W->h->io_set = set;
            auto tail = std::make_shared<FileTail>();
            // One finisher job per file.  With the MD5 wanted it starts INSIDE Pass 4, the moment the delivered s16 is complete
            // (jt_ctx::p4_output_hook): the PCM goes to the host and the signature's dependent chain (64 ms per ten minutes) runs
            // beside the output analysis, the encode and the image's download; the job then waits for the front's verdict and writes.
            struct Early { std::mutex m; std::condition_variable cv; int front = 0;          // 0 pending, 1 ok, -1 failed
                           bool started = false; const int16_t *pcm = nullptr; size_t n = 0; hipEvent_t ev = nullptr; int device = 0;
                           std::vector<double> st; double t_front = 0; };
            auto early = std::make_shared<Early>();
            Gate *g = &gate[set];
            const int fdev = W->device;
            auto finish = [tail, early, &r, &failed, g, t0, P, suspect, fdev] {
                NodeBinding nb(fdev);            // (a finisher thread serves whichever device's file comes next)
                std::string err; double ms[2] = {0, 0}; uint8_t md5[16]; bool md5_ok = true;
                if (early->started) {
                    const double m0 = wall_ms();
                    hipError_t e = hipSetDevice(early->device);
                    while (e == hipSuccess) {
                        e = hipEventQuery(early->ev);
                        if (e != hipErrorNotReady) break;
                        (void)hipGetLastError(); e = hipSuccess;
                        timespec ts{0, 50000}; nanosleep(&ts, nullptr);
                    }
                    md5_ok = e == hipSuccess;
                    if (md5_ok) jt_md5(early->pcm, sizeof(int16_t) * early->n, md5);
                    ms[0] = wall_ms() - m0;
                    std::unique_lock<std::mutex> l(early->m);
                    early->cv.wait(l, [&] { return early->front != 0; });
                    if (early->front < 0) { l.unlock(); g->release(); return; }          // (the worker has reported the front's error)
                }
                std::vector<double> sv(early->st); sv[5] = wall_ms() - early->t_front;
                int rc = JT_OK;
                if (early->started && !md5_ok) { rc = JT_E_HIP; err = "the PCM's copy for the STREAMINFO signature failed"; suspect->store(1); }
                else {
                    if (early->started) memcpy(tail->image + 26, md5, 16);                 // (tail->pcm is null: file_tail does not hash again)
                    double tms[2] = {0, 0};
                    rc = file_tail(*tail, &err, tms);
                    ms[0] += tms[0]; ms[1] = tms[1];
                }
                sv[6] = ms[0]; sv[7] = ms[1]; P->add_stats(sv.data());
                r.rc = rc;
                if (rc != JT_OK) { snprintf(r.error, sizeof r.error, "%s", err.c_str()); failed.fetch_add(1); }
                else { strncpy(r.output_path, tail->final_path.c_str(), sizeof r.output_path - 1); r.output_path[sizeof r.output_path - 1] = 0; }
                r.wall_ms = wall_ms() - t0;
                g->release();
            };
            if (flac_flags & JT_FLAC_MD5) {
                const int dev = W->device;
                W->h->p4_output_hook = [early, g, P, finish, set, dev](jt_ctx *hh) {
                    const size_t n = (size_t)hh->m_p4;
                    hh->pin_pcm().begin(sizeof(int16_t) * n + 64);
                    int16_t *pcm = hh->pin_pcm().take<int16_t>(n);
                    JT_HIP(hipMemcpyAsync(pcm, hh->s16_p4.p, sizeof(int16_t) * n, hipMemcpyDeviceToHost, hh->stream));
                    JT_HIP(hipEventRecord(hh->ev_pcm[set], hh->stream));
                    hh->pcm_early.pcm = pcm; hh->pcm_early.n = n;
                    early->pcm = pcm; early->n = n; early->ev = hh->ev_pcm[set]; early->device = dev; early->started = true;
                    { std::lock_guard<std::mutex> l(g->m); g->busy = true; }
                    P->finish_async(finish);
                };
            }
            r.rc = file_front(W->h, paths[i], base, frame_samples, flac_flags, nullptr, nullptr, &r.result, tail.get(), io);
            W->h->p4_output_hook = nullptr;
            const double t_front = wall_ms();
            st[1] = io[0]; st[2] = io[1]; st[4] = io[2]; st[3] = (t_front - t0) - st[0] - io[0] - io[1] - io[2]; st[8] = 1;
            early->st.assign(st, st + JT_POOL_STATS); early->t_front = t_front;
            if (r.rc != JT_OK) {
                snprintf(r.error, sizeof r.error, "%s", jt_last_error(W->h));
                r.wall_ms = wall_ms() - t0; failed.fetch_add(1);
                P->add_stats(st);
                if (early->started) { { std::lock_guard<std::mutex> l(early->m); early->front = -1; } early->cv.notify_all(); }
                if (!file_verdict(r.rc)) {
                    gate[0].wait_free(); gate[1].wait_free();          // (finisher jobs still read the old handle's pinned buffers and events)
                    jt_close(W->h); W->h = nullptr;
                    if (jt_open_ex(W->device, P->n_streams, P->open_flags, &W->h) != JT_OK) W->h = nullptr;
                }
                continue;
            }
            if (early->started) {
                // (an MD5 job that Pass 4 started but whose PCM the encoder did not adopt -- it cannot happen on this path -- would leave
                //  the signature to file_tail as well: tail->pcm is only null when the encoder found pcm_early)
                { std::lock_guard<std::mutex> l(early->m); early->front = 1; }
                early->cv.notify_all();
            } else {
                { std::lock_guard<std::mutex> l(g->m); g->busy = true; }
                P->finish_async(finish);
            }
            set ^= 1;
        }
        gate[0].wait_free(); gate[1].wait_free();
        if (W->h && suspect->exchange(0)) {                   // (a failure reported by the batch's last finisher jobs: the next batch gets a fresh handle)
            jt_close(W->h); W->h = nullptr;
            if (jt_open_ex(W->device, P->n_streams, P->open_flags, &W->h) != JT_OK) W->h = nullptr;
        }
    };
    const size_t nw = std::min<size_t>(P->workers.size(), (size_t)n_files);
    // with fewer files than workers, take one worker per device before a second on any (the workers are sorted by device)
    std::vector<size_t> pick;
    if (nw < P->workers.size()) {
        std::vector<char> used(P->workers.size(), 0);
        while (pick.size() < nw) {
            int last = INT32_MIN; bool any = false;
            for (size_t w = 0; w < P->workers.size() && pick.size() < nw; ++w)
                if (!used[w] && P->workers[w].device != last) { used[w] = 1; pick.push_back(w); last = P->workers[w].device; any = true; }
            if (!any) break;
        }
    } else for (size_t w = 0; w < nw; ++w) pick.push_back(w);
    std::vector<std::thread> th;
    for (size_t k = 1; k < pick.size(); ++k) th.emplace_back(worker, &P->workers[pick[k]]);
    worker(&P->workers[pick[0]]);
    for (auto &t : th) t.join();
    // workers whose handle could not be replaced leave the pool; files nobody took (every worker gone) keep JT_E_STATE and count as failed
    P->workers.erase(std::remove_if(P->workers.begin(), P->workers.end(), [](const jt_handle_pool::Worker &w) { return !w.h; }), P->workers.end());
    for (int i = 0; i < n_files; ++i) if (results[i].rc == JT_E_STATE && !results[i].error[0]) { snprintf(results[i].error, sizeof results[i].error, "no worker left to take the file"); failed.fetch_add(1); }
    return failed.load();
}

extern "C" int jt_process_files_multi(const int *devices, int n_devices, const char *const *paths, int n_files, int in_flight_per_device,
                                      const jt_host_config *base, int frame_samples, int flac_flags, jt_file_result *results, int *device_of_file)
{
    if (!devices || n_devices < 1 || !paths || !results || n_files < 0 || in_flight_per_device < 1) return JT_E_INVAL;
    if (n_files == 0) return 0;
    // a pool for the length of the call (a caller with more than one batch keeps its own: jt_handle_pool_open)
    jt_handle_pool *P = nullptr;
    const int rc = jt_handle_pool_open(devices, n_devices, in_flight_per_device, n_files, &P);
    if (rc != JT_OK) return rc;
    const int failed = jt_handle_pool_process_files(P, paths, n_files, base, frame_samples, flac_flags, results, device_of_file);
    jt_handle_pool_close(P);
    return failed;
}

extern "C" int jt_process_files(int device, const char *const *paths, int n_files, int in_flight, const jt_host_config *base,
                                int frame_samples, int flac_flags, jt_file_result *results)
{
    if (in_flight < 1) return JT_E_INVAL;
    return jt_process_files_multi(&device, 1, paths, n_files, in_flight, base, frame_samples, flac_flags, results, nullptr);
}
