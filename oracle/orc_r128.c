/*
 * orc_r128.c — ORACLE (test infrastructure only; see jt_oracle.h).
 * libswresample polyphase resampler (libswresample/resample.c), the ebur128 filter
 * (libavfilter/f_ebur128.c) and the loudnorm input measurement (libavfilter/ebur128.c as
 * used by af_loudnorm.c), restated from FFmpeg 8.1.
 * Reference call sites: filters.go:626,684-689 (ebur128=metadata=1:peak=sample+true:dualmono=true),
 * filters.go:706-710 (aformat 44100/s16), normalise.go:256-264 (loudnorm measure),
 * analyser_output.go:18 (region ebur128 without dualmono).
 * parity unpinned at the FFmpeg boundary; pinned to BS.1770/EBU 3341/3342 KATs in tests/.
 */
#include "jt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define FFMIN(a,b) ((a) < (b) ? (a) : (b))
#define FFMAX(a,b) ((a) > (b) ? (a) : (b))

/* =============================================================== swresample */
static double bessel_i0(double x)
{
    /* I0 by power series to convergence (resample.c uses an equivalent-accuracy evaluation) */
    double v = 1, lastv = 0, t = 1;
    int i;
    x = x * x / 4;
    for (i = 1; v != lastv; i++) {
        lastv = v;
        t *= x / (i * i);
        v += t;
    }
    return v;
}

typedef struct {
    int phase_count, filter_length, center;
    int64_t den, step;             /* output m has filter index floor(m*step/den): sample index / phase_count, phase % phase_count */
    double *bank;                  /* [phase_count][filter_length] */
} SwrPlan;

static int64_t gcd64(int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; }

/* swri_resample_init() + build_filter(): kaiser (beta 9), filter_size 32, phase_shift 10,
 * cutoff 0.97, exact_rational. */
static void swr_plan(SwrPlan *p, int in_rate, int out_rate)
{
    const int filter_size = 32;
    const double cutoff = 0.97, kaiser_beta = 9;
    double factor = FFMIN(out_rate * cutoff / in_rate, 1.0);
    int phase_count = 1 << 10;
    int filter_length = FFMAX((int)ceil(filter_size / factor), 1);
    if (filter_length > 1)
        filter_length = (filter_length + 1) & ~1;
    int64_t g = gcd64(out_rate, in_rate);
    int64_t pc_exact = out_rate / g;
    if (pc_exact <= phase_count)
        phase_count = (int)pc_exact;
    p->phase_count = phase_count;
    p->filter_length = filter_length;
    p->center = (filter_length - 1) / 2;
    /* index advances by in_rate*phase_count/out_rate per output sample: an integer when the exact phase count fits the 1024-entry
     * bank.  Otherwise (22050 or 11025 -> 192000: 1280 / 2560 exact phases) resample.c keeps 1024 phases and steps
     * index += dst_incr_div; frac += dst_incr_mod; carry at src_incr  (swri_resample, linear = 0: the bank row is NOT interpolated),
     * i.e. index_m = floor(m * in_rate * 1024 / out_rate). */
    if (pc_exact <= 1024) { p->step = (int64_t)in_rate / g * (phase_count / (out_rate / g)); p->den = 1; }
    else { int64_t gg = gcd64((int64_t)in_rate * phase_count, out_rate); p->step = (int64_t)in_rate * phase_count / gg; p->den = out_rate / gg; }
    p->bank = malloc(sizeof(double) * (size_t)phase_count * filter_length);
    const int tap_count = filter_length;
    const int center = p->center;
    for (int ph = 0; ph < phase_count; ph++) {
        double norm = 0;
        double *tab = p->bank + (size_t)ph * tap_count;
        for (int i = 0; i < tap_count; i++) {
            double x = M_PI * ((double)(i - center) - (double)ph / phase_count) * factor;
            double y, w;
            if (x == 0) y = 1.0;
            else        y = sin(x) / x;
            w = 2.0 * x / (factor * tap_count * M_PI);
            y *= bessel_i0(kaiser_beta * sqrt(FFMAX(1 - w * w, 0)));
            tab[i] = y;
            norm += y;
        }
        for (int i = 0; i < tap_count; i++)
            tab[i] = tab[i] / norm;          /* scale = 1 for FLTP/DBLP */
    }
}

static inline double swr_in_f64(const double *in, int64_t n, int64_t k, int flush)
{
    if (k < 0) k = -k;                           /* invert_initial_buffer(): in[-j] = in[j] */
    if (k >= n) {
        if (!flush) return 0.0;
        k = 2 * n - 1 - k;                       /* resample_flush(): in[n+j] = in[n-1-j] */
        if (k < 0) return 0.0;
    }
    return in[k];
}

int64_t orc_swr_resample_f64(const double *in, int64_t n, int in_rate, int out_rate,
                             double *out, int64_t out_cap, int flush)
{
    SwrPlan p;
    swr_plan(&p, in_rate, out_rate);
    int64_t m = 0;
    for (;; m++) {
        int64_t idx = m * p.step / p.den;
        int64_t si = idx / p.phase_count;
        int ph = (int)(idx % p.phase_count);
        if (flush) { if (si >= n) break; }
        else       { if (si - p.center + p.filter_length - 1 > n - 1) break; }
        if (m >= out_cap) break;
        const double *f = p.bank + (size_t)ph * p.filter_length;
        double val = 0;
        for (int i = 0; i < p.filter_length; i++)
            val += swr_in_f64(in, n, si - p.center + i, flush) * f[i];
        out[m] = val;
    }
    free(p.bank);
    return m;
}

int64_t orc_swr_resample_f32(const float *in, int64_t n, int in_rate, int out_rate,
                             float *out, int64_t out_cap, int flush)
{
    SwrPlan p;
    swr_plan(&p, in_rate, out_rate);
    int64_t m = 0;
    for (;; m++) {
        int64_t idx = m * p.step / p.den;
        int64_t si = idx / p.phase_count;
        int ph = (int)(idx % p.phase_count);
        if (flush) { if (si >= n) break; }
        else       { if (si - p.center + p.filter_length - 1 > n - 1) break; }
        if (m >= out_cap) break;
        const double *f = p.bank + (size_t)ph * p.filter_length;
        float val = 0;
        for (int i = 0; i < p.filter_length; i++) {
            int64_t k = si - p.center + i;
            float x;
            if (k < 0) k = -k;
            if (k >= n) { if (!flush) x = 0.f; else { k = 2 * n - 1 - k; x = k < 0 ? 0.f : in[k]; } }
            else x = in[k];
            val += x * (float)f[i];              /* FLTP: float taps, float accumulate */
        }
        out[m] = val;
    }
    free(p.bank);
    return m;
}

void orc_f64_to_s16(const double *in, int16_t *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        long v = lrint(in[i] * (1 << 15));
        if (v < -32768) v = -32768;
        if (v > 32767) v = 32767;
        out[i] = (int16_t)v;
    }
}

/* ============================================================ f_ebur128.c */
#define ABS_THRES    (-70)
#define ABS_UP_THRES 10
#define HIST_GRAIN   100
#define HIST_SIZE    ((ABS_UP_THRES - ABS_THRES) * HIST_GRAIN + 1)
#define LOUDNESS(energy) (-0.691 + 10 * log10(energy))
#define ENERGY(loudness) (pow(10., ((loudness) + 0.691) / 10.))
#define HIST_POS(power) (int)(((power) - ABS_THRES) * HIST_GRAIN)

typedef struct { double loudness, energy; unsigned count; } hist_entry;
typedef struct {
    double *cache; int cache_pos, cache_size; double sum; int filled;
    double rel_threshold, sum_kept_powers; int64_t nb_kept_powers;
    hist_entry *histogram;
} integrator;

static hist_entry *get_histogram(void)
{
    hist_entry *h = calloc(HIST_SIZE, sizeof(*h));
    for (int i = 0; i < HIST_SIZE; i++) {
        h[i].loudness = i / (double)HIST_GRAIN + ABS_THRES;
        h[i].energy = ENERGY(h[i].loudness);
    }
    return h;
}

static int clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static int gate_update(integrator *integ, double power, double loudness, int gate_thres)
{
    int ipower = clipi(HIST_POS(loudness), 0, HIST_SIZE - 1);
    double relative_threshold;
    integ->histogram[ipower].count++;
    integ->sum_kept_powers += power;
    integ->nb_kept_powers++;
    relative_threshold = integ->sum_kept_powers / integ->nb_kept_powers;
    if (!relative_threshold)
        relative_threshold = 1e-12;
    integ->rel_threshold = LOUDNESS(relative_threshold) + gate_thres;
    return clipi(HIST_POS(integ->rel_threshold), 0, HIST_SIZE - 1);
}

void orc_ebur128_kweight_coeffs(int sample_rate, double pre_b[3], double pre_a[3], double rlb_b[3], double rlb_a[3])
{
    double f0 = 1681.974450955533, G = 3.999843853973347, Q = 0.7071752369554196;
    double K = tan(M_PI * f0 / (double)sample_rate);
    double Vh = pow(10.0, G / 20.0);
    double Vb = pow(Vh, 0.4996667741545416);
    double a0 = 1.0 + K / Q + K * K;
    pre_b[0] = (Vh + Vb * K / Q + K * K) / a0;
    pre_b[1] = 2.0 * (K * K - Vh) / a0;
    pre_b[2] = (Vh - Vb * K / Q + K * K) / a0;
    pre_a[0] = 1.0;
    pre_a[1] = 2.0 * (K * K - 1.0) / a0;
    pre_a[2] = (1.0 - K / Q + K * K) / a0;
    f0 = 38.13547087602444; Q = 0.5003270373238773;
    K = tan(M_PI * f0 / (double)sample_rate);
    rlb_b[0] = 1.0; rlb_b[1] = -2.0; rlb_b[2] = 1.0;
    rlb_a[0] = 1.0;
    rlb_a[1] = 2.0 * (K * K - 1.0) / (1.0 + K / Q + K * K);
    rlb_a[2] = (1.0 - K / Q + K * K) / (1.0 + K / Q + K * K);
}

void orc_ebur128_mono(const double *in, int64_t n, int sample_rate, int dualmono, int true_peak,
                      orc_ebur128_out *o, double *m_series, double *s_series,
                      double *tp_series, double *sp_series, int64_t cap)
{
    double pre_b[3], pre_a[3], rlb_b[3], rlb_a[3];
    orc_ebur128_kweight_coeffs(sample_rate, pre_b, pre_a, rlb_b, rlb_a);
    const double pan_law = -3.01029995663978;
    integrator i400, i3000;
    memset(&i400, 0, sizeof(i400)); memset(&i3000, 0, sizeof(i3000));
    i400.cache_size = sample_rate * 4 / 10;  i3000.cache_size = sample_rate * 3;
    i400.cache = calloc(i400.cache_size, sizeof(double));
    i3000.cache = calloc(i3000.cache_size, sizeof(double));
    i400.histogram = get_histogram(); i3000.histogram = get_histogram();
    double x[3] = {0,0,0}, y[3] = {0,0,0}, z[3] = {0,0,0};
    double sample_peak = 0, tp = 0;
    double integrated_loudness = ABS_THRES, loudness_range = 0, lra_low = 0, lra_high = 0;
    double last_m = -HUGE_VAL, last_s = -HUGE_VAL;
    int sample_count = 0;
    int64_t nb = 0;
    const int blk = sample_rate / 10;

    /* true peak: swr to 192 kHz in DBLP, streaming (never flushed) */
    double *tp_up = NULL; int64_t tp_n = 0; int up = 1;
    SwrPlan plan; memset(&plan, 0, sizeof(plan));
    if (true_peak) {
        int64_t capu = (int64_t)ceil((double)n * 192000.0 / sample_rate) + 8;
        tp_up = malloc(sizeof(double) * capu);
        tp_n = orc_swr_resample_f64(in, n, sample_rate, 192000, tp_up, capu, 0);
        swr_plan(&plan, sample_rate, 192000);
        (void)up;
    }
    int64_t tp_cursor = 0;

    for (int64_t idx = 0; idx < n; idx++) {
        const int bin_id_400 = i400.cache_pos, bin_id_3000 = i3000.cache_pos;
        double bin;
        if (++i400.cache_pos == i400.cache_size)   { i400.filled = 1;  i400.cache_pos = 0; }
        if (++i3000.cache_pos == i3000.cache_size) { i3000.filled = 1; i3000.cache_pos = 0; }

        sample_peak = FFMAX(sample_peak, fabs(in[idx]));
        x[0] = in[idx];
        /* FILTER(y, x, pre) */
        y[2] = y[1]; y[1] = y[0];
        y[0] = x[0] * pre_b[0] + x[1] * pre_b[1] + x[2] * pre_b[2] - y[1] * pre_a[1] - y[2] * pre_a[2];
        x[2] = x[1]; x[1] = x[0];
        /* FILTER(z, y, rlb) */
        z[2] = z[1]; z[1] = z[0];
        z[0] = y[0] * rlb_b[0] + y[1] * rlb_b[1] + y[2] * rlb_b[2] - z[1] * rlb_a[1] - z[2] * rlb_a[2];
        bin = z[0] * z[0];
        i400.sum  = i400.sum  + bin - i400.cache[bin_id_400];
        i3000.sum = i3000.sum + bin - i3000.cache[bin_id_3000];
        i400.cache[bin_id_400] = bin;
        i3000.cache[bin_id_3000] = bin;

        if (++sample_count == blk) {
            double loudness_400, loudness_3000;
            double power_400 = 1e-12, power_3000 = 1e-12;
            sample_count = 0;
            if (i400.filled)  { power_400  += 1.0 * i400.sum;  power_400  /= i400.cache_size; }
            loudness_400 = LOUDNESS(power_400);
            if (i3000.filled) { power_3000 += 1.0 * i3000.sum; power_3000 /= i3000.cache_size; }
            loudness_3000 = LOUDNESS(power_3000);

            if (loudness_400 >= ABS_THRES) {
                double integrated_sum = 0.0;
                uint64_t nb_integrated = 0;
                int gate_hist_pos = gate_update(&i400, power_400, loudness_400, -10);
                for (int i = gate_hist_pos; i < HIST_SIZE; i++) {
                    const unsigned nb_v = i400.histogram[i].count;
                    nb_integrated += nb_v;
                    integrated_sum += nb_v * i400.histogram[i].energy;
                }
                if (nb_integrated) {
                    integrated_loudness = LOUDNESS(integrated_sum / nb_integrated);
                    if (dualmono) integrated_loudness -= pan_law;
                }
            }
            if (loudness_3000 >= ABS_THRES) {
                uint64_t nb_powers = 0;
                int gate_hist_pos = gate_update(&i3000, power_3000, loudness_3000, -20);
                for (int i = gate_hist_pos; i < HIST_SIZE; i++)
                    nb_powers += i3000.histogram[i].count;
                if (nb_powers) {
                    uint64_t nn, nb_pow;
                    nn = 0;
                    nb_pow = (uint64_t)(10 * nb_powers * 0.01 + 0.5);
                    for (int i = gate_hist_pos; i < HIST_SIZE; i++) {
                        nn += i3000.histogram[i].count;
                        if (nn >= nb_pow) { lra_low = i3000.histogram[i].loudness; break; }
                    }
                    nn = nb_powers;
                    nb_pow = (uint64_t)(95 * nb_powers * 0.01 + 0.5);
                    for (int i = HIST_SIZE - 1; i >= 0; i--) {
                        nn -= FFMIN(nn, i3000.histogram[i].count);
                        if (nn < nb_pow) { lra_high = i3000.histogram[i].loudness; break; }
                    }
                    loudness_range = lra_high - lra_low;
                }
            }
            if (dualmono) { loudness_400 -= pan_law; loudness_3000 -= pan_law; }
            last_m = loudness_400; last_s = loudness_3000;

            /* true peaks: ebur128 feeds swr one 100 ms frame at a time; outputs available so far
             * are those whose taps fit inside the samples pushed (streaming, no flush). */
            if (true_peak) {
                int64_t pushed = idx + 1;
                while (tp_cursor < tp_n) {
                    int64_t ix = tp_cursor * plan.step / plan.den;
                    int64_t si = ix / plan.phase_count;
                    if (si - plan.center + plan.filter_length - 1 > pushed - 1) break;
                    tp = FFMAX(tp, fabs(tp_up[tp_cursor]));
                    tp_cursor++;
                }
            }
            if (nb < cap) {
                if (m_series) m_series[nb] = loudness_400;
                if (s_series) s_series[nb] = loudness_3000;
                if (tp_series) tp_series[nb] = tp;
                if (sp_series) sp_series[nb] = sample_peak;
            }
            nb++;
        }
    }
    /* trailing partial frame: samples update peaks (sample peak per-sample, true peak per frame) */
    if (true_peak) {
        while (tp_cursor < tp_n) { tp = FFMAX(tp, fabs(tp_up[tp_cursor])); tp_cursor++; }
        free(tp_up); free(plan.bank);
    }
    o->integrated = integrated_loudness; o->lra = loudness_range;
    o->lra_low = lra_low; o->lra_high = lra_high;
    o->momentary_last = last_m; o->shortterm_last = last_s;
    o->sample_peak = sample_peak; o->true_peak = tp;
    o->target_threshold = i400.rel_threshold;
    o->nblocks = nb;
    free(i400.cache); free(i3000.cache); free(i400.histogram); free(i3000.histogram);
}

/* ======================================== libavfilter/ebur128.c (loudnorm) */
static double histogram_energies[1000];
static double histogram_energy_boundaries[1001];
static int hist_init_done = 0;
static void init_histogram(void)
{
    if (hist_init_done) return;
    histogram_energy_boundaries[0] = pow(10.0, (-70.0 + 0.691) / 10.0);
    for (int i = 0; i < 1000; ++i)
        histogram_energies[i] = pow(10.0, ((double)i / 10.0 - 69.95 + 0.691) / 10.0);
    for (int i = 1; i < 1001; ++i)
        histogram_energy_boundaries[i] = pow(10.0, ((double)i / 10.0 - 70.0 + 0.691) / 10.0);
    hist_init_done = 1;
}
static size_t find_histogram_index(double energy)
{
    size_t index_min = 0, index_max = 1000, index_mid;
    do {
        index_mid = (index_min + index_max) / 2;
        if (energy >= histogram_energy_boundaries[index_mid]) index_min = index_mid;
        else index_max = index_mid;
    } while (index_max - index_min != 1);
    return index_min;
}
static double energy_to_loudness(double energy) { return 10 * (log(energy) / log(10.0)) - 0.691; }

void orc_loudnorm_measure_mono(const double *in, int64_t n, int sample_rate, int dual_mono, orc_loudnorm_in *o)
{
    init_histogram();
    /* ebur128_init_filter(): 4th-order combined K-weighting, DF2 */
    double f0 = 1681.974450955533, G = 3.999843853973347, Q = 0.7071752369554196;
    double K = tan(M_PI * f0 / (double)sample_rate);
    double Vh = pow(10.0, G / 20.0), Vb = pow(Vh, 0.4996667741545416);
    double pb[3] = {0,0,0}, pa[3] = {1,0,0}, rb[3] = {1,-2,1}, ra[3] = {1,0,0};
    double a0 = 1.0 + K / Q + K * K;
    pb[0] = (Vh + Vb * K / Q + K * K) / a0;
    pb[1] = 2.0 * (K * K - Vh) / a0;
    pb[2] = (Vh - Vb * K / Q + K * K) / a0;
    pa[1] = 2.0 * (K * K - 1.0) / a0;
    pa[2] = (1.0 - K / Q + K * K) / a0;
    f0 = 38.13547087602444; Q = 0.5003270373238773;
    K = tan(M_PI * f0 / (double)sample_rate);
    ra[1] = 2.0 * (K * K - 1.0) / (1.0 + K / Q + K * K);
    ra[2] = (1.0 - K / Q + K * K) / (1.0 + K / Q + K * K);
    double b[5], a[5];
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

    const int64_t s100 = (sample_rate + 5) / 10;
    const int64_t ring = s100 * 30;                 /* MODE_S: 3 s of audio_data */
    double *audio = calloc(ring, sizeof(double));
    unsigned long *bhist = calloc(1000, sizeof(unsigned long));
    unsigned long *shist = calloc(1000, sizeof(unsigned long));
    double v[5] = {0,0,0,0,0};
    double peak = 0;
    int64_t audio_index = 0, needed = s100 * 4, have = 0, st_counter = 0;

    /* af_loudnorm.c: at EOF flush_frame() hands the last 2.9 s of the look-ahead buffer to filter_frame(), whose first statement is
     * ff_ebur128_add_frames_double(r128_in, ...): a stream of 3 s or more is metered with its last 556 800 samples (at 192 kHz) twice */
    const int64_t f3000 = (int64_t)llround(sample_rate * 3.0), fin = f3000 - (int64_t)llround(sample_rate * 0.1);
    const int64_t n_meter = n >= f3000 ? n + fin : n;
    for (int64_t ii = 0; ii < n_meter; ii++) {
        const int64_t i_src = ii < n ? ii : n - fin + (ii - n);
        const double x_in = in[i_src];
        if (fabs(x_in) > peak) peak = fabs(x_in);
        v[0] = x_in - a[1] * v[1] - a[2] * v[2] - a[3] * v[3] - a[4] * v[4];
        audio[audio_index] = b[0] * v[0] + b[1] * v[1] + b[2] * v[2] + b[3] * v[3] + b[4] * v[4];
        v[4] = v[3]; v[3] = v[2]; v[2] = v[1]; v[1] = v[0];
        /* libebur128 flushes denormals in v at the end of each add_frames call; values this small
         * never influence a block energy above the -70 LUFS gate */
        audio_index = (audio_index + 1) % ring;
        if (++have == needed) {
            /* ebur128_calc_gating_block(frames_per_block = s100*4) */
            double sum = 0.0;
            int64_t fpb = s100 * 4;
            if (audio_index < fpb) {
                for (int64_t k = 0; k < audio_index; k++) sum += audio[k] * audio[k];
                for (int64_t k = ring - (fpb - audio_index); k < ring; k++) sum += audio[k] * audio[k];
            } else {
                for (int64_t k = audio_index - fpb; k < audio_index; k++) sum += audio[k] * audio[k];
            }
            if (dual_mono) sum *= 2.0;
            sum /= (double)fpb;
            if (sum >= histogram_energy_boundaries[0])
                ++bhist[find_histogram_index(sum)];
            /* MODE_LRA short-term blocks */
            st_counter += needed;
            if (st_counter == s100 * 30) {
                double st = 0.0;
                for (int64_t k = 0; k < ring; k++) st += audio[k] * audio[k];
                if (dual_mono) st *= 2.0;
                st /= (double)ring;
                if (st >= histogram_energy_boundaries[0])
                    ++shist[find_histogram_index(st)];
                st_counter = s100 * 20;
            }
            needed = s100;
            have = 0;
        }
    }
    /* ff_ebur128_loudness_global + relative threshold */
    double rel = 0.0; long above = 0;
    for (int j = 0; j < 1000; ++j) { rel += bhist[j] * histogram_energies[j]; above += bhist[j]; }
    if (above) { rel /= (double)above; rel *= 0.1; }
    o->input_thresh = above ? energy_to_loudness(rel) : -70.0;
    if (!above) o->input_i = -HUGE_VAL;
    else {
        size_t start_index;
        double gated = 0.0; long cnt = 0;
        if (rel < histogram_energy_boundaries[0]) start_index = 0;
        else { start_index = find_histogram_index(rel); if (rel > histogram_energies[start_index]) ++start_index; }
        for (size_t j = start_index; j < 1000; ++j) { gated += bhist[j] * histogram_energies[j]; cnt += bhist[j]; }
        o->input_i = cnt ? energy_to_loudness(gated / cnt) : -HUGE_VAL;
    }
    /* ff_ebur128_loudness_range */
    {
        size_t stl_size = 0; double stl_power = 0.0;
        for (int j = 0; j < 1000; ++j) { stl_size += shist[j]; stl_power += shist[j] * histogram_energies[j]; }
        if (!stl_size) o->input_lra = 0.0;
        else {
            size_t index, j;
            double stl_integrated;
            stl_power /= stl_size;
            stl_integrated = 0.01 * stl_power;
            if (stl_integrated < histogram_energy_boundaries[0]) index = 0;
            else { index = find_histogram_index(stl_integrated); if (stl_integrated > histogram_energies[index]) ++index; }
            stl_size = 0;
            for (j = index; j < 1000; ++j) stl_size += shist[j];
            if (!stl_size) o->input_lra = 0.0;
            else {
                size_t pl = (size_t)((stl_size - 1) * 0.1 + 0.5), ph = (size_t)((stl_size - 1) * 0.95 + 0.5);
                double l_en, h_en;
                stl_size = 0; j = index;
                while (stl_size <= pl) stl_size += shist[j++];
                l_en = histogram_energies[j - 1];
                while (stl_size <= ph) stl_size += shist[j++];
                h_en = histogram_energies[j - 1];
                o->input_lra = energy_to_loudness(h_en) - energy_to_loudness(l_en);
            }
        }
    }
    o->input_tp = 20 * log10(peak);
    free(audio); free(bhist); free(shist);
}
