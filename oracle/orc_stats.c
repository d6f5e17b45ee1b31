/*
 * orc_stats.c — ORACLE (test infrastructure only; see jt_oracle.h).
 * astats (libavfilter/af_astats.c) and aspectralstats (libavfilter/af_aspectralstats.c)
 * restated from FFmpeg 8.1; formulas cross-checked against the in-repo statement
 * /root/reference/docs/Spectral-Metrics-Reference.md:9-56.
 * Reference call sites: filters.go:624-625; analyser_output.go:18; keys read at
 * analyser_metrics.go:432-474.
 */
#include "jt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define FFMIN(a,b) ((a) < (b) ? (a) : (b))
#define FFMAX(a,b) ((a) > (b) ? (a) : (b))
#define LINEAR_TO_DB(x) (log10(x) * 20)
#define HISTOGRAM_SIZE 8192
#define HISTOGRAM_MAX (HISTOGRAM_SIZE - 1)

/* libavutil FFSIGN(a) = ((a) > 0 ? 1 : -1): NaN and 0 both map to -1 */
static int fsign(double d) { return d > 0 ? 1 : -1; }

void orc_astats_mono(const double *in, int64_t n, int sample_rate, orc_astats_out *o)
{
    const double time_constant = 0.05;
    const double mult = exp((-1 / time_constant / sample_rate));
    const int tc_samples = (int)FFMAX(time_constant * sample_rate + .5, 1);
    double min = DBL_MAX, max = -DBL_MAX, nmin = DBL_MAX, nmax = -DBL_MAX;
    double min_non_zero = DBL_MAX, min_diff = DBL_MAX, max_diff = 0;
    double sigma_x = 0, sigma_x2 = 0, avg_sigma_x2 = 0;
    double min_sigma_x2 = DBL_MAX, max_sigma_x2 = 0, diff1_sum = 0, diff1_sum_x2 = 0;
    double last = NAN, last_non_zero = NAN;
    double min_run = 0, max_run = 0, min_runs = 0, max_runs = 0;
    uint64_t min_count = 0, max_count = 0, zero_runs = 0, nb_samples = 0;
    double abs_peak = 0; uint64_t abs_peak_count = 0;
    double noise_floor = NAN; uint64_t noise_floor_count = 0;
    uint64_t *ehistogram = calloc(HISTOGRAM_SIZE, sizeof(uint64_t));
    /* sliding-window local peak (calc_noise_floor): monotonic deque over |x| of the last tc_samples */
    int64_t *dq = malloc(sizeof(int64_t) * (size_t)(tc_samples + 1));
    int dq_head = 0, dq_len = 0;

    for (int64_t i = 0; i < n; i++) {
        const double d = in[i], nd = in[i];
        const double abs_d = fabs(d);
        int index;
        if (abs_peak < abs_d) { abs_peak = abs_d; abs_peak_count = 1; }
        else if (abs_peak == abs_d) abs_peak_count++;
        if (d < min) { min = d; nmin = nd; min_run = 1; min_runs = 0; min_count = 1; }
        else if (d == min) { min_count++; min_run = d == last ? min_run + 1 : 1; }
        else if (last == min) { min_runs += min_run * min_run; }
        if (d != 0 && fabs(d) < min_non_zero) min_non_zero = fabs(d);
        if (d > max) { max = d; nmax = nd; max_run = 1; max_runs = 0; max_count = 1; }
        else if (d == max) { max_count++; max_run = d == last ? max_run + 1 : 1; }
        else if (last == max) { max_runs += max_run * max_run; }
        if (d != 0) {
            zero_runs += fsign(d) != fsign(last_non_zero);
            last_non_zero = d;
        }
        sigma_x += nd;
        sigma_x2 += nd * nd;
        avg_sigma_x2 = avg_sigma_x2 * mult + (1.0 - mult) * nd * nd;
        if (!isnan(last)) {
            min_diff = FFMIN(min_diff, fabs(d - last));
            max_diff = FFMAX(max_diff, fabs(d - last));
            diff1_sum += fabs(d - last);
            diff1_sum_x2 += (d - last) * (d - last);
        }
        last = d;
        index = (int)lrint(FFMIN(FFMAX(fabs(nd), 0.0), 1.0) * HISTOGRAM_MAX);
        if (index < 0) index = 0;
        if (index > HISTOGRAM_MAX) index = HISTOGRAM_MAX;
        ehistogram[index]++;

        /* sliding max of |nd| over the last tc_samples samples */
        while (dq_len > 0 && fabs(in[dq[(dq_head + dq_len - 1) % (tc_samples + 1)]]) <= abs_d) dq_len--;
        dq[(dq_head + dq_len) % (tc_samples + 1)] = i; dq_len++;
        if (dq[dq_head] <= i - tc_samples) { dq_head = (dq_head + 1) % (tc_samples + 1); dq_len--; }

        if (nb_samples >= (uint64_t)tc_samples) {
            max_sigma_x2 = FFMAX(max_sigma_x2, avg_sigma_x2);
            min_sigma_x2 = FFMIN(min_sigma_x2, avg_sigma_x2);
        }
        nb_samples++;
        if (nb_samples >= (uint64_t)tc_samples) {
            double local_peak = fabs(in[dq[dq_head]]);
            if (isnan(noise_floor)) { noise_floor = local_peak; noise_floor_count = 1; }
            else if (local_peak < noise_floor) { noise_floor = local_peak; noise_floor_count = 1; }
            else if (local_peak == noise_floor) noise_floor_count++;
        }
    }
    /* print-time closing of the open runs (af_astats.c set_metadata/print_stats) */
    if (last == min) min_runs += min_run * min_run;
    if (last == max) max_runs += max_run * max_run;

    memset(o, 0, sizeof(*o));
    if (nb_samples == 0) { free(ehistogram); free(dq); return; }
    o->dc_offset = sigma_x / nb_samples;
    o->min_level = min; o->max_level = max;
    o->min_difference = min_diff; o->max_difference = max_diff;
    o->mean_difference = diff1_sum / (nb_samples - 1);
    o->rms_difference = sqrt(diff1_sum_x2 / (nb_samples - 1));
    o->peak_level_db = LINEAR_TO_DB(FFMAX(-nmin, nmax));
    o->rms_level_db = LINEAR_TO_DB(sqrt(sigma_x2 / nb_samples));
    o->rms_peak_db = LINEAR_TO_DB(sqrt(max_sigma_x2));
    if (min_sigma_x2 != 1) o->rms_trough_db = LINEAR_TO_DB(sqrt(min_sigma_x2));
    o->crest_factor = sigma_x2 ? FFMAX(-min, max) / sqrt(sigma_x2 / nb_samples) : 1;
    o->flat_factor = LINEAR_TO_DB((min_runs + max_runs) / (min_count + max_count));
    o->peak_count = (double)(min_count + max_count);
    o->abs_peak_count = (double)abs_peak_count;
    o->noise_floor_db = LINEAR_TO_DB(noise_floor);
    o->noise_floor_count = (double)noise_floor_count;
    {
        double entropy = 0.;
        for (int i = 0; i < HISTOGRAM_SIZE; i++) {
            double entry = ehistogram[i] / ((double)nb_samples);
            if (entry > 1e-8) entropy += entry * log2(entry);
        }
        o->entropy = -entropy / log2((double)FFMIN(nb_samples, (uint64_t)HISTOGRAM_SIZE));
    }
    o->dynamic_range = LINEAR_TO_DB(2 * FFMAX(fabs(min), fabs(max)) / min_non_zero);
    o->zero_crossings = (double)zero_runs;
    o->zero_crossings_rate = zero_runs / (double)nb_samples;
    o->number_of_samples = (double)nb_samples;
    free(ehistogram); free(dq);
}

/* ----------------------------------------------------------------- FFT (f32) */
/* Iterative radix-2 complex FFT, float arithmetic with double-computed twiddles (test oracle;
 * av_tx float FFT differs only in rounding order). */
static void fft_c2c_f32(float *re, float *im, int n)
{
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { float t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1;
        for (int k = 0; k < half; k++) {
            double ang = -2.0 * M_PI * k / len;
            float wr = (float)cos(ang), wi = (float)sin(ang);
            for (int i = k; i < n; i += len) {
                int j = i + half;
                float xr = re[j] * wr - im[j] * wi;
                float xi = re[j] * wi + im[j] * wr;
                re[j] = re[i] - xr; im[j] = im[i] - xi;
                re[i] += xr; im[i] += xi;
            }
        }
    }
}

void orc_fft_c2c_f32(float *re, float *im, int n) { fft_c2c_f32(re, im, n); }

void orc_rfft_mag_f32(const float *in, int n_fft, float *mag_half)
{
    float *re = malloc(sizeof(float) * n_fft), *im = calloc(n_fft, sizeof(float));
    memcpy(re, in, sizeof(float) * n_fft);
    fft_c2c_f32(re, im, n_fft);
    for (int i = 0; i < n_fft / 2; i++) mag_half[i] = hypotf(re[i], im[i]);
    free(re); free(im);
}

/* ------------------------------------------------------------- aspectralstats */
/* af_aspectralstats.c: win_func=hann (window_func.h: 0.5*(1-cos(2*pi*n/(N-1)))), overlap=0.5,
 * FFT scaled by 1/win_size, statistics over size = win_size/2 magnitude bins (float math). */
int64_t orc_aspectralstats_mono(const float *in, int64_t n, int sample_rate, int win_size,
                                double *stats, int64_t cap_hops)
{
    const int hop = win_size / 2;               /* win_size * (1 - 0.5) */
    const int size = win_size / 2;
    const float max_freq = sample_rate / 2;
    const float scale = max_freq / (float)size;
    float *lut = malloc(sizeof(float) * win_size);
    float *window = calloc(win_size, sizeof(float));
    float *re = malloc(sizeof(float) * win_size), *im = malloc(sizeof(float) * win_size);
    float *mag = calloc(size, sizeof(float)), *prev = calloc(size, sizeof(float));
    for (int i = 0; i < win_size; i++)
        lut[i] = (float)(.5 * (1 - cos(2 * M_PI * i / (win_size - 1))));
    int64_t nh = 0;
    for (int64_t pos = 0; pos < n; pos += hop, nh++) {
        if (nh >= cap_hops) break;
        int cnt = (int)FFMIN((int64_t)hop, n - pos);
        memmove(window, window + hop, sizeof(float) * (win_size - hop));
        memcpy(window + (win_size - hop), in + pos, sizeof(float) * cnt);
        if (cnt < hop) memset(window + (win_size - hop) + cnt, 0, sizeof(float) * (hop - cnt));
        for (int i = 0; i < win_size; i++) { re[i] = window[i] * lut[i]; im[i] = 0.f; }
        fft_c2c_f32(re, im, win_size);
        const float fscale = 1.f / win_size;
        for (int i = 0; i < size; i++) mag[i] = hypotf(re[i] * fscale, im[i] * fscale);
        double *o = stats + nh * 13;
        /* mean, variance */
        float mean = 0.f, var = 0.f;
        for (int i = 0; i < size; i++) mean += mag[i];
        mean /= size;
        for (int i = 0; i < size; i++) var += (mag[i] - mean) * (mag[i] - mean);
        var /= size;
        /* centroid */
        float num = 0.f, den = 0.f, centroid, spread, skew, kurt;
        for (int i = 0; i < size; i++) { num += mag[i] * i * scale; den += mag[i]; }
        centroid = den <= FLT_EPSILON ? 1.f : num / den;
        num = 0.f;
        for (int i = 0; i < size; i++) { float d = i * scale - centroid; num += mag[i] * d * d; }
        spread = den <= FLT_EPSILON ? 1.f : sqrtf(num / den);
        num = 0.f;
        for (int i = 0; i < size; i++) { float d = i * scale - centroid; num += mag[i] * d * d * d; }
        { float d3 = den * spread * spread * spread; skew = d3 <= FLT_EPSILON ? 1.f : num / d3; }
        num = 0.f;
        for (int i = 0; i < size; i++) { float d = i * scale - centroid; num += mag[i] * d * d * d * d; }
        { float d4 = den * spread * spread * spread * spread; kurt = d4 <= FLT_EPSILON ? 1.f : num / d4; }
        /* entropy */
        float ent = 0.f;
        for (int i = 0; i < size; i++) ent += mag[i] * logf(mag[i] + FLT_EPSILON);
        ent = -ent / logf(size);
        /* flatness */
        float lnum = 0.f, fden = 0.f, flat;
        for (int i = 0; i < size; i++) { float v = FLT_EPSILON + mag[i]; lnum += logf(v); fden += v; }
        lnum /= size; fden /= size; lnum = expf(lnum);
        flat = fden <= FLT_EPSILON ? 0.f : lnum / fden;
        /* crest */
        float mx = 0.f, msum = 0.f, crest;
        for (int i = 0; i < size; i++) { mx = fmaxf(mx, mag[i]); msum += mag[i]; }
        msum /= size;
        crest = msum <= FLT_EPSILON ? 0.f : mx / msum;
        /* flux */
        float fl = 0.f;
        for (int i = 0; i < size; i++) fl += (mag[i] - prev[i]) * (mag[i] - prev[i]);
        fl = sqrtf(fl);
        /* slope */
        float sm = 0.f, snum = 0.f, sden = 0.f, slope;
        const float m = size * 0.5f;
        for (int i = 0; i < size; i++) sm += mag[i];
        sm /= size;
        for (int i = 0; i < size; i++) { float a = (i - m) / m; snum += a * (mag[i] - sm); sden += a * a; }
        slope = fabsf(sden) <= FLT_EPSILON ? 0.f : snum / sden;
        /* decrease */
        float dnum = 0.f, dden = 0.f, decr;
        for (int i = 1; i < size; i++) { dnum += (mag[i] - mag[0]) / i; dden += mag[i]; }
        decr = dden <= FLT_EPSILON ? 0.f : dnum / dden;
        /* rolloff */
        float rs = 0.f, norm = 0.f, roll = 0.f;
        for (int i = 0; i < size; i++) norm += mag[i];
        norm *= 0.85f;
        { int idx = 0; for (int i = 0; i < size; i++) { rs += mag[i]; if (rs >= norm) { idx = i; break; } } roll = idx * scale; }
        o[0] = mean; o[1] = var; o[2] = centroid; o[3] = spread; o[4] = skew; o[5] = kurt;
        o[6] = ent; o[7] = flat; o[8] = crest; o[9] = fl; o[10] = slope; o[11] = decr; o[12] = roll;
        memcpy(prev, mag, sizeof(float) * size);
    }
    free(lut); free(window); free(re); free(im); free(mag); free(prev);
    return nh;
}
