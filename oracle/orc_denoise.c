/*
 * orc_denoise.c — ORACLE (test infrastructure only; see jt_oracle.h).
 * anlmdn (libavfilter/af_anlmdn.c) and afftdn (libavfilter/af_afftdn.c, tn=0) restated
 * from FFmpeg 8.1.  Reference call sites: filters.go:95-100,804-861 (anlmdn=s:p:r:m,
 * afftdn=nr:nt[:bn]:tn[:nf]); adaptive.go:133-170 (nf / custom profile).
 * parity unpinned at the FFmpeg boundary (see jt_oracle.h); afftdn is the highest-risk
 * restatement (transform size and bark-band mapping are recalled, not verified).
 *
 * Alignment conventions (documented choices; FFmpeg compensates latency through pts):
 *   anlmdn: output sample n corresponds to input sample n; hops of H=2K+1 outputs start at
 *           n = m*H - (K+S); input outside [0,N) is zero.
 *   afftdn: output aligned with input; frame t covers input [t*A - (W-A), t*A + A), A = rate/80,
 *           W = 3A; zero history before 0; tail flushed with zeros.
 */
#include "jt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FFMIN(a,b) ((a) < (b) ? (a) : (b))
#define FFMAX(a,b) ((a) > (b) ? (a) : (b))

/* ================================================================== anlmdn */
#define WEIGHT_LUT_NBITS 20
#define WEIGHT_LUT_SIZE  (1 << WEIGHT_LUT_NBITS)

static float compute_distance_ssd(const float *f1, const float *f2, ptrdiff_t K)
{
    float distance = 0.f;
    for (ptrdiff_t k = -K; k <= K; k++) {
        float d = f1[k] - f2[k];
        distance += d * d;
    }
    return distance;
}

static void compute_cache(float *cache, const float *f, ptrdiff_t S, ptrdiff_t K, ptrdiff_t i, ptrdiff_t jj)
{
    int v = 0;
    for (ptrdiff_t j = jj; j < jj + S; j++, v++) {
        float a = f[i - K - 1] - f[j - K - 1];
        float b = f[i + K] - f[j + K];
        cache[v] += -(a * a) + (b * b);
    }
}

void orc_anlmdn_f32(const float *in, float *out, int64_t n, int sample_rate,
                    double strength, double patch_s, double research_s, double smooth_opt)
{
    /* config_filter(): durations are AV_OPT_TYPE_DURATION (microseconds) rescaled to samples */
    const int K = (int)llrint((double)llrint(patch_s * 1e6) * sample_rate / 1e6);
    const int S = (int)llrint((double)llrint(research_s * 1e6) * sample_rate / 1e6);
    const int H = K * 2 + 1;
    const int N = H + (K + S) * 2;
    const float a = (float)strength;
    const float m = (float)smooth_opt;
    const float pdiff_lut_scale = 1.f / m * WEIGHT_LUT_SIZE;
    const float sw = (65536.f / (4 * K + 2)) / sqrtf(a);
    const float smooth = m;
    float *window = malloc(sizeof(float) * N);
    float *cache = malloc(sizeof(float) * 2 * S);
    const int64_t lead = K + S;
    for (int64_t hop = -lead; hop < n; hop += H) {
        /* window[w] = x[hop - (K+S) + w]; f = window + K; centre i in [S, S+H) maps to x[hop + (i-S)] */
        for (int w = 0; w < N; w++) {
            int64_t k = hop - lead + w;
            window[w] = (k >= 0 && k < n) ? in[k] : 0.f;
        }
        const float *f = window + K;
        for (int i = S; i < H + S; i++) {
            float P = 0.f, Q = 0.f;
            if (i == S) {
                int v = 0;
                for (int j = i - S; j <= i + S; j++) {
                    if (i == j) continue;
                    cache[v++] = compute_distance_ssd(f + i, f + j, K);
                }
            } else {
                compute_cache(cache, f, S, K, i, i - S);
                compute_cache(cache + S, f, S, K, i, i + 1);
            }
            for (int j = 0; j < 2 * S; j++) {
                float distance = cache[j];
                unsigned weight_lut_idx;
                float w;
                if (distance < 0.f)
                    cache[j] = distance = 0.f;
                w = distance * sw;
                if (w >= smooth)
                    continue;
                weight_lut_idx = (unsigned)(w * pdiff_lut_scale);
                w = expf(-(float)weight_lut_idx / pdiff_lut_scale);   /* weight_lut[idx] */
                P += w * f[i - S + j + (j >= S)];
                Q += w;
            }
            P += f[i];
            Q += 1;
            int64_t o = hop + (i - S);
            if (o >= 0 && o < n)
                out[o] = P / Q;
        }
    }
    free(window); free(cache);
}

/* ================================================================== afftdn */
#define C_LN (M_LN10 * 0.1)
#define NB_PROFILE_BANDS 15
#define SOLVE_SIZE 5

void orc_fft_c2c_f32(float *re, float *im, int n);

static const int band_centre_tab[NB_PROFILE_BANDS] = {
    /* analyser_noise_bands.go:15-17 ("verified against the ffmpeg 8.1 af_afftdn.c source") */
    80, 125, 195, 290, 440, 660, 1000, 1500, 2250, 3350, 5000, 7500, 11200, 16000, 24000
};

typedef struct {
    int sample_rate, sample_advance, window_length, fft_length, bin_count, number_of_bands;
    int *bin2band;
    double *window, *band_alpha, *band_beta;
    double band_noise[NB_PROFILE_BANDS];
    double *amt, *band_amt, *band_excit, *gain, *prior, *prior_band_excit, *clean_data, *noisy_data,
           *spread_function, *abs_var, *rel_var, *min_abs_var;
    double noise_reduction, noise_floor, max_gain, max_var, gain_scale, floor;
    double matrix_a[SOLVE_SIZE * SOLVE_SIZE], vector_b[SOLVE_SIZE], matrix_b[SOLVE_SIZE * NB_PROFILE_BANDS];
} Afftdn;

static double freq2bark(double x)
{
    double d = x / 7500.0;
    return 13.0 * atan(7.6E-4 * x) + 3.5 * atan(d * d);
}

static void factor(double *array, int size)
{
    for (int i = 0; i < size - 1; i++) {
        for (int j = i + 1; j < size; j++) {
            double d = array[j + i * size] / array[i + i * size];
            array[j + i * size] = d;
            for (int k = i + 1; k < size; k++)
                array[j + k * size] -= d * array[i + k * size];
        }
    }
}

static void solve(double *matrix, double *vector, int size)
{
    for (int i = 0; i < size - 1; i++)
        for (int j = i + 1; j < size; j++) {
            double d = matrix[j + i * size];
            vector[j] -= d * vector[i];
        }
    vector[size - 1] /= matrix[size * size - 1];
    for (int i = size - 2; i >= 0; i--) {
        double d = vector[i];
        for (int j = i + 1; j < size; j++)
            d -= matrix[i + j * size] * vector[j];
        vector[i] = d / matrix[i + i * size];
    }
}

static double process_get_band_noise(Afftdn *s, int band)
{
    double product, sum, f;
    int i = 0;
    if (band < NB_PROFILE_BANDS)
        return s->band_noise[band];
    for (int j = 0; j < SOLVE_SIZE; j++) {
        sum = 0.0;
        for (int k = 0; k < NB_PROFILE_BANDS; k++)
            sum += s->matrix_b[i++] * s->band_noise[k];
        s->vector_b[j] = sum;
    }
    solve(s->matrix_a, s->vector_b, SOLVE_SIZE);
    f = (0.5 * s->sample_rate) / band_centre_tab[NB_PROFILE_BANDS - 1];
    f = 15.0 + log(f / 1.5) / log(1.5);
    sum = 0.0;
    product = 1.0;
    for (int j = 0; j < SOLVE_SIZE; j++) {
        sum += product * s->vector_b[j];
        product *= f;
    }
    return sum;
}

static void set_band_parameters(Afftdn *s)
{
    double band_noise, d2 = 1, d3, d4, d5;
    int i = 0, j = 0, k = 0;
    d5 = 0.0;
    band_noise = process_get_band_noise(s, 0);
    for (int m = j; m < s->bin_count; m++) {
        if (m == j) {
            i = j;
            d5 = band_noise;
            if (k >= NB_PROFILE_BANDS)
                j = s->bin_count;
            else
                j = (int)((double)s->fft_length * band_centre_tab[k] / s->sample_rate);
            d2 = j - i;
            band_noise = process_get_band_noise(s, k);
            k++;
        }
        d3 = (j - m) / d2;
        d4 = (m - i) / d2;
        s->rel_var[m] = exp((d5 * d3 + band_noise * d4) * C_LN);
    }
}

static void set_parameters(Afftdn *s)
{
    s->max_var = s->floor * exp((100.0 + s->noise_floor) * C_LN);
    s->max_gain = exp(s->noise_reduction * (0.5 * C_LN));
    s->gain_scale = 1.0 / (s->max_gain * s->max_gain);
    set_band_parameters(s);
    for (int i = 0; i < s->bin_count; i++) {
        s->abs_var[i] = fmax(s->max_var * s->rel_var[i], 1.0);
        s->min_abs_var[i] = s->gain_scale * s->abs_var[i];
    }
}

static double limit_gain(double a, double b)
{
    if (a > 1.0) return (b * a - 1.0) / (b + a - 2.0);
    if (a < 1.0) return (b * a - 2.0 * a + 1.0) / (b - a);
    return 1.0;
}

static void afftdn_init(Afftdn *s, int sample_rate, double nr, double nf, const double *bn)
{
    memset(s, 0, sizeof(*s));
    s->sample_rate = sample_rate;
    s->sample_advance = sample_rate / 80;
    s->window_length = 3 * s->sample_advance;
    { int v = s->window_length, bits = 0; while (v) { bits++; v >>= 1; } s->fft_length = 1 << bits; }
    s->bin_count = s->fft_length / 2 + 1;

    for (int j = 0; j < SOLVE_SIZE; j++)
        for (int k = 0; k < SOLVE_SIZE; k++) {
            s->matrix_a[j + k * SOLVE_SIZE] = 0.0;
            for (int m = 0; m < NB_PROFILE_BANDS; m++)
                s->matrix_a[j + k * SOLVE_SIZE] += pow(m, j + k);
        }
    factor(s->matrix_a, SOLVE_SIZE);
    { int i = 0; for (int j = 0; j < SOLVE_SIZE; j++) for (int k = 0; k < NB_PROFILE_BANDS; k++) s->matrix_b[i++] = pow(k, j); }

    s->bin2band = calloc(s->bin_count, sizeof(int));
    const double sdiv = 1.25;  /* band_multiplier default */
    for (int i = 0; i < s->bin_count; i++)
        s->bin2band[i] = (int)lrint(sdiv * freq2bark(((double)i * s->sample_rate) / s->fft_length));
    s->number_of_bands = s->bin2band[s->bin_count - 1] + 1;
    const int nb = s->number_of_bands, bc = s->bin_count;
    s->window = calloc(s->window_length, sizeof(double));
    s->band_alpha = calloc(nb, sizeof(double)); s->band_beta = calloc(nb, sizeof(double));
    s->amt = calloc(bc, sizeof(double)); s->band_amt = calloc(nb, sizeof(double));
    s->band_excit = calloc(nb, sizeof(double)); s->gain = calloc(bc, sizeof(double));
    s->prior = calloc(bc, sizeof(double)); s->prior_band_excit = calloc(nb, sizeof(double));
    s->clean_data = calloc(bc, sizeof(double)); s->noisy_data = calloc(bc, sizeof(double));
    s->spread_function = calloc((size_t)nb * nb, sizeof(double));
    s->abs_var = calloc(bc, sizeof(double)); s->rel_var = calloc(bc, sizeof(double));
    s->min_abs_var = calloc(bc, sizeof(double));

    /* noise profile: white = zeros, custom = clip(+-24); both mean-reduced */
    for (int i = 0; i < NB_PROFILE_BANDS; i++)
        s->band_noise[i] = bn ? FFMIN(FFMAX((double)(float)bn[i], -24.), 24.) : 0.;
    { double mean = 0; for (int i = 0; i < NB_PROFILE_BANDS; i++) mean += s->band_noise[i];
      mean /= NB_PROFILE_BANDS; for (int i = 0; i < NB_PROFILE_BANDS; i++) s->band_noise[i] -= mean; }

    for (int i = 0; i < bc; i++) s->prior[i] = 1.0 - 1.0 + 0.0;  /* calloc'd: prior starts at 0 */
    {
        double p1 = pow(0.1, 2.5 / sdiv), p2 = pow(0.1, 1.0 / sdiv);
        int j = 0;
        for (int m = 0; m < nb; m++)
            for (int n2 = 0; n2 < nb; n2++) {
                if (n2 < m) s->spread_function[j++] = pow(p2, m - n2);
                else if (n2 > m) s->spread_function[j++] = pow(p1, n2 - m);
                else s->spread_function[j++] = 1.0;
            }
        for (int m = 0; m < nb; m++) { s->band_excit[m] = 0.0; s->prior_band_excit[m] = 0.0; }
        for (int m = 0; m < bc; m++) s->band_excit[s->bin2band[m]] += 1.0;
        j = 0;
        for (int m = 0; m < nb; m++)
            for (int n2 = 0; n2 < nb; n2++)
                s->prior_band_excit[m] += s->spread_function[j++] * s->band_excit[n2];
        double mn = pow(0.1, 2.5), mx = pow(0.1, 1.0);
        for (int i = 0; i < nb; i++) {
            if (i < lrint(12.0 * sdiv)) s->band_excit[i] = pow(0.1, 1.45 + 0.1 * i / sdiv);
            else s->band_excit[i] = pow(0.1, 2.5 - 0.2 * (i / sdiv - 14.0));
            s->band_excit[i] = FFMIN(FFMAX(s->band_excit[i], mn), mx);
        }
        j = 0;
        for (int i = 0; i < nb; i++)
            for (int k = 0; k < nb; k++)
                s->spread_function[j++] *= s->band_excit[i] / s->prior_band_excit[i];
    }
    {
        int j = 0;
        double sar = s->sample_advance / (double)s->sample_rate;
        for (int i = 0; i < bc; i++) {
            if ((i == s->fft_length / 2) || (s->bin2band[i] > j)) {
                double d6 = (i - 1) * (double)s->sample_rate / s->fft_length;
                double d7 = fmin(0.008 + 2.2 / d6, 0.03);
                s->band_alpha[j] = exp(-sar / d7);
                s->band_beta[j] = 1.0 - s->band_alpha[j];
                j = s->bin2band[i];
            }
        }
    }
    {
        double wscale = sqrt(8.0 / (9.0 * s->fft_length)), sum = 0.0;
        for (int i = 0; i < s->window_length; i++) {
            double d10 = sin(i * M_PI / s->window_length);
            d10 *= wscale * d10;
            s->window[i] = d10;
            sum += d10 * d10;
        }
        double window_weight = 0.5 * sum;
        s->floor = (double)(1LL << 48) * exp(-23.025558369790467) * window_weight;
    }
    s->noise_reduction = nr;
    s->noise_floor = nf;
    set_parameters(s);
}

static void afftdn_free(Afftdn *s)
{
    free(s->bin2band); free(s->window); free(s->band_alpha); free(s->band_beta); free(s->amt);
    free(s->band_amt); free(s->band_excit); free(s->gain); free(s->prior); free(s->prior_band_excit);
    free(s->clean_data); free(s->noisy_data); free(s->spread_function); free(s->abs_var);
    free(s->rel_var); free(s->min_abs_var);
}

/* process_frame(): per-bin a-priori-SNR gain with decision-directed prior, bark-band masking
 * limit, gain_smooth=0.  ratio = adaptivity (0.5) except on the very first frame (1.0). */
/* track_noise (tn=1, af_afftdn.c process_frame): after the first-stage gains, when the frame's magnitude spectrum is flat
 * (geometric / arithmetic mean of the bins above s->floor > 0.8) the noise floor moves a tenth of the way to
 *   clip(10 log10(mean) - 100 + floor_offset * max|S - mean| / mean, -90, -20)        (floor_offset option 'fo' = 1.0)
 * and set_parameters() re-derives abs_var / min_abs_var (rel_var, the band shape, stays); the masking limits of the SAME frame
 * already use the new variances.  The reference emits tn=1 whenever Noise.Floor == 0 (adaptive.go:147-151; the default golden chain
 * filters_test.go:298-311 carries it), starting from afftdn's default nf = -50 dB. */
static void track_noise_update(Afftdn *s)
{
    double num = 0., den = 0., mx = 0.; int size = 0;
    for (int n = 0; n < s->bin_count; n++) {
        const double v = s->noisy_data[n];
        if (v > s->floor) { num += log(v); den += v; size++; }
    }
    if (size < 1) size = 1;
    num /= size; den /= size; num = exp(num);
    if (num / den > 0.8) {
        for (int n = 0; n < s->bin_count; n++) { const double p = fabs(s->noisy_data[n] - den); if (p > mx) mx = p; }
        const double offset = 1.0 * (mx / den);
        double new_floor = 10.0 * log10(den) - 100.0 + offset;
        new_floor = FFMIN(FFMAX(new_floor, -90.), -20.);       /* av_clipd = FFMIN(FFMAX(a, amin), amax): a NaN (silent frame) becomes -90 */
        s->noise_floor = 0.1 * new_floor + s->noise_floor * 0.9;
        s->max_var = s->floor * exp((100.0 + s->noise_floor) * C_LN);
        for (int i = 0; i < s->bin_count; i++) {
            s->abs_var[i] = fmax(s->max_var * s->rel_var[i], 1.0);
            s->min_abs_var[i] = s->gain_scale * s->abs_var[i];
        }
    }
}

static void process_frame(Afftdn *s, float *re, float *im, int first, int track)
{
    const double ratio = first ? 1.0 : 0.5, rratio = 1. - ratio;
    const int nb = s->number_of_bands;
    for (int i = 0; i < s->bin_count; i++) {
        double mag = hypot(re[i], im[i]);
        double power = mag * mag;
        double mag_abs_var = power / s->abs_var[i];
        double new_mag_abs_var = ratio * s->prior[i] + rratio * fmax(mag_abs_var - 1.0, 0.0);
        double new_gain = new_mag_abs_var / (1.0 + new_mag_abs_var);
        double sqr_new_gain = new_gain * new_gain;
        s->noisy_data[i] = mag;
        s->prior[i] = mag_abs_var * sqr_new_gain;
        s->clean_data[i] = power * sqr_new_gain;
        s->gain[i] = new_gain;
    }
    if (track) track_noise_update(s);
    for (int i = 0; i < nb; i++) { s->band_excit[i] = 0.0; s->band_amt[i] = 0.0; }
    for (int i = 0; i < s->bin_count; i++) s->band_excit[s->bin2band[i]] += s->clean_data[i];
    for (int i = 0; i < nb; i++) {
        s->band_excit[i] = fmax(s->band_excit[i],
                                s->band_alpha[i] * s->band_excit[i] + s->band_beta[i] * s->prior_band_excit[i]);
        s->prior_band_excit[i] = s->band_excit[i];
    }
    for (int j = 0, i = 0; j < nb; j++)
        for (int k = 0; k < nb; k++)
            s->band_amt[j] += s->spread_function[i++] * s->band_excit[k];
    for (int i = 0; i < s->bin_count; i++) s->amt[i] = s->band_amt[s->bin2band[i]];
    for (int i = 0; i < s->bin_count; i++) {
        if (s->amt[i] > s->abs_var[i]) s->gain[i] = 1.0;
        else if (s->amt[i] > s->min_abs_var[i]) {
            const double limit = sqrt(s->abs_var[i] / s->amt[i]);
            s->gain[i] = limit_gain(s->gain[i], limit);
        } else s->gain[i] = limit_gain(s->gain[i], s->max_gain);
    }
    for (int i = 0; i < s->bin_count; i++) {
        const float g = (float)s->gain[i];
        re[i] *= g; im[i] *= g;
    }
}

void orc_afftdn_tn_f32(const float *in, float *out, int64_t n, int sample_rate,
                       double nr_db, double nf_db, const double *band_noise, int track, double *floor_series, int64_t cap);
void orc_afftdn_f32(const float *in, float *out, int64_t n, int sample_rate,
                    double nr_db, double nf_db, const double *band_noise)
{
    orc_afftdn_tn_f32(in, out, n, sample_rate, nr_db, nf_db, band_noise, 0, NULL, 0);
}

/* floor_series (optional): the noise floor in dB after every frame (cap entries) */
void orc_afftdn_tn_f32(const float *in, float *out, int64_t n, int sample_rate,
                       double nr_db, double nf_db, const double *band_noise, int track, double *floor_series, int64_t cap)
{
    Afftdn s;
    afftdn_init(&s, sample_rate, nr_db, nf_db, band_noise);
    const int A = s.sample_advance, W = s.window_length, L = s.fft_length;
    float *re = malloc(sizeof(float) * L), *im = malloc(sizeof(float) * L);
    double *acc = calloc((size_t)L * 2, sizeof(double));   /* out_samples (buffer_length) */
    int64_t nframes = (n + A - 1) / A + (W - A) / A;         /* input hops + flush hops */
    for (int64_t t = 0; t < nframes; t++) {
        int64_t start = t * A - (W - A);
        for (int m = 0; m < W; m++) {
            int64_t k = start + m;
            float x = (k >= 0 && k < n) ? in[k] : 0.f;
            re[m] = (float)(s.window[m] * x * (double)(1LL << 23));
            im[m] = 0.f;
        }
        for (int m = W; m < L; m++) { re[m] = 0.f; im[m] = 0.f; }
        orc_fft_c2c_f32(re, im, L);
        process_frame(&s, re, im, t == 0, track);
        if (floor_series && t < cap) floor_series[t] = s.noise_floor;
        /* inverse real transform: rebuild the conjugate half, inverse FFT via conj trick */
        for (int k = 1; k < L / 2; k++) { re[L - k] = re[k]; im[L - k] = -im[k]; }
        im[0] = 0.f; im[L / 2] = 0.f;
        for (int k = 0; k < L; k++) im[k] = -im[k];
        orc_fft_c2c_f32(re, im, L);                          /* unnormalised inverse (real part in re) */
        for (int m = 0; m < W; m++)
            acc[m] += s.window[m] * re[m] / (double)(1LL << 23);
        /* emit the first A samples of the accumulator: they correspond to input [start, start+A) */
        for (int m = 0; m < A; m++) {
            int64_t k = start + m;
            if (k >= 0 && k < n) out[k] = (float)acc[m];
        }
        memmove(acc, acc + A, sizeof(double) * ((size_t)L * 2 - A));
        memset(acc + ((size_t)L * 2 - A), 0, sizeof(double) * A);
    }
    free(re); free(im); free(acc);
    afftdn_free(&s);
}
