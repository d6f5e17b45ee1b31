/*
 * orc_declick.c — CPU oracle for FFmpeg af_adeclick.c (adeclick), the click/pop repair the reference inserts in Pass 4
 * between loudnorm and the brickwall limiter (normalise.go:1306-1311; "adeclick=t=%.1f:w=%.0f:o=%.0f:m=%s",
 * filters.go:947-962; production defaults t=1.7 w=55 o=50 m=s, filters.go:513-521).
 * TEST INFRASTRUCTURE ONLY (see jt_oracle.h).  Restated from knowledge of libavfilter/af_adeclick.c (FFmpeg 8.1): per
 * window an autoregressive model (Levinson-Durbin on the biased autocorrelation), click detection on the prediction error
 * against threshold * sigma_e with burst fusion, least-squares AR interpolation of the flagged samples (LDL^T of the banded
 * normal matrix), windows advanced by hop = w*(1-o/100) with overlap-save ('s') or overlap-add ('a') output.
 * Double precision throughout, plain sequential loops, mul-then-add (compiled -ffp-contract=off).
 */
#include "jt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static void autocorrelation(const double *input, int order, int size, double *output, double scale)
{
    for (int i = 0; i <= order; i++) {
        double value = 0.;
        for (int j = i; j < size; j++) value += input[j] * input[j - i];
        output[i] = value * scale;
    }
}

/* k[0..order] <- AR polynomial (k[0] = 1); r <- autocorrelation; a = scratch[order]; returns sigma_e */
static double autoregression(const double *samples, int ar_order, int nb_samples, double *k, double *r, double *a)
{
    double alpha;
    memset(a, 0, ar_order * sizeof(*a));
    autocorrelation(samples, ar_order, nb_samples, r, 1. / nb_samples);
    /* Levinson-Durbin */
    k[0] = a[0] = -r[1] / r[0];
    alpha = r[0] * (1. - k[0] * k[0]);
    for (int i = 1; i < ar_order; i++) {
        double epsilon = 0.;
        for (int j = 0; j < i; j++) epsilon += a[j] * r[i - j];
        epsilon += r[i + 1];
        k[i] = -epsilon / alpha;
        alpha *= (1. - k[i] * k[i]);
        for (int j = i - 1; j >= 0; j--) k[j] = a[j] + k[i] * a[i - j - 1];
        for (int j = 0; j <= i; j++) a[j] = k[j];
    }
    k[0] = 1.;
    for (int i = 1; i <= ar_order; i++) k[i] = a[i - 1];
    return sqrt(alpha);
}

static int isfinite_array(const double *samples, int nb_samples)
{
    for (int i = 0; i < nb_samples; i++) if (!isfinite(samples[i])) return 0;
    return 1;
}

static int find_index(const int *index, int value, int size)
{
    if ((value < index[0]) || (value > index[size - 1])) return 1;
    int i = 0, j = size - 1;
    while (i <= j) {
        int k = (i + j) / 2;
        if (index[k] == value) return 0;
        if (index[k] < value) i = k + 1; else j = k - 1;
    }
    return 1;
}

static int cholesky_decomposition(double *matrix, int n)
{
    for (int i = 0; i < n; i++) {
        const int in = i * n;
        double value = matrix[in + i];
        for (int j = 0; j < i; j++) value -= matrix[j * n + j] * matrix[in + j] * matrix[in + j];
        if (value == 0.) return -1;
        matrix[in + i] = value;
        for (int j = i + 1; j < n; j++) {
            const int jn = j * n;
            double x = matrix[jn + i];
            for (int k = 0; k < i; k++) x -= matrix[k * n + k] * matrix[in + k] * matrix[jn + k];
            matrix[jn + i] = x / matrix[in + i];
        }
    }
    return 0;
}

static int do_interpolation(double *matrix, double *vector, int n, double *out)
{
    if (cholesky_decomposition(matrix, n) < 0) return -1;
    double *y = malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; i++) {
        const int in = i * n;
        double value = vector[i];
        for (int j = 0; j < i; j++) value -= matrix[in + j] * y[j];
        y[i] = value;
    }
    for (int i = n - 1; i >= 0; i--) {
        out[i] = y[i] / matrix[i * n + i];
        for (int j = i + 1; j < n; j++) out[i] -= matrix[j * n + i] * out[j];
    }
    free(y);
    return 0;
}

static int interpolation(const double *src, int ar_order, const double *acoefficients, const int *index, int nb_errors,
                         double *auxiliary, double *interpolated)
{
    double *matrix = malloc(sizeof(double) * (size_t)nb_errors * nb_errors);
    double *vector = malloc(sizeof(double) * (size_t)nb_errors);
    autocorrelation(acoefficients, ar_order, ar_order + 1, auxiliary, 1.);
    for (int i = 0; i < nb_errors; i++) {
        const int im = i * nb_errors;
        for (int j = i; j < nb_errors; j++) {
            if (abs(index[j] - index[i]) <= ar_order)
                matrix[j * nb_errors + i] = matrix[im + j] = auxiliary[abs(index[j] - index[i])];
            else
                matrix[j * nb_errors + i] = matrix[im + j] = 0;
        }
    }
    for (int i = 0; i < nb_errors; i++) {
        double value = 0.;
        for (int j = -ar_order; j <= ar_order; j++)
            if (find_index(index, index[i] - j, nb_errors)) value -= src[index[i] - j] * auxiliary[abs(j)];
        vector[i] = value;
    }
    int ret = do_interpolation(matrix, vector, nb_errors, interpolated);
    free(matrix); free(vector);
    return ret;
}

static int detect_clicks(int window_size, int ar_order, int nb_burst_samples, double threshold, double sigmae, double *detection,
                         const double *acoefficients, unsigned char *click, int *index, const double *src, double *dst)
{
    int nb_clicks = 0, prev = -1;
    memset(detection, 0, window_size * sizeof(*detection));
    for (int i = ar_order; i < window_size; i++)
        for (int j = 0; j <= ar_order; j++) detection[i] += acoefficients[j] * src[i - j];
    for (int i = 0; i < window_size; i++) {
        click[i] = fabs(detection[i]) > sigmae * threshold;
        dst[i] = src[i];
    }
    for (int i = 0; i < window_size; i++) {
        if (!click[i]) continue;
        if (prev >= 0 && (i > prev + 1) && (i <= nb_burst_samples + prev))
            for (int j = prev + 1; j < i; j++) click[j] = 1;
        prev = i;
    }
    memset(click, 0, ar_order * sizeof(*click));
    memset(click + (window_size - ar_order), 0, ar_order * sizeof(*click));
    for (int i = ar_order; i < window_size - ar_order; i++) if (click[i]) index[nb_clicks++] = i;
    return nb_clicks;
}

/* method: 0 = overlap-add ('a'), 1 = overlap-save ('s').  ar_pct = arorder option (default 2), burst default 2.
 * n_clicks_out (optional) receives the total number of repaired samples.  Returns 0, or -1 when a window's normal matrix
 * is singular (the filter would fail). */
int orc_adeclick_f64(const double *in, double *out, int64_t n, int sample_rate, double threshold, double window_ms,
                     double overlap_pct, double ar_pct, double burst, int method, int64_t *n_clicks_out)
{
    int window_size = (int)(sample_rate * window_ms / 1000.);
    if (window_size < 100) window_size = 100;
    int ar_order = (int)(window_size * ar_pct / 100.); if (ar_order < 1) ar_order = 1;
    const int nb_burst_samples = (int)(window_size * burst / 1000.);
    int hop_size = (int)(window_size * (1. - (overlap_pct / 100.))); if (hop_size < 1) hop_size = 1;
    const int overlap_skip = method ? (window_size - hop_size) / 2 : 0;
    double *wlut = malloc(sizeof(double) * window_size);
    for (int i = 0; i < window_size; i++) wlut[i] = sin(M_PI * i / window_size) * (1. - (overlap_pct / 100.)) * M_PI_2;
    double *src = calloc(window_size, sizeof(double)), *dst = calloc(window_size, sizeof(double));
    double *buf = calloc((size_t)window_size * 2, sizeof(double));
    double *detection = calloc(window_size, sizeof(double)), *interpolated = calloc(window_size, sizeof(double));
    double *acoef = calloc(ar_order + 1, sizeof(double)), *acorr = calloc(ar_order + 1, sizeof(double));
    double *aux = calloc(ar_order + 1, sizeof(double)), *tmp = calloc(ar_order, sizeof(double));
    unsigned char *click = calloc(window_size, 1); int *index = calloc(window_size, sizeof(int));
    int64_t total_clicks = 0; int rc = 0;
    /* the fifo holds overlap_skip zeros followed by the input; at EOF the last windows are zero-padded */
    const int64_t fifo_len = n + overlap_skip;
    int64_t produced = 0;
    for (int64_t w0 = 0; produced < n; w0 += hop_size) {
        for (int j = 0; j < window_size; j++) {
            const int64_t f = w0 + j;                       /* fifo position */
            const int64_t p = f - overlap_skip;             /* input position */
            src[j] = (f < fifo_len && p >= 0) ? in[p] : 0.0;
        }
        const double sigmae = autoregression(src, ar_order, window_size, acoef, acorr, tmp);
        if (isfinite_array(acoef, ar_order + 1)) {
            const int nb_errors = detect_clicks(window_size, ar_order, nb_burst_samples, threshold, sigmae, detection, acoef, click,
                                                index, src, dst);
            if (nb_errors > 0) {
                if (interpolation(src, ar_order, acoef, index, nb_errors, aux, interpolated) < 0) { rc = -1; break; }
                for (int j = 0; j < nb_errors; j++) dst[index[j]] = interpolated[j];
                total_clicks += nb_errors;
            }
        } else {
            memcpy(dst, src, window_size * sizeof(*dst));
        }
        if (method == 0) { for (int j = 0; j < window_size; j++) buf[j] += dst[j] * wlut[j]; }
        else { for (int j = 0; j < hop_size; j++) buf[j] = dst[overlap_skip + j]; }
        for (int j = 0; j < hop_size && produced < n; j++) out[produced++] = buf[j];
        memmove(buf, buf + hop_size, ((size_t)window_size * 2 - hop_size) * sizeof(*buf));
        memset(buf + (size_t)window_size * 2 - hop_size, 0, hop_size * sizeof(*buf));
    }
    if (n_clicks_out) *n_clicks_out = total_clicks;
    free(wlut); free(src); free(dst); free(buf); free(detection); free(interpolated); free(acoef); free(acorr); free(aux); free(tmp);
    free(click); free(index);
    return rc;
}
