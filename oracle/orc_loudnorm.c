/*
 * orc_loudnorm.c — ORACLE (test infrastructure only; see jt_oracle.h).
 * af_loudnorm.c's DYNAMIC mode (FFmpeg 8.1, libavfilter/af_loudnorm.c) restated sequentially for one channel: the path loudnorm
 * takes in the reference's Pass 4 when its second-pass preconditions fail (measured LRA above the LRA target, a measured LRA that
 * prints as 0.00, ...; the reference notices it from the stats and logs a warning: normalise.go:687-693), and always takes in
 * Pass 3 (normalise.go:226-346, whose audio output the reference discards).  In that mode the filter forces its links to 192 kHz,
 * so `in` is the stream after the auto-inserted aresample and the reference's own `aresample=<source rate>` follows
 * (normalise.go:1293-1310).
 *
 * parity unpinned at the FFmpeg boundary (see jt_oracle.h): restated from the filter's source as remembered, ring buffers, frame
 * types and limiter state machine kept as they are there (3 s look-ahead buffer, 210 ms limiter buffer, 100 ms frames, the 21-tap
 * gaussian over the 30 most recent frame gains, attack 10 ms / release 100 ms, the `continue` in detect_peak that leaves prev_smp
 * stale after a rejected candidate, the FINAL_FRAME that refills the limiter buffer from the look-ahead buffer with one gain).
 * Channel count 1, dual_mono as given.  The measurement side is libavfilter/ebur128.c (histogram gating) as in orc_r128.c.
 */
#include "jt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_ebur128_kweight_coeffs(int sample_rate, double pre_b[3], double pre_a[3], double rlb_b[3], double rlb_a[3]);   /* orc_r128.c */

/* ------------------------------------------------------------------ incremental ebur128.c state (MODE_I | S | LRA | SAMPLE_PEAK) */
typedef struct {
    double b[5], a[5], v[5];
    double *audio; int64_t ring, idx, s100, needed, have, st_counter;
    unsigned long bhist[1000], shist[1000];
    double peak; int dual;
} lr128;
static double l_hist_energy[1000], l_hist_bound[1001];
static int l_hist_done = 0;
static void l_hist_init(void)
{
    if (l_hist_done) return;
    l_hist_bound[0] = pow(10.0, (-70.0 + 0.691) / 10.0);
    for (int i = 0; i < 1000; ++i) l_hist_energy[i] = pow(10.0, ((double)i / 10.0 - 69.95 + 0.691) / 10.0);
    for (int i = 1; i < 1001; ++i) l_hist_bound[i] = pow(10.0, ((double)i / 10.0 - 70.0 + 0.691) / 10.0);
    l_hist_done = 1;
}
static size_t l_hist_index(double energy)
{
    size_t lo = 0, hi = 1000, mid;
    do { mid = (lo + hi) / 2; if (energy >= l_hist_bound[mid]) lo = mid; else hi = mid; } while (hi - lo != 1);
    return lo;
}
static double l_e2l(double e) { return 10 * (log(e) / log(10.0)) - 0.691; }
static void lr128_init(lr128 *s, int rate, int dual)
{
    l_hist_init();
    memset(s, 0, sizeof(*s));
    double pb[3], pa[3] = {1, 0, 0}, rb[3] = {1, -2, 1}, ra[3] = {1, 0, 0};
    orc_ebur128_kweight_coeffs(rate, pb, pa, rb, ra);
    s->b[0] = pb[0] * rb[0];
    s->b[1] = pb[0] * rb[1] + pb[1] * rb[0];
    s->b[2] = pb[0] * rb[2] + pb[1] * rb[1] + pb[2] * rb[0];
    s->b[3] = pb[1] * rb[2] + pb[2] * rb[1];
    s->b[4] = pb[2] * rb[2];
    s->a[0] = pa[0] * ra[0];
    s->a[1] = pa[0] * ra[1] + pa[1] * ra[0];
    s->a[2] = pa[0] * ra[2] + pa[1] * ra[1] + pa[2] * ra[0];
    s->a[3] = pa[1] * ra[2] + pa[2] * ra[1];
    s->a[4] = pa[2] * ra[2];
    s->s100 = (rate + 5) / 10; s->ring = s->s100 * 30; s->needed = s->s100 * 4; s->dual = dual;
    s->audio = calloc((size_t)s->ring, sizeof(double));
}
static void lr128_free(lr128 *s) { free(s->audio); }
static double lr128_energy_last(const lr128 *s, int64_t frames)          /* ebur128_calc_gating_block over the last `frames` */
{
    double sum = 0.0;
    if (s->idx < frames) {
        for (int64_t k = 0; k < s->idx; k++) sum += s->audio[k] * s->audio[k];
        for (int64_t k = s->ring - (frames - s->idx); k < s->ring; k++) sum += s->audio[k] * s->audio[k];
    } else {
        for (int64_t k = s->idx - frames; k < s->idx; k++) sum += s->audio[k] * s->audio[k];
    }
    if (s->dual) sum *= 2.0;
    return sum / (double)frames;
}
static void lr128_add(lr128 *s, const double *x, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        if (fabs(x[i]) > s->peak) s->peak = fabs(x[i]);
        s->v[0] = x[i] - s->a[1] * s->v[1] - s->a[2] * s->v[2] - s->a[3] * s->v[3] - s->a[4] * s->v[4];
        s->audio[s->idx] = s->b[0] * s->v[0] + s->b[1] * s->v[1] + s->b[2] * s->v[2] + s->b[3] * s->v[3] + s->b[4] * s->v[4];
        s->v[4] = s->v[3]; s->v[3] = s->v[2]; s->v[2] = s->v[1]; s->v[1] = s->v[0];
        s->idx = (s->idx + 1) % s->ring;
        if (++s->have == s->needed) {
            const double e = lr128_energy_last(s, s->s100 * 4);
            if (e >= l_hist_bound[0]) ++s->bhist[l_hist_index(e)];
            s->st_counter += s->needed;
            if (s->st_counter == s->s100 * 30) {
                const double st = lr128_energy_last(s, s->ring);
                if (st >= l_hist_bound[0]) ++s->shist[l_hist_index(st)];
                s->st_counter = s->s100 * 20;
            }
            s->needed = s->s100; s->have = 0;
        }
    }
}
static double lr128_shortterm(const lr128 *s)
{
    const double e = lr128_energy_last(s, s->ring);
    return e <= 0.0 ? -HUGE_VAL : l_e2l(e);
}
static double lr128_relthr_energy(const lr128 *s, long *above)
{
    double rel = 0.0; long cnt = 0;
    for (int j = 0; j < 1000; ++j) { rel += s->bhist[j] * l_hist_energy[j]; cnt += (long)s->bhist[j]; }
    *above = cnt;
    if (!cnt) return 0.0;
    return rel / (double)cnt * 0.1;
}
static double lr128_relative_threshold(const lr128 *s)
{
    long above; const double rel = lr128_relthr_energy(s, &above);
    return above ? l_e2l(rel) : -70.0;
}
static double lr128_global(const lr128 *s)
{
    long above; const double rel = lr128_relthr_energy(s, &above);
    if (!above) return -HUGE_VAL;
    size_t start;
    if (rel < l_hist_bound[0]) start = 0;
    else { start = l_hist_index(rel); if (rel > l_hist_energy[start]) ++start; }
    double g = 0.0; long cnt = 0;
    for (size_t j = start; j < 1000; ++j) { g += s->bhist[j] * l_hist_energy[j]; cnt += (long)s->bhist[j]; }
    return cnt ? l_e2l(g / (double)cnt) : -HUGE_VAL;
}
static double lr128_lra(const lr128 *s)
{
    size_t stl_size = 0; double stl_power = 0.0;
    for (int j = 0; j < 1000; ++j) { stl_size += s->shist[j]; stl_power += s->shist[j] * l_hist_energy[j]; }
    if (!stl_size) return 0.0;
    stl_power /= (double)stl_size;
    const double integ = 0.01 * stl_power;
    size_t index, j;
    if (integ < l_hist_bound[0]) index = 0;
    else { index = l_hist_index(integ); if (integ > l_hist_energy[index]) ++index; }
    stl_size = 0;
    for (j = index; j < 1000; ++j) stl_size += s->shist[j];
    if (!stl_size) return 0.0;
    const size_t pl = (size_t)((stl_size - 1) * 0.1 + 0.5), ph = (size_t)((stl_size - 1) * 0.95 + 0.5);
    stl_size = 0; j = index;
    while (stl_size <= pl) stl_size += s->shist[j++];
    const double l_en = l_hist_energy[j - 1];
    while (stl_size <= ph) stl_size += s->shist[j++];
    const double h_en = l_hist_energy[j - 1];
    return l_e2l(h_en) - l_e2l(l_en);
}

/* ------------------------------------------------------------------ af_loudnorm.c */
enum { LN_FIRST, LN_INNER, LN_FINAL, LN_LINEAR };
enum { LIM_OUT, LIM_ATTACK, LIM_SUSTAIN, LIM_RELEASE };
typedef struct {
    double target_i, target_lra, target_tp, measured_i, measured_lra, measured_tp, measured_thresh, offset;
    double *buf; int buf_size, buf_index, prev_buf_index;
    double delta[30], weights[21], prev_delta; int index;
    double gain_reduction[2];
    double *limiter_buf, prev_smp; int limiter_buf_index, limiter_buf_size, limiter_state, peak_index, env_index, env_cnt;
    int attack_length, release_length, frame_type, above_threshold, prev_nb_samples;
    lr128 in, out;
} LN;

static int ln_frame_size(int rate, int msec) { const int fs = (int)round((double)rate * (msec / 1000.0)); return fs + (fs % 2); }

static double ln_gaussian(const LN *s, int index)
{
    double result = 0.;
    index = index - 10 > 0 ? index - 10 : index + 20;
    for (int i = 0; i < 21; i++) result += s->delta[((index + i) < 30) ? (index + i) : (index + i - 30)] * s->weights[i];
    return result;
}

static void ln_detect_peak(LN *s, int offset, int nb_samples, int *peak_delta, double *peak_value)
{
    const double ceiling = s->target_tp;
    double *buf = s->limiter_buf;
    *peak_delta = -1;
    int index = s->limiter_buf_index + offset + 1920;
    if (index >= s->limiter_buf_size) index -= s->limiter_buf_size;
    if (s->frame_type == LN_FIRST) s->prev_smp = fabs(buf[index - 1]);
    for (int n = 0; n < nb_samples; n++) {
        const double this = fabs(buf[index < s->limiter_buf_size ? index : index - s->limiter_buf_size]);
        double next = fabs(buf[(index + 1) < s->limiter_buf_size ? (index + 1) : (index + 1 - s->limiter_buf_size)]);
        if ((s->prev_smp <= this) && (next <= this) && (this > ceiling) && (n > 0)) {
            int detected = 1;
            for (int i = 2; i < 12; i++) {
                next = fabs(buf[(index + i) < s->limiter_buf_size ? (index + i) : (index + i - s->limiter_buf_size)]);
                if (next > this) { detected = 0; break; }
            }
            if (detected) {
                const double max_peak = fabs(buf[index]);
                s->prev_smp = fabs(buf[index < s->limiter_buf_size ? index : index - s->limiter_buf_size]);
                *peak_delta = n; s->peak_index = index; *peak_value = max_peak;
                return;
            }
            /* rejected: the filter `continue`s past the prev_smp update (one channel: straight to the index advance) */
        } else {
            s->prev_smp = this;
        }
        index += 1;
        if (index >= s->limiter_buf_size) index -= s->limiter_buf_size;
    }
}

static void ln_true_peak_limiter(LN *s, double *out, int nb_samples)
{
    double *buf = s->limiter_buf;
    const double ceiling = s->target_tp;
    const int index0 = s->limiter_buf_index;
    int smp_cnt = 0, peak_delta; double peak_value;

    if (s->frame_type == LN_FIRST) {
        double max = 0.;
        for (int n = 0; n < 1920; n++) max = fabs(buf[n]) > max ? fabs(buf[n]) : max;
        if (max > ceiling) {
            s->gain_reduction[1] = ceiling / max;
            s->limiter_state = LIM_SUSTAIN;
            for (int n = 0; n < 1920; n++) buf[n] *= s->gain_reduction[1];
        }
    }
    do {
        switch (s->limiter_state) {
        case LIM_OUT:
            ln_detect_peak(s, smp_cnt, nb_samples - smp_cnt, &peak_delta, &peak_value);
            if (peak_delta != -1) {
                s->env_cnt = 0;
                smp_cnt += (peak_delta - s->attack_length);
                s->gain_reduction[0] = 1.;
                s->gain_reduction[1] = ceiling / peak_value;
                s->limiter_state = LIM_ATTACK;
                s->env_index = s->peak_index - s->attack_length;
                if (s->env_index < 0) s->env_index += s->limiter_buf_size;
                s->env_index += s->env_cnt;
                if (s->env_index > s->limiter_buf_size) s->env_index -= s->limiter_buf_size;
            } else {
                smp_cnt = nb_samples;
            }
            break;
        case LIM_ATTACK:
            for (; s->env_cnt < s->attack_length; s->env_cnt++) {
                const double env = s->gain_reduction[0] - ((double)s->env_cnt / (s->attack_length - 1) * (s->gain_reduction[0] - s->gain_reduction[1]));
                buf[s->env_index] *= env;
                s->env_index += 1;
                if (s->env_index >= s->limiter_buf_size) s->env_index -= s->limiter_buf_size;
                smp_cnt++;
                if (smp_cnt >= nb_samples) { s->env_cnt++; break; }
            }
            if (smp_cnt < nb_samples) { s->env_cnt = 0; s->attack_length = 1920; s->limiter_state = LIM_SUSTAIN; }
            break;
        case LIM_SUSTAIN:
            ln_detect_peak(s, smp_cnt, nb_samples, &peak_delta, &peak_value);
            if (peak_delta == -1) {
                s->limiter_state = LIM_RELEASE;
                s->gain_reduction[0] = s->gain_reduction[1];
                s->gain_reduction[1] = 1.;
                s->env_cnt = 0;
                break;
            } else {
                const double gain_reduction = ceiling / peak_value;
                if (gain_reduction < s->gain_reduction[1]) {
                    s->limiter_state = LIM_ATTACK;
                    s->attack_length = peak_delta;
                    if (s->attack_length <= 1) s->attack_length = 2;
                    s->gain_reduction[0] = s->gain_reduction[1];
                    s->gain_reduction[1] = gain_reduction;
                    s->env_cnt = 0;
                    break;
                }
                for (s->env_cnt = 0; s->env_cnt < peak_delta; s->env_cnt++) {
                    buf[s->env_index] *= s->gain_reduction[1];
                    s->env_index += 1;
                    if (s->env_index >= s->limiter_buf_size) s->env_index -= s->limiter_buf_size;
                    smp_cnt++;
                    if (smp_cnt >= nb_samples) { s->env_cnt++; break; }
                }
            }
            break;
        case LIM_RELEASE:
            for (; s->env_cnt < s->release_length; s->env_cnt++) {
                const double env = s->gain_reduction[0] + (((double)s->env_cnt / (s->release_length - 1)) * (s->gain_reduction[1] - s->gain_reduction[0]));
                buf[s->env_index] *= env;
                s->env_index += 1;
                if (s->env_index >= s->limiter_buf_size) s->env_index -= s->limiter_buf_size;
                smp_cnt++;
                if (smp_cnt >= nb_samples) { s->env_cnt++; break; }
            }
            if (smp_cnt < nb_samples) { s->env_cnt = 0; s->limiter_state = LIM_OUT; }
            break;
        }
    } while (smp_cnt < nb_samples);

    int index = index0;
    for (int n = 0; n < nb_samples; n++) {
        out[n] = buf[index];
        if (fabs(out[n]) > ceiling) out[n] = ceiling * (out[n] < 0 ? -1 : 1);
        index += 1;
        if (index >= s->limiter_buf_size) index -= s->limiter_buf_size;
    }
}

/* filter_frame(); returns the number of output samples written to dst */
static int ln_filter_frame(LN *s, int rate, const double *src, int nb_in, double *dst)
{
    double *buf = s->buf, *limiter_buf = s->limiter_buf;
    double global, shortterm, relative_threshold, gain, gain_next, env_global, env_shortterm;
    int nb_out = nb_in, subframe_length;

    lr128_add(&s->in, src, nb_in);          /* unconditional in the filter: the flush frame (the last 2.9 s again) is metered a second time */

    if (s->frame_type == LN_FIRST && nb_in < ln_frame_size(rate, 3000)) {
        const double g = lr128_global(&s->in), true_peak = s->in.peak;
        const double offset = pow(10., (s->target_i - g) / 20.);
        const double offset_tp = true_peak * offset;
        s->offset = offset_tp < s->target_tp ? offset : s->target_tp - true_peak;
        s->frame_type = LN_LINEAR;
    }
    switch (s->frame_type) {
    case LN_FIRST:
        for (int n = 0; n < nb_in; n++) { buf[s->buf_index] = src[n]; s->buf_index += 1; }
        shortterm = lr128_shortterm(&s->in);
        if (shortterm < s->measured_thresh) {
            s->above_threshold = 0;
            env_shortterm = shortterm <= -70. ? 0. : s->target_i - s->measured_i;
        } else {
            s->above_threshold = 1;
            env_shortterm = shortterm <= -70. ? 0. : s->target_i - shortterm;
        }
        for (int n = 0; n < 30; n++) s->delta[n] = pow(10., env_shortterm / 20.);
        s->prev_delta = s->delta[s->index];
        s->buf_index = s->limiter_buf_index = 0;
        for (int n = 0; n < s->limiter_buf_size; n++) {
            limiter_buf[s->limiter_buf_index] = buf[s->buf_index] * s->delta[s->index] * s->offset;
            s->limiter_buf_index += 1;
            if (s->limiter_buf_index == s->limiter_buf_size) s->limiter_buf_index = 0;
            s->buf_index += 1;
        }
        subframe_length = ln_frame_size(rate, 100);
        ln_true_peak_limiter(s, dst, subframe_length);
        lr128_add(&s->out, dst, subframe_length);
        nb_out = subframe_length;
        s->frame_type = LN_INNER;
        break;
    case LN_INNER:
        gain = ln_gaussian(s, s->index + 10 < 30 ? s->index + 10 : s->index + 10 - 30);
        gain_next = ln_gaussian(s, s->index + 11 < 30 ? s->index + 11 : s->index + 11 - 30);
        for (int n = 0; n < nb_in; n++) {
            buf[s->prev_buf_index] = src[n];
            limiter_buf[s->limiter_buf_index] = buf[s->buf_index] * (gain + (((double)n / nb_in) * (gain_next - gain))) * s->offset;
            s->limiter_buf_index += 1;
            if (s->limiter_buf_index == s->limiter_buf_size) s->limiter_buf_index = 0;
            s->prev_buf_index += 1;
            if (s->prev_buf_index == s->buf_size) s->prev_buf_index = 0;
            s->buf_index += 1;
            if (s->buf_index == s->buf_size) s->buf_index = 0;
        }
        subframe_length = ln_frame_size(rate, 100) - nb_in;
        s->limiter_buf_index = s->limiter_buf_index + subframe_length < s->limiter_buf_size ? s->limiter_buf_index + subframe_length
                                                                                             : s->limiter_buf_index + subframe_length - s->limiter_buf_size;
        ln_true_peak_limiter(s, dst, nb_in);
        lr128_add(&s->out, dst, nb_in);
        global = lr128_global(&s->in);
        shortterm = lr128_shortterm(&s->in);
        relative_threshold = lr128_relative_threshold(&s->in);
        if (s->above_threshold == 0) {
            if (shortterm > s->measured_thresh) s->prev_delta *= 1.0058;
            const double shortterm_out = lr128_shortterm(&s->out);
            if (shortterm_out >= s->target_i) s->above_threshold = 1;
        }
        if (shortterm < relative_threshold || shortterm <= -70. || s->above_threshold == 0) {
            s->delta[s->index] = s->prev_delta;
        } else {
            env_global = fabs(shortterm - global) < (s->target_lra / 2.) ? shortterm - global : (s->target_lra / 2.) * ((shortterm - global) < 0 ? -1 : 1);
            env_shortterm = s->target_i - shortterm;
            s->delta[s->index] = pow(10., (env_global + env_shortterm) / 20.);
        }
        s->prev_delta = s->delta[s->index];
        s->index++;
        if (s->index >= 30) s->index -= 30;
        s->prev_nb_samples = nb_in;
        break;
    case LN_FINAL: {
        gain = ln_gaussian(s, s->index + 10 < 30 ? s->index + 10 : s->index + 10 - 30);
        s->limiter_buf_index = 0;
        int src_index = 0;
        for (int n = 0; n < s->limiter_buf_size; n++) {
            s->limiter_buf[s->limiter_buf_index] = src[src_index] * gain * s->offset;
            src_index += 1;
            s->limiter_buf_index += 1;
            if (s->limiter_buf_index == s->limiter_buf_size) s->limiter_buf_index = 0;
        }
        subframe_length = ln_frame_size(rate, 100);
        double *d = dst;
        for (int i = 0; i < nb_in / subframe_length; i++) {
            ln_true_peak_limiter(s, d, subframe_length);
            for (int n = 0; n < subframe_length; n++) {
                if (src_index < nb_in) { limiter_buf[s->limiter_buf_index] = src[src_index] * gain * s->offset; src_index += 1; }
                else limiter_buf[s->limiter_buf_index] = 0.;
                s->limiter_buf_index += 1;
                if (s->limiter_buf_index == s->limiter_buf_size) s->limiter_buf_index = 0;
            }
            d += subframe_length;
        }
        lr128_add(&s->out, dst, nb_in);
        break; }
    case LN_LINEAR:
        for (int n = 0; n < nb_in; n++) dst[n] = src[n] * s->offset;
        lr128_add(&s->out, dst, nb_in);
        break;
    }
    return nb_out;
}

int64_t orc_loudnorm_dynamic_mono(const double *in, int64_t n, int rate, const orc_loudnorm_params *p, double *out, orc_loudnorm_stats *st)
{
    LN s; memset(&s, 0, sizeof(s));
    s.target_i = p->target_i; s.target_lra = p->target_lra; s.target_tp = p->target_tp;
    s.measured_i = p->measured_i; s.measured_lra = p->measured_lra; s.measured_tp = p->measured_tp; s.measured_thresh = p->measured_thresh;
    s.offset = p->offset;
    /* init(): linear only when every measured value is there and the projected peak / LRA fit */
    s.frame_type = LN_FIRST;
    if (p->linear) {
        const double offset = s.target_i - s.measured_i, offset_tp = s.measured_tp + offset;
        if (s.measured_tp != 99 && s.measured_thresh != -70 && s.measured_lra != 0 && s.measured_i != 0)
            if (offset_tp <= s.target_tp && s.measured_lra <= s.target_lra) { s.frame_type = LN_LINEAR; s.offset = offset; }
    }
    /* config_input() */
    lr128_init(&s.in, rate, p->dual_mono); lr128_init(&s.out, rate, p->dual_mono);
    s.buf_size = ln_frame_size(rate, 3000);
    s.buf = calloc((size_t)s.buf_size, sizeof(double));
    s.limiter_buf_size = ln_frame_size(rate, 210);
    s.limiter_buf = calloc((size_t)s.buf_size, sizeof(double));
    {
        double total = 0.0; const double sigma = 3.5, c1 = 1.0 / (sigma * sqrt(2.0 * M_PI)), c2 = 2.0 * pow(sigma, 2.0);
        for (int i = 0; i < 21; i++) { const int x = i - 10; s.weights[i] = c1 * exp(-(pow(x, 2.0) / c2)); total += s.weights[i]; }
        const double adjust = 1.0 / total;
        for (int i = 0; i < 21; i++) s.weights[i] *= adjust;
    }
    s.index = 1; s.limiter_state = LIM_OUT;
    s.offset = pow(10., s.offset / 20.);
    s.target_tp = pow(10., s.target_tp / 20.);
    s.attack_length = ln_frame_size(rate, 10);
    s.release_length = ln_frame_size(rate, 100);

    /* activate(): 3 s, then 100 ms frames (a shorter one at the end of the stream), then the flush frame */
    int64_t pos = 0, produced = 0;
    const int f3000 = ln_frame_size(rate, 3000), f100 = ln_frame_size(rate, 100);
    while (pos < n) {
        int want = s.frame_type == LN_FIRST ? f3000 : (s.frame_type == LN_LINEAR ? f100 : f100);
        if (want > n - pos) want = (int)(n - pos);
        produced += ln_filter_frame(&s, rate, in + pos, want, out + produced);
        pos += want;
    }
    if (s.frame_type == LN_INNER) {                                                 /* flush_frame() */
        int nb_samples = s.buf_size - s.prev_nb_samples;
        nb_samples -= (f100 - s.prev_nb_samples);
        double *src = malloc(sizeof(double) * (size_t)nb_samples);
        int offset = (s.limiter_buf_size - s.prev_nb_samples);
        offset -= (f100 - s.prev_nb_samples);
        s.buf_index = s.buf_index - offset < 0 ? s.buf_index - offset + s.buf_size : s.buf_index - offset;
        for (int k = 0; k < nb_samples; k++) {
            src[k] = s.buf[s.buf_index];
            s.buf_index += 1;
            if (s.buf_index >= s.buf_size) s.buf_index -= s.buf_size;
        }
        s.frame_type = LN_FINAL;
        produced += ln_filter_frame(&s, rate, src, nb_samples, out + produced);
        free(src);
    }
    if (st) {                                                                       /* uninit(): the stats loudnorm prints */
        st->input_i = lr128_global(&s.in); st->input_lra = lr128_lra(&s.in); st->input_thresh = lr128_relative_threshold(&s.in);
        st->input_tp = 20. * log10(s.in.peak);
        st->output_i = lr128_global(&s.out); st->output_lra = lr128_lra(&s.out); st->output_thresh = lr128_relative_threshold(&s.out);
        st->output_tp = 20. * log10(s.out.peak);
        st->dynamic = s.frame_type == LN_LINEAR ? 0 : 1;                           /* "normalization_type" */
        st->target_offset = s.target_i - st->output_i;
    }
    lr128_free(&s.in); lr128_free(&s.out); free(s.buf); free(s.limiter_buf);
    return produced;
}
