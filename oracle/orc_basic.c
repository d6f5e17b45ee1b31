/*
 * orc_basic.c — ORACLE (test infrastructure only; see jt_oracle.h).
 * Biquads (af_biquads.c), agate (af_agate.c), acompressor (af_sidechaincompress.c),
 * deesser (af_deesser.c), alimiter (af_alimiter.c) restated from FFmpeg 8.1.
 * Reference call sites: filters.go:740-769 (biquads), :869-894 (agate), :900-916
 * (acompressor), :921-932 (deesser); normalise.go:446-480 (alimiter).
 * parity unpinned (FFmpeg source/binary absent) — see jt_oracle.h.
 */
#include "jt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FFMIN(a,b) ((a) < (b) ? (a) : (b))
#define FFMAX(a,b) ((a) > (b) ? (a) : (b))

/* ------------------------------------------------------------------ biquads */
/* af_biquads.c config_filter(): RBJ cookbook, width_type=q => alpha = sin(w0)/(2Q);
 * all coefficients divided by a0; normalize=1 rescales b by (sum a)/(sum b) when |sum b|>1e-6. */
static void biquad_cookbook(int type, double freq, double q, int sample_rate, double b[3], double a[3])
{
    double w0 = 2 * M_PI * freq / sample_rate;
    double alpha = sin(w0) / (2 * q);
    a[0] = 1 + alpha;
    a[1] = -2 * cos(w0);
    a[2] = 1 - alpha;
    if (type == 0) { /* highpass, poles=2 */
        b[0] = (1 + cos(w0)) / 2;
        b[1] = -(1 + cos(w0));
        b[2] = (1 + cos(w0)) / 2;
    } else {         /* lowpass, poles=2 */
        b[0] = (1 - cos(w0)) / 2;
        b[1] = 1 - cos(w0);
        b[2] = (1 - cos(w0)) / 2;
    }
    a[1] /= a[0]; a[2] /= a[0];
    b[0] /= a[0]; b[1] /= a[0]; b[2] /= a[0];
    a[0] /= a[0];
}

void orc_biquad_coeffs(int type, double freq, double q, int sample_rate, double b[3], double a[3])
{
    biquad_cookbook(type, freq, q, sample_rate, b, a);
    if (fabs(b[0] + b[1] + b[2]) > 1e-6) {
        double factor = (a[0] + a[1] + a[2]) / (b[0] + b[1] + b[2]);
        b[0] *= factor; b[1] *= factor; b[2] *= factor;
    }
}

/* Band RMS of a region (analyser_bands.go:33, analyser_noise_bands.go:65-119): the graph
 * "atrim=...,highpass=f=lo:p=2,lowpass=f=hi:p=2,astats=...measure_perchannel=0" with af_biquads.c's defaults: width_type q = 0.707,
 * normalize = 0, transform di.  BIQUAD_FILTER(flt): float coefficients and state, one expression per sample
 *   o = i2*b2 + i1*b1 + in*b0 + o2*a2 + o1*a1   (a1, a2 negated at configuration)
 * evaluated left to right; astats' Overall RMS_level = 20 log10 sqrt(sum(x^2) / n) with the sum in double. */
double orc_band_rms_db(const float *in, int64_t n, int sample_rate, double lo_hz, double hi_hz)
{
    double bh[3], ah[3], bl[3], al[3];
    biquad_cookbook(0, lo_hz, 0.707, sample_rate, bh, ah);
    biquad_cookbook(1, hi_hz, 0.707, sample_rate, bl, al);
    const float hb0 = (float)bh[0], hb1 = (float)bh[1], hb2 = (float)bh[2], ha1 = -(float)ah[1], ha2 = -(float)ah[2];
    const float lb0 = (float)bl[0], lb1 = (float)bl[1], lb2 = (float)bl[2], la1 = -(float)al[1], la2 = -(float)al[2];
    float i1 = 0, i2 = 0, o1 = 0, o2 = 0, j1 = 0, j2 = 0, p1 = 0, p2 = 0;
    double acc = 0.0;
    for (int64_t k = 0; k < n; k++) {
        const float x = in[k];
        const float y = (((i2 * hb2 + i1 * hb1) + x * hb0) + o2 * ha2) + o1 * ha1;
        i2 = i1; i1 = x; o2 = o1; o1 = y;
        const float z = (((j2 * lb2 + j1 * lb1) + y * lb0) + p2 * la2) + p1 * la1;
        j2 = j1; j1 = y; p2 = p1; p1 = z;
        acc += (double)z * (double)z;
    }
    return n > 0 ? 20.0 * log10(sqrt(acc / (double)n)) : -INFINITY;
}

/* The same band graph when libavfilter negotiates an INTEGER sample format for it.  Nothing in
 *   aformat=channel_layouts=mono,atrim,asetpts,highpass,lowpass,astats      (analyser_bands.go:33)
 * accepts only float, so the link formats are the intersection of abuffer's native decoder format with af_biquads.c's list
 * {s16p, s32p, fltp, dblp}: a 16-bit FLAC/WAV source runs the biquads in s16p, a 24/32-bit one in s32p (avfiltergraph.c
 * swap_sample_fmts keeps the source's width; DESIGN.md section 3 has the derivation).  af_biquads.c then instantiates
 *   BIQUAD_FILTER(s16, int16_t, float,  INT16_MIN, INT16_MAX, 1)   BIQUAD_FILTER(s32, int32_t, double, INT32_MIN, INT32_MAX, 1)
 * i.e. float (s16) / double (s32) coefficients and state working on the raw integer values, the recursion fed with the
 * UNQUANTISED o1/o2, and every stage's output stored by a C cast (truncation toward zero) after clipping to the integer range.
 * af_astats.c normalises integer samples by INT16_MAX / INT32_MAX (not 2^15 / 2^31) before squaring.
 * mode: 0 = fltp (orc_band_rms_db), 1 = s16p, 2 = s32p.  `in` holds the mono region as float = integer * 2^(1-bits). */
double orc_band_rms_db_fmt(const float *in, int64_t n, int sample_rate, double lo_hz, double hi_hz, int mode)
{
    if (mode == 0) return orc_band_rms_db(in, n, sample_rate, lo_hz, hi_hz);
    double bh[3], ah[3], bl[3], al[3];
    biquad_cookbook(0, lo_hz, 0.707, sample_rate, bh, ah);
    biquad_cookbook(1, hi_hz, 0.707, sample_rate, bl, al);
    double acc = 0.0;
    if (mode == 1) {
        const float hb0 = (float)bh[0], hb1 = (float)bh[1], hb2 = (float)bh[2], ha1 = -(float)ah[1], ha2 = -(float)ah[2];
        const float lb0 = (float)bl[0], lb1 = (float)bl[1], lb2 = (float)bl[2], la1 = -(float)al[1], la2 = -(float)al[2];
        float i1 = 0, i2 = 0, o1 = 0, o2 = 0, j1 = 0, j2 = 0, p1 = 0, p2 = 0;
        for (int64_t k = 0; k < n; k++) {
            const float x = (float)(int16_t)lrintf(in[k] * 32768.0f);
            const float y = (((i2 * hb2 + i1 * hb1) + x * hb0) + o2 * ha2) + o1 * ha1;
            i2 = i1; i1 = x; o2 = o1; o1 = y;
            const int16_t q1 = y < (float)INT16_MIN ? INT16_MIN : (y > (float)INT16_MAX ? INT16_MAX : (int16_t)y);
            const float x2 = (float)q1;
            const float z = (((j2 * lb2 + j1 * lb1) + x2 * lb0) + p2 * la2) + p1 * la1;
            j2 = j1; j1 = x2; p2 = p1; p1 = z;
            const int16_t q2 = z < (float)INT16_MIN ? INT16_MIN : (z > (float)INT16_MAX ? INT16_MAX : (int16_t)z);
            const double nd = (double)q2 / (double)INT16_MAX;
            acc += nd * nd;
        }
    } else {
        const double hb0 = bh[0], hb1 = bh[1], hb2 = bh[2], ha1 = -ah[1], ha2 = -ah[2];
        const double lb0 = bl[0], lb1 = bl[1], lb2 = bl[2], la1 = -al[1], la2 = -al[2];
        double i1 = 0, i2 = 0, o1 = 0, o2 = 0, j1 = 0, j2 = 0, p1 = 0, p2 = 0;
        for (int64_t k = 0; k < n; k++) {
            double xs = (double)in[k] * 2147483648.0;
            if (xs > (double)INT32_MAX) xs = (double)INT32_MAX;
            const double x = (double)(int32_t)llrint(xs);
            const double y = (((i2 * hb2 + i1 * hb1) + x * hb0) + o2 * ha2) + o1 * ha1;
            i2 = i1; i1 = x; o2 = o1; o1 = y;
            const int32_t q1 = y < (double)INT32_MIN ? INT32_MIN : (y > (double)INT32_MAX ? INT32_MAX : (int32_t)y);
            const double x2 = (double)q1;
            const double z = (((j2 * lb2 + j1 * lb1) + x2 * lb0) + p2 * la2) + p1 * la1;
            j2 = j1; j1 = x2; p2 = p1; p1 = z;
            const int32_t q2 = z < (double)INT32_MIN ? INT32_MIN : (z > (double)INT32_MAX ? INT32_MAX : (int32_t)z);
            const double nd = (double)q2 / (double)INT32_MAX;
            acc += nd * nd;
        }
    }
    return n > 0 ? 20.0 * log10(sqrt(acc / (double)n)) : -INFINITY;
}

/* aformat=channel_layouts=mono on a stereo source = the aresample libavfilter inserts in front of it; libswresample/rematrix.c:
 * swr_build_matrix2 gives FRONT_CENTER <- M_SQRT1_2 * FL + M_SQRT1_2 * FR, and auto_matrix normalises the row to sum 1
 * (maxval = 1.0) ONLY when the converter's OUTPUT or INTERNAL sample format is an integer one; for float/double internal +
 * output formats maxval = INT_MAX and the coefficients stay 1/sqrt(2).  swr_init picks the internal format from the byte
 * widths: S16P when input and output are both <= 2 bytes, else FLTP for inputs up to 4 bytes, else DBLP.
 *   mode 0: output fltp (every graph with a float-only filter behind the down-mix: Pass 1, Pass 2 -- DESIGN.md section 3):
 *           internal FLTP, float coefficients (float)M_SQRT1_2, mix_2_1 = fl(fl(c*L) + fl(c*R))      [swresample rematrix_template.c]
 *   mode 1: s16 source in the band graph (output s16p): internal S16P, integer coefficients lrintf(0.5 * 32768) = 16384,
 *           sum2_s16: (L*16384 + R*16384 + 16384) >> 15  (arithmetic shift)
 *   mode 2: s32 source in the band graph (output s32p): internal FLTP (input is 4 bytes wide), normalised float coefficients 0.5,
 *           then flt -> s32 by audioconvert: av_clipl_int32(llrintf(x * 2^31))
 * in: interleaved stereo as float = integer * 2^(1-bits) (or the float samples themselves); out: mono in the same scaling. */
void orc_downmix_stereo(const float *in, int64_t frames, int mode, float *out)
{
    const float c = (float)M_SQRT1_2;
    for (int64_t i = 0; i < frames; i++) {
        const float l = in[2 * i], r = in[2 * i + 1];
        if (mode == 0) {
            const float a = c * l, b = c * r;
            out[i] = a + b;
        } else if (mode == 1) {
            const int li = (int)lrintf(l * 32768.0f), ri = (int)lrintf(r * 32768.0f);
            const int m = (li * 16384 + ri * 16384 + 16384) >> 15;
            out[i] = (float)m * (1.0f / 32768.0f);
        } else {
            const float a = 0.5f * l, b = 0.5f * r;
            const float s = a + b;
            double v = (double)s * 2147483648.0;
            long long q = llrint(v);
            if (q > INT32_MAX) q = INT32_MAX;
            if (q < INT32_MIN) q = INT32_MIN;
            out[i] = (float)((double)q / 2147483648.0);
        }
    }
}

/* aformat=channel_layouts=mono on ANY source layout (filters.go:607-615): libswresample/rematrix.c swr_build_matrix2 with the default
 * mix levels (center_mix_level = surround_mix_level = M_SQRT1_2, lfe_mix_level = 0, no matrix encoding) and FRONT_CENTER as the one
 * output channel:
 *   FC <- 1.0 * FC                      (identity; with FL / FR present the entry is overwritten by center_mix_level * sqrt(2))
 *   FC <- M_SQRT1_2 * (FL + FR)         ("unaccounted & AV_CH_LAYOUT_STEREO")
 *   FC <- surround_mix_level * M_SQRT1_2 * BC, likewise each of BL / BR and SL / SR     (the branches for an output without back /
 *         side / front-left channels)
 *   FC <- M_SQRT1_2 * (FLC + FRC),  FC <- lfe_mix_level (0) * LFE
 * then auto_matrix's normalisation: the row is divided by the sum of its |coefficients| when that sum exceeds maxval = 1.0, which is
 * the case ONLY when the converter's output or internal sample format is an integer one (as orc_downmix_stereo above; float graphs
 * keep the raw coefficients).  A source without a layout (PCM WAV without a channel mask) gets av_channel_layout_default(channels)
 * in swr_init: 2.1, 4.0, 5.0(back), 5.1(back), 6.1, 7.1 for 3 .. 8 channels.  Channels arrive in native order (ascending mask bit).
 * swri_rematrix for one output channel with k non-zero coefficients: k = 1 copy / mix_1_1, k = 2 mix_2_1 (in1 * c1 + in2 * c2),
 * k >= 3 the generic loop  v = 0; v += in_j * matrix_flt[j] (float products, float sums, channel order)  or, in S16P,
 * v += in_j * matrix32[j]; out = (v + 16384) >> 15 with matrix32 = lrintf(coefficient * 32768) (mix_2_1's native_matrix carries
 * lrintf's remainder from one coefficient to the next; with two equal halves there is none).
 * mask 0 = the default layout of `channels`.  modes as orc_downmix_stereo.  Returns 0, or -1 for a layout this restatement does not
 * cover (channels beyond SIDE_RIGHT, or a channel count that does not match the mask). */
uint64_t orc_default_layout(int channels)
{
    static const uint64_t def[9] = {0, 0x4, 0x3, 0xB, 0x107, 0x37, 0x3F, 0x70F, 0x63F};
    return channels >= 1 && channels <= 8 ? def[channels] : 0;
}
int orc_downmix_coeffs(int channels, uint64_t mask, int normalise, double coef[8])
{
    if (!mask) mask = orc_default_layout(channels);
    if (!mask || (mask >> 11)) return -1;
    int nb = 0; for (int b = 0; b < 11; b++) nb += (int)((mask >> b) & 1);
    if (nb != channels || channels > 8) return -1;
    const int stereo = (mask & 3) != 0;
    int c = 0; double sum = 0.0;
    for (int b = 0; b < 11; b++) {
        if (!((mask >> b) & 1)) continue;
        double v;
        switch (b) {
        case 0: case 1: v = M_SQRT1_2; break;                                   /* FL FR */
        case 2: v = stereo ? M_SQRT1_2 * sqrt(2.0) : 1.0; break;                /* FC */
        case 3: v = 0.0; break;                                                 /* LFE */
        case 6: case 7: v = M_SQRT1_2; break;                                   /* FLC FRC */
        default: v = M_SQRT1_2 * M_SQRT1_2; break;                              /* BL BR BC SL SR */
        }
        coef[c++] = v; sum += fabs(v);
    }
    if (normalise && sum > 1.0) for (int i = 0; i < channels; i++) coef[i] /= sum;
    return 0;
}
int orc_downmix_layout(const float *in, int64_t frames, int channels, uint64_t mask, int mode, float *out)
{
    double coef[8];
    if (orc_downmix_coeffs(channels, mask, mode != 0, coef) != 0) return -1;
    int nz[8], k = 0;
    for (int c = 0; c < channels; c++) if (coef[c] != 0.0) nz[k++] = c;
    float cf[8]; int ci[8]; double rem = 0.0;
    for (int j = 0; j < k; j++) cf[j] = (float)coef[nz[j]];
    if (k == 2) {                     /* native_matrix of mix_2_1 (swri_rematrix_init): the remainder is carried over ALL inputs of the row */
        for (int c = 0; c < channels; c++) { const double target = coef[c] * 32768 + rem; const int q = (int)lrintf((float)target); rem += target - q; for (int j = 0; j < k; j++) if (nz[j] == c) ci[j] = q; }
    } else for (int j = 0; j < k; j++) ci[j] = (int)lrintf((float)(coef[nz[j]] * 32768));
    for (int64_t i = 0; i < frames; i++) {
        const float *x = in + i * channels;
        if (mode == 1) {
            int v = 0;
            if (k == 1) v = ((int)lrintf(x[nz[0]] * 32768.0f) * ci[0] + 16384) >> 15;
            else { for (int j = 0; j < k; j++) v += (int)lrintf(x[nz[j]] * 32768.0f) * ci[j]; v = (v + 16384) >> 15; }
            out[i] = (float)(int16_t)v * (1.0f / 32768.0f);
            continue;
        }
        float v;
        if (k == 1) v = cf[0] == 1.0f ? x[nz[0]] : x[nz[0]] * cf[0];
        else if (k == 2) { const float a = x[nz[0]] * cf[0], b = x[nz[1]] * cf[1]; v = a + b; }
        else { v = 0.0f; for (int j = 0; j < k; j++) { const float p = x[nz[j]] * cf[j]; v = v + p; } }
        if (mode == 0) { out[i] = v; continue; }
        double d = (double)v * 2147483648.0;
        long long q = llrint(d);
        if (q > INT32_MAX) q = INT32_MAX;
        if (q < INT32_MIN) q = INT32_MIN;
        out[i] = (float)((double)q / 2147483648.0);
    }
    return 0;
}

/* BIQUAD_TDII_FILTER(flt, float, float, ...): float coefficients, float state. */
void orc_biquad_tdii_f32(const float *in, float *out, int64_t n, const double b[3], const double a[3])
{
    float a1 = -(float)a[1], a2 = -(float)a[2];
    float b0 = (float)b[0], b1 = (float)b[1], b2 = (float)b[2];
    float w1 = 0.f, w2 = 0.f;
    for (int64_t i = 0; i < n; i++) {
        float x = in[i];
        float y = b0 * x + w1;
        w1 = b1 * x + w2 + a1 * y;
        w2 = b2 * x + a2 * y;
        out[i] = y;             /* mix = 1: out*wet + in*dry with wet=1, dry=0 */
    }
}

void orc_biquad_tdii_f64(const double *in, double *out, int64_t n, const double b[3], const double a[3])
{
    double a1 = -a[1], a2 = -a[2], b0 = b[0], b1 = b[1], b2 = b[2];
    double w1 = 0., w2 = 0.;
    for (int64_t i = 0; i < n; i++) {
        double x = in[i];
        double y = b0 * x + w1;
        w1 = b1 * x + w2 + a1 * y;
        w2 = b2 * x + a2 * y;
        out[i] = y;
    }
}

/* ------------------------------------------------- hermite_interpolation.h */
static double hermite_interpolation(double x, double x0, double x1,
                                    double p0, double p1, double m0, double m1)
{
    double width = x1 - x0;
    double t = (x - x0) / width;
    double t2, t3, ct0, ct1, ct2, ct3;
    m0 *= width;
    m1 *= width;
    t2 = t * t;
    t3 = t2 * t;
    ct0 = p0;
    ct1 = m0;
    ct2 = -3 * p0 - 2 * m0 + 3 * p1 - m1;
    ct3 = 2 * p0 + m0 - 2 * p1 + m1;
    return ct3 * t3 + ct2 * t2 + ct1 * t + ct0;
}

#define FAKE_INFINITY (65536.0 * 65536.0)
#define IS_FAKE_INFINITY(v) (fabs((v) - FAKE_INFINITY) < 1.0)

/* -------------------------------------------------------------------- agate */
/* af_agate.c: mode=downward, link=average (mono), level_in=level_sc=1. */
static double gate_output_gain(double lin_slope, double ratio, double thres, double knee,
                               double knee_start, double knee_stop, double range)
{
    double slope = log(lin_slope);
    double tratio = ratio;
    double gain, delta;
    if (IS_FAKE_INFINITY(ratio))
        tratio = 1000.;
    gain = (slope - thres) * tratio + thres;
    delta = tratio;
    if (knee > 1. && slope > knee_start)
        gain = hermite_interpolation(slope, knee_start, knee_stop,
                                     ((knee_start - thres) * tratio + thres), knee_stop, delta, 1.);
    return FFMAX(range, exp(gain - slope));
}

void orc_agate_f64(const double *in, double *out, int64_t n, int sample_rate, const orc_gate_params *p)
{
    double lin_threshold = p->threshold;
    double lin_knee_sqrt = sqrt(p->knee);
    if (p->detection_rms)
        lin_threshold *= lin_threshold;
    double attack_coeff  = FFMIN(1., 1. / (p->attack_ms * sample_rate / 4000.));
    double release_coeff = FFMIN(1., 1. / (p->release_ms * sample_rate / 4000.));
    double lin_knee_stop  = lin_threshold * lin_knee_sqrt;
    double lin_knee_start = lin_threshold / lin_knee_sqrt;
    double thres = log(lin_threshold);
    double knee_start = log(lin_knee_start);
    double knee_stop = log(lin_knee_stop);
    double lin_slope = 0.;
    for (int64_t i = 0; i < n; i++) {
        double abs_sample = fabs(in[i]), gain = 1.0;
        if (p->detection_rms)
            abs_sample *= abs_sample;
        lin_slope += (abs_sample - lin_slope) * (abs_sample > lin_slope ? attack_coeff : release_coeff);
        int detected = lin_slope < lin_knee_stop;
        if (lin_slope > 0.0 && detected)
            gain = gate_output_gain(lin_slope, p->ratio, thres, p->knee, knee_start, knee_stop, p->range);
        out[i] = in[i] * (1.0 * gain * p->makeup);
    }
}

/* -------------------------------------------------------------- acompressor */
static double comp_output_gain(double lin_slope, double ratio, double thres, double knee,
                               double knee_start, double knee_stop,
                               double compressed_knee_start, double compressed_knee_stop, int detection)
{
    double slope = log(lin_slope);
    double gain, delta;
    (void)compressed_knee_start;
    if (detection)
        slope *= 0.5;
    if (IS_FAKE_INFINITY(ratio)) {
        gain = thres;
        delta = 0.0;
    } else {
        gain = (slope - thres) / ratio + thres;
        delta = 1.0 / ratio;
    }
    if (knee > 1.0 && slope < knee_stop)
        gain = hermite_interpolation(slope, knee_start, knee_stop,
                                     knee_start, compressed_knee_stop, 1.0, delta);
    return exp(gain - slope);
}

void orc_acompressor_f64(const double *in, double *out, int64_t n, int sample_rate, const orc_comp_params *p)
{
    double thres = log(p->threshold);
    double lin_knee_start = p->threshold / sqrt(p->knee);
    double lin_knee_stop  = p->threshold * sqrt(p->knee);
    double adj_knee_start = lin_knee_start * lin_knee_start;
    double knee_start = log(lin_knee_start);
    double knee_stop  = log(lin_knee_stop);
    double compressed_knee_start = (knee_start - thres) / p->ratio + thres;
    double compressed_knee_stop  = (knee_stop - thres) / p->ratio + thres;
    double attack_coeff  = FFMIN(1., 1. / (p->attack_ms * sample_rate / 4000.));
    double release_coeff = FFMIN(1., 1. / (p->release_ms * sample_rate / 4000.));
    double lin_slope = 0.;
    for (int64_t i = 0; i < n; i++) {
        double abs_sample = fabs(in[i]), gain = 1.0;
        if (p->detection_rms)
            abs_sample *= abs_sample;
        lin_slope += (abs_sample - lin_slope) * (abs_sample > lin_slope ? attack_coeff : release_coeff);
        double detector = p->detection_rms ? adj_knee_start : lin_knee_start;
        int detected = lin_slope > detector;
        if (lin_slope > 0.0 && detected)
            gain = comp_output_gain(lin_slope, p->ratio, thres, p->knee, knee_start, knee_stop,
                                    compressed_knee_start, compressed_knee_stop, p->detection_rms);
        out[i] = in[i] * 1.0 * (gain * p->makeup * p->mix + (1. - p->mix));
    }
}

/* ------------------------------------------------------------------ deesser */
void orc_deesser_f64(const double *in, double *out, int64_t n, int sample_rate,
                     double intensity_opt, double max_opt, double frequency_opt)
{
    double s1 = 0, s2 = 0, s3 = 0, m1, m2;
    double ratioA = 1.0, ratioB = 1.0, iirSampleA = 0, iirSampleB = 0;
    int flip = 0;
    double overallscale = sample_rate < 44100 ? 44100.0 / sample_rate : sample_rate / 44100.0;
    double intensity = pow(intensity_opt, 5) * (8192 / overallscale);
    double maxdess = 1.0 / pow(10.0, ((max_opt - 1.0) * 48.0) / 20);
    double iirAmount = pow(frequency_opt, 2) / overallscale;
    for (int64_t i = 0; i < n; i++) {
        double sample = in[i];
        double offset, sense, recovery, attackspeed;
        s3 = s2; s2 = s1; s1 = sample;
        m1 = (s1 - s2) * ((s1 - s2) / 1.3);
        m2 = (s2 - s3) * ((s1 - s2) / 1.3);
        sense = (m1 - m2) * ((m1 - m2) / 1.3);
        attackspeed = 7.0 + sense * 1024;
        sense = 1.0 + intensity * intensity * sense;
        sense = FFMIN(sense, intensity);
        recovery = 1.0 + (0.01 / sense);
        offset = 1.0 - fabs(sample);
        if (flip) {
            iirSampleA = (iirSampleA * (1.0 - (offset * iirAmount))) + (sample * (offset * iirAmount));
            if (ratioA < sense)
                ratioA = ((ratioA * attackspeed) + sense) / (attackspeed + 1.0);
            else
                ratioA = 1.0 + ((ratioA - 1.0) / recovery);
            ratioA = FFMIN(ratioA, maxdess);
            sample = iirSampleA + ((sample - iirSampleA) / ratioA);
        } else {
            iirSampleB = (iirSampleB * (1.0 - (offset * iirAmount))) + (sample * (offset * iirAmount));
            if (ratioB < sense)
                ratioB = ((ratioB * attackspeed) + sense) / (attackspeed + 1.0);
            else
                ratioB = 1.0 + ((ratioB - 1.0) / recovery);
            ratioB = FFMIN(ratioB, maxdess);
            sample = iirSampleB + ((sample - iirSampleB) / ratioB);
        }
        flip = !flip;
        out[i] = sample;
    }
}

/* ----------------------------------------------------------------- alimiter */
/* af_alimiter.c, mono, level_in=level_out=1, level=0 (auto_level off), asc=1,
 * latency=1 (output aligned with input: first buffer_size-1 outputs trimmed, tail flushed with zeros). */
typedef struct {
    double limit, release, att, asc, asc_coeff, delta;
    int asc_c, asc_pos, asc_changed, auto_release;
    double *buffer, *nextdelta;
    int *nextpos;
    int buffer_size, pos, nextiter, nextlen;
    int sample_rate;
} Limiter;

static double get_rdelta(Limiter *s, double release, int sample_rate,
                         double peak, double limit, double patt, int asc)
{
    double rdelta = (1.0 - patt) / (sample_rate * release);
    (void)peak;
    if (asc && s->auto_release && s->asc_c > 0) {
        double a_att = limit / (s->asc_coeff * s->asc) * (double)s->asc_c;
        if (a_att > patt) {
            double delta = FFMAX((a_att - patt) / (sample_rate * release), rdelta / 10);
            if (delta < rdelta)
                rdelta = delta;
        }
    }
    return rdelta;
}

static double limiter_step(Limiter *s, double x)
{
    const int channels = 1;
    const int buffer_size = s->buffer_size;
    double *buffer = s->buffer, *nextdelta = s->nextdelta;
    int *nextpos = s->nextpos;
    const double limit = s->limit, release = s->release;
    double peak = 0, out;
    int i;

    buffer[s->pos] = x;
    peak = FFMAX(peak, fabs(x));

    if (s->auto_release && peak > limit) {
        s->asc += peak;
        s->asc_c++;
    }

    if (peak > limit) {
        double patt = FFMIN(limit / peak, 1.);
        double rdelta = get_rdelta(s, release, s->sample_rate, peak, limit, patt, 0);
        double delta = (limit / peak - s->att) / buffer_size * channels;
        int found = 0;

        if (delta < s->delta) {
            s->delta = delta;
            nextpos[0] = s->pos;
            nextpos[1] = -1;
            nextdelta[0] = rdelta;
            s->nextlen = 1;
            s->nextiter = 0;
        } else {
            for (i = s->nextiter; i < s->nextiter + s->nextlen; i++) {
                int j = i % buffer_size;
                double ppeak = 0, pdelta;
                ppeak = FFMAX(ppeak, fabs(buffer[nextpos[j]]));
                pdelta = (limit / peak - limit / ppeak) /
                         (((buffer_size - nextpos[j] + s->pos) % buffer_size) / channels);
                if (pdelta < nextdelta[j]) {
                    nextdelta[j] = pdelta;
                    found = 1;
                    break;
                }
            }
            if (found) {
                s->nextlen = i - s->nextiter + 1;
                nextpos[(s->nextiter + s->nextlen) % buffer_size] = s->pos;
                nextdelta[(s->nextiter + s->nextlen) % buffer_size] = rdelta;
                nextpos[(s->nextiter + s->nextlen + 1) % buffer_size] = -1;
                s->nextlen++;
            }
        }
    }

    {
        double *buf = &s->buffer[(s->pos + channels) % buffer_size];
        peak = FFMAX(0, fabs(buf[0]));

        if (s->pos == s->asc_pos && !s->asc_changed)
            s->asc_pos = -1;

        if (s->auto_release && s->asc_pos == -1 && peak > limit) {
            s->asc -= peak;
            s->asc_c--;
        }

        s->att += s->delta;
        out = buf[0] * s->att;
    }

    if ((s->pos + channels) % buffer_size == nextpos[s->nextiter]) {
        if (s->auto_release) {
            s->delta = get_rdelta(s, release, s->sample_rate, peak, limit, s->att, 1);
            if (s->nextlen > 1) {
                double ppeak = 0, pdelta;
                int pnextpos = nextpos[(s->nextiter + 1) % buffer_size];
                ppeak = FFMAX(ppeak, fabs(buffer[pnextpos]));
                pdelta = (limit / ppeak - s->att) /
                         (((buffer_size + pnextpos - ((s->pos + channels) % buffer_size)) % buffer_size) / channels);
                if (pdelta < s->delta)
                    s->delta = pdelta;
            }
        } else {
            s->delta = nextdelta[s->nextiter];
            s->att = limit / peak;
        }
        s->nextlen -= 1;
        nextpos[s->nextiter] = -1;
        s->nextiter = (s->nextiter + 1) % buffer_size;
    }

    if (s->att > 1.) {
        s->att = 1.;
        s->delta = 0.;
        s->nextiter = 0;
        s->nextlen = 0;
        nextpos[0] = -1;
    }
    if (s->att <= 0.) {
        s->att = 0.0000000000001;
        s->delta = (1.0 - s->att) / (s->sample_rate * release);
    }
    if (s->att != 1. && (1. - s->att) < 0.0000000000001)
        s->att = 1.;
    if (s->delta != 0. && fabs(s->delta) < 0.00000000000001)
        s->delta = 0.;

    out = (out < -limit ? -limit : (out > limit ? limit : out)) * 1.0 /* level */ * 1.0 /* level_out */;
    s->pos = (s->pos + channels) % buffer_size;
    return out;
}

void orc_alimiter_f64(const double *in, double *out, int64_t n, int sample_rate,
                      double limit, double attack_ms, double release_ms, double asc_level)
{
    Limiter s;
    memset(&s, 0, sizeof(s));
    double attack = attack_ms / 1000., release = release_ms / 1000.;
    s.limit = limit; s.release = release; s.att = 1.; s.asc_pos = -1;
    s.asc_coeff = pow(0.5, asc_level - 0.5) * 2 * -1;   /* af_alimiter.c init() */
    s.auto_release = 1;
    s.sample_rate = sample_rate;
    int obuffer_size = (int)(sample_rate * 1 * 100 / 1000. + 1);
    s.buffer = calloc(obuffer_size, sizeof(double));
    s.nextdelta = calloc(obuffer_size, sizeof(double));
    s.nextpos = malloc(obuffer_size * sizeof(int));
    for (int i = 0; i < obuffer_size; i++) s.nextpos[i] = -1;
    s.buffer_size = (int)(sample_rate * attack * 1);
    if (s.buffer_size < 1) s.buffer_size = 1;
    int trim = s.buffer_size - 1;   /* in_trim = out_pad = buffer_size/channels - 1 */
    for (int64_t i = 0; i < n + trim; i++) {
        double x = i < n ? in[i] : 0.0;
        double y = limiter_step(&s, x);
        if (i >= trim)
            out[i - trim] = y;
    }
    free(s.buffer); free(s.nextdelta); free(s.nextpos);
}
