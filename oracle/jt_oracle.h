/*
 * jt_oracle.h — CPU ORACLE for the jivetalking four-pass speech-mastering path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (jivetalking_amd/,
 * include/) may link, import or call this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * What it restates.  The reference (linuxmatters/jivetalking) contains no DSP of
 * its own: internal/processor builds FFmpeg filter-graph strings
 * (filters.go:607-962, normalise.go:257-264,446-480,1231-1334,
 * analyser_bands.go:33, analyser_output.go:18) and pumps frames through
 * libavfilter via the cgo binding github.com/linuxmatters/ffmpeg-statigo
 * (go.mod:13,17 — a `replace` to an EMPTY un-vendored submodule, no version
 * pin; bundled FFmpeg = 8.1 per docs/Spectral-Metrics-Reference.md:5).  The
 * arithmetic therefore lives in a third-party dependency that is ABSENT from
 * /root/reference.  This oracle restates the published algorithms of the
 * FFmpeg 8.1 filters the reference instantiates (libavfilter/af_biquads.c,
 * af_anlmdn.c, af_afftdn.c, af_agate.c, af_sidechaincompress.c, af_deesser.c,
 * af_alimiter.c, af_loudnorm.c + ebur128.c, f_ebur128.c, af_astats.c,
 * af_aspectralstats.c, af_volume.c, libswresample/resample.c) from knowledge
 * of that source, plain sequential C, double/float exactly where FFmpeg uses
 * double/float.
 *
 * PARITY STATUS: **parity unpinned** at the FFmpeg boundary.  Neither Go nor
 * FFmpeg exists in the build container, so the restatement cannot be checked
 * against the reference executable.  It IS pinned against: (a) the in-repo
 * formula statement docs/Spectral-Metrics-Reference.md:9-56 (aspectralstats,
 * astats), (b) standards known-answer tests (ITU-R BS.1770-4 / EBU Tech
 * 3341/3342) in tests/test_oracle_kat.py, (c) the reference's own range
 * assertions (analyser_test.go:185-207).  The scalar control logic around
 * FFmpeg (VAD, AdaptConfig, limiter planning) is NOT here: it is product host
 * code checked directly against the reference's golden tables (tests/golden/).
 */
#ifndef JT_ORACLE_H
#define JT_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- biquads: highpass/lowpass, RBJ, a=tdii, flt (af_biquads.c) ---- */
/* type: 0 = highpass, 1 = lowpass.  poles=2, width_type=q, normalize=1. */
void orc_biquad_coeffs(int type, double freq, double q, int sample_rate, double b[3], double a[3]);
double orc_band_rms_db(const float *in, int64_t n, int sample_rate, double lo_hz, double hi_hz);   /* analyser_bands.go:33 */
/* the band graph in the sample format libavfilter negotiates for the source: 0 fltp, 1 s16p, 2 s32p (orc_basic.c) */
double orc_band_rms_db_fmt(const float *in, int64_t n, int sample_rate, double lo_hz, double hi_hz, int mode);
/* aformat=channel_layouts=mono of a stereo source (libswresample rematrix): 0 float 1/sqrt2, 1 s16 integer 0.5, 2 s32 via float 0.5 */
void orc_downmix_stereo(const float *in, int64_t frames, int mode, float *out);
/* any layout up to 8 channels -> mono (swr_build_matrix2's default matrix): see orc_basic.c; mask 0 = av_channel_layout_default(channels) */
uint64_t orc_default_layout(int channels);
int orc_downmix_coeffs(int channels, uint64_t mask, int normalise, double coef[8]);
int orc_downmix_layout(const float *in, int64_t frames, int channels, uint64_t mask, int mode, float *out);
void orc_biquad_tdii_f32(const float *in, float *out, int64_t n, const double b[3], const double a[3]);
/* double-precision variant (band-RMS graphs run biquads on the decoder format; dbl used for s16/flt-agnostic checks) */
void orc_biquad_tdii_f64(const double *in, double *out, int64_t n, const double b[3], const double a[3]);

/* ---- anlmdn (af_anlmdn.c) ---- */
void orc_anlmdn_f32(const float *in, float *out, int64_t n, int sample_rate,
                    double strength, double patch_s, double research_s, double smooth);

/* ---- afftdn (af_afftdn.c), tn=0 static floor ---- */
/* band_noise: 15 custom band values (dB) or NULL for white. */
void orc_afftdn_f32(const float *in, float *out, int64_t n, int sample_rate,
                    double nr_db, double nf_db, const double *band_noise);
/* the same with tn=1 (track_noise): the floor follows spectrally flat frames; floor_series (optional) = floor in dB after each frame */
void orc_afftdn_tn_f32(const float *in, float *out, int64_t n, int sample_rate,
                       double nr_db, double nf_db, const double *band_noise, int track, double *floor_series, int64_t cap);

/* ---- agate / acompressor / deesser (dbl) ---- */
typedef struct {
    double threshold, ratio, attack_ms, release_ms, range, knee, makeup;
    int detection_rms;
} orc_gate_params;
typedef struct {
    double threshold, ratio, attack_ms, release_ms, makeup, knee, mix;
    int detection_rms;
} orc_comp_params;
void orc_agate_f64(const double *in, double *out, int64_t n, int sample_rate, const orc_gate_params *p);
void orc_acompressor_f64(const double *in, double *out, int64_t n, int sample_rate, const orc_comp_params *p);
void orc_deesser_f64(const double *in, double *out, int64_t n, int sample_rate,
                     double intensity, double max_deess, double frequency);

/* ---- adeclick (af_adeclick.c): method 0 = overlap-add, 1 = overlap-save; returns -1 if a window's normal matrix is singular ---- */
int orc_adeclick_f64(const double *in, double *out, int64_t n, int sample_rate, double threshold, double window_ms,
                     double overlap_pct, double ar_pct, double burst, int method, int64_t *n_clicks_out);

/* ---- alimiter (af_alimiter.c): level=0 (no auto level), latency=1, asc on ---- */
void orc_alimiter_f64(const double *in, double *out, int64_t n, int sample_rate,
                      double limit, double attack_ms, double release_ms, double asc_level);

/* ---- libswresample polyphase resampler (resample.c), kaiser beta 9, filter_size 32, cutoff 0.97 ---- */
/* out_cap >= ceil(n*out_rate/in_rate)+2.  flush!=0: aresample-style (right edge mirrored, M=ceil(n*out/in));
 * flush==0: streaming, only outputs whose taps are inside the input (ebur128 true-peak use).  Returns #outputs. */
int64_t orc_swr_resample_f64(const double *in, int64_t n, int in_rate, int out_rate, double *out, int64_t out_cap, int flush);
int64_t orc_swr_resample_f32(const float *in, int64_t n, int in_rate, int out_rate, float *out, int64_t out_cap, int flush);
/* dbl -> s16 as swresample's audioconvert: clip_int16(lrint(x*32768)) */
void orc_f64_to_s16(const double *in, int16_t *out, int64_t n);

/* ---- ebur128 filter (f_ebur128.c) on mono ---- */
typedef struct {
    double integrated, lra, lra_low, lra_high;
    double momentary_last, shortterm_last;   /* last 100 ms block values */
    double sample_peak, true_peak;           /* linear */
    double target_threshold;                 /* rel gate threshold of I (LUFS) */
    int64_t nblocks;
} orc_ebur128_out;
/* m_series/s_series: per-100ms-block M and S (caller buffers, cap entries), may be NULL.
 * tp_series/sp_series: cumulative linear peaks at each block, may be NULL. */
void orc_ebur128_mono(const double *in, int64_t n, int sample_rate, int dualmono, int true_peak,
                      orc_ebur128_out *out, double *m_series, double *s_series,
                      double *tp_series, double *sp_series, int64_t cap);

/* ---- loudnorm input measurement: libebur128 port (ebur128.c) at the given rate, mono, dual_mono ---- */
typedef struct { double input_i, input_tp, input_lra, input_thresh; } orc_loudnorm_in;
void orc_loudnorm_measure_mono(const double *in, int64_t n, int sample_rate, int dual_mono, orc_loudnorm_in *out);

/* ---- loudnorm, dynamic mode (af_loudnorm.c) on a mono stream already at the filter's internal 192 kHz; `out` has room for n ---- */
typedef struct {
    double target_i, target_lra, target_tp;                            /* I=, LRA=, TP= (dB) */
    double measured_i, measured_lra, measured_tp, measured_thresh;     /* 0 / 0 / 99 / -70 when not given (the filter's defaults) */
    double offset;                                                     /* offset= (dB) */
    int linear, dual_mono;
} orc_loudnorm_params;
typedef struct {
    double input_i, input_tp, input_lra, input_thresh, output_i, output_tp, output_lra, output_thresh, target_offset;
    int dynamic;                                                       /* normalization_type: 1 "dynamic", 0 "linear" */
} orc_loudnorm_stats;
int64_t orc_loudnorm_dynamic_mono(const double *in, int64_t n, int rate, const orc_loudnorm_params *p, double *out, orc_loudnorm_stats *st);

/* ---- astats (af_astats.c) on mono, length=0.05 ---- */
typedef struct {
    double dc_offset, min_level, max_level, min_difference, max_difference, mean_difference,
           rms_difference, peak_level_db, rms_level_db, rms_peak_db, rms_trough_db, crest_factor,
           flat_factor, peak_count, noise_floor_db, noise_floor_count, entropy, dynamic_range,
           zero_crossings, zero_crossings_rate, number_of_samples, abs_peak_count;
} orc_astats_out;
void orc_astats_mono(const double *in, int64_t n, int sample_rate, orc_astats_out *out);

/* ---- aspectralstats (af_aspectralstats.c), hann, overlap 0.5, mono ---- */
/* stats: nhops x 13 doubles in key order mean,variance,centroid,spread,skewness,kurtosis,
 * entropy,flatness,crest,flux,slope,decrease,rolloff. Returns number of hops written. */
int64_t orc_aspectralstats_mono(const float *in, int64_t n, int sample_rate, int win_size,
                                double *stats, int64_t cap_hops);

/* ---- helpers ---- */
void orc_rfft_mag_f32(const float *in, int n_fft, float *mag_half);  /* |FFT|, n_fft/2 bins, unscaled */

/* ---- FLAC (RFC 9639): sequential decoder with CRC-8 / CRC-16 / MD5 verification, and a format-coverage encoder that makes
 * decoder test streams (orc_flac.c) ---- */
typedef struct {
    int sample_rate, channels, bps, min_blocksize, max_blocksize, min_framesize, max_framesize;
    int64_t total_samples, decoded_samples, frames, audio_offset;
    int metadata_blocks, variable_blocksize, crc8_errors, crc16_errors;
    int obs_min_framesize, obs_max_framesize, obs_max_blocksize, last_blocksize;
    uint8_t md5_stored[16], md5_decoded[16];
} orc_flac_info;
/* out: interleaved int32 (may be NULL), cap_frames inter-channel samples.  0 = stream valid (all CRCs right); <0 = error. */
int orc_flac_decode(const uint8_t *data, int64_t len, int32_t *out, int64_t cap_frames, orc_flac_info *info);
int64_t orc_flac_encode(const int32_t *pcm, int64_t nframes, int channels, int bps, int sample_rate, int blocksize, int mode,
                        int lpc_order, uint8_t *out, int64_t cap);
uint8_t orc_flac_crc8(const uint8_t *p, int64_t n);
uint16_t orc_flac_crc16(const uint8_t *p, int64_t n);
void orc_md5(const uint8_t *data, int64_t len, uint8_t out[16]);

#ifdef __cplusplus
}
#endif
#endif
