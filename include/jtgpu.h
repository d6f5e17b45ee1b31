/*
 * jtgpu.h — C ABI of libjtgpu.so: the MI355X (gfx950) engine for jivetalking's four-pass
 * speech-mastering hot path.
 *
 * Boundary.  The reference has no plugin ABI; its engine seam is
 *   setupFilterGraph(decCtx, spec) + runFilterGraph(ctx, reader, src, sink, FrameLoopConfig)
 *   (internal/processor/frame_processor.go:64,164; abstracted as loudnormDeps,
 *   normalise.go:172-188)
 * i.e. "an FFmpeg filter-spec string in, frames carrying lavfi.* metadata + a loudnorm JSON out".
 * This library replaces the four runFilterGraph sweeps and the band/region re-measures; PCM
 * decode/encode stays on the host side.  Every entry point below names the reference
 * interface it replaces.  Plain C types only; one handle per worker goroutine / GPU stream;
 * every call returns 0 on success or a negative JT_E_* code, with text in jt_last_error().
 *
 * Values are returned as raw doubles.  FFmpeg hands the reference the same quantities as
 * formatted strings (astats "%f", ebur128 "%.3f", loudnorm "%.2f"); the host mirror
 * (jt_host.h) applies those roundings where the reference's control flow depends on them.
 */
#ifndef JTGPU_H
#define JTGPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define JT_OK               0
#define JT_E_INVAL         -1   /* bad argument */
#define JT_E_NOGPU         -2   /* no HIP device / device init failed */
#define JT_E_HIP           -3   /* HIP runtime error (text in jt_last_error) */
#define JT_E_STATE         -4   /* call out of order (e.g. pass2 before upload) */
#define JT_E_UNSUPPORTED   -5   /* parameter combination not implemented on the GPU path */
#define JT_E_CANCELLED     -6   /* jt_cancel() observed (ctx.Err() analogue, frame_processor.go:116-118) */
#define JT_E_SILENT        -7   /* "cannot normalise silent audio" (normalise.go:840-842) */

typedef struct jt_ctx jt_ctx;

/* ---- lifecycle ---- */
int  jt_device_count(void);                       /* HIP devices visible to this process (0 when there is none or HIP cannot initialise) */
int  jt_open(int device_id, jt_ctx **out);
/* jt_open for a handle that shares its GPU with other handles (pool.go:122-153: a bounded number of ProcessAudio calls in flight).
 * n_streams: 0 (= jt_open) or >= 8: a HIP stream per chain (fastest for ONE file at a time); 1: every chain on one stream; 2: main +
 * one low-priority stream; 3 .. 7 are taken as 2 (there is no layout between "two" and "one per chain"); negative: JT_E_INVAL.  The runtime multiplexes all streams of a process onto a handful of hardware queues, and one file's event
 * waits stall every stream that shares a queue with them: several handles on one GPU want ONE stream each (ten-minute files, eight in
 * flight: 10 ms per file against 15-100 with eight streams each).  flags: JT_OPEN_BLOCKING_SYNC = host waits POLL the
 * completion event (hipEventQuery: a few yields, then 50 us sleeps) instead of spinning on the completion signal inside the runtime --
 * every wait of the HIP runtime on this part spins, with or without hipEventBlockingSync, so a pool otherwise burns a host core per
 * handle; a wait ends at most 50 us late.  Results are identical either way. */
#define JT_OPEN_BLOCKING_SYNC 0x1
int  jt_open_ex(int device_id, int n_streams, int flags, jt_ctx **out);
void jt_close(jt_ctx *h);
const char *jt_last_error(const jt_ctx *h);
const char *jt_version(void);
/* ctx cancellation (frame_processor.go:116-118: ctx.Err() is checked per frame and is sticky for the whole ProcessAudio call).
 * jt_cancel may be called from another thread at any time; every pass / measurement on the handle then returns JT_E_CANCELLED
 * (a pass already queued on the GPU returns it when its kernels have drained, a few tens of ms) until a NEW JOB starts:
 * jt_upload_pcm / jt_attach_device_pcm / jt_load_audio / jt_process_file clear the flag, or jt_reset_cancel does.  The passes
 * themselves never clear it, so a cancel that lands between two passes is not lost.  The handle stays usable. */
void jt_cancel(jt_ctx *h);
void jt_reset_cancel(jt_ctx *h);
/* A caller that arms its cancellation source BEFORE the first call of a job (Go: context.AfterFunc, then jt_process_file) brackets the
 * job: jt_begin_job clears the flag once, and until jt_end_job no entry point clears it any more - a cancel that fires between the
 * arming and the first call is then observed by that call instead of being wiped by its "new job" reset.  Brackets do not nest. */
void jt_begin_job(jt_ctx *h);
void jt_end_job(jt_ctx *h);

/* ---- options.  The library reads no environment variable and writes none: what used to be JT_* variables is set here, per handle,
 * and read by the launchers (a Go host runs many goroutines; getenv / setenv beside them is a data race).  key / value are short
 * ASCII strings; a boolean takes "1" / "0" (or "true" / "false", "on" / "off"; NULL or "" = on).
 *   schedule switches (same results either way; the tests run both): no_pass2_prefetch, no_early_pass3, no_early_plan, no_lim_keep,
 *     no_staged_finish, no_r128_first, region_rot (0..3, -1 = default), region_full_astats, no_spec_direct (aspectralstats' records
 *     through a device buffer and a copy instead of straight into the pinned arena), nf_low (Pass 2: astats' noise-floor chain among the
 *     low-priority statistics instead of behind the K-weighting job), no_early_biquad (the early Pass-2 head's biquads behind Pass 1's
 *     analysis instead of beside it), as_avg_behind_spec (astats' exponential-average chain behind aspectralstats instead of behind
 *     its reduce chain), dk_unsorted (adeclick's solver lists in the order the front kernels appended them instead of longest window
 *     first), p2_device_join / dk_device_join (Pass 2's main stream waits for the early head / Pass 4's for the wide-band solver
 *     inside the queue -- hipStreamWaitEvent -- instead of on the host thread, which launches ~0.1 ms sooner on this part; handles
 *     that poll with sleeps and one-stream handles always wait inside the queue)
 *   kernel selection: adeclick_exact (the sequential-order adeclick kernel, bit-exact to af_adeclick.c's summation order),
 *     nlm_generic (the any-geometry anlmdn kernel), p3_unfused (Pass 3 as stand-alone upsampler + K-weighting kernels: what the fused
 *     sweep is tested against), limiter_lanes (alimiter with a lane per segment: what the wave-per-segment kernel is tested against),
 *     tp_unpruned (ebur128's true peak by the exhaustive kernels: what the branch-and-bound path is tested against), tp_prune_min
 *     (integer: the shortest signal, in samples, that takes the branch-and-bound path; default 1048576), brickwall_f64 (Pass 4's
 *     brickwall as doubles + a dbl -> flt -> s16 sweep instead of writing the float and the s16 itself), ln_no_batch (dynamic-mode
 *     loudnorm: the limiter's harmless peaks one detector call at a time instead of a frame's worth in one step; same bytes),
 *     ln_no_stream (dynamic-mode loudnorm: every frame through the one-workgroup kernel instead of the data-parallel stream path; same bytes),
 *     swr_untiled (the dynamic mode's 192 kHz -> source-rate aresample one thread per output instead of the LDS-tiled kernel; same bytes),
 *     flac_no_ahead (mono FLAC input: a parse walk and a decode walk per frame, as for stereo, instead of one walk that does both; same samples),
 *     nf_unpruned (astats' noise floor with every 50 ms window evaluated instead of by branch and bound; same value)
 *   diagnostics: host_timing (host-stage timings on stderr; PROCESS-WIDE although it is set through a handle: the host stages are
 *     plain functions without one, the last writer wins for every handle of the process)
 *   process-wide (h == NULL): graveyard_gb (gigabytes of superseded buffers parked before they are freed, default 24, 0 = free at
 *     once), poison_alloc (fill every new device allocation with 0xFF bytes; test switch), pool_streams (HIP streams per handle of a
 *     handle pool with three or more handles per device: default 1, see jt_open_ex), pool_blocking_sync (those handles' host waits
 *     sleep instead of spinning: default 1), host_timing, early_temp_min_kb (jt_process_file creates, reserves and maps its temporary
 *     output while the passes run when the output is expected to reach this size: default 32768 = 32 MiB).  Each library flavour (libjtgpu.so, libjtgpu_ab.so) keeps its own copy.
 * Superseded kernel generations and tuning knobs (nlm_old, afftdn_old, adeclick_fused, dyn_one_wave, kw_two_sweeps, follow_tiles,
 * dk_waves, ...) exist only in the A/B build of the library (make ab -> libjtgpu_ab.so, jt_build_flags() & 1): the default build
 * answers JT_E_UNSUPPORTED for them, an unknown key or a malformed value is JT_E_INVAL.
 * GPU_MAX_HW_QUEUES (ROCclr's hardware-queue count, read when the process first touches HIP) is the HOST's to set before that moment:
 * 8 gives every stream of a handle its own queue; the default 4 works, a few per cent slower. */
int jt_set_option(jt_ctx *h, const char *key, const char *value);
int jt_build_flags(void);                                   /* bit 0: A/B build (JT_AB) */

/* ---- input: replaces audio.Reader.ReadFrame feeding abuffer (reader.go:129, frame_processor.go:131-146) ---- */
/* interleaved f32 PCM in host memory; copied to HBM. channels 1 .. 8.  2 => the aformat=channel_layouts=mono down-mix
 * (filters.go:607-615) = libswresample's rematrix in the aresample libavfilter auto-inserts: Pass 1 and Pass 2 both hold a
 * float-only filter (aspectralstats / anlmdn), so that converter's internal and output formats are FLTP whatever the decoder
 * produced, auto_matrix does not normalise (maxval = INT_MAX), and mono = fl(fl(c*L) + fl(c*R)), c = (float)M_SQRT1_2
 * (DESIGN.md section 3 has the derivation; a stereo file with L == R therefore comes out 3.01 dB hotter, as with FFmpeg). */
int jt_upload_pcm(jt_ctx *h, const float *interleaved, int64_t frames, int sample_rate, int channels);
/* The same with the source's channel layout (libavutil / WAVEFORMATEXTENSIBLE bit order: FL 0x1, FR 0x2, FC 0x4, LFE 0x8, BL 0x10,
 * BR 0x20, FLC 0x40, FRC 0x80, BC 0x100, SL 0x200, SR 0x400; channels interleaved in ascending bit order).  channel_mask = 0: the
 * default layout of the channel count, which is what swr_init gives a source without one (av_channel_layout_default: mono, stereo,
 * 2.1, 4.0, 5.0, 5.1, 6.1, 7.1) -- jt_upload_pcm.  The mono down-mix is libswresample's default rematrix row for a FRONT_CENTER
 * output (swr_build_matrix2, center_mix_level = surround_mix_level = M_SQRT1_2, lfe_mix_level = 0):
 *   FC 1.0;  FL, FR, FLC, FRC 1/sqrt 2 each;  BL, BR, BC, SL, SR 0.5 each;  LFE 0 -- unnormalised in the float graphs of Pass 1 / 2
 * (as the stereo case above), divided by the sum of the coefficients in the band graphs of an integer source (jt_set_source_format);
 * products and sums in float, channel order (swri_rematrix's generic loop; two non-zero inputs: mix_2_1).
 * JT_E_UNSUPPORTED: channels beyond SIDE_RIGHT, or a mask whose bit count is not `channels`. */
int jt_upload_pcm_layout(jt_ctx *h, const float *interleaved, int64_t frames, int sample_rate, int channels, uint64_t channel_mask);
/* The decoder's native sample format of the PCM just uploaded / attached (they reset it to float): integer sources were scaled
 * by 2^(1-bits) to f32.  It matters in one place: the band-RMS graphs (analyser_bands.go:33) contain no float-only filter, so
 * libavfilter runs their biquads in the source's own width -- s16p (float state, output truncated and clipped per stage) for
 * <= 16-bit sources, s32p (double state) for 24/32-bit ones -- with an integer-normalised 0.5/0.5 down-mix, and astats divides
 * by INT16_MAX / INT32_MAX.  jt_load_audio sets this from the file. */
int jt_set_source_format(jt_ctx *h, int bits_per_sample, int is_float);
/* same, but the PCM is already resident in device memory (hipMalloc'd by the caller, e.g. a torch tensor);
 * not copied, must stay alive until jt_close / next upload. */
int jt_attach_device_pcm(jt_ctx *h, const void *dev_ptr, int64_t frames, int sample_rate, int channels);
/* s16 PCM at `sample_rate` (the Pass-2 temp FLAC as re-read by Pass 3/4 and region measures) */
int jt_upload_s16(jt_ctx *h, const int16_t *pcm, int64_t frames, int sample_rate);

/* ---- shared measurement structs (analyser.go:140-184, analyser_metrics.go:694-711) ---- */
typedef struct {
    double mean, variance, centroid, spread, skewness, kurtosis, entropy, flatness,
           crest, flux, slope, decrease, rolloff;
} jt_spectral;                               /* lavfi.aspectralstats.1.* */

typedef struct {
    double dc_offset, min_level, max_level, min_difference, max_difference, mean_difference,
           rms_difference, peak_level, rms_level, rms_peak, rms_trough, crest_factor, flat_factor,
           peak_count, noise_floor, noise_floor_count, entropy, dynamic_range, zero_crossings,
           zero_crossings_rate, number_of_samples, bit_depth;
} jt_astats;                                 /* lavfi.astats.1.* (dB where FFmpeg prints dB, linear otherwise) */

typedef struct {
    double integrated, lra, lra_low, lra_high, momentary, shortterm;  /* last values (LUFS / LU) */
    double true_peak, sample_peak;           /* LINEAR, as lavfi.r128.true_peak / sample_peak */
    double target_threshold;                 /* lavfi.r128.target_threshold analogue: relative gate of I */
} jt_r128;

/* one 100 ms ebur128 output frame as the Go OnFrame callback sees it (analyser.go:621-630) */
typedef struct {
    double momentary, shortterm;             /* lavfi.r128.M / S */
    double true_peak, sample_peak;           /* cumulative linear peaks */
    jt_spectral spectral;                    /* aspectralstats frame whose props survive re-framing */
} jt_frame_meta;

typedef struct {
    jt_astats   astats;                      /* whole-file (cumulative "latest wins") */
    jt_r128     r128;
    jt_spectral spectral_mean;               /* mean over the 100 ms output frames (analyser_metrics.go:715-744) */
    int64_t     n_frames_meta;               /* number of 100 ms output frames */
    int64_t     n_input_frames;              /* number of decoder frames of frame_samples */
} jt_analysis;

/* ---- Pass 1: replaces collectAnalysisFrames' runFilterGraph sweep (analyser.go:538-650) over
 * "aformat=channel_layouts=mono,astats=...,aspectralstats=win_size=2048:win_func=hann:measure=all,
 *  ebur128=metadata=1:peak=sample+true:dualmono=true:target=-16" (filters.go:42-45,624-626,684-689).
 * frame_samples = decoder frame size: what Reader.ReadFrame (reader.go:129) hands the frame loop per call.  0 = THE INPUT'S OWN
 * CADENCE (jt_input_frame_layout below): the block size(s) of the FLAC stream / the WAV demuxer's packet jt_load_audio saw, 4096 for
 * PCM that was uploaded.  frame_sumsq/frame_peak: per decoder frame
 * sum(x^2) and max|x| on the RAW (pre-downmix, all channels) samples as frameSumSquaresAndPeak computes
 * them (analyser_metrics.go:273-358); caller arrays of n_input_frames entries (or NULL).
 * meta: caller array for the 100 ms output frames (cap_meta entries, or NULL). */
int jt_pass1(jt_ctx *h, int frame_samples, jt_analysis *out,
             double *frame_sumsq, double *frame_peak, int64_t cap_frames,
             jt_frame_meta *meta, int64_t cap_meta);

/* ---- band RMS: replaces measureSpeechBandRMS / measureNoiseBands region graphs
 * "aformat=channel_layouts=mono,atrim=start:duration,asetpts,highpass=f=lo:p=2,lowpass=f=hi:p=2,
 *  astats=metadata=1:measure_perchannel=0" (analyser_bands.go:33, analyser_noise_bands.go:65-119).
 * out_db[i] = lavfi.astats.Overall.RMS_level of band i over the region of the uploaded input. */
int jt_band_rms(jt_ctx *h, double start_s, double dur_s, const double *lo_hz, const double *hi_hz,
                int n_bands, double *out_db, int *ok);

/* ---- Pass 2 parameters: numeric content of EffectiveFilterConfig (filters.go:111-255) at the
 * string-formatted precision BuildFilterSpec emits (filters.go:755,811,844,883,906,927). ---- */
typedef struct {
    int    hp_enabled;  double hp_freq, hp_q;                 /* highpass=f:poles=2:width_type=q:width:normalize=1:a=tdii */
    int    lp_enabled;  double lp_freq, lp_q;                 /* lowpass=... */
    int    nlm_enabled; double nlm_strength, nlm_patch_s, nlm_research_s, nlm_smooth;   /* anlmdn=s:p:r:m */
    int    fft_enabled; double fft_nr, fft_nf;                /* afftdn=nr:...:nf (nf 0 => FFmpeg default -50) */
    int    fft_custom;  double fft_band_noise[15];            /* nt=custom:bn= */
    int    fft_track_noise;                                   /* tn=1: the floor follows spectrally flat frames (adaptive.go:147-151) */
    int    gate_enabled; double gate_threshold, gate_ratio, gate_attack_ms, gate_release_ms,
                                gate_range, gate_knee, gate_makeup;                     /* agate, detection=rms */
    int    comp_enabled; double comp_threshold, comp_ratio, comp_attack_ms, comp_release_ms,
                                comp_makeup, comp_knee, comp_mix;                       /* acompressor, linear thr/makeup */
    int    deess_enabled; double deess_i, deess_m, deess_f;                             /* deesser=i:m:f */
    int    out_rate;                                           /* 44100 (aformat=sample_rates=44100:...:sample_fmts=s16) */
    int    out_frame_samples;                                  /* asetnsamples=n=4096 */
} jt_filter_params;

/* ---- Pass 2: replaces processWithFilters' runFilterGraph sweep (processor.go:255-373) over
 * BuildFilterSpec() (filters.go:968-989; golden chain filters_test.go:298-311).  Leaves the s16 @ out_rate
 * result resident on the device as stage 2. */
int jt_pass2(jt_ctx *h, const jt_filter_params *p, jt_analysis *out);
/* Optional: starts the head of the Pass-2 chain -- the highpass / lowpass cascade and anlmdn, whose parameters AdaptConfig takes
 * from the base configuration, not from the Pass-1 measurements (adaptive.go: tuneBandlimitLowPass is a constant, the rumble
 * filter and anlmdn are not tuned) -- on a stream of its own and returns at once, so that it runs while the host is still busy
 * with interval building, VAD and the band measurements.  The next jt_pass2 continues from that intermediate signal when its
 * parameters for those stages are identical (compared as the derived coefficients); otherwise the work is discarded and Pass 2
 * runs from the input as usual.  It may be
 * called as soon as the input is on the device: jt_pass1, jt_band_rms and jt_region_prefetch leave it running, any other call
 * on the handle retires it first. */
int jt_pass2_prefetch(jt_ctx *h, const jt_filter_params *p);
/* The same head, queued by the next jt_pass1 right behind its own kernels instead of at once: the analysis the host is waiting
 * for then takes its CU slots first and the head's large grid fills in around it. */
int jt_pass2_prefetch_after_pass1(jt_ctx *h, const jt_filter_params *p);

/* ---- region re-measure: replaces measureOutputRegionFromReader (analyser_output.go:95-227) over
 * "atrim=start:duration,asetpts=PTS-STARTPTS,astats=metadata=1:measure_perchannel=0,
 *  aspectralstats=measure=all,ebur128=metadata=1:peak=sample+true" (analyser_output.go:18).
 * stage: 2 = Pass-2 output, 4 = Pass-4 output (both s16 on device). */
typedef struct {
    double rms_level, peak_level, crest_factor;   /* astats Overall (crest linear) */
    jt_spectral spectral;                          /* mean over output frames */
    double momentary, shortterm;                   /* last lavfi.r128.M / S */
    double true_peak, sample_peak;                 /* last, LINEAR */
    int64_t frames;
} jt_region_sample;
int jt_region_measure(jt_ctx *h, int stage, double start_s, double dur_s, jt_region_sample *out);
/* MeasureOutputRegions (analyser_output.go:276-317) measures the room-tone and the speech region of one output back to
 * back: both analyses in one call, one synchronisation.  A region with dur_s[i] <= 0 is skipped (out[i] zeroed). */
int jt_region_measure_pair(jt_ctx *h, int stage, const double start_s[2], const double dur_s[2], jt_region_sample out[2]);
/* Announces the regions MeasureOutputRegions will ask for BEFORE the stage's output exists: the next jt_pass2 (stage 2) /
 * jt_pass4 (stage 4) then measures them in its own tail (same kernels, same arithmetic, no extra synchronisation), and a later
 * jt_region_measure_pair with the identical stage / start_s / dur_s returns that stored result without device work.  The
 * announcement is consumed by that one pass; any other request is measured on demand exactly as before. */
int jt_region_prefetch(jt_ctx *h, int stage, const double start_s[2], const double dur_s[2]);

/* ---- limiter prefix shared by Pass 3 and Pass 4 (normalise.go:446-465 buildPreLimiterPrefix) ---- */
typedef struct {
    int    needed;            /* emit alimiter */
    double pre_gain_db;       /* volume=%.1fdB when > 0 */
    double limit;             /* alimiter limit= (linear, %.6f) ; attack=5 release=100 asc=1 asc_level=0.8 latency=1 level=0 */
} jt_limiter_plan;

typedef struct {              /* loudnorm print_format=json (normalise.go:64-75); doubles, shim rounds to %.2f */
    double input_i, input_tp, input_lra, input_thresh;
    double output_i, output_tp, output_lra, output_thresh;
    double target_offset;
    int    normalization_type_dynamic;   /* 0 = "linear", 1 = "dynamic" */
} jt_loudnorm_stats;

/* Pass 3 started inside Pass 2 with the plan the caller WILL pass (normalise.go:373-561: the limiter plan is a function of Pass 2's
 * integrated loudness and true peak alone).  Announce a planner before jt_pass2: as soon as those two values exist (the other analysis
 * chains of Pass 2 are still running) jt_pass2 calls fn(user, integrated LUFS, true peak as a linear ratio, &plan) on the calling thread
 * and, if the plan needs the limiter prefix, queues volume -> alimiter -> the 192 kHz measurement right away; jt_pass3 with an equal
 * plan then collects the result instead of running it after Pass 2 has ended.  A schedule change only: same kernels, same inputs, same
 * numbers (JT_NO_EARLY_PLAN=1 disables it; tests/test_gpu_round3.py compares).  The announcement lasts for one jt_pass2. */
typedef int (*jt_plan_fn)(void *user, double integrated_lufs, double true_peak_linear, jt_limiter_plan *plan);
int jt_pass3_plan_hook(jt_ctx *h, jt_plan_fn fn, void *user);

/* ---- Pass 3: replaces measureWithLoudnorm's sweep (normalise.go:226-346) over
 * "<prefix>,loudnorm=I=-16.0:TP=-1.0:LRA=20.0:dual_mono=true:print_format=json:stats_file=..." (:257-268).
 * Only the input_* fields and target_offset are filled (the dynamic-mode output is discarded by the
 * reference; target_offset is NOT consumed: normalise.go:861-873 derives its own offset) —
 * output_* = NaN, target_offset = NaN. */
int jt_pass3(jt_ctx *h, const jt_limiter_plan *lim, double target_i, double target_tp, double target_lra,
             jt_loudnorm_stats *out);

typedef struct {              /* loudnorm second pass options (normalise.go:1269-1291), values as %.2f-formatted */
    double target_i, target_tp, target_lra;
    double measured_i, measured_tp, measured_lra, measured_thresh;
    double offset;
    int    adeclick_enabled;  double adeclick_threshold, adeclick_window_ms, adeclick_overlap_pct;   /* filters.go:947-962 */
    int    adeclick_method;   /* 1 = m=s (overlap-save, the reference default filters.go:513-521), 0 = option omitted (overlap-add) */
    double brickwall_limit;   /* alimiter limit= (linear) attack=1 release=50 (normalise.go:474-480) */
} jt_loudnorm_apply;

/* ---- Pass 4: replaces applyLoudnormAndMeasure's sweep (normalise.go:924-1190) over
 * buildLoudnormFilterSpec() (normalise.go:1231-1334; golden normalise_test.go:2135-2223):
 * [volume,alimiter,]loudnorm(linear),aresample,adeclick,alimiter(brickwall),astats,aspectralstats,
 * ebur128,aformat(s16).  Leaves the final s16 on device as stage 4.  When the second-pass values do not satisfy af_loudnorm's
 * init() (a measured value missing / 0, projected peak above TP, measured LRA above LRA) the filter's dynamic mode runs instead
 * (192 kHz, k_loudnorm.hip) and stats->normalization_type_dynamic = 1, as the reference's stats file would say. */
int jt_pass4(jt_ctx *h, const jt_limiter_plan *lim, const jt_loudnorm_apply *ap,
             jt_analysis *out, jt_loudnorm_stats *stats);

/* ---- input leg: replaces audio.OpenAudioFile + Reader.ReadFrame (reader.go:29-169), which every pass of the reference runs
 * again over the file.  `file` is the whole file image in host memory: FLAC (RFC 9639: any block size, 4..24 bits, 1..8
 * channels, all predictor / residual / stereo modes; an ID3v2 tag in front is skipped) is decoded on the GPU; RIFF/WAVE PCM
 * (u8, s16, s24, s32, f32, f64, WAVE_FORMAT_EXTENSIBLE; RF64 / BW64 with their ds64 sizes) is unpacked on the GPU.  Afterwards the handle is in the state
 * jt_upload_pcm_layout() leaves it in — up to eight channels in the file's layout (FLAC: libavcodec's flac_channel_layouts for the
 * channel count; WAVE_FORMAT_EXTENSIBLE: dwChannelMask; plain WAV: the default layout of the channel count), layouts with channels
 * beyond SIDE_RIGHT are refused with JT_E_UNSUPPORTED (jt_op_decode_audio decodes any layout) — (interleaved f32 at the file's rate; integer PCM scaled by 2^(1-bits) exactly as
 * libswresample's s16/s32 -> flt conversion does).  Errors: JT_E_INVAL for a damaged or truncated stream (frame CRC-16 /
 * header CRC-8 / sample count), JT_E_UNSUPPORTED for other containers or 32-bit FLAC. */
typedef struct {
    int format;                  /* 1 = FLAC, 2 = WAV */
    int sample_rate, channels, bits_per_sample, is_float;
    int64_t frames;              /* inter-channel samples */
    double duration_s;           /* Metadata.Duration (reader.go:22-27) */
    int64_t flac_frames; int flac_candidates;
    double gpu_ms, total_ms;
    /* The decoder's frame cadence -- what Reader.ReadFrame (reader.go:129-169) would deliver for this file, which is what closes the
     * reference's 250 ms analysis intervals (analyser.go:588-600) and paces its progress ticks (analyser.go:602, processor.go:274):
     *   FLAC: one AVFrame per FLAC frame = the stream's block size (every frame but the last); a stream whose frames differ in length
     *         (variable blocking strategy, or a fixed-blocksize stream with odd frames) has decoder_frames_variable = 1,
     *         decoder_frame_samples = its longest frame, and its per-frame lengths in jt_input_frame_layout;
     *   WAV:  libavformat/wavdec.c wav_read_packet: packets of max_size = 4096 BYTES rounded down to whole sample blocks
     *         (block_align = channels x bytes per sample; one block when that is larger), one AVFrame per packet:
     *         1024 samples of mono f32, 2048 of mono s16, 1365 of mono s24, 512 of stereo f32 ... (restated from FFmpeg's
     *         sources, which are absent here: DESIGN section 3, assumption 12). */
    int decoder_frame_samples, decoder_frames_variable;
    int64_t decoder_frames;      /* how many frames ReadFrame would deliver */
    uint64_t channel_mask;       /* the layout the down-mix used (libavutil / WAVEFORMATEXTENSIBLE bit order): the file's, or the default of its channel count */
} jt_audio_meta;
int jt_load_audio(jt_ctx *h, const uint8_t *file, int64_t len, jt_audio_meta *meta);
/* The cadence of the handle's current input (set by jt_load_audio; 4096 and constant after jt_upload_pcm / jt_attach_device_pcm):
 * *frame_samples = the constant frame length (the longest frame of a variable stream), *n_frames = the frame count; when the frames
 * differ in length *variable = 1 and lens (cap entries, or NULL) receives min(cap, n_frames) per-frame lengths.  Any pointer may be NULL. */
int jt_input_frame_layout(jt_ctx *h, int *frame_samples, int *variable, int64_t *n_frames, int32_t *lens, int64_t cap);
/* operator-level entry (parity tests): the decoded samples back on the host, interleaved; either pointer may be NULL */
int jt_op_decode_audio(jt_ctx *h, const uint8_t *file, int64_t len, int32_t *pcm_i32, float *pcm_f32, int64_t cap_values,
                       jt_audio_meta *meta);

/* ---- output: replaces Encoder.WriteFrame's input (encoder.go:145): raw s16, or the finished FLAC file below ---- */
int jt_output_len(jt_ctx *h, int stage, int64_t *n);
int jt_download_s16(jt_ctx *h, int stage, int16_t *dst, int64_t cap, int64_t *n);

/* calculateFrameLevel (encoder.go:235-257) for every frame_samples-long frame of a stage output: 20 log10(rms), clamped to the VU
 * meter's [-70, 0] dB -- the Level the reference's progress ticks carry (processor.go:336-338, normalise.go:288,1119-1121). */
int jt_output_frame_levels(jt_ctx *h, int stage, int frame_samples, double *levels_db, int64_t cap, int64_t *n_frames);

/* ---- FLAC output leg: replaces createOutputEncoder + Encoder.WriteFrame/Flush/Close (encoder.go:54-110,145-215): the
 * reference hands every 4096-sample s16 frame to FFmpeg's flac encoder (compression_level 5) and muxes a .flac file.  Here
 * the stage output already in HBM is encoded on the GPU, one frame per wavefront, and the finished file image (fLaC marker,
 * STREAMINFO, VORBIS_COMMENT, frames) is returned in a pinned host buffer owned by the handle (valid until the next
 * jt_flac_encode / jt_op_flac_encode_s16 / jt_close on it).  Same container contract as the reference's files: mono, 16 bit,
 * fixed block size 4096, STREAMINFO with total samples, min/max frame size and (JT_FLAC_MD5) the MD5 of the PCM.  FLAC is
 * lossless, so parity = the decoded PCM equals jt_download_s16() bit for bit; the compressed bytes themselves are this
 * encoder's (LPC order 1..8 by exhaustive search, 12-bit coefficients, Rice partition order 0..8). */
#define JT_FLAC_MD5 1            /* compute the STREAMINFO MD5 (host, one core, ~0.6 GB/s); without it the field is 0 = "unknown" */
typedef struct {
    int64_t bytes, frames, total_samples;
    int sample_rate, channels, bits_per_sample, block_size, min_frame_bytes, max_frame_bytes, header_bytes;
    double gpu_ms, md5_ms, total_ms;
    uint8_t md5[16];
} jt_flac_info;
int jt_flac_encode(jt_ctx *h, int stage, int flags, const uint8_t **data, int64_t *len, jt_flac_info *info);
/* operator-level entry (host PCM in, used by the parity tests): any length >= 1, any rate FLAC can describe */
int jt_op_flac_encode_s16(jt_ctx *h, const int16_t *pcm, int64_t n, int sample_rate, int flags,
                          const uint8_t **data, int64_t *len, jt_flac_info *info);

/* ---- per-pass device timers (ms, HIP events on the engine stream) for the roofline report ---- */
typedef struct { double pass1_ms, pass2_ms, pass3_ms, pass4_ms; double nlm_ms; int64_t nlm_launches;
                 int64_t declick_repaired;   /* samples adeclick re-interpolated in the last Pass 4 (diagnostic) */
                 double declick_ms;          /* adeclick kernel time in the last Pass 4 */
                 int64_t declick_heavy_windows; /* windows that needed the full-capacity second pass */
                 int64_t tp_units_total,     /* last long analysis, branch-and-bound true peak: units of the signal ... */
                         tp_units_evaluated; /* ... and those whose 192 kHz outputs were evaluated (seeds + kept units) */
                 int64_t ln_stream_frames,   /* last dynamic-mode loudnorm: 100 ms frames its data-parallel stream path covered ... */
                         ln_stream_why;      /* ... and why its attempts ended, bit r = reason r: 0 ran to the last full frame, 1 limiter state not in
                                                its steady form, 2 peak list full, 3 segment list full, 4 ring-end corner, 5 test switch */ } jt_timers;
int jt_get_timers(jt_ctx *h, jt_timers *out);

/* =====================================================================================
 * Operator-level entry points: one per FFmpeg filter on the path, operating on caller host buffers
 * (copied to/from HBM).  Used by the parity tests to compare each HIP kernel with the oracle.
 * ===================================================================================== */
int jt_op_biquad_f32(jt_ctx *h, const float *in, float *out, int64_t n, int sample_rate,
                     int hp_enabled, double hp_freq, double hp_q, int lp_enabled, double lp_freq, double lp_q);
int jt_op_anlmdn_f32(jt_ctx *h, const float *in, float *out, int64_t n, int sample_rate,
                     double strength, double patch_s, double research_s, double smooth);
int jt_op_afftdn_f32(jt_ctx *h, const float *in, float *out, int64_t n, int sample_rate,
                     double nr, double nf, const double *band_noise /* 15 or NULL */);
/* agate -> acompressor -> deesser chain in double on float input, output rounded to float
 * (the dbl->flt conversion FFmpeg inserts before aspectralstats) */
/* the same with tn=1 (track_noise); final_floor_db (optional) receives the tracked floor after the last frame */
int jt_op_afftdn_tn_f32(jt_ctx *h, const float *in, float *out, int64_t n, int sample_rate,
                        double nr, double nf, const double *band_noise, int track_noise, double *final_floor_db);
int jt_op_dynamics(jt_ctx *h, const float *in, float *out, int64_t n, int sample_rate, const jt_filter_params *p);
int jt_op_alimiter_f64(jt_ctx *h, const double *in, double *out, int64_t n, int sample_rate,
                       double limit, double attack_ms, double release_ms);
/* loudnorm's DYNAMIC mode (af_loudnorm.c) on a mono stream already at the filter's internal 192 kHz: what jt_pass4 runs, between an
 * up- and a down-resample, when the second-pass preconditions fail (measured LRA above the LRA target, a measured value missing or
 * printed as 0.00: the reference's "fell back to dynamic" case, normalise.go:687-693).  ap: I / TP / LRA, measured_*, offset (dB) as in
 * the filter's options; the adeclick / brickwall fields are ignored.  Shorter than 3 s: the filter's own one-gain branch. */
int jt_op_loudnorm_dynamic_f64(jt_ctx *h, const double *in192, int64_t n, const jt_loudnorm_apply *ap, double *out192, jt_loudnorm_stats *stats);
/* adeclick=t:w:o[:m=s] on a double stream (af_adeclick.c; arorder=2, burst=2 defaults).  method: 1 = overlap-save (m=s, what
 * filters.go:513-521 configures), 0 = overlap-add (m=a, FFmpeg's own default: the sequential-order kernel + a product buffer). */
int jt_op_adeclick_f64(jt_ctx *h, const double *in, double *out, int64_t n, int sr, double threshold, double window_ms,
                       double overlap_pct, int method, int64_t *n_repaired);
int jt_op_resample_f32_to_s16(jt_ctx *h, const float *in, int64_t n, int in_rate, int out_rate,
                              int16_t *out, int64_t cap, int64_t *n_out);
int jt_op_ebur128(jt_ctx *h, const float *in, int64_t n, int sample_rate, int dualmono, jt_r128 *out,
                  double *m_series, double *s_series, double *tp_series, double *sp_series, int64_t cap, int64_t *n_blocks);
int jt_op_astats(jt_ctx *h, const float *in, int64_t n, int sample_rate, jt_astats *out);
int jt_op_aspectralstats(jt_ctx *h, const float *in, int64_t n, int sample_rate, jt_spectral *hops, int64_t cap, int64_t *n_hops);
int jt_op_loudnorm_measure_s16(jt_ctx *h, const int16_t *in, int64_t n, int sample_rate,
                               const jt_limiter_plan *lim, jt_loudnorm_stats *out);

#ifdef __cplusplus
}
#endif
#endif
