/*
 * jt_host.h — host-side mirror of the reference's per-file control logic for the four-pass path, as a C ABI
 * (same shared library, libjtgpu.so).  The reference keeps this logic in Go (internal/processor); Go is not
 * available in the build image, so it is restated in C++ (jivetalking_amd/csrc/jt_host.cpp) with the same
 * names, argument meaning and results, and exported here so the cgo shim / tests can call each step.
 *
 *   interval construction ........ collectAnalysisFrames + intervalAccumulator   (analyser.go:538-650, analyser_metrics.go:165-428)
 *   noise-floor seed ............. estimateNoiseFloorAndThreshold                 (analyser_noise_seed.go:78-241)
 *   voice-activity detection ..... detectVoiceActivity and helpers                (analyser_vad.go:108-813)
 *   speech election .............. findBestSpeechRegion, scoring, refinement      (analyser_candidates_{shared,speech}.go)
 *   adaptive tuning .............. AdaptConfig                                    (adaptive*.go)
 *   filter-spec strings .......... BuildFilterSpec and builders                   (filters.go:607-998)
 *   limiter / loudnorm planning .. planLimiterForLoudnorm, calculateLinearModeTarget, buildLoudnormFilterSpec (normalise.go)
 *   orchestration ................ ProcessAudio / AnalyseOnlyDetailed            (processor.go:29-216)
 */
#ifndef JT_HOST_H
#define JT_HOST_H
#include "jtgpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- IntervalSample (analyser_metrics.go:17-32); times are Go time.Duration nanoseconds ---- */
typedef struct {
    int64_t     timestamp_ns;
    double      rms_level, peak_level;
    jt_spectral spectral;
    int         spectral_found;
    double      momentary_lufs, shortterm_lufs, true_peak, sample_peak;
} jt_interval;

/* Build the 250 ms interval stream exactly as collectAnalysisFrames does from decoder frames and the 100 ms
 * output frames.  meta entries whose momentary is NaN carry no r128 keys (trailing partial frame).
 * quantize != 0 applies FFmpeg's metadata print formats (ebur128 "%.3f", aspectralstats "%g") before use.
 * Returns the number of intervals written (<= cap). */
int64_t jt_host_build_intervals(int sample_rate, int64_t n_samples, int frame_samples, int channels,
                                const double *frame_sumsq, const double *frame_peak, int64_t n_frames,
                                const jt_frame_meta *meta, int64_t n_meta, int quantize,
                                jt_interval *out, int64_t cap);
/* The same for decoder frames of DIFFERENT lengths (a variable-blocksize FLAC stream: analyser.go:588-600 reads every frame's own
 * NbSamples): frame_lens[f] = samples per channel of frame f (their sum = n_samples).  frame_lens == NULL: frames of frame_samples. */
int64_t jt_host_build_intervals_v(int sample_rate, int64_t n_samples, int frame_samples, const int32_t *frame_lens, int channels,
                                  const double *frame_sumsq, const double *frame_peak, int64_t n_frames,
                                  const jt_frame_meta *meta, int64_t n_meta, int quantize,
                                  jt_interval *out, int64_t cap);

/* ---- regions / profiles (analyser.go:25-135) ---- */
typedef struct { int64_t start_ns, end_ns, duration_ns; } jt_region;

typedef struct {
    double rms_level, peak_level, crest_factor;
    jt_spectral spectral;
    double momentary_lufs, shortterm_lufs, true_peak, sample_peak;
} jt_region_metrics;                           /* RegionSample */

typedef struct {
    int64_t start_ns, duration_ns;
    double  measured_noise_floor, peak_level, crest_factor, entropy;
    jt_spectral spectral;
    double  band_noise[15]; int band_noise_n; int bands_measured;
    int     warning;                           /* 0 none, 1 short region, 2 long region */
} jt_noise_profile;

typedef struct {
    jt_region region;
    jt_region_metrics sample;
    double voicing_density;
    double body_band_rms, sib_band_rms; int bands_measured;
    double score;
    int64_t original_start_ns, original_duration_ns; int was_refined;
} jt_speech_candidate;

#define JT_MAX_REGIONS 512
typedef struct {
    /* Loudness (InputLoudnessMetrics) */
    double input_i, input_tp, input_lra, input_thresh, target_offset, momentary, shortterm, sample_peak;
    /* Dynamics (astats with the Go-side conversions: crest in dB, min/max level in dBFS) + Noise.FloorAstats */
    jt_astats dynamics; double floor_astats;
    jt_spectral spectral;
    /* Noise */
    double floor; int floor_source;            /* 0 astats, 1 rms_estimate, 2 ebur128_estimate, 3 vad_percentile */
    double floor_prescan, room_tone_detect_level; int voice_activated; double floored_fraction, reduction_headroom;
    /* Regions */
    int n_speech_regions; jt_region speech_regions[JT_MAX_REGIONS];
    int n_candidates;     jt_speech_candidate candidates[JT_MAX_REGIONS];
    int has_speech_profile; jt_speech_candidate speech_profile;
    int has_noise_profile;  jt_noise_profile noise_profile;
    int has_room_tone_sample; jt_region_metrics room_tone_sample;
    double voiced_low_percentile, noise_high_percentile, gate_separation_db;
    double duration_s;
    /* diagnostics of the detector itself */
    double vad_split, vad_margin; int vad_gap_tol;
} jt_measurements;

/* buildInputMeasurements + detectVoiceActivity (analyser.go:364-406, analyser_vad.go:728-783) from the Pass-1
 * whole-file analysis (already Go-converted by this call: crest -> dB, levels -> dBFS, peaks -> dB) and intervals. */
int jt_host_detect(const jt_analysis *pass1, const jt_interval *intervals, int64_t n_intervals,
                   double duration_s, double target_i, int quantize, jt_measurements *out);
/* assignInputMeasurementSuggestions (analyser.go:513-528), after the band measurements */
void jt_host_finish_measurements(jt_measurements *m);
/* afftdnBandEdgesHz (analyser_noise_bands.go:29-51) */
void jt_host_afftdn_band_edges(int index, double *lo_hz, double *hi_hz);

/* ---- filter configuration (filters.go:111-255) ---- */
typedef struct { int enabled; double frequency; int poles; double width, mix; int transform_tdii; } jt_biquad_cfg;
typedef struct {
    int downmix_enabled, analysis_enabled;
    int resample_enabled, resample_rate, resample_frame;
    jt_biquad_cfg rumble_hp, bandlimit_lp;
    int nr_enabled; double nr_strength, nr_patch_s, nr_research_s, nr_smooth;
    int afftdn_enabled; double afftdn_nr; int afftdn_custom; int afftdn_track_noise; double afftdn_noise_floor; char afftdn_band_noise[256];
    int gate_enabled; double gate_threshold, gate_ratio, gate_attack, gate_release, gate_range, gate_knee, gate_makeup; int gate_detection_set;
    int comp_enabled; double comp_threshold_db, comp_ratio, comp_attack, comp_release, comp_makeup_db, comp_knee, comp_mix;
    int deess_enabled; double deess_intensity, deess_amount, deess_frequency;
    int adeclick_enabled; double adeclick_threshold, adeclick_window, adeclick_overlap;
    int adeclick_method_s;         /* AdeclickConfig.Method (filters.go:240-246,958-960): 1 = "s" (the default config), 0 = "" (option omitted: af_adeclick.c's m=a), 2 = "a"; 3 = "save", 4 = "add" (the option table's long names: same methods, printed verbatim in the spec) */
    int loudnorm_enabled; double target_i, target_tp, target_lra; int dual_mono, linear;
} jt_host_config;

typedef struct {
    double gate_quiet_speech_estimate, gate_separation, gate_speech_headroom, gate_threshold_unclamped, gate_depth_db;
    int gate_narrow_gap; int afftdn_enabled; double afftdn_noise_floor_db; int afftdn_disabled_voice_activated; int afftdn_custom;
} jt_adaptive_diag;

void jt_host_default_config(jt_host_config *c);                                  /* DefaultFilterConfig (filters.go:353) */
void jt_host_adapt(const jt_host_config *base, const jt_measurements *m, jt_host_config *effective, jt_adaptive_diag *diag); /* AdaptConfig (adaptive.go:13) */
/* BuildFilterSpec with Pass1FilterOrder (pass=1) or Pass2FilterOrder (pass=2) (filters.go:42-68,968-989); returns length */
int  jt_host_filter_spec(const jt_host_config *effective, int pass, char *buf, int cap);
/* numeric jt_filter_params at the precision the spec string carries */
void jt_host_filter_params(const jt_host_config *effective, jt_filter_params *out);

/* ---- normalisation planning (normalise.go) ---- */
typedef struct {
    double pre_gain_db, ceiling_db, gain_db, filtered_tp; int needed, clamped;
    char pass3_prefix[256];
} jt_limiter_decision;
void jt_host_calculate_limiter_ceiling(double measured_i, double measured_tp, double target_i, double target_tp,
                                       double *ceiling, int *needed, int *clamped);                 /* :373-396 */
void jt_host_calculate_pre_gain(double measured_i, double target_i, double target_tp, double *pre_gain, double *rederived); /* :411-431 */
void jt_host_plan_limiter(double output_i, double output_tp_db, const jt_host_config *cfg, jt_limiter_decision *out, jt_limiter_plan *plan); /* :539-561 */
void jt_host_calculate_linear_mode_target(double measured_i, double measured_tp, double desired_i, double target_tp,
                                          double *effective_i, double *offset, int *linear_possible);  /* :614-632 */
double jt_host_loudnorm_internal_target_tp(double target_i, double measured_tp, double measured_i);  /* :583-585 */
/* buildLoudnormFilterSpec (normalise.go:1231-1334) + the numeric jt_loudnorm_apply it implies.  Returns the spec's length, or
 * JT_E_INVAL when adeclick is enabled with a method code outside 0..4 (jt_process_audio / jt_process_file refuse it the same way) */
int  jt_host_pass4_spec(const jt_host_config *cfg, const jt_loudnorm_stats *measurement, double offset, const jt_limiter_decision *lim,
                        int source_rate, const char *stats_path, char *buf, int cap, jt_loudnorm_apply *apply);

/* ---- orchestration: ProcessAudio on PCM already uploaded to the engine (processor.go:78-216) ---- */
typedef struct {
    jt_measurements input;                     /* Pass 1 (+bands) */
    jt_host_config  effective; jt_adaptive_diag diag;
    jt_analysis     filtered;                  /* Pass 2 whole-file */
    jt_limiter_decision limiter;
    jt_loudnorm_stats measure;                 /* Pass 3 (values rounded to the JSON's %.2f) */
    double effective_target_i, offset; int linear_possible;
    jt_analysis     final_;                    /* Pass 4 whole-file */
    jt_loudnorm_stats loudnorm;                /* Pass 4 loudnorm JSON */
    jt_region_sample filtered_room_tone, filtered_speech, final_room_tone, final_speech; int has_region_samples;
    double output_lufs, output_tp_db, input_lufs, input_tp_db; int within_target;
    char pass2_spec[2048]; char pass4_spec[2048];
    double pass_ms[4];
    double stage_ms[10];     /* host wall-clock: pass1, intervals+VAD, bands, adapt, pass2, regions(2), plan, pass3, pass4, regions(4) */
} jt_process_result;

/* Output naming (processor.go:379-388): "<dir>/<name without its last extension>-LUFS-<n>-processed.flac",
 * n = round-half-away(|output LUFS|).  Returns the length written (excluding the NUL), or -1 when cap is too small. */
int jt_host_lufs_filename_value(double output_lufs);
int jt_host_output_path(const char *input_path, int lufs_value, char *out, int cap);

/* frame_samples (here and in every entry point below): the decoder's frame length, which closes the 250 ms analysis intervals
 * (analyser.go:588-600) and paces the progress ticks.  0 = the input's own cadence (jtgpu.h: jt_input_frame_layout) -- what the
 * reference sees for this file; a positive value overrides it (the caller decoded the file itself and knows its frames). */
int jt_process_audio(jt_ctx *h, const jt_host_config *base, int frame_samples, jt_process_result *out);

/* ProgressUpdate / ProgressCallback (progress.go:5-39).  The callback runs synchronously on the calling thread at the same
 * lifecycle points the reference emits (processor.go:80-158, normalise.go:737-772): pass start (progress 0) and pass end
 * (progress 1) of Analysing / Processing / Measuring / Normalising.  Pointers are valid only during the call:
 *   measurements : Pass-1 end, Pass-2 start and end;   config + diag : Pass-2 start only;   limiter_* : Pass-4 start only.
 * Intra-pass ticks (every 100 decoder frames in the reference) are not sent by this entry point; jt_process_audio_ticks replays them. */
typedef struct {
    int pass;                      /* 1 Analysing, 2 Processing, 3 Measuring, 4 Normalising (filters.go:340-345) */
    const char *pass_name;
    double progress, level, duration;
    const jt_measurements *measurements;
    const jt_host_config *config;
    const jt_adaptive_diag *diag;
    int has_limiter, limiter_enabled; double limiter_ceiling;
} jt_progress_update;
typedef void (*jt_progress_fn)(void *user, const jt_progress_update *u);
int jt_process_audio_cb(jt_ctx *h, const jt_host_config *base, int frame_samples, jt_progress_fn cb, void *user, jt_process_result *out);
/* The same plus the reference's intra-pass ticks: every 100th decoder frame of each pass (analyser.go:602-618, processor.go:320-335,
 * normalise.go:292-301,1108-1117) with Progress computed as the reference computes it (Pass 1 scaled by BandPhaseProgressStart =
 * 0.95, Passes 3/4 by samples and capped at 0.99) and Level = calculateFrameLevel (encoder.go:235-257) of the frame at that
 * position, and the 17 "Analysing frequency bands" ticks on 0.95 .. 1.0 (analyser_band_runner.go:47-88).  A pass is one launch
 * sequence on the GPU, so the ticks of a pass arrive together when it completes, before its end event. */
int jt_process_audio_ticks(jt_ctx *h, const jt_host_config *base, int frame_samples, jt_progress_fn cb, void *user, jt_process_result *out);
#define JT_FILE_PROGRESS_TICKS 0x100   /* jt_process_file flag (next to JT_FLAC_MD5): progress as jt_process_audio_ticks */
int jt_analyse_only(jt_ctx *h, const jt_host_config *base, int frame_samples, jt_process_result *out);   /* AnalyseOnlyDetailed (processor.go:29-69) */

/* ---- run record: the reference's per-file JSON document and its two sidecars (runrecord.go:15-51,
 * runrecord_write.go:37-45), built from a jt_process_result exactly as MarshalRunRecord prints them: a sanitised map tree (NaN / Inf
 * -> null, omitempty, embedded structs promoted, *_s seconds for durations), keys sorted at every level, two-space indent,
 * encoding/json's number and string formats.  Every call returns the length of the full text (excluding the NUL) and writes at most
 * cap - 1 bytes + NUL: call with cap = 0 to size the buffer.  `h` supplies the interval series of the handle's last Pass-1 analysis
 * (interval_summary, .intervals.jsonl). */
typedef struct {
    const char *input_file;      /* filepath.Base(result.OutputPath) (analysis-only: the input's base name) */
    const char *version;         /* RunVersion */
    const char *executable;      /* resolveExecutablePath() */
    const char *processed_at;    /* time.Now().Format(time.RFC3339) */
    double duration_s;           /* InputMetadata.DurationSecs (<= 0: the measured duration) */
    int sample_rate_hz, channels;
} jt_run_provenance;
int64_t jt_host_run_record_json(const jt_ctx *h, const jt_process_result *res, const jt_run_provenance *prov, int analysis_only, char *buf, int64_t cap);
int64_t jt_host_intervals_jsonl(const jt_ctx *h, char *buf, int64_t cap);                        /* WriteIntervalsSidecar */
/* The interval series itself (AudioMeasurements.Regions.IntervalSamples): returns the count, copies at most cap entries */
int64_t jt_host_last_intervals(const jt_ctx *h, jt_interval *out, int64_t cap);
int64_t jt_host_candidates_jsonl(const jt_process_result *res, char *buf, int64_t cap);          /* WriteCandidatesSidecar */
/* loudnorm's print_format=json body (the text parseLoudnormStatsFile reads, normalise.go:143-165): ten "%.2f" string fields */
int     jt_host_loudnorm_json(const jt_loudnorm_stats *stats, char *buf, int cap);

/* File in, file out — the reference's per-file entry point as the CLI calls it: ProcessAudio(ctx, inputPath, config, cb)
 * (processor.go:78-330) reads inputPath through libavformat for every pass, writes a temp FLAC after Pass 2 and the final
 * "<name>-LUFS-<n>-processed.flac" after Pass 4 (processor.go:379-388).  Here: read the file once, jt_load_audio (decode on
 * the GPU), the four passes, jt_flac_encode(stage 4) and one write of the finished image.  flac_flags: JT_FLAC_MD5 or 0.
 * The image is written to a hidden sibling ".processing-*.tmp.flac" and renamed over the final name only when complete
 * (createSiblingTempPath + publishOutput, file_write.go:13-53); on any error or cancellation no temp file remains and an existing
 * output of that name is left untouched.
 * output_path receives the written path (cap bytes).  io_ms: read, decode, encode, write (host wall clock).
 * Errors: JT_E_INVAL when the input cannot be opened/read or the output cannot be written, otherwise as the calls above. */
int jt_process_file(jt_ctx *h, const char *input_path, const jt_host_config *base, int frame_samples, int flac_flags,
                    jt_progress_fn cb, void *user, jt_process_result *out, char *output_path, int cap, double io_ms[4]);

/* Host topology: the NUMA node a device hangs off (its PCI address's numa_node in sysfs) and how many of that node's CPUs this process
 * may run on; -1 / 0 when the host does not say (one node, a VM).  A handle pool binds its worker threads and finisher jobs -- and,
 * through first touch, the pinned I/O sets they allocate -- to that node (option pool_numa, process-wide, default on). */
int jt_host_device_numa_node(int device, int *n_cpus);
/* Test entry: the loudnorm statistics bin a block energy into ebur128.c's 1000-bin histogram (find_histogram_index: a bisection over
 * the boundaries); the library finds the same bin from a table keyed by the double's top 20 bits plus a comparison against the
 * boundaries.  Returns how many of the n energies (and of the 1001 boundaries with their neighbours) land in a different bin: 0. */
int64_t jt_host_hist_index_check(const double *e, int64_t n);

/* Test seam: makes the next jt_process_file calls fail at temp creation / temp write / publish (the failures the reference injects
 * through processorCreateSiblingTempPath, a failing encoder and processorRename: processor_test.go:552-627) so that the
 * no-residue discipline can be tested.  All zero = normal operation. */
void jt_host_test_inject_fault(int create_temp, int write, int rename_);

/* Several files on one GPU — the reference's bounded worker pool (cmd/jivetalking/pool.go:122-228: `runBoundedPool` runs at most
 * N ProcessAudio calls at a time, one file's failure never stops the others, every file gets its own result).  Here a worker is a
 * host thread with its own handle (own streams and buffers) on `device`; files are taken from a shared queue in order.  Two to
 * three files in flight hide each file's host phases and launch gaps behind the others' kernels (47 700 xRT against
 * 40 200 for one file at a time).  results[i] belongs to paths[i]; rc is the code jt_process_file returned for it.
 * Returns the number of files that failed, or a negative JT_E_* when the arguments are unusable. */
typedef struct {
    int rc;
    char error[256];
    char output_path[1024];
    double wall_ms;
    jt_process_result result;
} jt_file_result;
int jt_process_files(int device, const char *const *paths, int n_files, int in_flight, const jt_host_config *base,
                     int frame_samples, int flac_flags, jt_file_result *results);

/* The same over several GPUs of the node (BASELINE configs[2] / configs[3]: files shard across the 8 x MI355X, no data-path
 * collective): ONE shared queue, n_devices x in_flight_per_device workers, longest file first.  A device id may repeat in
 * `devices` (two worker sets on one GPU).  device_of_file[i] (optional) receives the device that served paths[i].  This is the
 * in-process alternative to one process per GPU (bench.py / torchrun, jivetalking_amd/shard.py): same sharding unit, but the
 * queue is dynamic, so uneven file lengths do not leave a GPU idle. */
int jt_process_files_multi(const int *devices, int n_devices, const char *const *paths, int n_files, int in_flight_per_device,
                           const jt_host_config *base, int frame_samples, int flac_flags, jt_file_result *results, int *device_of_file);

/* A caller-owned handle pool for more than one batch (a long-running host: the Go shim pools its handles the same way): the handles
 * are opened once -- jt_process_files_multi opens and closes one pool per call, 0.1 s of a sub-second batch -- and every batch reuses
 * them with their buffers warm.  max_workers > 0 caps the worker count (devices x in_flight_per_device otherwise).  A device that
 * cannot be opened gets no worker (its worker tries the devices nobody serves); a pool without any worker fails every file of a batch
 * with the first jt_open error.  jt_handle_pool_workers returns the number of handles and writes the device of each (sorted by device).
 * One batch at a time per pool; different pools are independent.  Semantics of a batch: pool.go:122-153, as above. */
typedef struct jt_handle_pool jt_handle_pool;
int  jt_handle_pool_open(const int *devices, int n_devices, int in_flight_per_device, int max_workers, jt_handle_pool **out);
int  jt_handle_pool_workers(const jt_handle_pool *pool, int *devices_out, int cap);
int  jt_handle_pool_process_files(jt_handle_pool *pool, const char *const *paths, int n_files, const jt_host_config *base, int frame_samples, int flac_flags,
                           jt_file_result *results, int *device_of_file);
void jt_handle_pool_close(jt_handle_pool *pool);
/* Where the pool's last batch spent its time, summed over the files (ms; a diagnostic for the host that sizes in_flight_per_device):
 * [0] waiting for a free I/O set of the handle, [1] reading the input, [2] decode, [3] the four passes + host logic, [4] encode
 * (the handle's thread); [5] from the end of the handle's work on the file to the start of its write (a free finisher thread, or the
 * remainder of an MD5 that started inside Pass 4), [6] the STREAMINFO MD5 (it starts when Pass 4's s16 is complete and runs beside the
 * output analysis and the encode), [7] temp write + rename (a finisher thread);
 * [8] files counted.  Returns JT_POOL_STATS, writes at most cap values. */
#define JT_POOL_STATS 9
int  jt_handle_pool_stats(jt_handle_pool *pool, double *out, int cap);

/* ---- granular detector steps (the reference table-tests each of these: analyser_vad_test.go) ---- */
int    jt_host_vad_detect(const jt_interval *iv, int64_t n, double noise_floor_seed, jt_measurements *out);      /* detectVoiceActivity :728 */
void   jt_host_vad_split(const jt_interval *iv, int64_t n, double seed, double *otsu_raw, double *split, double *floor_, double *margin, int *tol);
int    jt_host_vad_speech_runs(const jt_interval *iv, int64_t n, double split, double margin, int tol, jt_region *out, int cap);  /* buildSpeechRuns :473 */
int    jt_host_vad_gap_tolerance(const int *flags, int64_t n);                                                     /* gapToleranceIntervals :405 */
void   jt_host_vad_gate_stats(const jt_interval *iv, int64_t n, double split, const jt_region *region,
                              double *voiced_low, double *noise_high, double *separation);                        /* deriveGateStatistics :220 */
int    jt_host_vad_noise_seed(const jt_interval *iv, int64_t n, double *noise_floor, double *threshold);           /* estimateNoiseFloorAndThreshold */
int    jt_host_vad_pick_low_cluster(const jt_interval *iv, int64_t n, double split, jt_region *out);               /* pickLowClusterRegion :630 */
double jt_host_vad_floored_fraction(const jt_interval *iv, int64_t n);                                             /* flooredFraction :708 */

/* speech election steps (the reference table-tests them: analyser_candidates_speech_test.go) */
double jt_host_score_speech_candidate(double rms_level, int64_t duration_ns, double noise_floor_db, double level_var);   /* scoreSpeechCandidateGrounded :326 */
double jt_host_level_variance(const jt_interval *iv, int64_t n, int axis /* 0 momentary LUFS, 1 RMS */);                   /* levelVariance, _shared.go:301 */
/* findBestSpeechRegion (:216): returns the number of candidates written (<= cap) and the elected region, or -1 when nothing is elected */
int    jt_host_find_best_speech_region(const jt_region *regions, int n_regions, const jt_interval *iv, int64_t n_iv,
                                       int has_noise_profile, double noise_floor_db, jt_region *best, jt_speech_candidate *cands, int cap);
/* the steps inside the election, as the reference table-tests them (analyser_test.go:264-1026) */
int64_t jt_host_intervals_in_range(const jt_interval *iv, int64_t n, int64_t start_ns, int64_t end_ns, jt_interval *out, int64_t cap);  /* getIntervalsInRange */
double jt_host_score_interval_window(const jt_interval *iv, int64_t n);                                  /* scoreIntervalWindow */
double jt_host_score_speech_interval_window(const jt_interval *iv, int64_t n);                           /* scoreSpeechIntervalWindow */
int    jt_host_measure_speech_candidate(const jt_region *region, const jt_interval *iv, int64_t n, jt_speech_candidate *out);   /* measureSpeechCandidateFromIntervals; 0 = nil */
int    jt_host_refine_golden_speech(const jt_region *cand, const jt_interval *iv, int64_t n, jt_region *out);                  /* refineToGoldenSpeechSubregion; 1 = refined */
/* calculateFrameLevel (encoder.go:235-257): VU level of one s16 frame for ProgressUpdate.Level, dB clamped to [-70, 0] */
double jt_host_frame_level_s16(const int16_t *pcm, int n);

/* sizeof() of the ABI structs, for binding self-checks (0 interval, 1 measurements, 2 host_config, 3 process_result, ...) */
int64_t jt_host_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif
