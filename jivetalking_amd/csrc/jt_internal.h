// jt_internal.h — shared declarations for libjtgpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <cmath>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>
#include <utility>
#include <ctime>
#include <sched.h>
#include "../../include/jtgpu.h"
#include "../../include/jt_host.h"

struct JtError { int code; std::string msg; };

#define JT_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { \
    throw JtError{JT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)}; } } while (0)
#define JT_REQUIRE(cond, code, text) do { if (!(cond)) throw JtError{(code), (text)}; } while (0)

// Superseded allocations.  hipFree / hipHostFree wait for the WHOLE DEVICE to go idle before they return: a handle that outgrows a buffer
// in the middle of a pass (a longer file than it has seen, a pass that needs more than the previous one) used to stall until every
// other handle's queued work had finished -- several workers on one GPU (jt_process_files_multi) synchronised each other at every
// such growth, and a worker behind a 4-second dynamic-loudnorm kernel waited for all of it (tools/probe_dynamic_concurrency.py: N
// handles took N times as long).  A buffer that is replaced is parked here instead, still valid for whatever is in flight on it, and
// freed when a handle closes or when more than graveyard_gb (jt_set_option, default 24) have piled up -- a free does its own device-wide wait,
// which is what makes it safe at any time.
struct DevGraveyard {
    std::mutex m; std::vector<std::pair<void *, int>> v; size_t bytes = 0;
    void put(void *p, size_t b, int host) { std::lock_guard<std::mutex> g(m); v.emplace_back(p, host); bytes += b; }
    size_t parked() { std::lock_guard<std::mutex> g(m); return bytes; }
    void drain() {
        std::vector<std::pair<void *, int>> w;
        { std::lock_guard<std::mutex> g(m); w.swap(v); bytes = 0; }
        for (auto &e : w) { if (e.second) (void)hipHostFree(e.first); else (void)hipFree(e.first); }
    }
    // bytes parked before a drain (jt_set_option(NULL, "graveyard_gb", ...); 0 = free at once)
    static std::atomic<size_t> &limit_ref() { static std::atomic<size_t> l{(size_t)24 << 30}; return l; }
    static size_t limit() { return limit_ref().load(std::memory_order_relaxed); }
    static void set_limit_gb(double gb) { if (!(gb >= 0)) gb = 0; if (gb > 4096) gb = 4096; limit_ref().store((size_t)(gb * (double)((size_t)1 << 30))); }
    // pinned allocations park here as well: a hipHostMalloc that fails while gigabytes sit parked drains and tries once more
    static hipError_t host_malloc(void **p, size_t bytes);
};
inline DevGraveyard &jt_graveyard() { static DevGraveyard *g = new DevGraveyard(); return *g; }    // (never destroyed: the runtime may be gone at exit)
inline hipError_t DevGraveyard::host_malloc(void **p, size_t bytes)
{
    hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (e != hipSuccess && jt_graveyard().parked() > 0) { (void)hipGetLastError(); jt_graveyard().drain(); e = hipHostMalloc(p, bytes, hipHostMallocDefault); }
    return e;
}
// jt_set_option(NULL, "poison_alloc", "1"): every device allocation is filled with 0xFF bytes (tools/stale_memory.py, the GPU suite under it)
inline std::atomic<int> &jt_poison_alloc() { static std::atomic<int> v{0}; return v; }
// host-stage timings on stderr from the handle-less host functions (set with any handle's host_timing option)
inline std::atomic<int> &jt_host_timing() { static std::atomic<int> v{0}; return v; }
// what jt_handle_pool_open asks of jt_open_ex when a device gets three or more handles: streams per handle (default 1) and whether
// their host waits sleep (default yes); jt_set_option(NULL, "pool_streams" / "pool_blocking_sync", ..)
inline std::atomic<int> &jt_pool_streams() { static std::atomic<int> v{1}; return v; }
inline std::atomic<int> &jt_pool_blocking() { static std::atomic<int> v{1}; return v; }
inline std::atomic<int> &jt_pool_numa() { static std::atomic<int> v{1}; return v; }       // option pool_numa: a pool's threads run on their GPU's NUMA node
// jt_process_file creates, reserves and maps its temporary output while the passes run when the output is expected to reach this many
// KiB (default 32 MiB: about 35 minutes of speech); jt_set_option(NULL, "early_temp_min_kb", ..) lets tests reach that path with short files
inline std::atomic<long long> &jt_early_temp_min_kb() { static std::atomic<long long> v{32 << 10}; return v; }

// Simple owning device buffer
template <typename T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    void retire() { if (p) { jt_graveyard().put(p, n * sizeof(T), 0); p = nullptr; n = 0; } }     // still valid for work in flight; freed later
    void ensure(size_t count) {
        if (count <= n && p) return;
        DevGraveyard &gy = jt_graveyard();
        if (p) { gy.put(p, n * sizeof(T), 0); p = nullptr; n = 0; if (gy.parked() > DevGraveyard::limit()) gy.drain(); }
        if (count == 0) count = 1;
        hipError_t e_ = hipMalloc((void **)&p, count * sizeof(T));
        if (e_ == hipErrorOutOfMemory) { (void)hipGetLastError(); gy.drain(); e_ = hipMalloc((void **)&p, count * sizeof(T)); }
        if (e_ != hipSuccess) { p = nullptr; throw JtError{JT_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e_)}; }
        // Nothing may depend on what an allocation holds: new pages happen to be zero, recycled ones are not, and a handle that has
        // processed a longer file keeps its stale samples behind a shorter one.  The process-wide option poison_alloc fills every allocation with 0xFF bytes
        // (NaNs / -1); tools/stale_memory.py and the whole GPU suite give the same results with it (tests/test_gpu_round2.py runs the
        // former).  The fill runs on the null stream, which the handle's non-blocking streams do not wait for: synchronise before use.
        if (jt_poison_alloc().load(std::memory_order_relaxed)) { JT_HIP(hipMemset(p, 0xFF, count * sizeof(T))); JT_HIP(hipStreamSynchronize(nullptr)); }
        n = count;
    }
    void zero(hipStream_t s) { if (p) JT_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
};

// ---------------------------------------------------------------- configuration surface
// The library reads NO environment variable on the per-file path (a Go host runs dozens of goroutines: getenv / setenv races are
// undefined behaviour).  Everything that used to be a JT_* variable is a field here, set through jt_set_option(h, key, value) and
// read by the launchers.  Default build: the schedule switches the tests run both ways, the parity kernels, diagnostics.  JT_AB
// build (make ab -> libjtgpu_ab.so): also the superseded kernel generations and the tuning knobs, compiled in for A/B runs; only
// that build imports JT_<KEY> variables, once, inside jt_open.
#define JT_OPT_BOOLS(X) \
    X(no_pass2_prefetch) X(no_early_biquad) X(no_early_pass3) X(no_early_plan) X(no_lim_keep) X(no_staged_finish) X(no_r128_first) \
    X(region_full_astats) X(host_timing) X(adeclick_exact) X(nlm_generic) X(p3_unfused) X(limiter_lanes) X(tp_unpruned) X(no_spec_direct) X(brickwall_f64) X(nf_low) X(nf_unpruned) X(as_avg_behind_spec) X(ln_no_batch) X(ln_no_stream) X(swr_untiled) X(flac_no_ahead) X(dk_unsorted) X(p2_device_join) X(dk_device_join)
#define JT_OPT_INTS(X) X(region_rot) X(tp_prune_min) X(ln_stream_stop)
#define JT_OPT_AB_BOOLS(X) \
    X(nlm_old) X(afftdn_old) X(adeclick_fused) X(dk_levinson_in_kernel) X(dk_no_xcd) X(dk_serial) X(dk_profile) X(dyn_one_wave) \
    X(dyn_no_cu_reserve) X(kw_two_sweeps) X(follow_tiles) X(follow_one_wave) X(tp_old) X(lim_profile) X(ups_no_stream8) X(ups_no_stream16) X(edge_polyphase) X(no_lim_s16)
#define JT_OPT_AB_INTS(X) X(dk_waves) X(dyn_steps) X(deess_chunk) X(deess_halo) X(follow_div) X(follow_dbg)
struct JtOpts {
#define X(k) bool k = false;
    JT_OPT_BOOLS(X) JT_OPT_AB_BOOLS(X)
#undef X
#define X(k) int k = 0;
    JT_OPT_AB_INTS(X)
#undef X
    int tp_prune_min = 1 << 20;          // signals at least this long take the branch-and-bound true peak (k_resample.hip); shorter ones the exhaustive kernels
    int ln_stream_stop = 0;              // test switch: the dynamic mode's stream path ends an attempt before every frame whose number is a multiple of this (as its
                                         // ring-end corner does); -1: a peak list of 64 entries, -N: a segment list of N entries (its two "list full" ways out)
    int region_rot = -1;                 // -1: the announced regions' chains on adeclick's second stream; r: region chain i behind full chain (i + r) % 4
};
#ifdef JT_AB
#define JT_AB_ON(expr) (expr)
#else
#define JT_AB_ON(expr) false
#endif
// 0 = set, JT_E_INVAL = unknown key / bad value, JT_E_UNSUPPORTED = a key of the JT_AB build asked of the default build
int jt_opts_set(JtOpts *o, const char *key, const char *value);

// ---------------------------------------------------------------- host-side plans
struct BiquadF32 { float b0, b1, b2, a1, a2; };      // TDII, a1/a2 already negated
struct BiquadF64 { double b0, b1, b2, a1, a2; };     // DF1 (f_ebur128.c FILTER macro), a1/a2 as in the difference equation

void jt_biquad_design(int type /*0 hp,1 lp*/, double freq, double q, int sr, double b[3], double a[3], int normalize);
void jt_kweight_design(int sr, BiquadF64 *pre, BiquadF64 *rlb);

struct SwrPlanHost {
    int phase_count = 0, filter_length = 0, center = 0; int64_t step = 0;
    std::vector<double> bank;   // [phase][tap]
};
void jt_swr_plan(SwrPlanHost *p, int in_rate, int out_rate);

// R128 host finishing (f_ebur128.c gating / LRA on per-100ms block energies)
struct R128Series { std::vector<double> M, S; double integrated, lra, lra_low, lra_high, rel_threshold; };
void jt_r128_finish(const double *block_sums, int64_t nblocks, int blk, int sr, bool dualmono, R128Series *out, bool integrated_only = false);
// libebur128-style (af_loudnorm.c) finishing on per-100ms block energies
void jt_loudnorm_finish(const double *block_sums, int64_t nblocks, int64_t s100, bool dual_mono, double scale_energy,
                        double *i, double *lra, double *thresh);

// ---------------------------------------------------------------- kernel launchers (k_*.hip)
// lane-serial family
void launch_frame_stats(const float *in, int64_t n_total /*frames*channels*/, int samples_per_frame,
                        double *sumsq, double *peak, int64_t nframes, hipStream_t s);
// mode: 0 = float 1/sqrt2 (Pass 1 / Pass 2), 1 = s16 integer matrix, 2 = s32 via float 0.5 (the band graphs of integer sources)
void launch_frame_sumsq_s16(const int16_t *in, int64_t n, int spf, double *sumsq, int64_t nframes, hipStream_t s);
// frames of different lengths: off[f] .. off[f + 1] = frame f's range in samples per channel (nframes + 1 entries on the device)
void launch_frame_stats_var(const float *in, int channels, const int64_t *off, double *sumsq, double *peak, int64_t nframes, hipStream_t s);
// aformat=channel_layouts=mono: libswresample's default rematrix row for one FRONT_CENTER output (k_lane.hip, k_downmix).  k = inputs with
// a non-zero coefficient (native channel order), nz[] their channel indices, cf[] the float coefficients, ci[] the S16P integer ones
struct DownmixRow { int k; int stereo; int nz[8]; float cf[8]; int ci[8]; };
// false: a layout the restatement does not cover (channels beyond SIDE_RIGHT, a mask that does not match the channel count)
bool jt_downmix_row(int channels, unsigned long long mask, int mode, DownmixRow *row);
unsigned long long jt_default_layout(int channels);          // av_channel_layout_default: what swr_init gives a source without a layout
void launch_downmix(const float *in, float *out, int64_t frames, int channels, int mode, const DownmixRow &row, hipStream_t s);
void launch_s16_to_f32(const int16_t *in, float *out, int64_t n, hipStream_t s);
void launch_s16_to_f32_pair(const int16_t *in0, int64_t n0, const int16_t *in1, int64_t n1, float *out0, float *out1, hipStream_t s);   // two ranges, one launch
void launch_s16_to_f64(const int16_t *in, double *out, int64_t n, double gain, int gain_in_float, hipStream_t s);
void launch_biquad_f32(const float *in, float *out, int64_t n, int nstages, const BiquadF32 *st, hipStream_t s);
struct DynParams {
    int gate_on, comp_on, deess_on;
    // gate
    double g_attack, g_release, g_lin_knee_stop, g_thres, g_knee_start, g_knee_stop, g_ratio, g_knee, g_range, g_makeup;
    // comp
    double c_attack, c_release, c_thres, c_knee_start, c_knee_stop, c_adj_knee_start, c_ckstop, c_ratio, c_knee, c_makeup, c_mix;
    // deesser
    double d_intensity, d_maxdess, d_iir;
};
void jt_dyn_design(const jt_filter_params *p, int sr, DynParams *d);
// in_has_slack: 16 readable bytes behind in[n-1] and tmp64[n-1] (the LDS-streamed followers read whole 16-byte groups)
void launch_dynamics(const float *in, float *out_f32, double *tmp64, double *tmp64b, double *states, int64_t n, const DynParams &d, hipStream_t s,
                     const JtOpts &o, bool in_has_slack = false);

// astats (k_astats.hip)
struct jt_ctx;
void launch_biquad_di_f32(const float *in, float *out, int64_t n, BiquadF32 hp, BiquadF32 lp, hipStream_t s);
// hp/lp: b0 b1 b2 -a1 -a2 as af_biquads.c's doubles; mode: negotiated sample format of the band graph (0 fltp, 1 s16p, 2 s32p)
void launch_band_rms(const float *in, int64_t n, int nbands, const double (*hp)[5], const double (*lp)[5], int mode, double *sums, hipStream_t s);

// limiter (exact, chunked at provably clean points)
// limiter, first sweep (blk == 256): block maxima of |in| and out = in * gain; launch_limiter_f64 expects `out` to hold that copy
struct LimOut16 { int16_t *s16; float *f32; };        // the limiter's output as k_f64_to_s16(round_via_float = 1) would convert it (the brickwall of Pass 4)
void launch_absmax_copy_f64(const double *in, double *out, int64_t n, double gain, double *out_max, int64_t nblk, hipStream_t s, const LimOut16 *o16 = nullptr);
void launch_absmax_conv_s16(const int16_t *in, double *conv, double *out, int64_t n, double vol, int vol_in_float, double gain, double *out_max,
                            int64_t nblk, hipStream_t s);
struct LimSrc16 { const int16_t *p; double vol; int vol_in_float; };
void launch_limiter_f64(const double *in, double *out, int64_t n, int sr, double limit, int buffer_size,
                        double release_s, double asc_coeff, const double *block_max, int64_t nblk, int blk, int need, int target,
                        int64_t *cand, int64_t ntargets, double in_gain, double *scratch_delta, int64_t *scratch_pos, hipStream_t s, double *scratch_lp,
                        bool lane_per_segment = false, bool lim_profile = false, const LimSrc16 *src16 = nullptr, const LimOut16 *o16 = nullptr);
// lane_per_segment: k_limiter_f64 instead of k_limiter_wave; src16: the wave kernel converts its hot segments from the s16 source (in may be null)
bool jt_limiter_wave_ok(int buffer_size);

// resampler / true peak
void launch_resample_to_s16(const float *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                            int64_t step, int16_t *out, int64_t m, hipStream_t s, const JtOpts &o);
int64_t launch_resample_range_to_s16(const float *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                                     int64_t step, int64_t m, int64_t m_first, int64_t m_count, int16_t *dst, int64_t dst_cap, hipStream_t s);
int64_t jt_resample_range_cap(int64_t n, int phase_count, int filter_length, int64_t step, int64_t m, int64_t m_count);
void launch_true_peak_f32(const float *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                          int64_t step, int blk, double *block_tp, int64_t nblocks_alloc, int64_t m_total, hipStream_t s, const JtOpts *o = nullptr);
// branch-and-bound true peak (block maxima of the evaluated units only: the PREFIX maximum is what equals the exhaustive kernels');
// false = plan / length not served.  norms = SwrDev::tp_norms
size_t jt_tp_prune_scratch_bytes(int64_t n, int phase_count, int filter_length, int64_t step, int blk);
bool launch_true_peak_f32_pruned(const float *in, int64_t n, const double *bank, int phase_count, int filter_length, int center, int64_t step,
                                 int blk, double *block_tp, int64_t nblocks_alloc, int64_t m_total, const double norms[3], void *scratch,
                                 size_t scratch_bytes, hipStream_t s, const int **kept_dev = nullptr, int64_t *units = nullptr, int64_t *seeds = nullptr);
void launch_true_peak_f64(const double *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                          int64_t step, int blk, double *block_tp, int64_t nblocks_alloc, int64_t m_total, hipStream_t s, const JtOpts *o = nullptr);
// Pass-3 192 kHz streams (FLT path from s16, DBL path after the limiter prefix); K-weighted afterwards by launch_kweight_blocks_*
void launch_resample_stream_s16_f32(const int16_t *in, int64_t n, const float *bankf, const float *bankf_scaled, int phase_count, int filter_length, int center,
                                    int64_t step, int64_t m_total, float *out, hipStream_t s);
void launch_resample_stream_f64(const double *in, int64_t n, const double *bank, int phase_count, int filter_length, int center,
                                int64_t step, int64_t m_total, double *out, hipStream_t s, const JtOpts &o);
// Pass 3 in one sweep (k_p3_fused, k_resample.hip): the sweep of a K-weighting job whose chunks are the resampler's periods; the 192 kHz
// stream is never stored.  flush: loudnorm's flush frame, the last `flush` outputs metered again behind the stream (0: none).
struct KwSweep; struct KwCoef;
bool jt_p3_fused_supported(int phase_count, int filter_length, int64_t step, int blk, int64_t flush);
int  jt_p3_fold_powers(const KwCoef &k, int P, double *out);
void launch_p3_fused_s16(const int16_t *in, int64_t n, const float *bankf, const float *bankf_scaled, int P, int center, int64_t step, int64_t m_total,
                         int64_t flush, const KwSweep &W, const double *mpow_dev, hipStream_t s);
void launch_p3_fused_f64(const double *in, int64_t n, const double *bank, int P, int center, int64_t step, int64_t m_total, int64_t flush, const KwSweep &W,
                         const double *mpow_dev, hipStream_t s);
void launch_f64_to_s16(const double *in, int16_t *out, float *out_f32, int64_t n, int round_via_float, hipStream_t s);
void launch_f32_to_f64(const float *in, double *out, int64_t n, hipStream_t s);

// spectral
void launch_aspectralstats(const float *in, int64_t n, int sr, int win_size, const float2 *twiddle, const float *hann,
                           jt_spectral *hops, int64_t nhops, int sel_blk, int64_t nframes, hipStream_t s);
// afftdn
struct AfftdnPlanHost {
    int sr, A, W, L, bins, nbands;
    std::vector<int> bin2band; std::vector<double> window, alpha, beta, spread, abs_var, min_abs_var, rel_var;
    double max_gain, floor = 0, noise_floor = 0;
};
void jt_afftdn_plan(AfftdnPlanHost *pl, int sr, double nr, double nf, const double *band_noise);
struct AfftdnDev {
    int A, W, L, bins, nbands; double max_gain;
    const int *bin2band; const double *window, *alpha, *beta, *spread, *abs_var, *min_abs_var; const float2 *twiddle;
    // noise tracking (tn=1): band shape, the magnitude floor of the flatness measure, max_var before frame t (mvseq[t]) and after it
    // (mvseq[t + 1]); track_out[t] = the floor a spectrally flat frame votes for (NaN = frame not flat), written by the analysis mode
    const double *rel_var = nullptr; double floor = 0; const double *mvseq = nullptr; double *track_out = nullptr;
    // most bark bands any run of 64 consecutive bins (a wave's segment of either half of the spectrum) touches: row length of the
    // (segment, band) partial sums in k_afftdn_grp
    int seg_span = 0;
};
// mode 0: static noise floor; 1: per-frame variances from d.mvseq (tn=1, second sweep); 2: tn=1 first sweep, writes d.track_out only
void launch_afftdn(const float *in, float *out, int64_t n, const AfftdnDev &d, int frames_per_chunk, int warm_frames, hipStream_t s, const JtOpts &o, int mode = 0);
int64_t jt_afftdn_nframes(int64_t n, int A, int W);
// anlmdn
void launch_anlmdn(const float *in, float *out, int64_t n, int K, int S, float sw, float smooth, float lut_scale, hipStream_t s, const JtOpts &o);

// adeclick (k_declick.hip)
struct jt_ctx;
bool jt_adeclick_supported(int sample_rate, double window_ms, double overlap_pct, double ar_pct, int method, std::string *why);
// loudnorm, dynamic mode (k_loudnorm.hip): what the frame loop needs besides the stream and the per-frame series
struct LoudnormDynParams {
    double target_i, target_lra, target_tp_lin, measured_thresh, offset_lin, delta0;
    double weights[21], kwb[5], kwa[5];
    int above0, dual_mono;
    int64_t n_inner; int final_len;
    int stream_stop;   // option ln_stream_stop (test switch)
    int no_batch;      // option ln_no_batch: the limiter's harmless peaks one detector call at a time (what the batched step is tested against)
};
void jt_loudnorm_series(const double *bs, int64_t nfull, int64_t s100, bool dual_mono, int64_t n_inner, double *out);
void jt_kweight_coeffs5(int sr, double b[5], double a[5]);
// the stream path of the dynamic mode (k_loudnorm.hip): an envelope segment the limiter's state machine decided on, the control block
// the kernels of one attempt share, and the scratch they work in (jt_lns_scratch_bytes carves one allocation)
struct LnsSeg { int t, len, kind, c0, al, pad; double g0, g1; };                 // t: relative to the attempt's first output sample (LnsCtl::tbase); kind: 0 constant g0, 1 attack ramp, 2 release ramp; +256: second layer
struct LnsCtl { int active, ok, ka, kbe, npk, nseg, why, attempts; long long frames; int iters, refills; long long cycles; int why_mask, pad; long long tbase; };   // why_mask: bit r = an attempt ended for reason r   // why: 1 state not in its steady form, 2 peak list full, 3 segment list full, 4 ring-end corner
// what a frame's scan range (its output position + 1920, one frame long) holds of the peak list: entries [fi, fe), the last one's time, the
// next entry's time and gain reduction, the smallest gain reduction inside (k_lns_frames; the machine's O(1) test for a frame SUSTAIN passes)
struct LnsFrame { int fi, fe, l_t, e_t; double ming, e_g; };
struct LnsBufs { LnsCtl *ctl; double *G, *Gn, *E; unsigned long long *bm; unsigned short *woff; int *bcnt, *boff; int *pk_t; double *pk_v, *pk_g; LnsSeg *seg; int pk_cap, seg_cap; LnsFrame *fr; };
size_t jt_lns_scratch_bytes(int64_t n, int64_t n_inner, LnsBufs *B, unsigned char *base);
// carry: LN_CARRY = 256 doubles of device memory (the state one launch of the workgroup kernel hands to the next)
void launch_loudnorm_dynamic(const double *x, int64_t n, const LoudnormDynParams &P, const double *series, double *ring, double *y, double *dbg, hipStream_t s,
                             double *carry, const JtOpts &o, const LnsBufs *stream = nullptr);
void launch_scale_f64(const double *in, double *out, int64_t n, double g, hipStream_t s);
void launch_swr_plain_f64(const double *in, int64_t n, const double *bank, int P, int L, int center, int64_t step, int64_t m_total, double *out, hipStream_t s,
                          const double *bankT = nullptr);     // bankT[i * P + phase]: the bank transposed (the tiled kernel)
void launch_adeclick(jt_ctx *h, const double *in, double *out, int64_t n, int sample_rate, double threshold, double window_ms,
                     double overlap_pct, double ar_pct, double burst, double gain, unsigned long long *d_stats, hipStream_t s, int method = 1);

// FLAC encoder (k_flac.hip): analyse + scan fill recs/offs/summary, emit writes the frames at their byte offsets
size_t jt_flac_rec_bytes(int64_t nframes);
void launch_flac_analyse(const int16_t *pcm, int64_t n, int sr_code, int sr_extra_bytes, int sr_extra_val, void *recs,
                         long long *offs, void *summary, hipStream_t s);
void launch_flac_emit(const int16_t *pcm, int64_t n, int sr_code, int sr_extra_bytes, int sr_extra_val, const void *recs,
                      const long long *offs, uint8_t *out, void *summary, hipStream_t s);
// fork-join over [0, n) on the process-wide worker pool (jt_plan.cpp); fn(lo, hi, part)
void jt_parallel_for(int64_t n, const std::function<void(int64_t, int64_t, int)> &fn, int *nparts_out = nullptr);
#define JT_FLAC_MD5_DEFER 0x200      /* internal flag of jt_flac_encode: the PCM is copied to pinned memory, the caller hashes it and patches STREAMINFO (bytes 26..41 of the image) */
void jt_md5(const void *data, size_t len, uint8_t out[16]);
int jt_flac_encode_file(jt_ctx *h, int stage, int flags, const uint8_t **data, int64_t *len, jt_flac_info *info);      // jt_io.cpp      // RFC 1321 (jt_plan.cpp), STREAMINFO signature

// FLAC / WAV input leg (k_flacdec.hip); the structs mirror the kernels' layouts
struct JtFlacStream { int channels = 0, bps = 0, sample_rate = 0, min_blocksize = 0, max_blocksize = 0; long long total_samples = 0, audio_offset = 0, len = 0; };
struct JtFlacCand { long long pos, number; int blocksize, hdr_len, ch_assign, variable; };
struct JtFlacParsed { long long end; int ok, wasted_any; long long sub_bit[8]; };
struct JtFlacFrame { long long pos, out_offset; int blocksize, ch_assign; long long sub_bit[8]; };
void launch_flacdec_find(const uint8_t *file, const JtFlacStream &s, JtFlacCand *cands, int *count, int cap, hipStream_t st);
void launch_flacdec_parse(const uint8_t *file, const JtFlacStream &s, const JtFlacCand *cands, int ncand, JtFlacParsed *out, hipStream_t st);
// mono streams: every candidate parsed AND decoded by one walk, into its own row of max_blocksize samples; the chain's rows gathered
void launch_flacdec_decode_cand(const uint8_t *file, const JtFlacStream &s, const JtFlacCand *cands, int ncand, int *rows, JtFlacParsed *parsed, hipStream_t st);
void launch_flacdec_finish_cand(const JtFlacStream &s, const JtFlacFrame *frames, long long nframes, const int *rows, int *out_i32, float *out_f32, hipStream_t st);
void launch_flacdec_decode(const uint8_t *file, const JtFlacStream &s, const JtFlacFrame *frames, long long nframes, long long total,
                           int *planar, int *errs, int *out_i32, float *out_f32, hipStream_t st);
void launch_pcm_convert(const uint8_t *raw, long long nvals, int fmt, float *out_f32, int *out_i32, hipStream_t st);

// ---------------------------------------------------------------- pinned host staging
// Every device->host result of a pass lands in one pinned arena, so a pass enqueues all its kernels and copies back to back
// and synchronises ONCE (pageable destinations would serialise each copy through a bounce buffer).
struct HostArena {
    unsigned char *p = nullptr; size_t cap = 0, off = 0;
    HostArena() {}
    HostArena(const HostArena &) = delete; HostArena &operator=(const HostArena &) = delete;
    ~HostArena() { if (p) (void)hipHostFree(p); }
    // caller guarantees nothing is in flight (start of a pass)
    void begin(size_t bytes) {
        off = 0;
        if (bytes <= cap) return;
        if (p) { jt_graveyard().put(p, cap, 1); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 2 + (1u << 20);
        JT_HIP(DevGraveyard::host_malloc((void **)&p, want));
        cap = want;
    }
    template <typename T> T *take(size_t count) {
        off = (off + 63) & ~(size_t)63;
        if (off + sizeof(T) * count > cap) throw JtError{JT_E_HIP, "pinned host arena exhausted"};
        T *r = reinterpret_cast<T *>(p + off); off += sizeof(T) * count; return r;
    }
};

// cached resampler plans (the kaiser bank depends on the rate pair only) with their device copies
struct SwrDev {
    int in_rate = 0, out_rate = 0; SwrPlanHost pl; DevBuf<double> bank_d; DevBuf<float> bank_f;
    DevBuf<double> bank_dT;      // bank_d transposed, [tap][phase] (k_swr_tile_f64)
    double tp_norms[3] = {0, 0, 0};   // over the tap rows: max sum |t_i|, max |sum t_i|, max sum |t_i| |i - c| (c fixed): the true peak's branch-and-bound bounds
    DevBuf<float> bank_fs;       // bank_f * 2^-15 (s16 sources: the int -> float scale folded into the taps); empty unless that product is exact
    int64_t out_len(int64_t n) const { return (int64_t)(((__int128)n * pl.phase_count + pl.step - 1) / pl.step); }
};

// deferred results ("jobs"): enqueue = kernels + async copies into the arena; finish = host arithmetic after the pass's sync
struct jt_ctx;
struct AstatsJob {
    const unsigned char *hb = nullptr; const unsigned long long *eh = nullptr;
    size_t o_part = 0, o_runs = 0, o_nf = 0, o_smin = 0, o_smax = 0;
    int nparts = 0, nf_parts = 0, nsig = 0; bool have_nf = false; int64_t n = 0;
    bool levels_only = false;       // RMS / peak level and crest factor only (the announced regions' samples)
};
// sA: reduce -> min/max -> runs; sB: noise floor; sC: exp-average sigma chain (three independent chains; pass the same stream to serialise)
void jt_astats_enqueue(jt_ctx *h, const float *x, int64_t n, int sr, AstatsJob *job, hipStream_t sA, hipStream_t sB, hipStream_t sC,
                       unsigned long long *ehist = nullptr, bool levels_only = false);     // ehist: the job's own 8192-bin histogram (default: the shared one)
void jt_astats_finish(const AstatsJob *job, jt_astats *out);
struct KwJob { const double *hc = nullptr; int64_t nchunks = 0, nfull = 0; int m = 1; };
// BS.1770 K-weighting as two transposed-DF2 biquads (pre-filter shelf b/a, RLB high-pass c/d): k_lane.hip KW2_STEP
struct KwCoef { double b0, b1, b2, a1, a2, c0, c1, c2, d1, d2; };
// What a sweep over L-sample chunks leaves per chunk c (k_kw1, or a kernel that produces the signal itself): the zero-state end state
// zs[4c..], sum zs^2 in csum[c], max |x| in cpeak[c], the four sums zs_j g_j[k] in cross[4c..]; gtab = the homogeneous-response table
// g[j][k] (j < L) on the device and on the host, tail = samples of the last chunk.  (two_sweeps: the A/B build's older form.)
struct KwSweep {
    KwCoef k; int64_t L = 0, nchunks = 0, tail = 0; const double *gtab = nullptr, *gtab_host = nullptr;
    double *zs = nullptr, *csum = nullptr, *cpeak = nullptr, *cross = nullptr; const double *pw = nullptr; int nterms = 0; bool two_sweeps = false;
};
// scratch of a job that outlives the pass arenas (the Pass-3 measurement started early by Pass 2): device doubles / pinned doubles
struct KwScratch { double *dev = nullptr; double *pin = nullptr; };
void jt_kweight_scratch_sizes(int64_t n, int blk, size_t *dev_doubles, size_t *pin_doubles, int64_t chunk_len = 0);
// the K-weighting job around a caller-supplied sweep (chunk_len > 0: that chunk length, a divisor of blk)
void jt_kweight_enqueue_sweep(jt_ctx *h, int64_t n, int rate, int blk, int64_t chunk_len, KwJob *job, hipStream_t s, const KwScratch *ext,
                              const std::function<void(const KwSweep &)> &sweep);
void jt_kweight_enqueue_f32(jt_ctx *h, const float *in, int64_t n, int rate, int blk, KwJob *job, hipStream_t s, const KwScratch *ext = nullptr);
void jt_kweight_enqueue_f64(jt_ctx *h, const double *in, int64_t n, int rate, int blk, KwJob *job, hipStream_t s, const KwScratch *ext = nullptr);
// sums/peaks: nfull+1 entries (last = trailing partial block)
void jt_kweight_finish(const KwJob *job, std::vector<double> &sums, std::vector<double> &peaks);
size_t jt_arena_bytes_for(int64_t n);          // generous bound on the arena bytes one analysis of n samples stages

// ---------------------------------------------------------------- context
struct jt_ctx {
    int device = 0;
    JtOpts opts;
    hipStream_t stream = nullptr;
    int n_streams = 8; std::vector<hipStream_t> owned_streams;      // jt_open_ex: which of the streams below are aliases
    bool blocking = false; hipEvent_t ev_block = nullptr;          // JT_OPEN_BLOCKING_SYNC: host waits sleep instead of spinning (jt_stream_sync)
    // jt_pass4 calls this (when set) as soon as the delivered s16 is complete on the main stream, BEFORE the output analysis is queued: a
    // handle pool copies the PCM to the host there and starts the file's STREAMINFO MD5 (64 ms of a host core per ten minutes) while the
    // analysis, the FLAC encode and the download of the image are still to come.  pcm_early names that copy for jt_flac_encode.
    std::function<void(jt_ctx *)> p4_output_hook;
    struct PcmEarly { const int16_t *pcm = nullptr; size_t n = 0; } pcm_early;
    hipEvent_t ev_pcm[2] = {nullptr, nullptr};                      // end of that copy, per I/O set
    // auxiliary streams: the independent parts of an analysis (astats chains, true peak + K-weighting, spectral) are forked
    // onto them and joined back into `stream` with events, so latency-bound kernels overlap instead of queueing
    hipStream_t aux[8] = {};                   // [0..3] the analysis chains of a pass, [4..7] the chains of announced output regions
    hipEvent_t ev_fork = nullptr, ev_join[8] = {};
    // marks inside the full-length analysis of a pass (analysis_enqueue): [0..3] the ends of its four chains, [4] the astats part of
    // chain 2 (before aspectralstats), [5] the K-weighting job of chain 1 (before the noise floor); ev_stats: Pass 4's loudnorm statistics
    hipEvent_t ev_chain[7] = {}, ev_stats = nullptr, ev_nf = nullptr;
    std::string err;
    std::atomic<int> cancelled{0};
    bool hold_cancel = false;            // inside jt_process_file: loading the input must not clear a cancel that already arrived
    // input
    int sr = 0, channels = 0; int64_t n = 0;
    int src_fmt = 0;                    // the decoder's native sample format: 0 flt/dbl, 1 s16 (also u8), 2 s32 (24/32-bit integer)
    unsigned long long ch_mask = 0;     // the source's channel layout (WAVEFORMATEXTENSIBLE / libavutil bit order); 0 = the default layout of its channel count
    // the decoder's frame cadence of the current input (what frame_samples = 0 means; jtgpu.h: jt_input_frame_layout): jt_load_audio
    // sets it from the file, every other way of handing over PCM leaves 4096 / constant.  dec_frame_lens is non-empty only when the
    // frames differ in length (a variable-blocksize FLAC); d_frame_off = their start offsets (n + 1 entries) for k_frame_stats_var.
    int dec_frame_samples = 4096; int64_t dec_frames = 0; std::vector<int32_t> dec_frame_lens; DevBuf<int64_t> d_frame_off;
    DevBuf<float> band_mono;            // the band graphs' own down-mix of an integer stereo source
    const float *in_raw = nullptr;      // interleaved (owned or attached)
    DevBuf<float> in_owned;
    DevBuf<float> mono;                 // downmixed when channels == 2
    const float *in_mono = nullptr;
    // stage buffers
    DevBuf<float> work_a, work_b;       // f32 ping-pong at source rate
    DevBuf<int16_t> s16_p2, s16_p4;     // Pass-2 / Pass-4 outputs
    int64_t m_p2 = 0, m_p4 = 0; int out_rate = 0;
    DevBuf<double> f64_a, f64_b, f64_c; // f64 ping-pong at output rate (+ the brickwall's output when adeclick's input is still being metered)
    // Pass 3 leaves the alimiter prefix's output in f64_b; Pass 4 applies the same prefix to the same samples and takes it from there
    // (set by pass3_core, dropped by anything that rewrites the Pass-2 output or the f64 buffers)
    struct LimKeep { bool valid = false; const int16_t *src = nullptr; int64_t m = 0; int rate = 0; double pre_gain_db = 0, limit = 0; } lim_keep;
    DevBuf<float> stream_f; DevBuf<double> stream_d;   // 192 kHz loudnorm-measurement stream
    DevBuf<unsigned char> ln_scratch;                             // loudnorm dynamic mode, stream path: LnsBufs
    DevBuf<double> stream_y, ln_ring, ln_series, ln_carry;       // loudnorm dynamic mode: output stream, limiter ring, per-frame series
    // scratch
    DevBuf<double> d_scr0, d_scr1, d_scr2, d_scr3;
    DevBuf<unsigned long long> ehist; DevBuf<float> as_g, as_p;
    DevBuf<jt_spectral> spec_hops;
    DevBuf<float2> twiddle; int twiddle_n = 0; DevBuf<float> hann; int hann_n = 0;
    DevBuf<double> bank_d; DevBuf<float> bank_f;
    DevBuf<int64_t> lim_bounds, lim_pos; DevBuf<double> lim_delta, lim_lp;
    DevBuf<int> af_bin2band; DevBuf<double> af_tab, af_track; double af_last_floor = 0;      // af_track: tn=1 votes + per-frame max_var
    HostArena pin;
    // per-pass bump allocator over d_scr2 for the astats jobs: the three chains of different jobs run on different streams, so
    // every job needs its own scratch slice (sized at the start of the pass, never reallocated while work is queued)
    size_t as_off = 0;
    void as_begin(size_t bytes) { d_scr2.ensure((bytes + 7) / 8); as_off = 0; }
    unsigned char *as_take(size_t bytes) {
        if (as_off + bytes > d_scr2.n * sizeof(double)) throw JtError{JT_E_HIP, "astats scratch exhausted"};
        unsigned char *r = reinterpret_cast<unsigned char *>(d_scr2.p) + as_off; as_off += (bytes + 255) & ~(size_t)255; return r;
    }
    // per-pass bump allocator over d_scr0 for the K-weighting jobs (several can be in flight before the pass's sync)
    DevBuf<unsigned char> tp_scr; size_t tp_scr_off = 0;     // the branch-and-bound true peak's unit bounds / lists (one long analysis per pass)
    size_t tp_off = 0, tp_cap = 0;      // bump allocator over d_scr1: the per-block true-peak maxima of a pass (zeroed once by pass_begin)
    size_t kw_off = 0;
    void kw_begin(size_t doubles) { d_scr0.ensure(doubles); kw_off = 0; }
    double *kw_take(size_t doubles) {
        if (kw_off + doubles > d_scr0.n) throw JtError{JT_E_HIP, "K-weighting scratch exhausted"};
        double *r = d_scr0.p + kw_off; kw_off += (doubles + 7) & ~(size_t)7; return r;
    }
    SwrDev swr[4]; int swr_next = 0;
    // K-weighting: homogeneous-response tables of the one-sweep kernel, per (rate, chunk length)
    struct KwTab { int rate = 0; int64_t L = 0; std::vector<double> g; DevBuf<double> dev; int nterms = 1; double gram[10]; } kw_tab[8]; int kw_tab_next = 0;   // g: table [L][4] | powers of F^L
    DevBuf<double> p3_mpow; int p3_mpow_P = 0;         // the fused Pass-3 sweep's fold matrices (k_p3_fused), per period length
    DevBuf<float> region_f; DevBuf<int16_t> region_s16;
    // Pass 3's measurement of the Pass-2 output for the no-prefix plan (the usual one), queued by Pass 2 itself on a stream of
    // its own as soon as the s16 output exists: it runs beside Pass 2's analysis tail and the host work between the passes.
    // jt_pass3 collects it when the plan has no limiter prefix and measures normally otherwise.
    // Head of the Pass-2 chain (highpass / lowpass cascade + anlmdn: the stages whose parameters do not come from the Pass-1
    // measurements) started by jt_pass2_prefetch while the host still works on those measurements.
    struct SpecPass2 {
        bool pending = false; int nst = 0; BiquadF32 st[2]; bool nlm = false; double nlm_p[4] = {0, 0, 0, 0};
        int stages = 0; hipStream_t stream = nullptr; hipEvent_t done = nullptr;
        bool armed = false; jt_filter_params armed_p;      // jt_pass2_prefetch_after_pass1: started by the next jt_pass1
    } spec_p2;
    struct SpecLoudnorm {
        bool pending = false; KwJob kw; int64_t nfull = 0; int blk = 0;
        DevBuf<double> dev; double *pin = nullptr; size_t pin_cap = 0; hipStream_t stream = nullptr; hipEvent_t fork = nullptr;
    } spec_ln;
    // regions announced by jt_region_prefetch ([0] = stage 2, [1] = stage 4) and the results their pass stored
    struct RegionSlot { bool armed = false, valid = false; double start_s[2] = {0, 0}, dur_s[2] = {0, 0}; jt_region_sample out[2]; };
    RegionSlot region_slot[2];
    DevBuf<double> declick_scr; DevBuf<unsigned long long> declick_stats; DevBuf<int> declick_heavy;
    // adeclick's split pipeline (front kernel -> solver kernels, k_declick.hip): per-window flagged count, index list, right-hand side, aux
    DevBuf<int> declick_F, declick_lists; DevBuf<unsigned short> declick_idx; DevBuf<double> declick_rhs, declick_aux, declick_r, declick_wlut, declick_prod;
    DevBuf<unsigned long long> declick_ctl;
    hipStream_t dk_stream = nullptr; hipEvent_t dk_ev[2] = {nullptr, nullptr};
    // Pass 3 with the limiter prefix, started inside Pass 2 once its loudness / true peak are known (jt_pass3_plan_hook)
    struct EarlyPass3 {
        jt_plan_fn fn = nullptr; void *user = nullptr; bool armed = false, mark_kw = false;
        bool valid = false; jt_limiter_plan plan{}; KwJob kw; int64_t nfull = 0; int blk = 0;
        DevBuf<double> dev; double *pin = nullptr; size_t pin_cap = 0; hipEvent_t ev[2] = {nullptr, nullptr};
    } early_p3;      // the wide-band solver's stream (created on first use)
    // FLAC output leg: frame records, byte offsets, summary, encoded frames; pinned host copies of the PCM (MD5) and the file
    DevBuf<unsigned char> fl_rec, fl_out; DevBuf<long long> fl_off; DevBuf<int16_t> fl_pcm;
    // two sets: a handle pool hands a finished file's MD5 + write to a finisher thread and starts the next file in the other set
    HostArena io_pcm[2], io_flac[2]; int io_set = 0;
    HostArena io_small;                                           // the file legs' small device->host results (counts, summaries, frame candidates): pinned, so that the copies do not wait inside the runtime
    HostArena &pin_pcm() { return io_pcm[io_set]; }
    HostArena &pin_flac() { return io_flac[io_set]; }
    struct FlacDeferred { const int16_t *pcm = nullptr; size_t n = 0; } flac_deferred;      // JT_FLAC_MD5_DEFER: what the caller still has to hash
    // input leg: file image, frame candidates / parse results / frame table, planar and interleaved integer PCM
    DevBuf<unsigned char> in_file, in_tab; DevBuf<int> in_planar, in_i32;
    std::vector<jt_interval> last_intervals;      // the 250 ms interval series of the last Pass-1 analysis (run record, sidecar)
    std::vector<double> host_fss, host_fpk;       // analyse_core's per-decoder-frame sums / peaks and 100 ms frame records: grown, never cleared
    std::vector<jt_frame_meta> host_meta;
    jt_timers timers{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
};

// ---------------------------------------------------------------- C-ABI entry wrappers (jt_api.cpp, jt_io.cpp)
// Every entry point except the ones that belong between Pass 1 and Pass 2 (band RMS, region announcement, Pass 2 itself) first
// retires a Pass-2 head that jt_pass2_prefetch may have left running: it reads the input and writes the Pass-2 work buffers.
void jt_spec_pass2_cancel(jt_ctx *h);
#define JT_API_BEGIN_KEEP(h) if (!(h)) return JT_E_INVAL; try { JT_HIP(hipSetDevice((h)->device));
#define JT_API_BEGIN(h) JT_API_BEGIN_KEEP(h) jt_spec_pass2_cancel(h);
// Host waits of a handle.  Every wait of the HIP runtime on this part spins on the completion signal for as long as it lasts --
// hipStreamSynchronize, hipEventSynchronize (with or without hipEventBlockingSync), and copies into pageable memory alike
// (tools/ubench/wait_cpu.hip: 40 ms of CPU for a 40 ms kernel) -- so a pool of handles burns a host core per handle.  A handle opened
// with JT_OPEN_BLOCKING_SYNC polls instead: hipEventQuery, a few yields, then 50 us sleeps (a wait ends at most that much late; a
// file has a dozen of them).
inline hipError_t jt_event_wait(jt_ctx *h, hipEvent_t ev)
{
    if (!h->blocking) return hipEventSynchronize(ev);
    for (int i = 0;; ++i) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        (void)hipGetLastError();                               // (hipErrorNotReady is sticky in the thread's last-error slot)
        if (i < 4) sched_yield();
        else { timespec ts{0, 50000}; nanosleep(&ts, nullptr); }
    }
}
inline hipError_t jt_stream_sync(jt_ctx *h, hipStream_t s)
{
    if (!h->blocking) return hipStreamSynchronize(s);
    const hipError_t e = hipEventRecord(h->ev_block, s);      // (one host thread drives a handle: one event is enough)
    return e != hipSuccess ? e : jt_event_wait(h, h->ev_block);
}
// On failure the streams are drained before returning: a pass may have queued kernels and copies into the pinned arena, and the
// next call is allowed to resize or reuse both.
inline void jt_drain(jt_ctx *h)
{
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 8; ++i) if (h->aux[i]) (void)hipStreamSynchronize(h->aux[i]);
    if (h->spec_ln.stream) { (void)hipStreamSynchronize(h->spec_ln.stream); h->spec_ln.pending = false; }
    if (h->spec_p2.stream) { (void)hipStreamSynchronize(h->spec_p2.stream); h->spec_p2.pending = false; }
    // adeclick's second solver has a stream of its own (aux[4..7] alias it only after the first pass_begin and not under region_rot)
    if (h->dk_stream) (void)hipStreamSynchronize(h->dk_stream);
    // nothing a failed pass queued may be collected by a later call: Pass 3's early measurements, the kept limiter prefix, region slots
    h->early_p3.valid = false; h->early_p3.mark_kw = false; h->lim_keep.valid = false;
    h->region_slot[0].valid = h->region_slot[1].valid = false;
}
#define JT_API_END(h) } catch (const JtError &e) { jt_drain(h); (h)->err = e.msg; return e.code; } \
    catch (const std::exception &e) { jt_drain(h); (h)->err = e.what(); return JT_E_HIP; } return JT_OK;
// input state shared by jt_upload_pcm / jt_attach_device_pcm / jt_load_audio: h->in_raw is set, derive the mono signal
void jt_set_input_common(jt_ctx *h, int64_t frames, int sr, int ch, unsigned long long mask = 0);
