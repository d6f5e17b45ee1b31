// jt_api.cpp — C ABI of libjtgpu.so (see include/jtgpu.h for the reference interfaces each entry replaces).
// Orchestrates the four device-side sweeps over a file that stays resident in HBM as contiguous PCM.
#include "jt_internal.h"
#include <chrono>
#include <algorithm>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <exception>
#include <cctype>

static void check_cancel(jt_ctx *h) { if (h->cancelled.load()) throw JtError{JT_E_CANCELLED, "cancelled"}; }
// the early Pass-2 head (defined with jt_pass2_prefetch).  phase 0: all of it behind everything queued on the main stream; 1: the
// biquad cascade only, beside whatever is queued (jt_pass1 calls it BEFORE its analysis chains); 2: the rest (anlmdn) behind the main stream
static void spec_pass2_start(jt_ctx *h, const jt_filter_params *p, int phase = 0);

#ifdef JT_AB
static void opts_from_env(JtOpts *o);
#endif

extern "C" const char *jt_version(void) { return "jtgpu 0.1 (gfx950)"; }

extern "C" int jt_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return count < 0 ? 0 : count;
}

extern "C" int jt_open_ex(int device_id, int n_streams, int flags, jt_ctx **out)
{
    if (!out) return JT_E_INVAL;
    *out = nullptr;
    // A context drives up to eight streams (main, four analysis chains, two early-start streams, adeclick's second solver; half of them
    // at low priority, which has a queue pool of its own); ROCclr multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues
    // per priority, fixed when the runtime initialises.  8 gives every stream of ONE context a queue to itself -- but that variable
    // belongs to the HOST: it must be in the environment before the process touches HIP (the Go shim sets it in init(),
    // jivetalking_amd/_lib.py before it loads this library).  The library itself neither reads nor writes the
    // environment.  More than 8 is worse, not better: with 16 (32 queues in the process) several contexts on one GPU oversubscribe the
    // hardware queue slots and the driver time-slices them (tools/inflight_probe.py).
    // n_streams: 0 / >= 8 = all of them (one file alone is fastest that way: its chains overlap each other);
    // 1 = every chain on the main stream, 2 = main + one low-priority stream for the statistics chains and the early Pass-2 head.
    // Several contexts on one GPU (a handle pool) want FEW streams each: with six contexts x eight streams on eight hardware queues
    // every event wait of one file is a barrier packet in a queue that other files' streams share, and the files serialise each other.
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return JT_E_NOGPU;
    if (device_id < 0 || device_id >= count) return JT_E_INVAL;
    if (n_streams < 0 || (flags & ~JT_OPEN_BLOCKING_SYNC)) return JT_E_INVAL;
    jt_ctx *h = new jt_ctx();
    h->device = device_id;
    h->n_streams = (n_streams == 0 || n_streams >= 8) ? 8 : (n_streams >= 2 ? 2 : 1);
    // JT_OPEN_BLOCKING_SYNC: every host wait of this handle POLLS its event (jt_event_wait: hipEventQuery, a few yields, then 50 us
    // sleeps) instead of spinning on the completion signal inside the runtime -- a pool of handles otherwise burns one host core per
    // handle for the length of the batch.  (The events are still created with hipEventBlockingSync, which this runtime ignores:
    // tools/ubench/wait_cpu.hip.)  n_streams 3 .. 7 are taken as 2, as jtgpu.h says.
    h->blocking = (flags & JT_OPEN_BLOCKING_SYNC) != 0;
    const unsigned evb = h->blocking ? hipEventBlockingSync : 0u;
#ifdef JT_AB
    opts_from_env(&h->opts);
#endif
    try {
        JT_HIP(hipSetDevice(device_id));
        JT_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->owned_streams.push_back(h->stream);
        JT_HIP(hipEventCreateWithFlags(&h->ev0, evb)); JT_HIP(hipEventCreateWithFlags(&h->ev1, evb));
        JT_HIP(hipEventCreateWithFlags(&h->ev2, evb)); JT_HIP(hipEventCreateWithFlags(&h->ev3, evb));
        JT_HIP(hipEventCreateWithFlags(&h->ev_block, hipEventDisableTiming | evb));
        JT_HIP(hipEventCreateWithFlags(&h->ev_pcm[0], hipEventDisableTiming)); JT_HIP(hipEventCreateWithFlags(&h->ev_pcm[1], hipEventDisableTiming));
        JT_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming | evb));
        int prio_least = 0, prio_greatest = 0;
        JT_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        // the analysis chains run at the low priority too (the same queue class as the early Pass-2 head: no third pool of hardware queues):
        // inside Pass 2 the main stream then carries the critical chain -- output stage, limiter prefix, Pass 3's sweep -- past the
        // statistics nobody waits for (-0.45 ms per file); where a pass ends with its analysis the chains are alone anyway
#ifdef JT_AB
        const bool aux_low = getenv("JT_AUX_NORMAL") == nullptr;      // (A/B build only: the analysis chains back in the normal class)
        const bool tp_normal = getenv("JT_TP_LOW") == nullptr;
#else
        const bool aux_low = true, tp_normal = true;
#endif
        auto own = [&](hipStream_t *s, bool low) {
            if (low) JT_HIP(hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio_least));
            else JT_HIP(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
            h->owned_streams.push_back(*s);
        };
        hipStream_t second = h->stream;
        if (h->n_streams == 2) own(&second, true);
        // lowest priority: the small band-RMS launches the host is waiting for must get through beside it
        if (h->n_streams == 8) own(&h->spec_p2.stream, true); else h->spec_p2.stream = second;
        JT_HIP(hipEventCreateWithFlags(&h->spec_p2.done, hipEventDisableTiming | evb));
        if (h->n_streams == 8) own(&h->spec_ln.stream, false); else h->spec_ln.stream = h->stream;
        JT_HIP(hipEventCreateWithFlags(&h->spec_ln.fork, hipEventDisableTiming | evb));
        // the chains of announced output regions (aux[4..7]) are small: aliases of other streams, picked by pass_begin
        for (int i = 0; i < 8; ++i) {
            // (aux[3] carries the true-peak sweep, the longest chain of every analysis and, in Pass 2, what the limiter plan waits for: it
            // stays in the normal class so that its workgroups are placed before the other statistics')
            if (i < 4) {
                if (h->n_streams == 8) own(&h->aux[i], aux_low && !(i == 3 && tp_normal));
                else h->aux[i] = i == 3 ? h->stream : second;
            } else h->aux[i] = h->aux[(i + 3) % 4];
            JT_HIP(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming | evb));
        }
        if (h->n_streams == 8) own(&h->dk_stream, false); else h->dk_stream = h->stream;
        JT_HIP(hipEventCreateWithFlags(&h->dk_ev[0], hipEventDisableTiming | evb)); JT_HIP(hipEventCreateWithFlags(&h->dk_ev[1], hipEventDisableTiming | evb));
        for (int i = 0; i < 7; ++i) JT_HIP(hipEventCreateWithFlags(&h->ev_chain[i], hipEventDisableTiming | evb));
        JT_HIP(hipEventCreateWithFlags(&h->ev_stats, hipEventDisableTiming | evb));
        JT_HIP(hipEventCreateWithFlags(&h->ev_nf, hipEventDisableTiming | evb));
    } catch (const JtError &) { jt_close(h); return JT_E_NOGPU; }
    *out = h;
    return JT_OK;
}
extern "C" int jt_open(int device_id, jt_ctx **out) { return jt_open_ex(device_id, 0, 0, out); }

extern "C" void jt_close(jt_ctx *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (hipStream_t s : h->owned_streams) (void)hipStreamSynchronize(s);
    for (hipStream_t s : h->owned_streams) (void)hipStreamDestroy(s);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->ev3) (void)hipEventDestroy(h->ev3);
    if (h->ev_block) (void)hipEventDestroy(h->ev_block);
    if (h->ev_pcm[0]) (void)hipEventDestroy(h->ev_pcm[0]);
    if (h->ev_pcm[1]) (void)hipEventDestroy(h->ev_pcm[1]);
    if (h->spec_p2.done) (void)hipEventDestroy(h->spec_p2.done);
    if (h->early_p3.pin) (void)hipHostFree(h->early_p3.pin);
    if (h->early_p3.ev[0]) { (void)hipEventDestroy(h->early_p3.ev[0]); (void)hipEventDestroy(h->early_p3.ev[1]); }
    if (h->dk_ev[0]) (void)hipEventDestroy(h->dk_ev[0]);
    if (h->dk_ev[1]) (void)hipEventDestroy(h->dk_ev[1]);
    if (h->spec_ln.fork) (void)hipEventDestroy(h->spec_ln.fork);
    if (h->spec_ln.pin) (void)hipHostFree(h->spec_ln.pin);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (int i = 0; i < 8; ++i) if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    for (int i = 0; i < 7; ++i) if (h->ev_chain[i]) (void)hipEventDestroy(h->ev_chain[i]);
    if (h->ev_nf) (void)hipEventDestroy(h->ev_nf);
    if (h->ev_stats) (void)hipEventDestroy(h->ev_stats);
    delete h;
    jt_graveyard().drain();                                    // (the frees above have waited for the device anyway)
}

extern "C" const char *jt_last_error(const jt_ctx *h) { return h ? h->err.c_str() : "null handle"; }
extern "C" void jt_cancel(jt_ctx *h) { if (h) h->cancelled.store(1); }

// ---------------------------------------------------------------- options (include/jtgpu.h: jt_set_option)
static bool opt_bool(const char *v, bool *out)
{
    if (!v || !*v || !strcmp(v, "1") || !strcmp(v, "true") || !strcmp(v, "on")) { *out = true; return true; }
    if (!strcmp(v, "0") || !strcmp(v, "false") || !strcmp(v, "off")) { *out = false; return true; }
    return false;
}
static bool opt_int(const char *v, int *out)
{
    if (!v || !*v) return false;
    char *end = nullptr; const long x = strtol(v, &end, 10);
    if (*end || x < -1000000 || x > ((long)1 << 30)) return false;
    *out = (int)x; return true;
}
int jt_opts_set(JtOpts *o, const char *key, const char *value)
{
    if (!o || !key) return JT_E_INVAL;
#define X(k) if (!strcmp(key, #k)) return opt_bool(value, &o->k) ? JT_OK : JT_E_INVAL;
    JT_OPT_BOOLS(X)
#undef X
#define X(k) if (!strcmp(key, #k)) return opt_int(value, &o->k) ? JT_OK : JT_E_INVAL;
    JT_OPT_INTS(X)
#undef X
#ifdef JT_AB
#define X(k) if (!strcmp(key, #k)) return opt_bool(value, &o->k) ? JT_OK : JT_E_INVAL;
    JT_OPT_AB_BOOLS(X)
#undef X
#define X(k) if (!strcmp(key, #k)) return opt_int(value, &o->k) ? JT_OK : JT_E_INVAL;
    JT_OPT_AB_INTS(X)
#undef X
#else
#define X(k) if (!strcmp(key, #k)) return JT_E_UNSUPPORTED;
    JT_OPT_AB_BOOLS(X) JT_OPT_AB_INTS(X)
#undef X
#endif
    return JT_E_INVAL;
}
extern "C" int jt_build_flags(void)
{
#ifdef JT_AB
    return 1;
#else
    return 0;
#endif
}
extern "C" int jt_set_option(jt_ctx *h, const char *key, const char *value)
{
    if (!key) { if (h) h->err = "set_option: null key"; return JT_E_INVAL; }
    if (!h) {
        // process-wide keys
        if (!strcmp(key, "graveyard_gb")) {
            if (!value || !*value) return JT_E_INVAL;
            char *end = nullptr; const double gb = strtod(value, &end);
            if (*end || !(gb >= 0)) return JT_E_INVAL;
            DevGraveyard::set_limit_gb(gb); return JT_OK;
        }
        if (!strcmp(key, "pool_streams")) { int v = 0; if (!opt_int(value, &v) || v < 0) return JT_E_INVAL; jt_pool_streams().store(v); return JT_OK; }
        if (!strcmp(key, "pool_numa")) { bool b = false; if (!opt_bool(value, &b)) return JT_E_INVAL; jt_pool_numa().store(b ? 1 : 0); return JT_OK; }
        if (!strcmp(key, "pool_blocking_sync")) { bool b = false; if (!opt_bool(value, &b)) return JT_E_INVAL; jt_pool_blocking().store(b ? 1 : 0); return JT_OK; }
        if (!strcmp(key, "host_timing")) { bool b = false; if (!opt_bool(value, &b)) return JT_E_INVAL; jt_host_timing().store(b ? 1 : 0); return JT_OK; }
        if (!strcmp(key, "early_temp_min_kb")) { int v = 0; if (!opt_int(value, &v) || v < 0) return JT_E_INVAL; jt_early_temp_min_kb().store(v); return JT_OK; }
        if (!strcmp(key, "poison_alloc")) { bool b = false; if (!opt_bool(value, &b)) return JT_E_INVAL; jt_poison_alloc().store(b ? 1 : 0); return JT_OK; }
        return JT_E_INVAL;
    }
    const int rc = jt_opts_set(&h->opts, key, value);
    if (rc == JT_OK && !strcmp(key, "host_timing")) jt_host_timing().store(h->opts.host_timing ? 1 : 0);
    if (rc == JT_E_UNSUPPORTED) h->err = std::string("set_option: '") + key + "' selects a superseded kernel generation or a tuning knob: JT_AB build only (libjtgpu_ab.so)";
    else if (rc != JT_OK) h->err = std::string("set_option: unknown key or bad value: ") + key + "=" + (value ? value : "");
    return rc;
}
#ifdef JT_AB
// the A/B build keeps the tools' JT_<KEY>=value switches working: imported ONCE per handle at jt_open, under a lock
static void opts_from_env(JtOpts *o)
{
    static std::mutex m; std::lock_guard<std::mutex> g(m);
    auto imp = [&](const char *k) {
        std::string e = "JT_"; for (const char *c = k; *c; ++c) e += (char)toupper((unsigned char)*c);
        if (const char *v = getenv(e.c_str())) (void)jt_opts_set(o, k, v);
    };
#define X(k) imp(#k);
    JT_OPT_BOOLS(X) JT_OPT_INTS(X) JT_OPT_AB_BOOLS(X) JT_OPT_AB_INTS(X)
#undef X
    if (const char *v = getenv("JT_GRAVEYARD_GB")) DevGraveyard::set_limit_gb(atof(v));
    if (getenv("JT_POISON_ALLOC")) jt_poison_alloc().store(1);
}
#endif

// ---------------------------------------------------------------- helpers
static void ensure_twiddle(jt_ctx *h, int N)
{
    if (h->twiddle_n == N) return;
    std::vector<float2> tw(N / 2);
    for (int k = 0; k < N / 2; ++k) { double a = -2.0 * M_PI * k / N; tw[k] = make_float2((float)std::cos(a), (float)std::sin(a)); }
    h->twiddle.ensure(N / 2);
    JT_HIP(hipMemcpyAsync(h->twiddle.p, tw.data(), sizeof(float2) * (N / 2), hipMemcpyHostToDevice, h->stream));
    JT_HIP(jt_stream_sync(h, h->stream));
    h->twiddle_n = N;
}
static void ensure_hann(jt_ctx *h, int N)
{
    if (h->hann_n == N) return;
    std::vector<float> w(N);
    for (int i = 0; i < N; ++i) w[i] = (float)(.5 * (1 - std::cos(2 * M_PI * i / (N - 1))));
    h->hann.ensure(N);
    JT_HIP(hipMemcpyAsync(h->hann.p, w.data(), sizeof(float) * N, hipMemcpyHostToDevice, h->stream));
    JT_HIP(jt_stream_sync(h, h->stream));
    h->hann_n = N;
}

void jt_spec_pass2_cancel(jt_ctx *h)
{
    h->spec_p2.armed = false;
    if (h->spec_p2.pending) { h->spec_p2.pending = false; JT_HIP(jt_stream_sync(h, h->spec_p2.stream)); }
}

void jt_set_input_common(jt_ctx *h, int64_t frames, int sr, int ch, unsigned long long mask)
{
    JT_REQUIRE(frames > 0, JT_E_INVAL, "empty input");
    JT_REQUIRE(sr >= 8000 && sr <= 384000, JT_E_INVAL, "unsupported sample rate");
    JT_REQUIRE(ch >= 1 && ch <= 8, JT_E_INVAL, "unsupported channel count");
    // aformat=channel_layouts=mono (filters.go:607-615) goes through libswresample's layout-specific rematrix: restated for every
    // layout of FL FR FC LFE BL BR FLC FRC BC SL SR (k_lane.hip: jt_downmix_row); anything else is refused, not averaged with
    // the wrong weights
    DownmixRow row;
    JT_REQUIRE(jt_downmix_row(ch, mask, 0, &row), JT_E_UNSUPPORTED, "this channel layout is not down-mixed on the device (channels beyond SIDE_RIGHT, or a mask that does not match the channel count)");
    h->n = frames; h->sr = sr; h->channels = ch; h->ch_mask = mask;
    h->dec_frame_samples = 4096; h->dec_frames = (frames + 4095) / 4096; h->dec_frame_lens.clear();      // (jt_load_audio overrides: the file's own)
    h->m_p2 = h->m_p4 = 0;
    if (!h->hold_cancel) h->cancelled.store(0);           // a new job: jt_cancel() is sticky from here until the next input / jt_reset_cancel
    if (ch == 1 && row.k == 1 && row.cf[0] == 1.0f) h->in_mono = h->in_raw;      // (a lone FRONT_CENTER: the identity row)
    else {
        h->mono.ensure((size_t)frames);
        launch_downmix(h->in_raw, h->mono.p, frames, ch, 0, row, h->stream);
        h->in_mono = h->mono.p;
    }
}

extern "C" int jt_input_frame_layout(jt_ctx *h, int *frame_samples, int *variable, int64_t *n_frames, int32_t *lens, int64_t cap)
{
    JT_API_BEGIN_KEEP(h)
    JT_REQUIRE(h->n > 0, JT_E_STATE, "frame layout: no input");
    const bool var = !h->dec_frame_lens.empty();
    if (frame_samples) *frame_samples = h->dec_frame_samples;
    if (variable) *variable = var ? 1 : 0;
    if (n_frames) *n_frames = var ? (int64_t)h->dec_frame_lens.size() : (h->n + h->dec_frame_samples - 1) / h->dec_frame_samples;
    if (lens && var) std::copy(h->dec_frame_lens.begin(), h->dec_frame_lens.begin() + std::min<int64_t>(cap, (int64_t)h->dec_frame_lens.size()), lens);
    JT_API_END(h)
}

extern "C" int jt_upload_pcm_layout(jt_ctx *h, const float *pcm, int64_t frames, int sr, int ch, uint64_t channel_mask)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(pcm && frames > 0 && ch >= 1 && ch <= 8, JT_E_INVAL, "bad pcm arguments");
    h->in_owned.ensure((size_t)frames * ch);
    JT_HIP(hipMemcpyAsync(h->in_owned.p, pcm, sizeof(float) * (size_t)frames * ch, hipMemcpyHostToDevice, h->stream));
    h->in_raw = h->in_owned.p;
    h->src_fmt = 0;
    jt_set_input_common(h, frames, sr, ch, channel_mask);
    JT_HIP(jt_stream_sync(h, h->stream));
    JT_API_END(h)
}
extern "C" int jt_upload_pcm(jt_ctx *h, const float *pcm, int64_t frames, int sr, int ch)
{
    return jt_upload_pcm_layout(h, pcm, frames, sr, ch, 0);
}

extern "C" int jt_attach_device_pcm(jt_ctx *h, const void *dev_ptr, int64_t frames, int sr, int ch)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(dev_ptr && frames > 0, JT_E_INVAL, "bad device pcm arguments");
    h->in_raw = static_cast<const float *>(dev_ptr);
    h->src_fmt = 0;
    jt_set_input_common(h, frames, sr, ch);
    JT_HIP(jt_stream_sync(h, h->stream));
    JT_API_END(h)
}

// an early Pass-3 measurement still reading the Pass-2 output must finish before that output is replaced
static void spec_loudnorm_cancel(jt_ctx *h)
{
    if (h->spec_ln.pending) { JT_HIP(jt_stream_sync(h, h->spec_ln.stream)); h->spec_ln.pending = false; }
}

extern "C" int jt_upload_s16(jt_ctx *h, const int16_t *pcm, int64_t frames, int sr)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(pcm && frames > 0, JT_E_INVAL, "bad s16 arguments");
    spec_loudnorm_cancel(h);
    h->s16_p2.ensure((size_t)frames);
    JT_HIP(hipMemcpyAsync(h->s16_p2.p, pcm, sizeof(int16_t) * (size_t)frames, hipMemcpyHostToDevice, h->stream));
    JT_HIP(jt_stream_sync(h, h->stream));
    h->m_p2 = frames; h->out_rate = sr; h->m_p4 = 0;
    JT_API_END(h)
}

// ---------------------------------------------------------------- cached resampler plans
static SwrDev &get_swr(jt_ctx *h, int in_rate, int out_rate)
{
    for (auto &e : h->swr) if (e.in_rate == in_rate && e.out_rate == out_rate) return e;
    SwrDev &e = h->swr[h->swr_next]; h->swr_next = (h->swr_next + 1) % 4;
    JT_HIP(jt_stream_sync(h, h->stream));             // the slot's old banks may still be read by queued kernels
    jt_swr_plan(&e.pl, in_rate, out_rate);
    {   // norms of the tap rows for the true peak's bounds (k_resample.hip, k_tp_bounds)
        const int L = e.pl.filter_length;
        double a1 = 0, t0 = 0, m1 = 1e300;
        for (int ph = 0; ph < e.pl.phase_count; ++ph) {
            double a = 0, t = 0;
            for (int i = 0; i < L; ++i) { const double v = e.pl.bank[(size_t)ph * L + i]; a += std::fabs(v); t += v; }
            a1 = std::max(a1, a); t0 = std::max(t0, std::fabs(t));
        }
        for (int c = 0; c < L; ++c) {                         // one reference tap for every row: the one with the smallest worst case
            double worst = 0;
            for (int ph = 0; ph < e.pl.phase_count; ++ph) {
                double m = 0;
                for (int i = 0; i < L; ++i) m += std::fabs(e.pl.bank[(size_t)ph * L + i]) * std::abs(i - c);
                worst = std::max(worst, m);
            }
            m1 = std::min(m1, worst);
        }
        e.tp_norms[0] = a1; e.tp_norms[1] = t0; e.tp_norms[2] = m1;
    }
    std::vector<float> bf(e.pl.bank.size());
    for (size_t i = 0; i < bf.size(); ++i) bf[i] = (float)e.pl.bank[i];
    e.bank_d.ensure(e.pl.bank.size()); e.bank_f.ensure(bf.size()); e.bank_dT.ensure(e.pl.bank.size());
    JT_HIP(hipMemcpyAsync(e.bank_d.p, e.pl.bank.data(), sizeof(double) * e.pl.bank.size(), hipMemcpyHostToDevice, h->stream));
    {
        const int L = e.pl.filter_length, P = e.pl.phase_count;
        std::vector<double> bt(e.pl.bank.size());
        for (int ph = 0; ph < P; ++ph) for (int i = 0; i < L; ++i) bt[(size_t)i * P + ph] = e.pl.bank[(size_t)ph * L + i];
        JT_HIP(hipMemcpy(e.bank_dT.p, bt.data(), sizeof(double) * bt.size(), hipMemcpyHostToDevice));
    }
    JT_HIP(hipMemcpyAsync(e.bank_f.p, bf.data(), sizeof(float) * bf.size(), hipMemcpyHostToDevice, h->stream));
    // s16 sources are scaled by 2^-15 before the taps (swr's s16 -> flt conversion); a power of two commutes with the rounding of
    // every fused multiply-add, so the scale may live in the taps instead -- as long as no scaled tap leaves the normal range
    std::vector<float> bs(bf.size());
    bool exact = true;
    for (size_t i = 0; i < bf.size(); ++i) { bs[i] = std::ldexp(bf[i], -15); if (std::ldexp(bs[i], 15) != bf[i] || (bs[i] != 0 && !std::isnormal(bs[i]))) exact = false; }
    e.bank_fs.release();
    if (exact) { e.bank_fs.ensure(bs.size()); JT_HIP(hipMemcpyAsync(e.bank_fs.p, bs.data(), sizeof(float) * bs.size(), hipMemcpyHostToDevice, h->stream)); }
    JT_HIP(jt_stream_sync(h, h->stream));
    e.in_rate = in_rate; e.out_rate = out_rate;
    return e;
}

// start of a pass: size the pinned arena and the K-weighting scratch for everything the pass will stage (nothing in flight)
// (+ `extra_jobs` further analyses of at most `extra_samples` each: the announced output regions)
static void pass_begin(jt_ctx *h, int64_t max_samples, int analyses, int64_t extra_samples = 0, int extra_jobs = 0)
{
    ensure_twiddle(h, 2048); ensure_hann(h, 2048);
    {   // the chains of the announced output regions (analysis lanes 4..7): all four on the stream adeclick's second solver uses
        // (idle outside adeclick), so that they run beside the pass's full-length analysis.  Behind the full chains on their own four
        // streams (JT_REGION_ROT=r: region chain i behind full chain (i + r) % 4) they end 0.4 ms later at best (r = 3, the longest
        // region chain behind the shortest full chain) -- tools/ab_env.py JT_REGION_ROT=-,0,3
        const bool rv = h->opts.region_rot >= 0;
        const int rot = rv ? h->opts.region_rot & 3 : 3;
        for (int i = 0; i < 4; ++i) h->aux[4 + i] = rv ? h->aux[(i + rot) % 4] : h->dk_stream;
    }
    const size_t na = (size_t)std::max(1, analyses), ne = (size_t)std::max(0, extra_jobs);
    h->pin.begin(jt_arena_bytes_for(max_samples) * na + jt_arena_bytes_for(extra_samples) * ne + (4u << 20));
    h->kw_begin((size_t)(max_samples / 512 + 1024) * 12 * na + ((size_t)(extra_samples / 512 + 1024) * 8 + (size_t)extra_samples / 128 + 16384) * ne);   // + the regions' own scratch
    h->as_begin(((size_t)(2u << 20) + (size_t)max_samples / 6) * na + ((size_t)(2u << 20) + (size_t)extra_samples / 6) * ne);
    // per-100 ms true-peak maxima of every analysis of the pass (the full-length one and the announced regions'): atomicMax targets, so
    // they start from zero -- ONE fill for the pass, queued on the main stream ahead of every fork, instead of one per analysis
    {
        const size_t slab = (size_t)(max_samples / 400 + 16) * na + (size_t)(extra_samples / 400 + 16) * ne;
        h->d_scr1.ensure(slab);
        JT_HIP(hipMemsetAsync(h->d_scr1.p, 0, sizeof(double) * slab, h->stream));
        h->tp_off = 0; h->tp_cap = slab; h->tp_scr_off = 0;
    }
    h->spec_hops.ensure((size_t)(max_samples / 512 + 16));
    h->ehist.ensure(8192);
}

// ---------------------------------------------------------------- analysis of a mono f32 signal on device
// (astats + aspectralstats + ebur128 of the Pass-1 / Pass-2 / Pass-4 / region graphs).  enqueue: every kernel and every
// device->host copy of the analysis, no synchronisation; finish: the host arithmetic (gating, LRA, merges) after the sync.
struct AnalysisHost {
    jt_astats astats; R128Series r128; std::vector<double> tp_cum, sp_cum; double tp_final, sp_final;
    const jt_spectral *hops = nullptr; int64_t nblocks = 0, nhops = 0, nout = 0; int blk = 0;
    jt_spectral spec_sum; int64_t spec_cnt = -1;      // the frames' spectral records summed in frame order (analysis_finish), -1: not summed
};
struct AnalysisJob {
    AstatsJob as; KwJob kw; const double *btp = nullptr; const jt_spectral *hops = nullptr;
    const int *tp_kept = nullptr; int64_t tp_units = 0, tp_seeds = 0;      // branch-and-bound true peak: units kept by round 2 (pinned), units, seeds
    int64_t n = 0, nfull = 0, nhops = 0, nout = 0; int blk = 0, sr = 0; bool dualmono = false, want_astats = true, want_r128 = true, want_spec = true;
    bool astats_levels_only = false;
};

// sets = 2: also the chains of the announced output regions (aux[4..7])
static void analysis_join(jt_ctx *h, int sets = 1)
{
    // later main-stream work (and the pass's final synchronisation of the main stream) orders after every chain
    for (int i = 0; i < 4 * sets; ++i) { JT_HIP(hipEventRecord(h->ev_join[i], h->aux[i])); JT_HIP(hipStreamWaitEvent(h->stream, h->ev_join[i], 0)); }
}
// where one analysis runs: the four streams aux[first .. first+3] and its block-true-peak / hop / histogram scratch (null = the
// context's shared buffers, which serialise jobs on one stream set)
// `from`: the stream whose queued work produces the signal (null = the main stream)
struct AnalysisLanes { int first = 0; bool direct = false; hipStream_t from = nullptr; };      // direct: small results go straight into the pinned arena
static void fork_aux(jt_ctx *h, int first, int last, hipStream_t from = nullptr)
{
    JT_HIP(hipEventRecord(h->ev_fork, from ? from : h->stream));
    for (int i = first; i <= last; ++i) JT_HIP(hipStreamWaitEvent(h->aux[i], h->ev_fork, 0));
}

// join = false: the caller queues more independent main-stream work first and calls analysis_join() itself
static void analysis_enqueue(jt_ctx *h, const float *x, int64_t n, int sr, bool dualmono, int sel_blk, AnalysisJob *J, bool join = true,
                             const AnalysisLanes *ln = nullptr)
{
    // pass_begin sizes the K-weighting and true-peak slabs for 100 ms blocks of at least 800 samples (the rates jt_upload_pcm accepts)
    JT_REQUIRE(sr >= 8000 && sr <= 384000, JT_E_UNSUPPORTED, "analysis: sample rates outside 8-384 kHz are not measured on the device");
    J->n = n; J->sr = sr; J->dualmono = dualmono;
    // fork: x is ready once everything queued on the main stream so far has run
    const int f = ln ? ln->first : 0;
    hipStream_t a0 = h->aux[f], a1 = h->aux[f + 1], a2 = h->aux[f + 2], a3 = h->aux[f + 3];
    fork_aux(h, f, f + 3, ln ? ln->from : nullptr);
    // K-weighting goes FIRST on its stream (ahead of the noise-floor chain it shares it with): with the true peak it is what the limiter
    // plan of Pass 2 waits for (jt_pass3_plan_hook); the chain's total is the same either way
    const bool r128_first = h->early_p3.mark_kw && J->want_r128;
    hipStream_t a_nf = a1;          // the stream of astats' noise-floor chain
    if (J->want_r128) {
        // a planner waits for this job (Pass 2): it runs in the normal priority class, on the stream adeclick's second solver uses in
        // Pass 4 (idle here; the announced regions' chains queue up behind it), instead of among the low-priority statistics
        hipStream_t akw = a1;
        if (r128_first && !f && h->dk_stream && !h->opts.no_r128_first) { akw = h->dk_stream; JT_HIP(hipStreamWaitEvent(akw, h->ev_fork, 0)); }
        jt_kweight_enqueue_f32(h, x, n, sr, sr / 10, &J->kw, akw);
        if (!f) JT_HIP(hipEventRecord(h->ev_chain[5], akw));
        if (h->early_p3.mark_kw) { JT_HIP(hipEventRecord(h->early_p3.ev[0], akw)); h->early_p3.mark_kw = false; }
        if (akw != a1) {
            JT_HIP(hipStreamWaitEvent(a1, h->ev_chain[5], 0));          // (the pass's join of aux[1] then covers the job)
            // the noise-floor chain stays behind the job on that stream: at low priority beside Pass 3's sweep it ended last of all (3.3 ms
            // for 0.5 ms of work) and the pass with it
            if (!h->opts.nf_low) a_nf = akw;
        }
    }
    if (J->want_r128) {
        const int blk = sr / 10; const int64_t nfull = n / blk;
        J->blk = blk; J->nfull = nfull;
        SwrDev &sw = get_swr(h, sr, 192000);
        JT_REQUIRE(h->tp_off + (size_t)nfull + 2 <= h->tp_cap, JT_E_HIP, "true-peak scratch exhausted");
        double *d_tp = h->d_scr1.p + h->tp_off; h->tp_off += ((size_t)nfull + 2 + 7) & ~(size_t)7;      // (zeroed by pass_begin)
        // long signals: only the units that can move the running maximum are evaluated (k_resample.hip, branch and bound); the unit
        // bounds and lists of the pass's long analysis live in tp_scr (a second long analysis in the same pass runs exhaustively)
        bool pruned = false;
        if (!h->opts.tp_unpruned && n >= (int64_t)h->opts.tp_prune_min) {
            const size_t need = jt_tp_prune_scratch_bytes(n, sw.pl.phase_count, sw.pl.filter_length, sw.pl.step, blk);
            if (need && h->tp_scr_off == 0) {
                h->tp_scr.ensure(need);
                const int *kept = nullptr;
                pruned = launch_true_peak_f32_pruned(x, n, sw.bank_d.p, sw.pl.phase_count, sw.pl.filter_length, sw.pl.center, sw.pl.step, blk, d_tp,
                                                     nfull + 1, sw.out_len(n), sw.tp_norms, h->tp_scr.p, need, a3, &kept, &J->tp_units, &J->tp_seeds);
                if (pruned) {
                    h->tp_scr_off = need;
                    int *hk = h->pin.take<int>(2);
                    JT_HIP(hipMemcpyAsync(hk, kept, sizeof(int), hipMemcpyDeviceToHost, a3));
                    J->tp_kept = hk;
                }
            }
        }
        if (!pruned)
            launch_true_peak_f32(x, n, sw.bank_d.p, sw.pl.phase_count, sw.pl.filter_length, sw.pl.center, sw.pl.step, blk, d_tp, nfull + 1,
                                 sw.out_len(n), a3, &h->opts);
        double *btp = h->pin.take<double>((size_t)nfull + 2);
        JT_HIP(hipMemcpyAsync(btp, d_tp, sizeof(double) * (nfull + 2), hipMemcpyDeviceToHost, a3));
        J->btp = btp;
        if (r128_first && !h->opts.no_r128_first) {
            // a planner waits for the true peak: the statistics nobody waits for (astats chains, spectral) start behind the upsampler
            // instead of sharing the CUs with it (3.6 ms beside them, 1-1.5 ms without), and then run beside Pass 3's prefix chain
            JT_HIP(hipEventRecord(h->early_p3.ev[1], a3));
            JT_HIP(hipStreamWaitEvent(a0, h->early_p3.ev[1], 0)); JT_HIP(hipStreamWaitEvent(a1, h->early_p3.ev[1], 0));
            JT_HIP(hipStreamWaitEvent(a2, h->early_p3.ev[1], 0));
        }
    }
    // aspectralstats first on its stream: it is the longest kernel of an analysis (an in-wave FFT per 100 ms frame) and used to start
    // behind the two sweeps of the exponential-average chain; those now follow it
    if (J->want_spec) {
        const int win = 2048, hop = win / 2;
        const int64_t nhops = (n + hop - 1) / hop;
        int64_t nframes = 0;
        if (sel_blk > 0) nframes = n / sel_blk + ((n % sel_blk) ? 1 : 0);
        const int64_t nout = sel_blk > 0 ? nframes : nhops;
        J->nhops = nhops; J->nout = nout;
        jt_spectral *hops = h->pin.take<jt_spectral>((size_t)std::max<int64_t>(nout, 1));
        if ((ln && ln->direct) || (sel_blk > 0 && !h->opts.no_spec_direct)) {
            // selected frames (one 104-byte record per 100 ms: 3.7 MB for an hour; an announced region: a few dozen records): written
            // by the kernel straight into the pinned arena -- the copy behind the kernel was 0.4 ms at the end of the pass's longest chain
            launch_aspectralstats(x, n, sr, win, h->twiddle.p, h->hann.p, hops, nhops, sel_blk, nframes, a2);
        } else {
            h->spec_hops.ensure((size_t)std::max<int64_t>(nout, 1));
            launch_aspectralstats(x, n, sr, win, h->twiddle.p, h->hann.p, h->spec_hops.p, nhops, sel_blk, nframes, a2);
            if (nout > 0) JT_HIP(hipMemcpyAsync(hops, h->spec_hops.p, sizeof(jt_spectral) * nout, hipMemcpyDeviceToHost, a2));
        }
        J->hops = hops;
        if (!f) JT_HIP(hipEventRecord(h->ev_chain[6], a2));       // (the exponential-average chain of astats follows on this stream)
    }
    if (J->want_astats) {
        // astats' exponential-average chain (k_as_zs -> k_as_sigma) rides behind the REDUCE chain (round 6): behind aspectralstats, the
        // longest kernel of an analysis, it made that stream the last to end in Pass 1 and in the final analysis (2.2 + 0.2 + 0.2 ms against
        // 1.3 + 0.3 for the reduce chain); option as_avg_behind_spec = the old place.  The same launches, another queue
        jt_astats_enqueue(h, x, n, sr, &J->as, a0, a_nf, h->opts.as_avg_behind_spec ? a2 : a0, nullptr, J->astats_levels_only);
        if (a_nf != a1) { JT_HIP(hipEventRecord(h->ev_nf, a_nf)); JT_HIP(hipStreamWaitEvent(a1, h->ev_nf, 0)); }      // (joined and waited for through aux[1])
    }
    if (!f) JT_HIP(hipEventRecord(h->ev_chain[4], a2));
    if (!f) for (int i = 0; i < 4; ++i) JT_HIP(hipEventRecord(h->ev_chain[i], h->aux[i]));
    if (join) analysis_join(h, f ? 2 : 1);
}

static void spectral_add(jt_spectral *a, const jt_spectral &b)
{
    double *pa = &a->mean; const double *pb = &b.mean;
    for (int i = 0; i < 13; ++i) pa[i] += pb[i];
}

// staged: the job is the pass's full-length analysis and may still be running; each part of the host arithmetic waits for the chain
// that feeds it (ev_chain) and runs while the later chains -- true peak and aspectralstats end last -- are still on the GPU
// astats_later: the caller collects astats itself, after everything else (its noise-floor chain is usually the last to end)
static void analysis_finish(jt_ctx *h, const AnalysisJob &J, AnalysisHost *A, bool staged, bool astats_later = false)
{
    auto wait = [&](int i) { if (staged) JT_HIP(jt_event_wait(h, h->ev_chain[i])); };
    const double *bsum = J.kw.hc, *bpk = J.kw.hc ? J.kw.hc + (size_t)J.nfull + 1 : nullptr;      // (in place: KwJob::hc = (nfull + 1) sums, then as many peaks)
    if (J.want_r128) {
        wait(5);
        A->blk = J.blk; A->nblocks = J.nfull;
        jt_r128_finish(bsum, J.nfull, J.blk, J.sr, J.dualmono, &A->r128);
    }
    if (J.want_astats && !astats_later) { wait(0); wait(1); wait(4); jt_astats_finish(&J.as, &A->astats); }
    // The mean of the per-frame spectral records (33-36 k records of 13 doubles for an hour, summed in frame order: 0.15-0.2 ms) needs the
    // aspectralstats chain only.  In a pass's tail that chain is not the last to end, so the sum is taken as soon as it has, before the
    // true-peak chain is waited for if that one is still running: at the end of Pass 4 nothing but the astats chain is left behind it
    // (it used to be the last thing the host did before returning the file's result).
    auto spec_sum = [&] {
        wait(6);
        A->hops = J.hops; A->nhops = J.nhops; A->nout = J.nout;
        if (!J.want_r128) return;
        const int64_t nframes = J.nfull + ((J.n % J.blk) != 0 ? 1 : 0);
        std::memset(&A->spec_sum, 0, sizeof A->spec_sum);
        for (int64_t k = 0; k < nframes; ++k) spectral_add(&A->spec_sum, J.hops[(size_t)k]);
        A->spec_cnt = nframes;
    };
    bool spec_done = false;
    if (J.want_spec && staged && J.want_r128 && hipEventQuery(h->ev_chain[3]) != hipSuccess) { (void)hipGetLastError(); spec_sum(); spec_done = true; }
    if (J.want_r128) {
        wait(3);
        const int64_t nfull = J.nfull;
        A->tp_cum.assign(nfull, 0.0); A->sp_cum.assign(nfull, 0.0);
        double tp = 0, sp = 0;
        for (int64_t k = 0; k < nfull; ++k) {
            sp = std::max(sp, bpk[(size_t)k]); tp = std::max(tp, J.btp[(size_t)k]);
            A->sp_cum[k] = sp; A->tp_cum[k] = tp;
        }
        // trailing partial frame: peaks still update (f_ebur128.c runs the per-sample loop over it)
        sp = std::max(sp, bpk[(size_t)nfull]); tp = std::max(tp, J.btp[(size_t)nfull]);
        A->sp_final = sp; A->tp_final = tp;
        if (J.tp_kept) { h->timers.tp_units_total = J.tp_units; h->timers.tp_units_evaluated = J.tp_seeds + *J.tp_kept; }
    }
    if (J.want_spec && !spec_done) spec_sum();
}
static void spectral_scale(jt_spectral *a, double s)
{
    double *pa = &a->mean;
    for (int i = 0; i < 13; ++i) pa[i] *= s;
}

// Assemble what the Go OnFrame callbacks would have seen: one record per 100 ms ebur128 output frame
// (+ one trailing partial frame without r128 keys when n % blk != 0).  The aspectralstats props that
// survive ebur128's re-framing are those of the hop containing the frame's first sample.
static void assemble_analysis(const AnalysisHost &A, int64_t n, bool dualmono, jt_analysis *out, jt_frame_meta *meta, int64_t cap_meta)
{
    (void)dualmono;
    const int64_t nfull = A.nblocks;
    const bool partial = (n % A.blk) != 0;
    const int64_t nframes = nfull + (partial ? 1 : 0);
    jt_spectral mean; std::memset(&mean, 0, sizeof(mean)); int64_t cnt = 0;
    const bool summed = A.spec_cnt == nframes;            // (analysis_finish took the sum, same records in the same order)
    if (summed) { mean = A.spec_sum; cnt = nframes; }
    for (int64_t k = 0; k < nframes && !(summed && !meta); ++k) {
        const jt_spectral &sp = A.hops[(size_t)k];      // selected-frames mode: one record per output frame
        if (!summed) { spectral_add(&mean, sp); cnt++; }
        if (meta && k < cap_meta) {
            jt_frame_meta &m = meta[k];
            m.spectral = sp;
            if (k < nfull) { m.momentary = A.r128.M[k]; m.shortterm = A.r128.S[k]; m.true_peak = A.tp_cum[k]; m.sample_peak = A.sp_cum[k]; }
            else { m.momentary = NAN; m.shortterm = NAN; m.true_peak = NAN; m.sample_peak = NAN; }   // keys absent
        }
    }
    if (cnt) spectral_scale(&mean, 1.0 / cnt);
    out->spectral_mean = mean;
    out->n_frames_meta = nframes;
    out->r128.integrated = A.r128.integrated; out->r128.lra = A.r128.lra;
    out->r128.lra_low = A.r128.lra_low; out->r128.lra_high = A.r128.lra_high;
    out->r128.momentary = nfull ? A.r128.M[nfull - 1] : NAN;
    out->r128.shortterm = nfull ? A.r128.S[nfull - 1] : NAN;
    // "latest wins": the last frame that carried r128 keys is the last full block
    out->r128.true_peak = nfull ? A.tp_cum[nfull - 1] : 0.0;
    out->r128.sample_peak = nfull ? A.sp_cum[nfull - 1] : 0.0;
    out->r128.target_threshold = A.r128.rel_threshold;
}

static void analysis_complete(jt_ctx *h, const AnalysisJob &J, jt_analysis *out, jt_frame_meta *meta, int64_t cap_meta, bool staged = false)
{
    // one per host thread, keeping its vectors' storage (five series of an hour's 36 000 blocks: 1.4 MB of fresh, page-faulting memory
    // per analysis otherwise); everything else starts as in a fresh object
    static thread_local AnalysisHost A_tl;
    AnalysisHost &A = A_tl;
    A.hops = nullptr; A.nblocks = A.nhops = A.nout = 0; A.blk = 0; A.spec_cnt = -1; A.tp_cum.clear(); A.sp_cum.clear(); A.r128.M.clear(); A.r128.S.clear();
    const bool timing = h->opts.host_timing;
    std::chrono::steady_clock::time_point t0; if (timing) t0 = std::chrono::steady_clock::now();
    if (staged && h->opts.no_staged_finish) {          // (the round-2 order: everything after the last chain)
        for (int i = 0; i < 4; ++i) JT_HIP(jt_event_wait(h, h->ev_chain[i]));
        staged = false;
    }
    // astats last: its noise-floor chain (low priority, behind the K-weighting job) is usually the last chain of a pass to end, and the
    // per-frame assembly (36 000 records for an hour) needs none of it
    analysis_finish(h, J, &A, staged, true);
    std::chrono::steady_clock::time_point t1; if (timing) t1 = std::chrono::steady_clock::now();
    assemble_analysis(A, J.n, J.dualmono, out, meta, cap_meta);
    if (J.want_astats) {
        if (staged) { JT_HIP(jt_event_wait(h, h->ev_chain[0])); JT_HIP(jt_event_wait(h, h->ev_chain[1])); JT_HIP(jt_event_wait(h, h->ev_chain[4])); }
        jt_astats_finish(&J.as, &A.astats);
    }
    out->astats = A.astats;
    if (timing) fprintf(stderr, "analysis_complete (n = %lld): finish %.3f ms, assemble %.3f ms\n", (long long)J.n,
                        std::chrono::duration<double, std::milli>(t1 - t0).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
}

// ---------------------------------------------------------------- Pass 1
extern "C" int jt_pass1(jt_ctx *h, int frame_samples, jt_analysis *out, double *frame_sumsq, double *frame_peak,
                        int64_t cap_frames, jt_frame_meta *meta, int64_t cap_meta)
{
    JT_API_BEGIN_KEEP(h)                      // (reads the input only: a running Pass-2 head is left alone)
    JT_REQUIRE(h->n > 0 && h->in_raw, JT_E_STATE, "pass1: no input uploaded");
    JT_REQUIRE(out && frame_samples >= 0, JT_E_INVAL, "pass1: bad arguments");
    check_cancel(h);
    JT_HIP(hipEventRecord(h->ev0, h->stream));
    std::memset(out, 0, sizeof(*out));
    // frame_samples == 0: the input's own cadence (jt_input_frame_layout), frame by frame when the stream's frames differ in length
    const bool var_frames = frame_samples == 0 && !h->dec_frame_lens.empty();
    if (frame_samples == 0) frame_samples = h->dec_frame_samples;
    const int64_t nfr = var_frames ? (int64_t)h->dec_frame_lens.size() : (h->n + frame_samples - 1) / frame_samples;
    out->n_input_frames = nfr;
    pass_begin(h, h->n, 1);
    const double *fst = nullptr;
    AnalysisJob J;
    // Round 6: the announced Pass-2 head's BIQUAD cascade (a thin, latency-bound sweep that needs nothing but the input) goes in FIRST, on its
    // low-priority stream beside the analysis chains; anlmdn -- 150 k workgroups that would take every slot -- still waits for the analysis
    // (below).  Schedule only: option no_early_biquad keeps the cascade behind the analysis too.
    bool head_split = false;
    if (h->spec_p2.armed && !h->opts.no_early_biquad) {
        try { spec_pass2_start(h, &h->spec_p2.armed_p, 1); head_split = true; }
        catch (const JtError &e) { if (e.code != JT_E_INVAL) throw; h->spec_p2.armed = false; }      // (dropped as below: jt_pass2 raises the error)
    }
    // the analysis chains fork first: the per-decoder-frame sums then run beside them on the main stream instead of ahead of them
    // (the fork waits for everything queued on the main stream: 0.3 ms of every Pass 1 with the sums in front)
    analysis_enqueue(h, h->in_mono, h->n, h->sr, true, h->sr / 10, &J, false);
    if (frame_sumsq && frame_peak) {
        h->d_scr3.ensure((size_t)nfr * 2);
        if (var_frames) launch_frame_stats_var(h->in_raw, h->channels, h->d_frame_off.p, h->d_scr3.p, h->d_scr3.p + nfr, nfr, h->stream);
        else launch_frame_stats(h->in_raw, h->n * h->channels, frame_samples * h->channels, h->d_scr3.p, h->d_scr3.p + nfr, nfr, h->stream);
        double *tmp = h->pin.take<double>((size_t)nfr * 2);
        JT_HIP(hipMemcpyAsync(tmp, h->d_scr3.p, sizeof(double) * nfr * 2, hipMemcpyDeviceToHost, h->stream));
        fst = tmp;
    }
    analysis_join(h);
    // an announced Pass-2 head goes in behind the analysis: its 150 k workgroups would otherwise take every slot first
    if (h->spec_p2.armed) {
        h->spec_p2.armed = false;
        // best effort: a head that cannot be built (a source below 41 kHz puts the fixed 20.5 kHz low-pass past Nyquist: fill_biquads
        // throws EINVAL) is dropped here, Pass 1 completes as it does in the reference, and jt_pass2 raises the error
        try { spec_pass2_start(h, &h->spec_p2.armed_p, head_split ? 2 : 0); } catch (const JtError &e) { if (e.code != JT_E_INVAL) throw; }
    }
    analysis_complete(h, J, out, meta, cap_meta, true);       // (chain by chain, while the later chains still run)
    JT_HIP(jt_stream_sync(h, h->stream));
    check_cancel(h);
    if (fst) {
        const int64_t c = std::min(nfr, cap_frames);
        std::copy(fst, fst + c, frame_sumsq);
        std::copy(fst + nfr, fst + nfr + c, frame_peak);
    }
    out->n_input_frames = nfr;
    JT_HIP(hipEventRecord(h->ev1, h->stream));
    JT_HIP(jt_event_wait(h, h->ev1));
    float ms = 0; JT_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1)); h->timers.pass1_ms = ms;
    JT_API_END(h)
}


// atrim=start:duration (libavfilter/trim.c): option strings are "%f" seconds -> AV_OPT_TYPE_DURATION microseconds ->
// av_rescale_q to the 1/rate time base (nearest); sample i is kept iff start_pts <= i < start_pts + duration_tb.
static void trim_range(double start_s, double dur_s, int rate, int64_t total, int64_t *s0, int64_t *len)
{
    const long long s_us = std::llround(start_s * 1e6), d_us = std::llround(dur_s * 1e6);
    auto rescale = [&](long long us) { return (int64_t)(((__int128)us * rate + 500000) / 1000000); };
    int64_t a = rescale(s_us), d = rescale(d_us);
    a = std::max<int64_t>(0, std::min(a, total));
    int64_t b = std::max(a, std::min(a + d, total));
    *s0 = a; *len = b - a;
}

// ---- output regions (MeasureOutputRegions, analyser_output.go:276-317): the analysis of up to two atrim'd ranges of a stage's s16
// output.  Shared by jt_region_measure_pair (on demand) and the tails of Pass 2 / Pass 4 (regions announced by jt_region_prefetch).
struct RegionJobs { AnalysisJob J[2]; int64_t s0[2] = {0, 0}, len[2] = {0, 0}; bool want[2] = {false, false}; const int16_t *src[2] = {nullptr, nullptr}; };

// false when a wanted region is empty after atrim's rounding (the on-demand call reports that as an error)
static bool regions_resolve(const double start_s[2], const double dur_s[2], int rate, int64_t m, RegionJobs *R)
{
    for (int r = 0; r < 2; ++r) {
        R->want[r] = dur_s[r] > 0 && start_s[r] >= 0;
        R->s0[r] = R->len[r] = 0;
        if (R->want[r]) { trim_range(start_s[r], dur_s[r], rate, m, &R->s0[r], &R->len[r]); if (R->len[r] <= 0) return false; }
    }
    return true;
}
// queues both analyses (after everything already queued on the main stream); the caller joins.  own_lanes: on the second stream
// set with scratch of their own, so that they run beside a pass's full-length analysis instead of behind it
// src: the stage's s16 output; or null with R->src[r] = the first sample of region r (Pass 2 resamples its regions on `from`, a
// stream of the second set, while the main stream still produces the full output)
static void regions_enqueue(jt_ctx *h, const int16_t *src, int rate, RegionJobs *R, bool own_lanes, hipStream_t from = nullptr)
{
    h->region_f.ensure((size_t)(R->len[0] + R->len[1] + 1));
    {   // both regions' samples in one launch
        const int16_t *s0 = R->want[0] ? (src ? src + R->s0[0] : R->src[0]) : nullptr, *s1 = R->want[1] ? (src ? src + R->s0[1] : R->src[1]) : nullptr;
        launch_s16_to_f32_pair(s0, R->want[0] ? R->len[0] : 0, s1, R->want[1] ? R->len[1] : 0, h->region_f.p, h->region_f.p + R->len[0], from ? from : h->stream);
    }
    for (int r = 0; r < 2; ++r) {
        if (!R->want[r]) continue;
        AnalysisLanes ln; ln.from = from;
        if (own_lanes) { ln.first = 4; ln.direct = true; }
        R->J[r].astats_levels_only = !h->opts.region_full_astats;      // (regions_finish reads the levels and the crest factor only)
        analysis_enqueue(h, h->region_f.p + (r ? R->len[0] : 0), R->len[r], rate, false, rate / 10, &R->J[r], false, own_lanes ? &ln : nullptr);
    }
}
static void regions_finish(jt_ctx *h, int rate, const RegionJobs &R, jt_region_sample out[2])
{
    for (int r = 0; r < 2; ++r) {
        std::memset(&out[r], 0, sizeof(out[r]));
        if (!R.want[r]) continue;
        jt_analysis a; std::memset(&a, 0, sizeof(a));
        std::vector<jt_frame_meta> meta((size_t)(R.len[r] / (rate / 10) + 2));
        analysis_complete(h, R.J[r], &a, meta.data(), (int64_t)meta.size());
        out[r].rms_level = a.astats.rms_level; out[r].peak_level = a.astats.peak_level; out[r].crest_factor = a.astats.crest_factor;
        out[r].spectral = a.spectral_mean;
        out[r].momentary = a.r128.momentary; out[r].shortterm = a.r128.shortterm;
        out[r].true_peak = a.r128.true_peak; out[r].sample_peak = a.r128.sample_peak;
        out[r].frames = a.n_frames_meta;
    }
}
// upper bound of an announced region's length at `rate` (pass_begin sizes the arenas before the output length is known)
static int64_t region_slot_samples(const jt_ctx::RegionSlot &sl, int rate)
{
    double d = 0;
    for (int r = 0; r < 2; ++r) if (sl.dur_s[r] > 0) d = std::max(d, sl.dur_s[r]);
    return (int64_t)std::ceil(d * rate) + 4;
}

// ---------------------------------------------------------------- band RMS
extern "C" int jt_band_rms(jt_ctx *h, double start_s, double dur_s, const double *lo_hz, const double *hi_hz,
                           int n_bands, double *out_db, int *ok)
{
    JT_API_BEGIN_KEEP(h)
    JT_REQUIRE(h->n > 0 && h->in_mono, JT_E_STATE, "band_rms: no input uploaded");
    JT_REQUIRE(lo_hz && hi_hz && out_db && n_bands > 0 && dur_s > 0 && start_s >= 0, JT_E_INVAL, "band_rms: bad arguments");
    // atrim=start:duration selects samples with start <= t < start+duration (pts-based)
    int64_t s0, len; trim_range(start_s, dur_s, h->sr, h->n, &s0, &len);
    for (int b = 0; b < n_bands; ++b) { out_db[b] = NAN; if (ok) ok[b] = 0; }
    if (len <= 0) return JT_OK;
    JT_REQUIRE(n_bands <= 16, JT_E_INVAL, "band_rms: at most 16 bands per call");
    check_cancel(h);
    double hp[16][5], lp[16][5]; int slot[16]; int nb = 0;
    for (int b = 0; b < n_bands; ++b) {
        // a corner at/above Nyquist has no valid biquad: the reference treats those bands as unmeasurable (non-finite)
        if (hi_hz[b] >= h->sr * 0.5 || lo_hz[b] >= h->sr * 0.5) { out_db[b] = NAN; if (ok) ok[b] = 1; continue; }
        double bh[3], ah[3], bl[3], al[3];
        jt_biquad_design(0, lo_hz[b], 0.707, h->sr, bh, ah, 0);
        jt_biquad_design(1, hi_hz[b], 0.707, h->sr, bl, al, 0);
        for (int k = 0; k < 3; ++k) { hp[nb][k] = bh[k]; lp[nb][k] = bl[k]; }
        hp[nb][3] = -ah[1]; hp[nb][4] = -ah[2]; lp[nb][3] = -al[1]; lp[nb][4] = -al[2];
        slot[nb++] = b;
    }
    if (nb > 0) {
        // The band graph holds no float-only filter, so libavfilter keeps the decoder's integer width for it:
        // s16p / s32p biquads, and for a stereo source a down-mix of its own (integer matrix, not Pass 1's 1/sqrt2).
        const int mode = h->src_fmt;
        const float *src = h->in_mono + s0;
        if (mode != 0 && (h->channels >= 2 || h->in_mono != h->in_raw)) {
            DownmixRow row;
            JT_REQUIRE(jt_downmix_row(h->channels, h->ch_mask, mode, &row), JT_E_UNSUPPORTED, "band_rms: channel layout not covered");
            h->band_mono.ensure((size_t)len);
            launch_downmix(h->in_raw + s0 * h->channels, h->band_mono.p, len, h->channels, mode, row, h->stream);
            src = h->band_mono.p;
        }
        h->d_scr1.ensure(16);
        JT_HIP(hipMemsetAsync(h->d_scr1.p, 0, sizeof(double) * 16, h->stream));
        launch_band_rms(src, len, nb, hp, lp, mode, h->d_scr1.p, h->stream);
        double sums[16];
        JT_HIP(hipMemcpyAsync(sums, h->d_scr1.p, sizeof(double) * 16, hipMemcpyDeviceToHost, h->stream));
        JT_HIP(jt_stream_sync(h, h->stream));
        for (int k = 0; k < nb; ++k) {
            out_db[slot[k]] = 20 * std::log10(std::sqrt(sums[k] / (double)len));
            if (ok) ok[slot[k]] = 1;   // astats Overall.RMS_level key present (may be -inf on digital silence)
        }
    }
    JT_API_END(h)
}

extern "C" int jt_set_source_format(jt_ctx *h, int bits_per_sample, int is_float)
{
    if (!h) return JT_E_INVAL;
    if (is_float) { if (bits_per_sample != 32 && bits_per_sample != 64) { h->err = "source format: float PCM is 32 or 64 bits"; return JT_E_INVAL; } h->src_fmt = 0; return JT_OK; }
    if (bits_per_sample < 4 || bits_per_sample > 32) { h->err = "source format: integer PCM is 4..32 bits"; return JT_E_INVAL; }
    h->src_fmt = bits_per_sample <= 16 ? 1 : 2;         // libavcodec hands 8/16-bit PCM and <= 16-bit FLAC out as (u8/)s16, wider as s32
    return JT_OK;
}

extern "C" void jt_reset_cancel(jt_ctx *h) { if (h) h->cancelled.store(0); }
extern "C" void jt_begin_job(jt_ctx *h) { if (h) { h->cancelled.store(0); h->hold_cancel = true; } }
extern "C" void jt_end_job(jt_ctx *h) { if (h) h->hold_cancel = false; }

// ---------------------------------------------------------------- Pass 2
static void run_anlmdn(jt_ctx *h, const float *in, float *out, int64_t n, int sr, double strength, double patch_s, double research_s, double smooth,
                       hipStream_t st = nullptr)
{
    if (!st) st = h->stream;
    const int K = (int)std::llrint((double)std::llrint(patch_s * 1e6) * sr / 1e6);
    const int S = (int)std::llrint((double)std::llrint(research_s * 1e6) * sr / 1e6);
    JT_REQUIRE(K >= 1 && S >= 1, JT_E_INVAL, "anlmdn: patch/research too small");
    const float a = (float)strength, m = (float)smooth;
    const float lut_scale = 1.f / m * (float)(1 << 20);
    const float sw = (65536.f / (4 * K + 2)) / std::sqrt(a);
    JT_HIP(hipEventRecord(h->ev2, st));
    launch_anlmdn(in, out, n, K, S, sw, m, lut_scale, st, h->opts);
    JT_HIP(hipEventRecord(h->ev3, st));
}

// The filter's tables (window, band maps, variances: ~20 KB) are planned on the host and uploaded on the main stream.  jt_pass2 does
// that BEFORE it makes the main stream wait for the early head: queued behind the wait, the upload -- an SDMA copy with its own
// signalling -- sat between anlmdn's last workgroup and afftdn's first (0.19 ms of every step, profiles/r05_timeline_one_step.txt).
struct AfftdnPrep { AfftdnPlanHost pl; AfftdnDev d; };
static void afftdn_prepare(jt_ctx *h, int sr, double nr, double nf, const double *bn, AfftdnPrep *P)
{
    AfftdnPlanHost &pl = P->pl; jt_afftdn_plan(&pl, sr, nr, nf, bn);
    ensure_twiddle(h, pl.L);
    const size_t nb = pl.nbands, bins = pl.bins;
    const size_t ndbl = pl.W + nb + nb + nb * nb + bins + bins + bins;
    h->af_tab.ensure(ndbl + (bins + 1) / 2);
    // staged in the pinned arena: the copy stays valid until the pass's single synchronisation (tables | bin -> band map: one upload)
    double *tab = h->pin.take<double>(ndbl + (bins + 1) / 2); double *w = tab;
    w = std::copy(pl.window.begin(), pl.window.end(), w);
    w = std::copy(pl.alpha.begin(), pl.alpha.end(), w);
    w = std::copy(pl.beta.begin(), pl.beta.end(), w);
    w = std::copy(pl.spread.begin(), pl.spread.end(), w);
    w = std::copy(pl.abs_var.begin(), pl.abs_var.end(), w);
    w = std::copy(pl.min_abs_var.begin(), pl.min_abs_var.end(), w);
    w = std::copy(pl.rel_var.begin(), pl.rel_var.end(), w);
    int *b2b = reinterpret_cast<int *>(tab + ndbl);
    std::copy(pl.bin2band.begin(), pl.bin2band.end(), b2b);
    JT_HIP(hipMemcpyAsync(h->af_tab.p, tab, sizeof(double) * (ndbl + (bins + 1) / 2), hipMemcpyHostToDevice, h->stream));
    AfftdnDev d;
    d.A = pl.A; d.W = pl.W; d.L = pl.L; d.bins = pl.bins; d.nbands = pl.nbands; d.max_gain = pl.max_gain;
    d.bin2band = reinterpret_cast<const int *>(h->af_tab.p + ndbl);
    d.window = h->af_tab.p; d.alpha = d.window + pl.W; d.beta = d.alpha + nb; d.spread = d.beta + nb;
    d.abs_var = d.spread + nb * nb; d.min_abs_var = d.abs_var + bins; d.rel_var = d.min_abs_var + bins; d.twiddle = h->twiddle.p;
    d.floor = pl.floor;
    {
        const int H = pl.L / 2, HH = H / 2;
        int span = 1;
        for (int r = 0; r * 64 < HH; ++r) {
            span = std::max(span, pl.bin2band[(size_t)std::min(64 * r + 63, HH - 1)] - pl.bin2band[(size_t)(64 * r)] + 1);
            span = std::max(span, pl.bin2band[(size_t)(H - 64 * r)] - pl.bin2band[(size_t)std::max(H - 64 * r - 63, HH + 1)] + 1);
        }
        d.seg_span = span;
    }
    P->d = d;
}
static void run_afftdn(jt_ctx *h, const float *in, float *out, int64_t n, int sr, double nr, double nf, const double *bn, bool track = false,
                       const AfftdnPrep *prep = nullptr)
{
    AfftdnPrep local;
    if (!prep) { afftdn_prepare(h, sr, nr, nf, bn, &local); prep = &local; }
    const AfftdnPlanHost &pl = prep->pl; AfftdnDev d = prep->d;
    if (!track) { launch_afftdn(in, out, n, d, 0, 96, h->stream, h->opts); return; }     // chunk length chosen from the frame count; 96 warm-up frames
    // tn=1 (af_afftdn.c track_noise; what the reference emits when Noise.Floor == 0, adaptive.go:147-151).  The floor is a state
    // that survives arbitrarily long stretches of speech (only spectrally flat frames move it), so a warm-up halo cannot restore
    // it -- but the VOTE of a frame (is it flat, and which floor does it ask for) depends on that frame's magnitudes alone.  First
    // sweep: every frame's vote, in parallel; then the first-order recurrence nf <- 0.9 nf + 0.1 vote over the frames on the host
    // (N / 600 values); second sweep: the usual chunked kernel with the per-frame variances that recurrence implies.
    const int64_t nframes = jt_afftdn_nframes(n, pl.A, pl.W);
    h->af_track.ensure((size_t)(2 * nframes + 4));
    d.track_out = h->af_track.p; d.mvseq = h->af_track.p + nframes + 1;
    launch_afftdn(in, out, n, d, 0, 0, h->stream, h->opts, 2);
    std::vector<double> vote((size_t)nframes), mv((size_t)nframes + 1);
    JT_HIP(hipMemcpyAsync(vote.data(), d.track_out, sizeof(double) * nframes, hipMemcpyDeviceToHost, h->stream));
    JT_HIP(jt_stream_sync(h, h->stream));
    check_cancel(h);
    const double Cc = M_LN10 * 0.1;
    double nfl = pl.noise_floor;
    mv[0] = pl.floor * std::exp((100.0 + nfl) * Cc);
    for (int64_t t = 0; t < nframes; ++t) {
        if (!std::isnan(vote[(size_t)t])) { nfl = 0.1 * vote[(size_t)t] + nfl * 0.9; mv[(size_t)t + 1] = pl.floor * std::exp((100.0 + nfl) * Cc); }
        else mv[(size_t)t + 1] = mv[(size_t)t];
    }
    h->af_last_floor = nfl;
    JT_HIP(hipMemcpyAsync(h->af_track.p + nframes + 1, mv.data(), sizeof(double) * (nframes + 1), hipMemcpyHostToDevice, h->stream));
    launch_afftdn(in, out, n, d, 0, 96, h->stream, h->opts, 1);
    JT_HIP(jt_stream_sync(h, h->stream));                 // (mv is a host vector: the upload must have left it)
}

static void fill_biquads(const jt_filter_params *p, int sr, BiquadF32 st[2], int *nst)
{
    *nst = 0;
    // af_biquads.c config_filter(): w0 = 2*pi*f/rate; "if (w0 > M_PI || w0 <= 0.) return AVERROR(EINVAL)" -- the graph does not configure.
    // The reference's fixed 20.5 kHz band limit therefore fails Pass 2 for sources below 41 kHz (adaptive_bandlimit_lowpass.go:17-20
    // states the assumption and applies no guard); the same error is reported here instead of running a meaningless filter.
    if (p->hp_enabled) JT_REQUIRE(p->hp_freq > 0 && 2.0 * p->hp_freq <= (double)sr, JT_E_INVAL, "highpass: frequency outside (0, Nyquist] (af_biquads: EINVAL)");
    if (p->lp_enabled) JT_REQUIRE(p->lp_freq > 0 && 2.0 * p->lp_freq <= (double)sr, JT_E_INVAL, "lowpass: frequency outside (0, Nyquist] (af_biquads: EINVAL)");
    if (p->hp_enabled) {
        double b[3], a[3]; jt_biquad_design(0, p->hp_freq, p->hp_q, sr, b, a, 1);
        st[(*nst)++] = BiquadF32{(float)b[0], (float)b[1], (float)b[2], -(float)a[1], -(float)a[2]};
    }
    if (p->lp_enabled) {
        double b[3], a[3]; jt_biquad_design(1, p->lp_freq, p->lp_q, sr, b, a, 1);
        st[(*nst)++] = BiquadF32{(float)b[0], (float)b[1], (float)b[2], -(float)a[1], -(float)a[2]};
    }
}

static void run_resample_s16(jt_ctx *h, const float *x, int64_t n, int in_rate, int out_rate, DevBuf<int16_t> &dst, int64_t *m_out)
{
    if (in_rate == out_rate) {
        // aformat alone (44.1 kHz input): flt -> dbl -> s16 as swresample converts; the f64 ping-pong buffer of the dynamics stage is
        // free again at this point of the stream (a buffer allocated and freed here cost two device synchronisations per file)
        dst.ensure((size_t)n);
        h->f64_a.ensure((size_t)n + 16);
        launch_f32_to_f64(x, h->f64_a.p, n, h->stream);
        launch_f64_to_s16(h->f64_a.p, dst.p, nullptr, n, 0, h->stream);
        *m_out = n;
        return;
    }
    SwrDev &sw = get_swr(h, in_rate, out_rate);
    const int64_t m = sw.out_len(n);                      // ceil(n*out/in)
    dst.ensure((size_t)m);
    launch_resample_to_s16(x, n, sw.bank_d.p, sw.pl.phase_count, sw.pl.filter_length, sw.pl.center, sw.pl.step, dst.p, m, h->stream, h->opts);
    *m_out = m;
}

extern "C" int jt_pass2_prefetch(jt_ctx *h, const jt_filter_params *p)
{
    JT_API_BEGIN(h)                                           // (retires an earlier head first)
    JT_REQUIRE(h->n > 0 && h->in_mono, JT_E_STATE, "pass2_prefetch: no input uploaded");
    JT_REQUIRE(p, JT_E_INVAL, "pass2_prefetch: bad arguments");
    if (h->opts.no_pass2_prefetch) return JT_OK;              // (A/B switch: Pass 2 then runs every stage itself)
    spec_pass2_start(h, p);
    JT_API_END(h)
}
extern "C" int jt_pass2_prefetch_after_pass1(jt_ctx *h, const jt_filter_params *p)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(h->n > 0 && h->in_mono, JT_E_STATE, "pass2_prefetch: no input uploaded");
    JT_REQUIRE(p, JT_E_INVAL, "pass2_prefetch: bad arguments");
    if (h->opts.no_pass2_prefetch) return JT_OK;
    h->spec_p2.armed_p = *p; h->spec_p2.armed = true;
    JT_API_END(h)
}
static void spec_pass2_start(jt_ctx *h, const jt_filter_params *p, int phase)
{
    jt_ctx::SpecPass2 &sp = h->spec_p2;
    const int64_t n = h->n; const int sr = h->sr;
    if (phase != 2) {
        fill_biquads(p, sr, sp.st, &sp.nst);
        sp.nlm = p->nlm_enabled != 0;
        sp.nlm_p[0] = p->nlm_strength; sp.nlm_p[1] = p->nlm_patch_s; sp.nlm_p[2] = p->nlm_research_s; sp.nlm_p[3] = p->nlm_smooth;
        sp.stages = 0;
    }
    if (sp.nst == 0 && !sp.nlm) return;                       // nothing to start
    h->work_a.ensure((size_t)n + 16); h->work_b.ensure((size_t)n + 16);
    // the same buffer walk as jt_pass2: in_mono -> work_a -> work_b
    const float *cur = h->in_mono; float *nxt = h->work_a.p; float *oth = h->work_b.p;
    JT_HIP(hipEventRecord(h->ev_fork, h->stream));
    JT_HIP(hipStreamWaitEvent(sp.stream, h->ev_fork, 0));     // everything queued so far (the input upload / down-mix; phase 2: Pass 1's analysis) first
    if (sp.nst > 0) {
        if (phase != 2) { launch_biquad_f32(cur, nxt, n, sp.nst, sp.st, sp.stream); sp.stages++; }
        cur = nxt; std::swap(nxt, oth);
    }
    if (phase == 1) return;                                   // (anlmdn follows in phase 2, on the same stream)
    if (sp.nlm) { run_anlmdn(h, cur, nxt, n, sr, sp.nlm_p[0], sp.nlm_p[1], sp.nlm_p[2], sp.nlm_p[3], sp.stream); cur = nxt; std::swap(nxt, oth); sp.stages++; }
    JT_HIP(hipEventRecord(sp.done, sp.stream));
    sp.pending = true;
}

// af_loudnorm's flush frame re-enters filter_frame(), whose first statement feeds the frame to the INPUT meter: the last 2.9 s of a
// stream of 3 s or more are metered twice (libavfilter/af_loudnorm.c: flush_frame -> filter_frame -> ff_ebur128_add_frames_double).
// The measurement streams therefore carry a copy of their last 556 800 samples behind the end.
static int64_t loudnorm_meter_len(int64_t m_total) { return m_total >= 576000 ? m_total + 556800 : m_total; }
template <typename T> static void loudnorm_append_flush(T *stream, int64_t m_total, hipStream_t st)
{
    if (m_total >= 576000) JT_HIP(hipMemcpyAsync(stream + m_total, stream + (m_total - 556800), (size_t)556800 * sizeof(T), hipMemcpyDeviceToDevice, st));
}

// loudnorm's input meter over a signal at `rate` (the Pass-2 output as s16, or the limiter prefix's f64 output): swr -> 192 kHz,
// K-weighting, 100 ms block energies and sample peaks, the flush frame included.  One sweep that never stores the 192 kHz stream
// (k_p3_fused) when the rate pair allows it (44.1 kHz: the reference's output rate); the stand-alone pair -- upsampler into
// stream_f / stream_d, k_kw1 over it -- otherwise and under the option p3_unfused (what the fused sweep is tested against).
// ext: scratch that outlives the pass arenas (sized by p3_scratch_sizes); null = the pass arenas.
static bool p3_fused_ok(jt_ctx *h, const SwrDev &sw, int64_t m_total)
{
    const int blk = (192000 + 5) / 10;
    const int64_t flush = loudnorm_meter_len(m_total) - m_total;
    return !h->opts.p3_unfused && jt_p3_fused_supported(sw.pl.phase_count, sw.pl.filter_length, sw.pl.step, blk, flush);
}
static void p3_scratch_sizes(jt_ctx *h, int64_t n, int rate, size_t *dev_d, size_t *pin_d)
{
    SwrDev &sw = get_swr(h, rate, 192000);
    const int64_t m_total = sw.out_len(n);
    jt_kweight_scratch_sizes(loudnorm_meter_len(m_total), (192000 + 5) / 10, dev_d, pin_d, p3_fused_ok(h, sw, m_total) ? sw.pl.phase_count : 0);
}
static void p3_measure_enqueue(jt_ctx *h, const int16_t *s16, const double *f64, int64_t n, int rate, KwJob *kw, int64_t *nfull, int *blk_out,
                               hipStream_t st, const KwScratch *ext)
{
    SwrDev &sw = get_swr(h, rate, 192000);
    const int64_t m_total = sw.out_len(n);
    const int blk = (192000 + 5) / 10;
    const int64_t m_meter = loudnorm_meter_len(m_total);
    *blk_out = blk; *nfull = m_meter / blk;
    if (p3_fused_ok(h, sw, m_total)) {
        const int P = sw.pl.phase_count;
        if (h->p3_mpow_P != P) {
            // the fold's matrices F^(JW w) depend on the 192 kHz K-weighting coefficients and the period only: once per handle
            BiquadF64 pre, rlb; jt_kweight_design(192000, &pre, &rlb);
            KwCoef k{pre.b0, pre.b1, pre.b2, pre.a1, pre.a2, rlb.b0, rlb.b1, rlb.b2, rlb.a1, rlb.a2};
            double tab[16 * 16]; const int cnt = jt_p3_fold_powers(k, P, tab);
            h->p3_mpow.retire(); h->p3_mpow.ensure((size_t)cnt);
            JT_HIP(hipMemcpy(h->p3_mpow.p, tab, sizeof(double) * (size_t)cnt, hipMemcpyHostToDevice));
            h->p3_mpow_P = P;
        }
        const double *mpow = h->p3_mpow.p;
        jt_kweight_enqueue_sweep(h, m_meter, 192000, blk, P, kw, st, ext, [&](const KwSweep &W) {
            if (s16) launch_p3_fused_s16(s16, n, sw.bank_f.p, sw.bank_fs.p, P, sw.pl.center, sw.pl.step, m_total, m_meter - m_total, W, mpow, st);
            else launch_p3_fused_f64(f64, n, sw.bank_d.p, P, sw.pl.center, sw.pl.step, m_total, m_meter - m_total, W, mpow, st);
        });
        return;
    }
    if (s16) {
        h->stream_f.ensure((size_t)m_meter);
        launch_resample_stream_s16_f32(s16, n, sw.bank_f.p, sw.bank_fs.p, sw.pl.phase_count, sw.pl.filter_length, sw.pl.center, sw.pl.step, m_total, h->stream_f.p, st);
        loudnorm_append_flush(h->stream_f.p, m_total, st);
        jt_kweight_enqueue_f32(h, h->stream_f.p, m_meter, 192000, blk, kw, st, ext);
    } else {
        h->stream_d.ensure((size_t)m_meter);
        launch_resample_stream_f64(f64, n, sw.bank_d.p, sw.pl.phase_count, sw.pl.filter_length, sw.pl.center, sw.pl.step, m_total, h->stream_d.p, st, h->opts);
        loudnorm_append_flush(h->stream_d.p, m_total, st);
        jt_kweight_enqueue_f64(h, h->stream_d.p, m_meter, 192000, blk, kw, st, ext);
    }
}

// Pass 3 for a plan without limiter prefix (loudnorm's first-pass measurement of the s16 output: swr -> 192 kHz, K-weighting,
// 100 ms block energies), queued behind the output stage on a stream that Pass 2 does not wait for.  Scratch and result buffers
// of its own: the pass arenas are recycled by the next pass_begin.
static void spec_loudnorm_enqueue(jt_ctx *h, const int16_t *s16, int64_t n, int rate)
{
    if (rate == 192000 || n <= 0 || h->opts.no_early_pass3) return;
    auto &S = h->spec_ln;
    size_t dev_d = 0, pin_d = 0; p3_scratch_sizes(h, n, rate, &dev_d, &pin_d);
    S.dev.ensure(dev_d);
    if (pin_d > S.pin_cap) {
        if (S.pin) jt_graveyard().put(S.pin, sizeof(double) * S.pin_cap, 1);
        S.pin = nullptr; S.pin_cap = 0;
        JT_HIP(DevGraveyard::host_malloc((void **)&S.pin, sizeof(double) * pin_d));
        S.pin_cap = pin_d;
    }
    JT_HIP(hipEventRecord(S.fork, h->stream));
    JT_HIP(hipStreamWaitEvent(S.stream, S.fork, 0));
    const KwScratch ext{S.dev.p, S.pin};
    p3_measure_enqueue(h, s16, nullptr, n, rate, &S.kw, &S.nfull, &S.blk, S.stream, &ext);
    S.pending = true;
}

extern "C" int jt_pass3_plan_hook(jt_ctx *h, jt_plan_fn fn, void *user)
{
    if (!h) return JT_E_INVAL;
    h->early_p3.fn = fn; h->early_p3.user = user; h->early_p3.armed = fn != nullptr;
    return JT_OK;
}

// s16 != nullptr: `in` is WRITTEN first (the s16 -> dbl conversion with the volume stage, fused into the limiter's first sweep)
struct LimS16 { const int16_t *s16 = nullptr; double vol = 1.0; int vol_in_float = 0; };
static void run_limiter(jt_ctx *h, double *in, double *out, int64_t n, int sr, double limit, double attack_ms, double release_ms, double in_gain,
                        const LimS16 &src = LimS16{}, const LimOut16 *o16 = nullptr);
static bool limiter_can_emit16(jt_ctx *h, int sr, double attack_ms);

// Pass 3 for a plan WITH the limiter prefix, queued on the main stream inside Pass 2 (behind the output stage, before the join with the
// analysis chains): volume -> alimiter on the s16 output, swr (double) -> 192 kHz, K-weighting + block energies.  The same launches as
// pass3_core's prefix branch; scratch and results of its own (the pass arenas hold Pass 2's analysis).
static void early_pass3_enqueue(jt_ctx *h, const int16_t *s16, int64_t m, int rate, const jt_limiter_plan &lim)
{
    auto &E = h->early_p3;
    size_t dev_d = 0, pin_d = 0; p3_scratch_sizes(h, m, rate, &dev_d, &pin_d);
    E.dev.ensure(dev_d);
    if (pin_d > E.pin_cap) {
        if (E.pin) jt_graveyard().put(E.pin, sizeof(double) * E.pin_cap, 1);
        E.pin = nullptr; E.pin_cap = 0;
        JT_HIP(DevGraveyard::host_malloc((void **)&E.pin, sizeof(double) * pin_d));
        E.pin_cap = pin_d;
    }
    h->f64_a.ensure((size_t)m); h->f64_b.ensure((size_t)m);
    const bool pre = lim.pre_gain_db > 0;
    const double g = pre ? std::pow(10.0, lim.pre_gain_db / 20.0) : 1.0;
    run_limiter(h, h->f64_a.p, h->f64_b.p, m, rate, lim.limit, 5.0, 100.0, 1.0, LimS16{s16, g, pre ? 1 : 0});
    h->lim_keep = {true, s16, m, rate, lim.pre_gain_db, lim.limit};
    const KwScratch ext{E.dev.p, E.pin};
    p3_measure_enqueue(h, nullptr, h->f64_b.p, m, rate, &E.kw, &E.nfull, &E.blk, h->stream, &ext);
    E.plan = lim; E.valid = true;
}

extern "C" int jt_pass2(jt_ctx *h, const jt_filter_params *p, jt_analysis *out)
{
    JT_API_BEGIN_KEEP(h)
    JT_REQUIRE(h->n > 0 && h->in_mono, JT_E_STATE, "pass2: no input uploaded");
    JT_REQUIRE(p && out, JT_E_INVAL, "pass2: bad arguments");
    check_cancel(h);
    std::memset(out, 0, sizeof(*out));
    JT_HIP(hipEventRecord(h->ev0, h->stream));
    const int64_t n = h->n; const int sr = h->sr;
    const int out_rate = p->out_rate > 0 ? p->out_rate : 44100;
    spec_loudnorm_cancel(h);
    h->lim_keep.valid = false;
    h->early_p3.valid = false;
    const bool plan_hook = h->early_p3.armed && !h->opts.no_early_plan; h->early_p3.armed = false;
    jt_ctx::RegionSlot &slot = h->region_slot[0];
    slot.valid = false; h->region_slot[1].valid = false;            // both stage outputs are about to be replaced
    const bool announced = slot.armed; slot.armed = false;
    pass_begin(h, n, 1, announced ? region_slot_samples(slot, out_rate) : 0, announced ? 2 : 0);
    h->work_a.ensure((size_t)n + 16); h->work_b.ensure((size_t)n + 16);      // (+ slack: 16-byte group reads at the end of the signal)
    const float *cur = h->in_mono; float *nxt = h->work_a.p; float *oth = h->work_b.p;
    auto advance = [&]() { cur = nxt; std::swap(nxt, oth); };
    BiquadF32 st[2]; int nst = 0; fill_biquads(p, sr, st, &nst);
    bool nlm_timed = false;
    jt_ctx::SpecPass2 &sp = h->spec_p2;
    const double nlm_p[4] = {p->nlm_strength, p->nlm_patch_s, p->nlm_research_s, p->nlm_smooth};
    const bool head_ready = sp.pending && sp.nst == nst && (nst == 0 || !std::memcmp(sp.st, st, sizeof(BiquadF32) * nst)) &&
                            sp.nlm == (p->nlm_enabled != 0) && (!sp.nlm || !std::memcmp(sp.nlm_p, nlm_p, sizeof(nlm_p)));
    AfftdnPrep af_prep; bool af_ready = false;
    const double af_nf = p->fft_nf < 0 ? p->fft_nf : -50.0;
    if (p->fft_enabled) { afftdn_prepare(h, sr, p->fft_nr, af_nf, p->fft_custom ? p->fft_band_noise : nullptr, &af_prep); af_ready = true; }
    if (head_ready) {
        // jt_pass2_prefetch ran exactly these stages on this input: continue from its result
        sp.pending = false;
        // The head runs on a queue of its own; a queue that waits for another queue's event takes ~0.1 ms to notice it on this part
        // (anlmdn's end to afftdn's start: 100-140 us in every timeline), a host thread that spins on the event ~10 us, and this thread has
        // nothing to queue that could start before the head is done.  Option p2_device_join: the wait inside the queue, as before.
        if (h->opts.p2_device_join || sp.stream == h->stream) JT_HIP(hipStreamWaitEvent(h->stream, sp.done, 0));      // (a one-stream handle: nothing to hand over)
        else { JT_HIP(jt_event_wait(h, sp.done)); check_cancel(h); }
        for (int k = 0; k < sp.stages; ++k) advance();
        nlm_timed = sp.nlm;
    } else {
        jt_spec_pass2_cancel(h);
        if (nst > 0) { launch_biquad_f32(cur, nxt, n, nst, st, h->stream); advance(); }
        check_cancel(h);
        if (p->nlm_enabled) { run_anlmdn(h, cur, nxt, n, sr, p->nlm_strength, p->nlm_patch_s, p->nlm_research_s, p->nlm_smooth); advance(); nlm_timed = true; }
    }
    check_cancel(h);
    if (p->fft_enabled) {
        run_afftdn(h, cur, nxt, n, sr, p->fft_nr, af_nf, p->fft_custom ? p->fft_band_noise : nullptr, p->fft_track_noise != 0, af_ready ? &af_prep : nullptr); advance();
    }
    check_cancel(h);
    DynParams d; jt_dyn_design(p, sr, &d);
    if (d.gate_on || d.comp_on || d.deess_on) {
        h->f64_a.ensure((size_t)n + 16); if (d.deess_on) h->f64_b.ensure((size_t)n + 16);
        h->d_scr3.ensure((size_t)(n / 256 + 4));
        launch_dynamics(cur, nxt, h->f64_a.p, h->f64_b.p, h->d_scr3.p, n, d, h->stream, h->opts, cur != h->in_mono); advance();
    }
    check_cancel(h);
    AnalysisJob J;
    const bool plan_early = plan_hook && out_rate != 192000;
    if (plan_early) {
        // what the limiter plan waits for: the K-weighting job (first on aux[1]: analysis_enqueue marks its end) and the true peak
        // (aux[3], marked below before the region chains queue up behind it on the same stream)
        auto &E = h->early_p3;
        if (!E.ev[0]) { const unsigned evb = hipEventDisableTiming | (h->blocking ? hipEventBlockingSync : 0u); JT_HIP(hipEventCreateWithFlags(&E.ev[0], evb)); JT_HIP(hipEventCreateWithFlags(&E.ev[1], evb)); }
        E.mark_kw = true;
    }
    analysis_enqueue(h, cur, n, sr, true, sr / 10, &J, false);
    if (plan_early) JT_HIP(hipEventRecord(h->early_p3.ev[1], h->aux[3]));
    // announced output regions: their samples are resampled separately (the same tap sums as the full output's) on the second
    // stream set, so that their analysis runs beside the output stage instead of after it
    RegionJobs RJ;
    bool regions = false;
    if (announced && sr != out_rate) {
        SwrDev &sw = get_swr(h, sr, out_rate);
        const int64_t m = sw.out_len(n);
        regions = regions_resolve(slot.start_s, slot.dur_s, out_rate, m, &RJ);
        if (regions) {
            int64_t cap[2] = {0, 0};
            for (int r = 0; r < 2; ++r) if (RJ.want[r]) cap[r] = jt_resample_range_cap(n, sw.pl.phase_count, sw.pl.filter_length, sw.pl.step, m, RJ.len[r]);
            h->region_s16.ensure((size_t)(cap[0] + cap[1] + 1));
            fork_aux(h, 4, 4);
            if (plan_early && !h->opts.no_r128_first) {
                // nobody waits for the regions: their resampling (1.4 ms of the output stage's arithmetic) starts once the two chains the
                // limiter plan waits for have ended, instead of beside them
                JT_HIP(hipStreamWaitEvent(h->aux[4], h->early_p3.ev[0], 0)); JT_HIP(hipStreamWaitEvent(h->aux[4], h->early_p3.ev[1], 0));
            }
            for (int r = 0; r < 2; ++r) {
                if (!RJ.want[r]) continue;
                int16_t *dst = h->region_s16.p + (r ? cap[0] : 0);
                RJ.src[r] = dst + launch_resample_range_to_s16(cur, n, sw.bank_d.p, sw.pl.phase_count, sw.pl.filter_length, sw.pl.center, sw.pl.step,
                                                               m, RJ.s0[r], RJ.len[r], dst, cap[r], h->aux[4]);
            }
            regions_enqueue(h, nullptr, out_rate, &RJ, true, h->aux[4]);
        }
    }
    run_resample_s16(h, cur, n, sr, out_rate, h->s16_p2, &h->m_p2);     // main stream, concurrent with the analysis chains
    if (announced && sr == out_rate) {
        regions = regions_resolve(slot.start_s, slot.dur_s, out_rate, h->m_p2, &RJ);
        if (regions) regions_enqueue(h, h->s16_p2.p, out_rate, &RJ, true);
    }
    // Pass 3's measurement for the plan without a prefix, on its own stream (not joined): a guess when nobody can tell us the plan;
    // with a planner it is queued below, once the plan says so (a file that needs the prefix then does not pay 3 ms of GPU work for it)
    if (!plan_early) spec_loudnorm_enqueue(h, h->s16_p2.p, h->m_p2, out_rate);
    if (plan_early) {
        // The limiter plan is a function of this pass's integrated loudness and true peak (K-weighting chain on aux[1], true peak on
        // aux[3]): wait for those two chains only, ask the planner, and queue the prefix measurement while astats / aspectralstats / the
        // regions are still running.  The values are the ones analysis_complete() will report below (same arithmetic, same inputs).
        auto &E = h->early_p3;
        // (the gating of the loudness -- 36 000 blocks of an hour, a histogram and the LRA percentiles -- runs while the true-peak sweep,
        // which ends later than the K-weighting job, is still on the GPU)
        const bool timing = h->opts.host_timing;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto tq = now();
        JT_HIP(jt_event_wait(h, E.ev[0]));
        const auto t0 = now();
        check_cancel(h);
        // (the block sums are read where the kernel left them -- pinned host memory -- and the series keeps its storage from file to file:
        //  copies and fresh 0.3 MB vectors were a third of this stretch, which the main stream's next kernel waits for)
        static thread_local R128Series r;
        jt_r128_finish(J.kw.hc, J.nfull, J.blk, J.sr, J.dualmono, &r, true);
        const auto t1 = now();
        JT_HIP(jt_event_wait(h, E.ev[1]));
        const auto t2 = now();
        check_cancel(h);
        double tp = 0; for (int64_t k = 0; k < J.nfull; ++k) tp = std::max(tp, J.btp[(size_t)k]);
        jt_limiter_plan plan; std::memset(&plan, 0, sizeof plan);
        const bool planned = E.fn(E.user, r.integrated, tp, &plan) == JT_OK;
        const auto t3 = now();
        if (planned && plan.needed) early_pass3_enqueue(h, h->s16_p2.p, h->m_p2, out_rate, plan);
        else spec_loudnorm_enqueue(h, h->s16_p2.p, h->m_p2, out_rate);
        if (timing) fprintf(stderr, "plan hook: wait for K-weighting %.3f ms, gating %.3f, wait for true peak %.3f, plan %.3f, enqueue %.3f\n",
                            ms(tq, t0), ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
    }
    analysis_join(h, regions ? 2 : 1);
    h->out_rate = out_rate; h->m_p4 = 0;
    JT_HIP(hipEventRecord(h->ev1, h->stream));
    if (h->early_p3.valid) {
        // the main stream still carries Pass 3's prefix chain (several ms): the analysis chains end long before it, so their host
        // arithmetic (gating, merges, the per-frame assembly) runs while the GPU works instead of after it
        analysis_complete(h, J, out, nullptr, 0, true);         // (the full-length chains first: the regions' run behind them)
        for (int i = 0; i < 4 * (regions ? 2 : 1); ++i) JT_HIP(jt_event_wait(h, h->ev_join[i]));
        check_cancel(h);
        if (regions) { regions_finish(h, out_rate, RJ, slot.out); slot.valid = true; }
        JT_HIP(jt_event_wait(h, h->ev1));
        check_cancel(h);
    } else {
        analysis_complete(h, J, out, nullptr, 0, true);
        JT_HIP(jt_event_wait(h, h->ev1));
        check_cancel(h);
        if (regions) { regions_finish(h, out_rate, RJ, slot.out); slot.valid = true; }
    }
    out->n_input_frames = 0;
    float ms = 0; JT_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1)); h->timers.pass2_ms = ms;
    if (nlm_timed) { JT_HIP(hipEventElapsedTime(&ms, h->ev2, h->ev3)); h->timers.nlm_ms = ms; h->timers.nlm_launches = 1; }
    JT_API_END(h)
}

// ---------------------------------------------------------------- limiter driver (clean-point segmentation)
// the brickwall may leave as float + s16 (LimOut16) when the wave-per-segment kernel serves this look-ahead length
static bool limiter_can_emit16(jt_ctx *h, int sr, double attack_ms)
{
    int B = (int)(sr * (attack_ms / 1000.) * 1); if (B < 1) B = 1;
    return jt_limiter_wave_ok(B) && !h->opts.limiter_lanes && !h->opts.brickwall_f64;
}
static void run_limiter(jt_ctx *h, double *in, double *out, int64_t n, int sr, double limit, double attack_ms,
                        double release_ms, double in_gain, const LimS16 &src, const LimOut16 *o16)
{
    const double attack = attack_ms / 1000., release = release_ms / 1000.;
    int B = (int)(sr * attack * 1); if (B < 1) B = 1;
    const double asc_coeff = std::pow(0.5, 0.8 - 0.5) * 2 * -1;     // asc_level = 0.8 (normalise.go:459,477)
    const int blk = 256;
    const int64_t nblk = (n + blk - 1) / blk;
    h->d_scr3.ensure((size_t)nblk);
    // the wave-per-segment limiter reads the few samples of its hot segments from the s16 source itself: the converted signal is not stored
    const bool from16 = src.s16 && !JT_AB_ON(h->opts.no_lim_s16) && jt_limiter_wave_ok(B) && !h->opts.limiter_lanes;
    const LimSrc16 s16src{src.s16, src.vol, src.vol_in_float};
    if (from16) launch_absmax_conv_s16(src.s16, nullptr, out, n, src.vol, src.vol_in_float, in_gain, h->d_scr3.p, nblk, h->stream);
    else if (src.s16 && !JT_AB_ON(h->opts.no_lim_s16)) launch_absmax_conv_s16(src.s16, in, out, n, src.vol, src.vol_in_float, in_gain, h->d_scr3.p, nblk, h->stream);
    else {
        if (src.s16) launch_s16_to_f64(src.s16, in, n, src.vol, src.vol_in_float, h->stream);
        launch_absmax_copy_f64(in, out, n, in_gain, h->d_scr3.p, nblk, h->stream, o16);      // blk == 256
    }
    // a position p is clean when no sample in the previous (B + release*sr + 4) samples exceeds the limit; segment starts are
    // picked on the device (one candidate per 2048 samples), so the limiter needs no host round trip
    const int need = (int)std::ceil((B + release * sr + 4.0) / blk) + 1;
    const int target = 2048 / blk;
    const int64_t ntargets = (nblk + target - 1) / target;
    h->lim_bounds.ensure((size_t)ntargets);
    h->lim_delta.ensure((size_t)ntargets * B); h->lim_pos.ensure((size_t)ntargets * B); h->lim_lp.ensure((size_t)ntargets * B);
    launch_limiter_f64(in, out, n, sr, limit, B, release, asc_coeff, h->d_scr3.p, nblk, blk, need, target, h->lim_bounds.p, ntargets,
                       in_gain, h->lim_delta.p, h->lim_pos.p, h->stream, h->lim_lp.p, h->opts.limiter_lanes, JT_AB_ON(h->opts.lim_profile), from16 ? &s16src : nullptr, o16);
}

// loudnorm (dynamic-mode first pass) input statistics of a signal at `rate`, measured after swr -> 192 kHz
struct LoudnormJob { KwJob kw; int64_t nfull = 0; int blk = 0; };
static void loudnorm_measure_enqueue(jt_ctx *h, const int16_t *s16, const double *f64, int64_t n, int rate, LoudnormJob *J)
{
    p3_measure_enqueue(h, s16, f64, n, rate, &J->kw, &J->nfull, &J->blk, h->stream, nullptr);
}
static void loudnorm_measure_finish(const LoudnormJob &J, jt_loudnorm_stats *out)
{
    // the block sums and peaks are read where the kernel left them (pinned host memory: KwJob::hc = (nfull + 1) sums, then as many peaks);
    // copying the 0.6 MB of an hour's blocks into vectors first was a third of this function, which the GPU waits for
    const double *bsum = J.kw.hc, *bpk = J.kw.hc + (size_t)J.nfull + 1;
    double pk = 0; for (int64_t k = 0; k <= J.nfull; ++k) pk = std::max(pk, bpk[(size_t)k]);
    jt_loudnorm_finish(bsum, J.nfull, J.blk, true, 1.0, &out->input_i, &out->input_lra, &out->input_thresh);
    out->input_tp = 20 * std::log10(pk);
    out->output_i = out->output_tp = out->output_lra = out->output_thresh = NAN; out->target_offset = NAN;
    out->normalization_type_dynamic = 1;
}

static void pass3_core(jt_ctx *h, const int16_t *s16, int64_t m, int rate, const jt_limiter_plan *lim, jt_loudnorm_stats *out)
{
    if (h->early_p3.valid) {
        // Pass 2 ran this very measurement (jt_pass3_plan_hook) and synchronised behind it: take it if the plan is the one it was run with
        auto &E = h->early_p3;
        E.valid = false;
        if (s16 == h->s16_p2.p && m == h->m_p2 && lim && lim->needed && E.plan.needed && lim->pre_gain_db == E.plan.pre_gain_db && lim->limit == E.plan.limit) {
            if (h->spec_ln.pending) { JT_HIP(jt_stream_sync(h, h->spec_ln.stream)); h->spec_ln.pending = false; }   // (retire the no-prefix guess)
            LoudnormJob J; J.kw = E.kw; J.nfull = E.nfull; J.blk = E.blk;
            loudnorm_measure_finish(J, out);
            return;
        }
        h->lim_keep.valid = false;
    }
    if (h->spec_ln.pending) {
        // Pass 2 already queued this measurement of its output for the no-prefix plan: collect it -- or let it finish (it owns the
        // 192 kHz stream buffer) and measure what was asked for
        JT_HIP(jt_stream_sync(h, h->spec_ln.stream));
        h->spec_ln.pending = false;
        if (s16 == h->s16_p2.p && m == h->m_p2 && !(lim && lim->needed)) {
            LoudnormJob J; J.kw = h->spec_ln.kw; J.nfull = h->spec_ln.nfull; J.blk = h->spec_ln.blk;
            loudnorm_measure_finish(J, out);
            return;
        }
    }
    pass_begin(h, m * 192000 / rate + 600000, 1);
    LoudnormJob J;
    h->lim_keep.valid = false;
    if (lim && lim->needed) {
        h->f64_a.ensure((size_t)m); h->f64_b.ensure((size_t)m);
        const bool pre = lim->pre_gain_db > 0;
        const double g = pre ? std::pow(10.0, lim->pre_gain_db / 20.0) : 1.0;
        run_limiter(h, h->f64_a.p, h->f64_b.p, m, rate, lim->limit, 5.0, 100.0, 1.0, LimS16{s16, g, pre ? 1 : 0});
        h->lim_keep = {true, s16, m, rate, lim->pre_gain_db, lim->limit};
        loudnorm_measure_enqueue(h, nullptr, h->f64_b.p, m, rate, &J);
    } else {
        loudnorm_measure_enqueue(h, s16, nullptr, m, rate, &J);
    }
    JT_HIP(jt_stream_sync(h, h->stream));
    loudnorm_measure_finish(J, out);
}

extern "C" int jt_pass3(jt_ctx *h, const jt_limiter_plan *lim, double target_i, double target_tp, double target_lra,
                        jt_loudnorm_stats *out)
{
    JT_API_BEGIN(h)
    (void)target_i; (void)target_tp; (void)target_lra;
    JT_REQUIRE(h->m_p2 > 0, JT_E_STATE, "pass3: no Pass-2 output on device");
    JT_REQUIRE(out, JT_E_INVAL, "pass3: bad arguments");
    check_cancel(h);
    JT_HIP(hipEventRecord(h->ev0, h->stream));
    pass3_core(h, h->s16_p2.p, h->m_p2, h->out_rate, lim, out);
    JT_HIP(hipEventRecord(h->ev1, h->stream));
    JT_HIP(jt_event_wait(h, h->ev1));
    float ms = 0; JT_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1)); h->timers.pass3_ms = ms;
    JT_API_END(h)
}

// ---------------------------------------------------------------- loudnorm, dynamic mode (af_loudnorm.c; k_loudnorm.hip)
// x: the stream at 192 kHz on the device with room for m + 576000 more doubles behind it; y: m doubles.  Synchronises the stream.
struct LoudnormDynIn { double target_i, target_lra, target_tp, measured_i, measured_lra, measured_tp, measured_thresh, offset; bool linear, dual_mono; };
static void loudnorm_dynamic_run(jt_ctx *h, double *x, int64_t m, const LoudnormDynIn &in, double *y, jt_loudnorm_stats *st)
{
    constexpr int F100 = 19200, F3000 = 576000, LBS = 40320, FINAL = F3000 - F100;
    hipStream_t s = h->stream;
    const double ch = in.dual_mono ? 2.0 : 1.0;
    const double target_tp_lin = std::pow(10., in.target_tp / 20.);
    auto finish_stats = [&](const std::vector<double> &bs_in, int64_t nfull_in, double pk_in, bool dynamic) {
        KwJob oj; jt_kweight_enqueue_f64(h, y, m, 192000, F100, &oj, s);
        JT_HIP(jt_stream_sync(h, s));
        std::vector<double> bo, po; jt_kweight_finish(&oj, bo, po);
        const int64_t nfo = m / F100;
        double pko = 0; for (int64_t k = 0; k <= nfo; ++k) pko = std::max(pko, po[(size_t)k]);
        if (st) {
            jt_loudnorm_finish(bs_in.data(), nfull_in, F100, in.dual_mono, 1.0, &st->input_i, &st->input_lra, &st->input_thresh);
            jt_loudnorm_finish(bo.data(), nfo, F100, in.dual_mono, 1.0, &st->output_i, &st->output_lra, &st->output_thresh);
            st->input_tp = 20 * std::log10(pk_in); st->output_tp = 20 * std::log10(pko);
            st->target_offset = in.target_i - st->output_i;
            st->normalization_type_dynamic = dynamic ? 1 : 0;
        }
    };
    if (m < F3000) {
        // shorter than the 3 s the first frame asks for: the filter measures what it got and applies one gain (frame_type LINEAR_MODE)
        KwJob ij; jt_kweight_enqueue_f64(h, x, m, 192000, F100, &ij, s);
        JT_HIP(jt_stream_sync(h, s));
        std::vector<double> bs, pk; jt_kweight_finish(&ij, bs, pk);
        const int64_t nf = m / F100;
        double peak = 0; for (int64_t k = 0; k <= nf; ++k) peak = std::max(peak, pk[(size_t)k]);
        double gi, gl, gt; jt_loudnorm_finish(bs.data(), nf, F100, in.dual_mono, 1.0, &gi, &gl, &gt);
        const double offset = std::pow(10., (in.target_i - gi) / 20.);
        const double offset_tp = peak * offset;
        const double g = offset_tp < target_tp_lin ? offset : target_tp_lin - peak;
        launch_scale_f64(x, y, m, g, s);
        finish_stats(bs, nf, peak, false);
        return;
    }
    // input meter over the stream followed by its last 2.9 s again (the filter's flush frame passes through r128_in a second time)
    JT_HIP(hipMemcpyAsync(x + m, x + (m - FINAL), (size_t)FINAL * sizeof(double), hipMemcpyDeviceToDevice, s));
    const int64_t m_ext = m + FINAL;
    KwJob ij; jt_kweight_enqueue_f64(h, x, m_ext, 192000, F100, &ij, s);
    const int64_t rem = (m - F3000) % F100;
    KwJob pj; const int64_t pad = rem ? F100 - rem : 0;
    if (rem) {      // a trailing partial frame: its short-term window is not on block boundaries; measure a copy shifted so that it is
        JT_HIP(hipMemsetAsync(y, 0, (size_t)pad * sizeof(double), s));
        JT_HIP(hipMemcpyAsync(y + pad, x, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, s));
        jt_kweight_enqueue_f64(h, y, m + pad, 192000, F100, &pj, s);
    }
    JT_HIP(jt_stream_sync(h, s));
    std::vector<double> bs, pk; jt_kweight_finish(&ij, bs, pk);
    const int64_t nfull_x = m / F100, nfull_ext = m_ext / F100;
    double peak = 0; for (int64_t k = 0; k <= nfull_x; ++k) peak = std::max(peak, pk[(size_t)k]);
    // per-frame series of the input meter: histogram gating as ebur128.c keeps it, updated block by block
    const int64_t n_inner = (m - F3000 + F100 - 1) / F100;
    std::vector<double> series((size_t)std::max<int64_t>(1, n_inner) * 3);
    jt_loudnorm_series(bs.data(), nfull_x, F100, in.dual_mono, n_inner, series.data());
    if (rem) {
        std::vector<double> b2, p2; jt_kweight_finish(&pj, b2, p2);
        const int64_t nb2 = (m + pad) / F100;
        double e = 0; for (int q = 29; q >= 0; --q) e += b2[(size_t)(nb2 - 1 - q)];
        e = e * ch / (double)F3000;
        series[(size_t)(n_inner - 1) * 3] = e <= 0.0 ? -HUGE_VAL : 10 * (std::log(e) / std::log(10.0)) - 0.691;
    }
    LoudnormDynParams P; std::memset(&P, 0, sizeof(P));
    P.target_i = in.target_i; P.target_lra = in.target_lra; P.target_tp_lin = target_tp_lin; P.measured_thresh = in.measured_thresh;
    P.offset_lin = std::pow(10., in.offset / 20.);
    {
        double e = 0; for (int q = 0; q < 30; ++q) e += bs[(size_t)q];
        e = e * ch / (double)F3000;
        const double shortterm = e <= 0.0 ? -HUGE_VAL : 10 * (std::log(e) / std::log(10.0)) - 0.691;
        double env;
        if (shortterm < in.measured_thresh) { P.above0 = 0; env = shortterm <= -70. ? 0. : in.target_i - in.measured_i; }
        else { P.above0 = 1; env = shortterm <= -70. ? 0. : in.target_i - shortterm; }
        P.delta0 = std::pow(10., env / 20.);
    }
    {
        double total = 0.0; const double sigma = 3.5, c1 = 1.0 / (sigma * std::sqrt(2.0 * M_PI)), c2 = 2.0 * std::pow(sigma, 2.0);
        for (int i = 0; i < 21; i++) { const int xx = i - 10; P.weights[i] = c1 * std::exp(-(std::pow(xx, 2.0) / c2)); total += P.weights[i]; }
        const double adjust = 1.0 / total;
        for (int i = 0; i < 21; i++) P.weights[i] *= adjust;
    }
    jt_kweight_coeffs5(192000, P.kwb, P.kwa);
    P.dual_mono = in.dual_mono ? 1 : 0; P.n_inner = n_inner; P.final_len = FINAL; P.no_batch = h->opts.ln_no_batch ? 1 : 0; P.stream_stop = h->opts.ln_stream_stop;
    h->ln_ring.ensure(LBS); h->ln_series.ensure(series.size());
    JT_HIP(hipMemcpyAsync(h->ln_series.p, series.data(), series.size() * sizeof(double), hipMemcpyHostToDevice, s));
    // (the workgroup kernel runs as a sequence of launches of a few milliseconds each on the main stream: a launch that lasts a second
    // holds up every other handle's stream that shares its hardware queue, k_loudnorm.hip)
    h->ln_carry.ensure(256);
    // the stream path's scratch (k_loudnorm.hip): ~0.7 bytes per 192 kHz sample
    LnsBufs lns{}; const bool stream = !h->opts.ln_no_stream && !h->opts.ln_no_batch && n_inner > 48;
    if (stream) {
        h->ln_scratch.ensure(jt_lns_scratch_bytes(m, n_inner, nullptr, nullptr));
        jt_lns_scratch_bytes(m, n_inner, &lns, h->ln_scratch.p);
        JT_HIP(hipMemsetAsync(lns.ctl, 0, sizeof(LnsCtl), s));
        // (test switch, negative values: -1 = a peak list of 64 entries, -N = a segment list of N entries: the two "list full" ways out)
        if (h->opts.ln_stream_stop == -1) lns.pk_cap = 64;
        else if (h->opts.ln_stream_stop < -1) lns.seg_cap = std::min(lns.seg_cap, -h->opts.ln_stream_stop);
        if (h->opts.ln_stream_stop < 0) P.stream_stop = 0;
    }
    launch_loudnorm_dynamic(x, m, P, h->ln_series.p, h->ln_ring.p, y, nullptr, s, h->ln_carry.p, h->opts, stream ? &lns : nullptr);
    JT_HIP(hipGetLastError());
    finish_stats(bs, nfull_ext, peak, true);
    h->timers.ln_stream_frames = 0; h->timers.ln_stream_why = 0;
    if (stream) {
        // (diagnostic; the stream has been waited for)  frames the stream path covered, and why its last attempt stopped where it did
        h->io_small.begin(256);
        LnsCtl *c = h->io_small.take<LnsCtl>(1);
        JT_HIP(hipMemcpyAsync(c, lns.ctl, sizeof(LnsCtl), hipMemcpyDeviceToHost, s));
        JT_HIP(jt_stream_sync(h, s));
        h->timers.ln_stream_frames = c->frames; h->timers.ln_stream_why = c->why_mask;
        if (jt_host_timing().load(std::memory_order_relaxed))
            fprintf(stderr, "loudnorm dynamic, stream path: %lld frames in %d attempts; last attempt: frames [%d, %d), %d peaks, %d segments, %d machine steps, %d list windows, %.2f Mcycles, why %d\n",
                    (long long)c->frames, c->attempts, c->ka, c->kbe, c->npk, c->nseg, c->iters, c->refills, c->cycles / 1e6, c->why);
    }
}

// ---------------------------------------------------------------- Pass 4
extern "C" int jt_pass4(jt_ctx *h, const jt_limiter_plan *lim, const jt_loudnorm_apply *ap, jt_analysis *out, jt_loudnorm_stats *stats)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(h->m_p2 > 0, JT_E_STATE, "pass4: no Pass-2 output on device");
    JT_REQUIRE(ap && out, JT_E_INVAL, "pass4: bad arguments");
    check_cancel(h);
    JT_HIP(hipEventRecord(h->ev0, h->stream));
    std::memset(out, 0, sizeof(*out));
    const int64_t m = h->m_p2; const int rate = h->out_rate;
    // af_loudnorm.c init(): linear mode only when every measured_* is supplied and the projected peak / LRA fit
    const double offset_db = ap->target_i - ap->measured_i;
    const double offset_tp = ap->measured_tp + offset_db;
    const bool linear = (ap->measured_tp != 99 && ap->measured_thresh != -70 && ap->measured_lra != 0 && ap->measured_i != 0) &&
                        (offset_tp <= ap->target_tp) && (ap->measured_lra <= ap->target_lra);
    // not linear: the filter falls back to its dynamic mode at 192 kHz (the reference logs a warning and delivers the file,
    // normalise.go:687-693); the gain then comes from loudnorm_dynamic_run below instead of one multiplication
    const bool dyn = !linear;
    const double gain = dyn ? 1.0 : std::pow(10., offset_db / 20.);
    h->timers.declick_ms = 0; h->timers.declick_repaired = 0;
    jt_ctx::RegionSlot &slot = h->region_slot[1];
    slot.valid = false;
    const bool announced = slot.armed; slot.armed = false;
    const int64_t m192 = dyn ? get_swr(h, rate, 192000).out_len(m) : 0;
    pass_begin(h, dyn ? m192 + 700000 : m, dyn ? 6 : 2, announced ? region_slot_samples(slot, rate) : 0, announced ? 2 : 0);
    h->f64_a.ensure((size_t)m); h->f64_b.ensure((size_t)m);
    double *sig = h->f64_a.p, *tmp = h->f64_b.p;
    const bool pre = lim && lim->needed && lim->pre_gain_db > 0;
    const jt_ctx::LimKeep &lk = h->lim_keep;
    const bool kept = lk.valid && lim && lim->needed && lk.src == h->s16_p2.p && lk.m == m && lk.rate == rate &&
                      lk.pre_gain_db == lim->pre_gain_db && lk.limit == lim->limit && !h->opts.no_lim_keep;
    h->lim_keep.valid = false;                                  // (this pass overwrites both buffers)
    if (kept) std::swap(sig, tmp);                              // Pass 3 ran this prefix on these samples: its output is still in f64_b
    else {
        const double vol = pre ? std::pow(10.0, lim->pre_gain_db / 20.0) : 1.0;
        if (lim && lim->needed) { run_limiter(h, sig, tmp, m, rate, lim->limit, 5.0, 100.0, 1.0, LimS16{h->s16_p2.p, vol, pre ? 1 : 0}); std::swap(sig, tmp); }
        else launch_s16_to_f64(h->s16_p2.p, sig, m, vol, pre ? 1 : 0, h->stream);
    }
    check_cancel(h);
    // loudnorm r128_in / r128_out statistics (libavfilter/ebur128.c) at the stream rate
    KwJob sj; const int sblk = (rate + 5) / 10;
    const bool stats_lin = stats && !dyn;
    if (dyn) {
        SwrDev &up = get_swr(h, rate, 192000);
        h->stream_d.ensure((size_t)m192 + 576000 + 64); h->stream_y.ensure((size_t)m192 + 19200 + 64);
        launch_resample_stream_f64(sig, m, up.bank_d.p, up.pl.phase_count, up.pl.filter_length, up.pl.center, up.pl.step, m192, h->stream_d.p, h->stream, h->opts);
        LoudnormDynIn din{ap->target_i, ap->target_lra, ap->target_tp, ap->measured_i, ap->measured_lra, ap->measured_tp, ap->measured_thresh,
                          ap->offset, true, true};
        jt_loudnorm_stats dst; std::memset(&dst, 0, sizeof(dst));
        loudnorm_dynamic_run(h, h->stream_d.p, m192, din, h->stream_y.p, &dst);
        if (stats) *stats = dst;
        // the reference's own aresample back to the source rate (normalise.go:1293-1310); swr would deliver ceil(m192 * rate / 192000)
        // samples, one more than m for some lengths: the pass keeps m
        SwrDev &dn = get_swr(h, 192000, rate);
        launch_swr_plain_f64(h->stream_y.p, m192, dn.bank_d.p, dn.pl.phase_count, dn.pl.filter_length, dn.pl.center, dn.pl.step,
                             std::min<int64_t>(dn.out_len(m192), m), tmp, h->stream, h->opts.swr_untiled ? nullptr : dn.bank_dT.p);
        if (dn.out_len(m192) < m) JT_HIP(hipMemsetAsync(tmp + dn.out_len(m192), 0, (size_t)(m - dn.out_len(m192)) * sizeof(double), h->stream));
        std::swap(sig, tmp);
        check_cancel(h);
    }
    const double *stats_src = sig;                        // the gained-by-nothing stream the loudnorm meters see (r128_in; r128_out = gain * it)
    bool stats_queued = false, stats_on_spec = false;
    if (stats_lin && !ap->adeclick_enabled) {
        fork_aux(h, 3, 3); jt_kweight_enqueue_f64(h, stats_src, m, rate, sblk, &sj, h->aux[3]); stats_queued = true;
        JT_HIP(hipEventRecord(h->ev_stats, h->aux[3]));
    }
    // adeclick on the gained stream (af_adeclick.c), then the brickwall alimiter
    double brick_gain = gain;
    const unsigned long long *dk_stats = nullptr;
    if (ap->adeclick_enabled) {
        std::string why;
        JT_REQUIRE(jt_adeclick_supported(rate, ap->adeclick_window_ms, ap->adeclick_overlap_pct, 2.0, ap->adeclick_method, &why), JT_E_UNSUPPORTED, why);
        JT_HIP(hipEventRecord(h->ev2, h->stream));
        launch_adeclick(h, sig, tmp, m, rate, ap->adeclick_threshold, ap->adeclick_window_ms, ap->adeclick_overlap_pct, 2.0, 2.0, gain,
                        nullptr, h->stream, ap->adeclick_method);
        JT_HIP(hipEventRecord(h->ev3, h->stream));
        unsigned long long *hs = h->pin.take<unsigned long long>(16);
        JT_HIP(hipMemcpyAsync(hs, h->declick_ctl.p + 288, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        dk_stats = hs;
        std::swap(sig, tmp);
        brick_gain = 1.0;
        // the statistics job starts behind adeclick (beside it, its few workgroups waited for the persistent waves' CU slots for the whole
        // launch: 9.5 ms in the kernel trace for 0.3 ms of work) and reads adeclick's INPUT, which the brickwall must therefore not reuse
        if (stats_lin) {
            // on the stream Pass 2 uses for the early Pass-3 measurement (idle now), so that no analysis chain queues behind it
            JT_HIP(hipEventRecord(h->spec_ln.fork, h->stream));
            JT_HIP(hipStreamWaitEvent(h->spec_ln.stream, h->spec_ln.fork, 0));
            jt_kweight_enqueue_f64(h, stats_src, m, rate, sblk, &sj, h->spec_ln.stream);
            JT_HIP(hipEventRecord(h->ev_stats, h->spec_ln.stream));
            stats_queued = true; stats_on_spec = true;
        }
        h->f64_c.ensure((size_t)m);
        tmp = h->f64_c.p;
    }
    // dbl -> flt (aspectralstats) -> dbl (ebur128) -> s16.  Nothing reads the brickwall's doubles but that conversion: its two kernels
    // write the float and the s16 themselves (the first sweep wherever the limiter rests, the wave kernel its hot segments) instead of
    // 8 bytes per sample that a third sweep reads back (option brickwall_f64: the three-sweep form, same bytes)
    h->work_a.ensure((size_t)m);
    h->s16_p4.ensure((size_t)m);
    if (limiter_can_emit16(h, rate, 1.0)) {
        const LimOut16 o16{h->s16_p4.p, h->work_a.p};
        run_limiter(h, sig, tmp, m, rate, ap->brickwall_limit, 1.0, 50.0, brick_gain, LimS16{}, &o16);
        check_cancel(h);
    } else {
        run_limiter(h, sig, tmp, m, rate, ap->brickwall_limit, 1.0, 50.0, brick_gain);
        std::swap(sig, tmp);
        check_cancel(h);
        launch_f64_to_s16(sig, h->s16_p4.p, h->work_a.p, m, 1, h->stream);
    }
    h->m_p4 = m;
    h->pcm_early = {};
    if (h->p4_output_hook) h->p4_output_hook(h);                // (a handle pool: PCM to the host, MD5 started -- jt_internal.h)
    AnalysisJob J;
    analysis_enqueue(h, h->work_a.p, m, rate, true, rate / 10, &J, false);
    RegionJobs RJ;
    const bool regions = announced && regions_resolve(slot.start_s, slot.dur_s, rate, m, &RJ);
    if (regions) regions_enqueue(h, h->s16_p4.p, rate, &RJ, true);         // beside the full-length analysis
    analysis_join(h, regions ? 2 : 1);
    if (stats_on_spec) { JT_HIP(hipEventRecord(h->spec_ln.fork, h->spec_ln.stream)); JT_HIP(hipStreamWaitEvent(h->stream, h->spec_ln.fork, 0)); }
    JT_HIP(hipEventRecord(h->ev1, h->stream));
    // host tail, while the output analysis is still on the GPU: the loudnorm statistics (two gated integrations over the 100 ms blocks
    // of a job that ended before the analysis began), then the analysis itself chain by chain as its chains end
    if (stats_lin && stats_queued) {
        JT_HIP(jt_event_wait(h, h->ev_stats));
        const int64_t nfull = m / sblk;
        std::vector<double> bsum, bpk;
        jt_kweight_finish(&sj, bsum, bpk);
        double pk = 0; for (int64_t k = 0; k <= nfull; ++k) pk = std::max(pk, bpk[(size_t)k]);
        jt_loudnorm_finish(bsum.data(), nfull, sblk, true, 1.0, &stats->input_i, &stats->input_lra, &stats->input_thresh);
        jt_loudnorm_finish(bsum.data(), nfull, sblk, true, gain * gain, &stats->output_i, &stats->output_lra, &stats->output_thresh);
        stats->input_tp = 20 * std::log10(pk);
        stats->output_tp = 20 * std::log10(pk * gain);
        stats->target_offset = ap->target_i - stats->output_i;
        stats->normalization_type_dynamic = 0;
    }
    analysis_complete(h, J, out, nullptr, 0, true);
    JT_HIP(jt_event_wait(h, h->ev1));
    check_cancel(h);
    if (dk_stats) {
        h->timers.declick_repaired = (int64_t)dk_stats[0]; h->timers.declick_heavy_windows = (int64_t)dk_stats[2];
        if (JT_AB_ON(h->opts.dk_profile)) {
            fprintf(stderr, "adeclick: repaired %llu, second-pass windows %llu, third-pass windows %llu; phase clocks (JT_DK_PROFILE build):", dk_stats[0], dk_stats[2], dk_stats[3]);
            for (int i = 4; i < 12; ++i) fprintf(stderr, " %llu", dk_stats[i]);
            fprintf(stderr, "\n");
            std::vector<int> hist(192);
            JT_HIP(hipMemcpy(hist.data(), h->declick_heavy.p + 2 * ((m + 1211) / 1212), 192 * sizeof(int), hipMemcpyDeviceToHost));
            fprintf(stderr, "bwmax per window:"); for (int i = 0; i < 64; ++i) if (hist[i]) fprintf(stderr, " %d:%d", i, hist[i]);
            fprintf(stderr, "\nbw per pivot:"); for (int i = 0; i < 64; ++i) if (hist[64 + i]) fprintf(stderr, " %d:%d", i, hist[64 + i]);
            fprintf(stderr, "\nF/16 per window:"); for (int i = 0; i < 64; ++i) if (hist[128 + i]) fprintf(stderr, " %d:%d", i, hist[128 + i]);
            fprintf(stderr, "\n");
        }
        { float dms = 0; JT_HIP(hipEventElapsedTime(&dms, h->ev2, h->ev3)); h->timers.declick_ms = dms; }
        JT_REQUIRE(dk_stats[1] == 0, JT_E_HIP, "adeclick: singular interpolation matrix (af_adeclick.c would fail the graph)");
    }
    if (regions) { regions_finish(h, rate, RJ, slot.out); slot.valid = true; }
    float ms = 0; JT_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1)); h->timers.pass4_ms = ms;
    JT_API_END(h)
}

// ---------------------------------------------------------------- region re-measure
// MeasureOutputRegions (analyser_output.go:276-317) measures the room-tone and the speech region of one output back to
// back; jt_region_measure_pair enqueues both analyses and synchronises once.  A region with dur_s <= 0 is skipped (out zeroed).
extern "C" int jt_region_prefetch(jt_ctx *h, int stage, const double start_s[2], const double dur_s[2])
{
    if (!h || !start_s || !dur_s || (stage != 2 && stage != 4)) { if (h) h->err = "region_prefetch: bad arguments"; return JT_E_INVAL; }
    jt_ctx::RegionSlot &sl = h->region_slot[stage == 4];
    sl.armed = true; sl.valid = false;
    for (int r = 0; r < 2; ++r) { sl.start_s[r] = start_s[r]; sl.dur_s[r] = dur_s[r]; }
    return JT_OK;
}

extern "C" int jt_region_measure_pair(jt_ctx *h, int stage, const double start_s[2], const double dur_s[2], jt_region_sample out[2])
{
    JT_API_BEGIN(h)
    JT_REQUIRE(out && start_s && dur_s, JT_E_INVAL, "region_measure: bad arguments");
    const int16_t *src = nullptr; int64_t m = 0;
    if (stage == 2) { src = h->s16_p2.p; m = h->m_p2; } else if (stage == 4) { src = h->s16_p4.p; m = h->m_p4; }
    JT_REQUIRE(src && m > 0, JT_E_STATE, "region_measure: stage output not on device");
    // measured by the stage's own pass (jt_region_prefetch)?
    const jt_ctx::RegionSlot &sl = h->region_slot[stage == 4];
    if (sl.valid && sl.start_s[0] == start_s[0] && sl.start_s[1] == start_s[1] && sl.dur_s[0] == dur_s[0] && sl.dur_s[1] == dur_s[1]) {
        out[0] = sl.out[0]; out[1] = sl.out[1];
        return JT_OK;
    }
    const int rate = h->out_rate;
    RegionJobs RJ;
    for (int r = 0; r < 2; ++r) std::memset(&out[r], 0, sizeof(out[r]));
    JT_REQUIRE(regions_resolve(start_s, dur_s, rate, m, &RJ), JT_E_INVAL, "region_measure: empty region");
    pass_begin(h, std::max<int64_t>(RJ.len[0] + RJ.len[1], 1), 2);
    regions_enqueue(h, src, rate, &RJ, false);
    analysis_join(h);
    JT_HIP(jt_stream_sync(h, h->stream));
    regions_finish(h, rate, RJ, out);
    JT_API_END(h)
}

extern "C" int jt_region_measure(jt_ctx *h, int stage, double start_s, double dur_s, jt_region_sample *out)
{
    if (!h || !out || !(start_s >= 0) || !(dur_s > 0)) { if (h) h->err = "region_measure: bad arguments"; return JT_E_INVAL; }
    const double st[2] = {start_s, 0.0}, du[2] = {dur_s, 0.0};
    jt_region_sample o[2];
    const int rc = jt_region_measure_pair(h, stage, st, du, o);
    if (rc == JT_OK) *out = o[0];
    return rc;
}

// ---------------------------------------------------------------- output
extern "C" int jt_output_len(jt_ctx *h, int stage, int64_t *n)
{
    if (!h || !n) return JT_E_INVAL;
    *n = stage == 2 ? h->m_p2 : (stage == 4 ? h->m_p4 : 0);
    return JT_OK;
}

extern "C" int jt_download_s16(jt_ctx *h, int stage, int16_t *dst, int64_t cap, int64_t *n)
{
    JT_API_BEGIN(h)
    const int16_t *src = stage == 2 ? h->s16_p2.p : (stage == 4 ? h->s16_p4.p : nullptr);
    const int64_t m = stage == 2 ? h->m_p2 : (stage == 4 ? h->m_p4 : 0);
    JT_REQUIRE(src && m > 0, JT_E_STATE, "download: stage output not on device");
    JT_REQUIRE(dst && cap >= m, JT_E_INVAL, "download: buffer too small");
    JT_HIP(hipMemcpyAsync(dst, src, sizeof(int16_t) * m, hipMemcpyDeviceToHost, h->stream));
    JT_HIP(jt_stream_sync(h, h->stream));
    if (n) *n = m;
    JT_API_END(h)
}

// calculateFrameLevel (encoder.go:235-257) of every frame of a stage output: what the reference's OnFrame callbacks feed the VU
// meter with (processor.go:336-338, normalise.go:288,1119-1121)
extern "C" int jt_output_frame_levels(jt_ctx *h, int stage, int frame_samples, double *levels_db, int64_t cap, int64_t *n_frames)
{
    JT_API_BEGIN(h)
    const int16_t *src = stage == 2 ? h->s16_p2.p : (stage == 4 ? h->s16_p4.p : nullptr);
    const int64_t m = stage == 2 ? h->m_p2 : (stage == 4 ? h->m_p4 : 0);
    JT_REQUIRE(src && m > 0, JT_E_STATE, "frame_levels: stage output not on device");
    JT_REQUIRE(frame_samples > 0 && (levels_db || cap == 0), JT_E_INVAL, "frame_levels: bad arguments");
    const int64_t nfr = (m + frame_samples - 1) / frame_samples;
    if (n_frames) *n_frames = nfr;
    const int64_t c = std::min(nfr, cap);
    if (c > 0) {
        h->d_scr3.ensure((size_t)nfr);                        // (a scratch of the passes, idle between them: no allocation per call)
        launch_frame_sumsq_s16(src, m, frame_samples, h->d_scr3.p, nfr, h->stream);
        std::vector<double> ss((size_t)nfr);
        JT_HIP(hipMemcpyAsync(ss.data(), h->d_scr3.p, sizeof(double) * nfr, hipMemcpyDeviceToHost, h->stream));
        JT_HIP(jt_stream_sync(h, h->stream));
        for (int64_t f = 0; f < c; ++f) {
            const int64_t cnt = std::min<int64_t>(frame_samples, m - f * frame_samples);
            const double rms = std::sqrt(ss[(size_t)f] / (double)cnt);
            levels_db[f] = rms < 0.00001 ? -70.0 : std::max(-70.0, std::min(0.0, 20.0 * std::log10(rms)));
        }
    }
    JT_API_END(h)
}

extern "C" int jt_get_timers(jt_ctx *h, jt_timers *out)
{
    if (!h || !out) return JT_E_INVAL;
    *out = h->timers;
    return JT_OK;
}

// ---------------------------------------------------------------- operator-level entry points (host buffers)
template <typename T> static void h2d(jt_ctx *h, DevBuf<T> &b, const T *src, size_t n)
{
    b.ensure(n);
    JT_HIP(hipMemcpyAsync(b.p, src, sizeof(T) * n, hipMemcpyHostToDevice, h->stream));
}
template <typename T> static void d2h(jt_ctx *h, T *dst, const T *src, size_t n)
{
    JT_HIP(hipMemcpyAsync(dst, src, sizeof(T) * n, hipMemcpyDeviceToHost, h->stream));
    JT_HIP(jt_stream_sync(h, h->stream));
}

extern "C" int jt_op_biquad_f32(jt_ctx *h, const float *in, float *out, int64_t n, int sr,
                                int hp_en, double hp_f, double hp_q, int lp_en, double lp_f, double lp_q)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in && out && n > 0, JT_E_INVAL, "op_biquad: bad arguments");
    DevBuf<float> a, b; h2d(h, a, in, (size_t)n); b.ensure((size_t)n);
    jt_filter_params p{}; p.hp_enabled = hp_en; p.hp_freq = hp_f; p.hp_q = hp_q; p.lp_enabled = lp_en; p.lp_freq = lp_f; p.lp_q = lp_q;
    BiquadF32 st[2]; int nst = 0; fill_biquads(&p, sr, st, &nst);
    JT_REQUIRE(nst > 0, JT_E_INVAL, "op_biquad: no stage enabled");
    launch_biquad_f32(a.p, b.p, n, nst, st, h->stream);
    d2h(h, out, b.p, (size_t)n);
    JT_API_END(h)
}

extern "C" int jt_op_anlmdn_f32(jt_ctx *h, const float *in, float *out, int64_t n, int sr,
                                double strength, double patch_s, double research_s, double smooth)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in && out && n > 0, JT_E_INVAL, "op_anlmdn: bad arguments");
    DevBuf<float> a, b; h2d(h, a, in, (size_t)n); b.ensure((size_t)n);
    run_anlmdn(h, a.p, b.p, n, sr, strength, patch_s, research_s, smooth);
    d2h(h, out, b.p, (size_t)n);
    float ms = 0; JT_HIP(hipEventElapsedTime(&ms, h->ev2, h->ev3)); h->timers.nlm_ms = ms; h->timers.nlm_launches = 1;
    JT_API_END(h)
}

extern "C" int jt_op_afftdn_f32(jt_ctx *h, const float *in, float *out, int64_t n, int sr, double nr, double nf, const double *bn)
{
    return jt_op_afftdn_tn_f32(h, in, out, n, sr, nr, nf, bn, 0, nullptr);
}
extern "C" int jt_op_afftdn_tn_f32(jt_ctx *h, const float *in, float *out, int64_t n, int sr, double nr, double nf, const double *bn,
                                   int track_noise, double *final_floor_db)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in && out && n > 0, JT_E_INVAL, "op_afftdn: bad arguments");
    DevBuf<float> a, b; h2d(h, a, in, (size_t)n); b.ensure((size_t)n);
    pass_begin(h, 1 << 16, 1);
    run_afftdn(h, a.p, b.p, n, sr, nr, nf, bn, track_noise != 0);
    d2h(h, out, b.p, (size_t)n);
    if (final_floor_db) *final_floor_db = track_noise ? h->af_last_floor : nf;
    JT_API_END(h)
}

extern "C" int jt_op_dynamics(jt_ctx *h, const float *in, float *out, int64_t n, int sr, const jt_filter_params *p)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in && out && p && n > 0, JT_E_INVAL, "op_dynamics: bad arguments");
    DevBuf<float> a, b; h2d(h, a, in, (size_t)n); b.ensure((size_t)n);
    DynParams d; jt_dyn_design(p, sr, &d);
    DevBuf<double> t1, t2, st; t1.ensure((size_t)n); t2.ensure((size_t)n); st.ensure((size_t)(n / 256 + 4));
    launch_dynamics(a.p, b.p, t1.p, t2.p, st.p, n, d, h->stream, h->opts);
    d2h(h, out, b.p, (size_t)n);
    JT_API_END(h)
}

extern "C" int jt_op_alimiter_f64(jt_ctx *h, const double *in, double *out, int64_t n, int sr,
                                  double limit, double attack_ms, double release_ms)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in && out && n > 0, JT_E_INVAL, "op_alimiter: bad arguments");
    DevBuf<double> a, b; h2d(h, a, in, (size_t)n); b.ensure((size_t)n);
    run_limiter(h, a.p, b.p, n, sr, limit, attack_ms, release_ms, 1.0);
    d2h(h, out, b.p, (size_t)n);
    JT_API_END(h)
}

extern "C" int jt_op_adeclick_f64(jt_ctx *h, const double *in, double *out, int64_t n, int sr, double threshold, double window_ms,
                                  double overlap_pct, int method, int64_t *n_repaired)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in && out && n > 0, JT_E_INVAL, "op_adeclick: bad arguments");
    std::string why;
    JT_REQUIRE(jt_adeclick_supported(sr, window_ms, overlap_pct, 2.0, method, &why), JT_E_UNSUPPORTED, why);
    DevBuf<double> a, b; h2d(h, a, in, (size_t)n); b.ensure((size_t)n);
    launch_adeclick(h, a.p, b.p, n, sr, threshold, window_ms, overlap_pct, 2.0, 2.0, 1.0, nullptr, h->stream, method);
    unsigned long long st[16] = {0};
    JT_HIP(hipMemcpyAsync(st, h->declick_ctl.p + 288, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    d2h(h, out, b.p, (size_t)n);
    if (JT_AB_ON(h->opts.dk_profile)) { fprintf(stderr, "adeclick phase clocks:"); for (int i = 4; i < 12; ++i) fprintf(stderr, " %llu", st[i]); fprintf(stderr, " heavy %llu", st[2]); fprintf(stderr, "\n"); }
    if (n_repaired) *n_repaired = (int64_t)st[0];
    JT_REQUIRE(st[1] == 0, JT_E_HIP, "adeclick: singular interpolation matrix");
    JT_API_END(h)
}

extern "C" int jt_op_loudnorm_dynamic_f64(jt_ctx *h, const double *in192, int64_t n, const jt_loudnorm_apply *ap, double *out192, jt_loudnorm_stats *stats)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in192 && out192 && ap && n > 0, JT_E_INVAL, "op_loudnorm_dynamic: bad arguments");
    pass_begin(h, n + 700000, 6);
    h->stream_d.ensure((size_t)n + 576000 + 64); h->stream_y.ensure((size_t)n + 19200 + 64);
    JT_HIP(hipMemcpyAsync(h->stream_d.p, in192, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    LoudnormDynIn din{ap->target_i, ap->target_lra, ap->target_tp, ap->measured_i, ap->measured_lra, ap->measured_tp, ap->measured_thresh,
                      ap->offset, true, true};
    jt_loudnorm_stats st; std::memset(&st, 0, sizeof(st));
    loudnorm_dynamic_run(h, h->stream_d.p, n, din, h->stream_y.p, &st);
    if (stats) *stats = st;
    d2h(h, out192, h->stream_y.p, (size_t)n);
    JT_API_END(h)
}

extern "C" int jt_op_resample_f32_to_s16(jt_ctx *h, const float *in, int64_t n, int in_rate, int out_rate,
                                         int16_t *out, int64_t cap, int64_t *n_out)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in && out && n > 0, JT_E_INVAL, "op_resample: bad arguments");
    DevBuf<float> a; h2d(h, a, in, (size_t)n);
    DevBuf<int16_t> o; int64_t m = 0;
    run_resample_s16(h, a.p, n, in_rate, out_rate, o, &m);
    JT_REQUIRE(cap >= m, JT_E_INVAL, "op_resample: output buffer too small");
    d2h(h, out, o.p, (size_t)m);
    if (n_out) *n_out = m;
    JT_API_END(h)
}

extern "C" int jt_op_ebur128(jt_ctx *h, const float *in, int64_t n, int sr, int dualmono, jt_r128 *out,
                             double *ms, double *ss, double *tps, double *sps, int64_t cap, int64_t *n_blocks)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in && out && n > 0, JT_E_INVAL, "op_ebur128: bad arguments");
    DevBuf<float> a; h2d(h, a, in, (size_t)n);
    pass_begin(h, n, 1);
    AnalysisJob J; J.want_astats = false; J.want_spec = false;
    analysis_enqueue(h, a.p, n, sr, dualmono != 0, 0, &J);
    JT_HIP(jt_stream_sync(h, h->stream));
    AnalysisHost A; analysis_finish(h, J, &A, false);
    out->integrated = A.r128.integrated; out->lra = A.r128.lra; out->lra_low = A.r128.lra_low; out->lra_high = A.r128.lra_high;
    out->momentary = A.nblocks ? A.r128.M[A.nblocks - 1] : NAN; out->shortterm = A.nblocks ? A.r128.S[A.nblocks - 1] : NAN;
    out->true_peak = A.tp_final; out->sample_peak = A.sp_final; out->target_threshold = A.r128.rel_threshold;
    for (int64_t k = 0; k < std::min(cap, A.nblocks); ++k) {
        if (ms) ms[k] = A.r128.M[k];
        if (ss) ss[k] = A.r128.S[k];
        if (tps) tps[k] = A.tp_cum[k];
        if (sps) sps[k] = A.sp_cum[k];
    }
    if (n_blocks) *n_blocks = A.nblocks;
    JT_API_END(h)
}

extern "C" int jt_op_astats(jt_ctx *h, const float *in, int64_t n, int sr, jt_astats *out)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in && out && n > 0, JT_E_INVAL, "op_astats: bad arguments");
    DevBuf<float> a; h2d(h, a, in, (size_t)n);
    pass_begin(h, n, 1);
    AstatsJob J; jt_astats_enqueue(h, a.p, n, sr, &J, h->stream, h->stream, h->stream);
    JT_HIP(jt_stream_sync(h, h->stream));
    jt_astats_finish(&J, out);
    JT_API_END(h)
}

extern "C" int jt_op_aspectralstats(jt_ctx *h, const float *in, int64_t n, int sr, jt_spectral *hops, int64_t cap, int64_t *n_hops)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in && hops && n > 0, JT_E_INVAL, "op_aspectralstats: bad arguments");
    DevBuf<float> a; h2d(h, a, in, (size_t)n);
    pass_begin(h, n, 1);
    AnalysisJob J; J.want_astats = false; J.want_r128 = false;
    analysis_enqueue(h, a.p, n, sr, false, 0, &J);
    JT_HIP(jt_stream_sync(h, h->stream));
    AnalysisHost A; analysis_finish(h, J, &A, false);
    for (int64_t k = 0; k < std::min(cap, A.nhops); ++k) hops[k] = A.hops[(size_t)k];
    if (n_hops) *n_hops = A.nhops;
    JT_API_END(h)
}

extern "C" int jt_op_loudnorm_measure_s16(jt_ctx *h, const int16_t *in, int64_t n, int sr, const jt_limiter_plan *lim, jt_loudnorm_stats *out)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(in && out && n > 0, JT_E_INVAL, "op_loudnorm_measure: bad arguments");
    DevBuf<int16_t> a; h2d(h, a, in, (size_t)n);
    pass3_core(h, a.p, n, sr, lim, out);
    JT_API_END(h)
}
