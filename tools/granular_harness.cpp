// granular_harness.cpp — walks the C ABI in the order a cgo shim at the reference's ENGINE seam would (setupFilterGraph /
// runFilterGraph replaced call by call: frame_processor.go:64,164; loudnormDeps, normalise.go:172-188), with the scalar control
// logic between the calls, and checks the outcome against jt_process_audio (the one-call mirror of ProcessAudio) on a second
// handle: same effective configuration, same statistics, the same s16 samples.  Plain C++ over include/*.h: this is the call
// sequence integration/go/gpu_engine.go makes, compiled and run where Go cannot be.
//
//   granular_harness <f32-mono-pcm-file> <sample-rate>        exit 0 = identical
#include "jtgpu.h"
#include "jt_host.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(call) do { int rc__ = (call); if (rc__ != JT_OK) { fprintf(stderr, "%s -> %d (%s)\n", #call, rc__, jt_last_error(h)); return 2; } } while (0)
static double qf(const char *fmt, double v) { char b[128]; snprintf(b, sizeof b, fmt, v); return strtod(b, nullptr); }
static double lin2db(double v) { return v > 0 ? 20.0 * log10(v) : -INFINITY; }
static double secs(int64_t ns) { return qf("%f", (double)ns / 1e9); }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s pcm.f32 rate\n", argv[0]); return 64; }
    const int sr = atoi(argv[2]);
    FILE *f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 66; }
    fseek(f, 0, SEEK_END); const long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<float> pcm((size_t)bytes / 4);
    if (fread(pcm.data(), 4, pcm.size(), f) != pcm.size()) { fclose(f); return 66; }
    fclose(f);
    const int64_t n = (int64_t)pcm.size();
    const int frame = 4096;

    // ---- reference run: the one-call mirror
    jt_ctx *h = nullptr;
    if (jt_open(0, &h) != JT_OK) { fprintf(stderr, "jt_open failed\n"); return 3; }
    jt_host_config base; jt_host_default_config(&base);
    static jt_process_result one; memset(&one, 0, sizeof one);
    CK(jt_upload_pcm(h, pcm.data(), n, sr, 1));
    CK(jt_process_audio(h, &base, frame, &one));
    int64_t m1 = 0; CK(jt_output_len(h, 4, &m1));
    std::vector<int16_t> out1((size_t)m1); CK(jt_download_s16(h, 4, out1.data(), m1, &m1));
    jt_close(h);

    // ---- granular walk on a fresh handle
    if (jt_open(0, &h) != JT_OK) return 3;
    CK(jt_upload_pcm(h, pcm.data(), n, sr, 1));
    const int64_t nfr = (n + frame - 1) / frame, cap_meta = n / (sr / 10) + 2;
    std::vector<double> fss((size_t)nfr), fpk((size_t)nfr);
    std::vector<jt_frame_meta> meta((size_t)cap_meta);
    jt_analysis a1; CK(jt_pass1(h, frame, &a1, fss.data(), fpk.data(), nfr, meta.data(), cap_meta));           // collectAnalysisFrames
    std::vector<jt_interval> iv((size_t)(n / (sr / 5) + 16));
    int64_t niv = jt_host_build_intervals(sr, n, frame, 1, fss.data(), fpk.data(), nfr, meta.data(), a1.n_frames_meta, 1, iv.data(), (int64_t)iv.size());
    static jt_measurements ms; memset(&ms, 0, sizeof ms);
    CK(jt_host_detect(&a1, iv.data(), niv, (double)n / sr, base.target_i, 1, &ms));                                  // buildInputMeasurements + detectVoiceActivity
    if (ms.has_speech_profile && ms.speech_profile.region.duration_ns > 0) {                                         // measureSpeechBands
        const double lo[2] = {1000.0, 6000.0}, hi[2] = {3000.0, 9000.0}; double db[2]; int ok[2];
        CK(jt_band_rms(h, secs(ms.speech_profile.region.start_ns), secs(ms.speech_profile.region.duration_ns), lo, hi, 2, db, ok));
        if (ok[0]) ms.speech_profile.body_band_rms = qf("%f", db[0]);
        if (ok[1]) ms.speech_profile.sib_band_rms = qf("%f", db[1]);
        ms.speech_profile.bands_measured = ok[0] && ok[1];
    }
    if (ms.has_noise_profile && ms.noise_profile.duration_ns > 0) {                                                  // measureNoiseBands
        double lo[15], hi[15], db[15]; int ok[15], finite = 0;
        for (int i = 0; i < 15; ++i) jt_host_afftdn_band_edges(i, &lo[i], &hi[i]);
        CK(jt_band_rms(h, secs(ms.noise_profile.start_ns), secs(ms.noise_profile.duration_ns), lo, hi, 15, db, ok));
        for (int i = 0; i < 15; ++i) { ms.noise_profile.band_noise[i] = ok[i] ? qf("%f", db[i]) : 0.0; if (ok[i] && std::isfinite(db[i])) finite++; }
        ms.noise_profile.band_noise_n = 15; ms.noise_profile.bands_measured = finite >= 10;
    }
    jt_host_finish_measurements(&ms);
    jt_host_config eff; jt_adaptive_diag dg; jt_host_adapt(&base, &ms, &eff, &dg);                                   // AdaptConfig
    jt_filter_params fp; jt_host_filter_params(&eff, &fp);
    jt_analysis a2; CK(jt_pass2(h, &fp, &a2));                                                                       // processWithFilters
    double rst[2] = {0, 0}, rdu[2] = {0, 0};
    if (ms.has_noise_profile && ms.noise_profile.duration_ns > 0) { rst[0] = secs(ms.noise_profile.start_ns); rdu[0] = secs(ms.noise_profile.duration_ns); }
    if (ms.has_speech_profile && ms.speech_profile.region.duration_ns > 0) { rst[1] = secs(ms.speech_profile.region.start_ns); rdu[1] = secs(ms.speech_profile.region.duration_ns); }
    jt_region_sample r2[2], r4[2];
    if (rdu[0] > 0 || rdu[1] > 0) CK(jt_region_measure_pair(h, 2, rst, rdu, r2));                                   // MeasureOutputRegions
    jt_limiter_decision ld; jt_limiter_plan plan;
    jt_host_plan_limiter(qf("%.3f", a2.r128.integrated), lin2db(qf("%.3f", a2.r128.true_peak)), &eff, &ld, &plan);   // planLimiterForLoudnorm
    jt_loudnorm_stats m3; CK(jt_pass3(h, &plan, eff.target_i, eff.target_tp, eff.target_lra, &m3));                  // measureWithLoudnorm
    m3.input_i = qf("%.2f", m3.input_i); m3.input_tp = qf("%.2f", m3.input_tp); m3.input_lra = qf("%.2f", m3.input_lra); m3.input_thresh = qf("%.2f", m3.input_thresh);
    double effI, offs; int lin;
    jt_host_calculate_linear_mode_target(m3.input_i, m3.input_tp, eff.target_i, jt_host_loudnorm_internal_target_tp(eff.target_i, m3.input_tp, m3.input_i), &effI, &offs, &lin);
    jt_host_config effc = eff; effc.target_i = effI;
    jt_loudnorm_apply ap; char spec[2048];
    jt_host_pass4_spec(&effc, &m3, effI - m3.input_i, &ld, 44100, nullptr, spec, (int)sizeof spec, &ap);             // buildLoudnormFilterSpec
    jt_analysis a4; jt_loudnorm_stats ln; CK(jt_pass4(h, &plan, &ap, &a4, &ln));                                     // applyLoudnormAndMeasure
    if (rdu[0] > 0 || rdu[1] > 0) CK(jt_region_measure_pair(h, 4, rst, rdu, r4));
    int64_t m2 = 0; CK(jt_output_len(h, 4, &m2));
    std::vector<int16_t> out2((size_t)m2); CK(jt_download_s16(h, 4, out2.data(), m2, &m2));
    jt_close(h);

    // ---- the two must agree exactly
    int bad = 0;
#define SAME(what, a, b) do { if (!((a) == (b) || (std::isnan((double)(a)) && std::isnan((double)(b))))) { fprintf(stderr, "MISMATCH %s: %.17g vs %.17g\n", what, (double)(a), (double)(b)); bad++; } } while (0)
    SAME("samples", m1, m2);
    if (m1 == m2 && memcmp(out1.data(), out2.data(), (size_t)m1 * 2) != 0) { fprintf(stderr, "MISMATCH: s16 output differs\n"); bad++; }
    SAME("input_i", one.input.input_i, ms.input_i); SAME("floor", one.input.floor, ms.floor);
    SAME("gate threshold", one.effective.gate_threshold, eff.gate_threshold); SAME("comp threshold", one.effective.comp_threshold_db, eff.comp_threshold_db);
    SAME("afftdn nf", one.effective.afftdn_noise_floor, eff.afftdn_noise_floor); SAME("deesser", one.effective.deess_intensity, eff.deess_intensity);
    SAME("pass2 I", one.filtered.r128.integrated, a2.r128.integrated); SAME("pass2 TP", one.filtered.r128.true_peak, a2.r128.true_peak);
    SAME("pass3 I", one.measure.input_i, m3.input_i); SAME("pass3 TP", one.measure.input_tp, m3.input_tp);
    SAME("limiter needed", one.limiter.needed, ld.needed); SAME("offset", one.offset, effI - m3.input_i);
    SAME("final I", one.final_.r128.integrated, a4.r128.integrated); SAME("final TP", one.final_.r128.true_peak, a4.r128.true_peak);
    SAME("loudnorm out I", one.loudnorm.output_i, ln.output_i);
    if (rdu[0] > 0) { SAME("room tone rms (2)", one.filtered_room_tone.rms_level, r2[0].rms_level); SAME("room tone rms (4)", one.final_room_tone.rms_level, r4[0].rms_level); }
    if (rdu[1] > 0) { SAME("speech rms (2)", one.filtered_speech.rms_level, r2[1].rms_level); SAME("speech centroid (4)", one.final_speech.spectral.centroid, r4[1].spectral.centroid); }
    if (strcmp(one.pass4_spec, spec) != 0) { fprintf(stderr, "MISMATCH pass4 spec:\n %s\n %s\n", one.pass4_spec, spec); bad++; }
    printf("granular walk vs jt_process_audio: %s (%lld samples, final I %.3f LUFS)\n", bad ? "DIFFERENT" : "identical", (long long)m2, a4.r128.integrated);
    return bad ? 1 : 0;
}
