"""End-to-end parity: jt_process_audio (four passes on the GPU, C++ host logic between them) against the same
pipeline composed from the CPU oracle with the SAME effective parameters, plus the north_star landing targets."""
import ctypes as C
import numpy as np
import pytest

from jivetalking_amd import synth, hostlogic as H, _lib as L
from conftest import options

pytestmark = pytest.mark.gpu
SR = 48000


def oracle_pass2(orc, x, fp, sr):
    y = x
    if fp.hp_enabled:
        y = orc.biquad_f32(y, 0, fp.hp_freq, sr, fp.hp_q)
    if fp.lp_enabled:
        y = orc.biquad_f32(y, 1, fp.lp_freq, sr, fp.lp_q)
    if fp.nlm_enabled:
        y = orc.anlmdn(y, sr, fp.nlm_strength, fp.nlm_patch_s, fp.nlm_research_s, fp.nlm_smooth)
    if fp.fft_enabled:
        bn = list(fp.fft_band_noise) if fp.fft_custom else None
        y = orc.afftdn(y, sr, fp.fft_nr, fp.fft_nf if fp.fft_nf < 0 else -50.0, bn)
    yd = y.astype(np.float64)
    if fp.gate_enabled:
        yd = orc.agate(yd, sr, fp.gate_threshold, fp.gate_ratio, fp.gate_attack_ms, fp.gate_release_ms, fp.gate_range, fp.gate_knee, fp.gate_makeup)
    if fp.comp_enabled:
        yd = orc.acompressor(yd, sr, fp.comp_threshold, fp.comp_ratio, fp.comp_attack_ms, fp.comp_release_ms, fp.comp_makeup, fp.comp_knee, fp.comp_mix)
    if fp.deess_enabled:
        yd = orc.deesser(yd, sr, fp.deess_i, fp.deess_m, fp.deess_f)
    yf = yd.astype(np.float32)          # dbl -> flt before aspectralstats
    # (a 44.1 kHz source: libswresample sets up no resampler for equal rates, aresample=44100 only converts the format)
    s16 = orc.f64_to_s16(yf.astype(np.float64) if sr == 44100 else orc.swr_f64(yf.astype(np.float64), sr, 44100, True))
    return yf, s16


@pytest.fixture(scope="module")
def processed(engine):
    x = synth.speech_like(45.0, SR, seed=21)
    engine.upload_pcm(x, SR, 1)
    res = H.process_audio(engine)
    return x, res, engine.download_s16(2), engine.download_s16(4)


def test_pipeline_lands_on_target(processed, oracle):
    x, res, p2, p4 = processed
    # measured independently by the oracle on the delivered s16 PCM
    e = oracle.ebur128(p4.astype(np.float64) / 32768.0, 44100, True, True)
    assert abs(e["integrated"] - (-16.0)) <= 0.1, e["integrated"]          # north_star: -16.0 +/- 0.1 LUFS
    assert 20 * np.log10(e["true_peak"]) <= -1.0                            # <= -1.0 dBTP
    assert abs(res.output_lufs - e["integrated"]) < 0.01
    assert res.within_target == 1 and res.loudnorm.normalization_type_dynamic == 0
    assert p4.size == p2.size == int(np.ceil(x.size * 147 / 160))


def test_pass1_measurements_match_oracle(processed, oracle):
    x, res, _, _ = processed
    e = oracle.ebur128(x.astype(np.float64), SR, True, True)
    assert abs(res.input.input_i - e["integrated"]) < 0.002                 # %.3f-quantised on the host side
    assert abs(res.input.input_lra - e["lra"]) < 0.011
    assert abs(res.input.input_tp - 20 * np.log10(e["true_peak"])) < 0.05   # '%.3f' linear quantisation of the metadata
    a = oracle.astats(x.astype(np.float64), SR)
    assert abs(res.input.dynamics.rms_level - a["rms_level_db"]) < 1e-5
    assert abs(res.input.dynamics.peak_level - a["peak_level_db"]) < 1e-5


def test_pass2_output_matches_oracle_chain(processed, oracle):
    x, res, p2, _ = processed
    fp = L.FilterParams()
    H.lib().jt_host_filter_params(C.byref(res.effective), C.byref(fp))
    _, ref = oracle_pass2(oracle, x, fp, SR)
    assert ref.size == p2.size
    d = np.abs(ref.astype(np.int32) - p2.astype(np.int32))
    # afftdn (f32 FFT schedule) dominates the difference: <= 3 LSB of s16 anywhere, < 0.3 LSB on average
    assert d.max() <= 3, d.max()
    assert d.mean() < 0.3


def test_pass3_pass4_match_oracle(processed, oracle):
    x, res, p2, p4 = processed
    # Pass 3 (no limiter prefix expected on this material)
    assert res.limiter.needed == 0
    up = oracle.swr_f32(p2.astype(np.float32) / 32768.0, 44100, 192000, True).astype(np.float64)
    m = oracle.loudnorm_measure(up, 192000, True)
    assert abs(res.measure.input_i - m["input_i"]) <= 0.011                 # both rounded to the JSON's %.2f
    assert abs(res.measure.input_tp - m["input_tp"]) <= 0.011
    # Pass 4: linear gain -> adeclick (t=1.7 w=55 o=50 m=s) -> brickwall -> (flt) -> s16
    gain = 10 ** ((res.effective_target_i - res.measure.input_i) / 20.0)
    y, repaired = oracle.adeclick(p2.astype(np.float64) / 32768.0 * gain, 44100, 1.7, 55.0, 50.0, method="s", return_count=True)
    assert repaired > 0                                                     # the repair stage is active on this material
    z = oracle.alimiter(y, 44100, 0.803526, 1.0, 50.0)
    ref = oracle.f64_to_s16(z.astype(np.float32).astype(np.float64))
    # the limiter is the oracle's state machine lane for lane; adeclick's default kernel relaxes the summation order (1e-12 on the
    # f64 samples, tests/test_gpu_ops.py), which can only show at an exact .5 rounding tie of the s16 conversion
    d = np.abs(ref.astype(np.int32) - p4.astype(np.int32))
    assert d.max() <= 1 and np.count_nonzero(d) <= 2, (d.max(), np.count_nonzero(d))


def test_pass4_with_adeclick_method_omitted_runs_overlap_add(engine, oracle):
    """AdeclickConfig.Method == "" (filters.go:958-960 appends m= only when set): af_adeclick.c then runs its default, overlap-add.  The
    whole four-pass job with that setting against the oracle's Pass-4 chain composed with method 'a'."""
    x = synth.speech_like(20.0, SR, seed=21)
    engine.upload_pcm(x, SR, 1)
    base = H.default_config(); base.adeclick_method_s = 0
    res = H.process_audio(engine, base)
    spec = res.pass4_spec.decode()
    assert "adeclick=t=1.7:w=55:o=50," in spec and ":m=s" not in spec
    p2, p4 = engine.download_s16(2), engine.download_s16(4)
    assert res.limiter.needed == 0 and res.loudnorm.normalization_type_dynamic == 0
    gain = 10 ** ((res.effective_target_i - res.measure.input_i) / 20.0)
    y, repaired = oracle.adeclick(p2.astype(np.float64) / 32768.0 * gain, 44100, 1.7, 55.0, 50.0, method="a", return_count=True)
    assert repaired > 0
    ref = oracle.f64_to_s16(oracle.alimiter(y, 44100, 0.803526, 1.0, 50.0).astype(np.float32).astype(np.float64))
    assert np.array_equal(ref, p4)                                          # sequential-order kernel + ordered overlap-add: bit-exact
    engine.upload_pcm(x, SR, 1)
    assert not np.array_equal(p4, (H.process_audio(engine), engine.download_s16(4))[1])     # (m=s delivers different samples)


def test_region_samples_and_specs_present(processed):
    _, res, _, _ = processed
    assert res.pass2_spec.decode().startswith("aformat=channel_layouts=mono,highpass=f=80")
    assert "loudnorm=I=-16.00" in res.pass4_spec.decode()
    if res.input.has_noise_profile:
        assert res.filtered_room_tone.frames > 0 and res.final_room_tone.frames > 0
        # noise reduction + gate must not raise the room tone
        assert res.filtered_room_tone.rms_level <= res.input.room_tone_sample.rms_level + 1.0


def test_analyse_only_matches_process_pass1(engine):
    x = synth.speech_like(20.0, SR, seed=22)
    engine.upload_pcm(x, SR, 1)
    a = H.process_audio(engine, analyse_only=True)
    b = H.process_audio(engine)
    assert a.input.input_i == b.input.input_i and a.input.floor == b.input.floor
    assert H.filter_spec(a.effective, 2) == H.filter_spec(b.effective, 2)


def test_stereo_downmix_and_silence_guard(engine, oracle):
    x = synth.speech_like(15.0, SR, seed=23)
    st = np.stack([x, 0.5 * x], axis=1).reshape(-1)
    engine.upload_pcm(st, SR, 2)
    r = H.process_audio(engine)
    assert abs(r.output_lufs + 16.0) <= 0.1                                  # north_star: -16.0 +/- 0.1 LUFS
    # Pass 1 sees the oracle's statement of swresample's float rematrix (1/sqrt2 each, not normalised)
    e = oracle.ebur128(oracle.downmix_stereo(st, 0).astype(np.float64), SR, True, True)
    assert abs(r.input.input_i - e["integrated"]) < 0.002
    z = np.zeros(SR * 12, np.float32)
    engine.upload_pcm(z, SR, 1)
    with pytest.raises(L.JtError) as ei:
        H.process_audio(engine)
    assert ei.value.code in (L.JT_E_SILENT, L.JT_E_UNSUPPORTED)


# ---------------------------------------------------------------- BASELINE.json configs[4]: 96 kHz stereo input
def test_96k_stereo_downmix_and_resample_path(engine, oracle):
    """96 kHz stereo with L != R: rematrix downmix (float formats: 1/sqrt2 each), Pass-1/2 at 96 kHz (anlmdn K=576 S=192: six offsets per lane, afftdn 4096-point
    instance), 96 k -> 44.1 k polyphase (72 taps, 147 phases, step 320), true peak via 96 k -> 192 k; Pass 3/4 at 44.1 kHz."""
    sr = 96000
    a = synth.speech_like(40.0, sr, seed=31)
    b = np.roll(a, 37) * 0.8 + synth.speech_like(40.0, sr, seed=32) * 0.1
    st = np.empty(a.size * 2, np.float32); st[0::2] = a; st[1::2] = b
    engine.upload_pcm(st, sr, 2)
    res = H.process_audio(engine)
    p2, p4 = engine.download_s16(2), engine.download_s16(4)
    mono = oracle.downmix_stereo(st, 0)                                            # swresample rematrix, float path (1/sqrt2 each)
    # Pass 1 on the downmix
    e = oracle.ebur128(mono.astype(np.float64), sr, True, True)
    assert abs(res.input.input_i - e["integrated"]) < 0.002
    assert abs(res.input.input_tp - 20 * np.log10(e["true_peak"])) < 0.05
    # Pass 2 chain + 96k -> 44.1k
    fp = L.FilterParams()
    H.lib().jt_host_filter_params(C.byref(res.effective), C.byref(fp))
    _, ref2 = oracle_pass2(oracle, mono, fp, sr)
    assert ref2.size == p2.size == int(np.ceil(mono.size * 147 / 320))
    d = np.abs(ref2.astype(np.int32) - p2.astype(np.int32))
    assert d.max() <= 3 and d.mean() < 0.3
    # landing
    f = oracle.ebur128(p4.astype(np.float64) / 32768.0, 44100, True, True)
    assert abs(f["integrated"] + 16.0) <= 0.1 and 20 * np.log10(f["true_peak"]) <= -1.0


# ---------------------------------------------------------------- error behaviour and the smaller entry points
def test_error_codes_and_refusals():
    from jivetalking_amd import Engine
    from jivetalking_amd.engine import default_filter_params
    with Engine(0) as e:
        with pytest.raises(L.JtError) as ei:                        # nothing uploaded yet
            e.pass2(default_filter_params())
        assert ei.value.code == L.JT_E_STATE
        x = synth.speech_like(8.0, SR, seed=41)
        e.upload_pcm(x, SR, 1)
        p = default_filter_params()
        with pytest.raises(L.JtError) as ei:                        # Pass 3 before any Pass-2 output exists
            e.pass3()
        assert ei.value.code == L.JT_E_STATE
        p.lp_freq = 30000.0                                          # a corner above Nyquist: af_biquads does not configure (EINVAL)
        with pytest.raises(L.JtError) as ei:
            e.pass2(p)
        assert ei.value.code == L.JT_E_INVAL
        # the context stays usable after a refused call; the reference's default chain (afftdn tn=1, filters_test.go:298-311) runs
        p.lp_freq = 20500.0
        p.fft_enabled = 1; p.fft_track_noise = 1
        a = e.pass2(p)
        assert np.isfinite(a["r128"]["integrated"])


def test_region_pair_equals_two_single_calls(engine, oracle):
    x = synth.speech_like(25.0, SR, seed=42)
    engine.upload_pcm(x, SR, 1)
    H.process_audio(engine)
    p2 = engine.download_s16(2)
    # the region pair call must agree with two single-region calls bit for bit, and with the oracle on the same samples
    st = [3.0, 11.5]; du = [4.0, 6.25]
    pair = (L.RegionSample * 2)()
    rc = engine.lib.jt_region_measure_pair(engine.h, C.c_int(2), (C.c_double * 2)(*st), (C.c_double * 2)(*du), pair)
    assert rc == 0
    for k in range(2):
        one = engine.region_measure(2, st[k], du[k])
        assert one["rms_level"] == pair[k].rms_level and one["true_peak"] == pair[k].true_peak
        assert one["spectral"]["centroid"] == pair[k].spectral.centroid and one["frames"] == pair[k].frames
        # and with the oracle on the downloaded Pass-2 samples of that region
        s0 = int(round(st[k] * 44100)); n = int(round(du[k] * 44100))
        seg = p2[s0:s0 + n].astype(np.float64) / 32768.0
        a = oracle.astats(seg, 44100)
        assert abs(one["rms_level"] - a["rms_level_db"]) < 1e-6


def _same(a, b):
    return a == b or (a != a and b != b)


def test_announced_regions_equal_on_demand_measurement(engine):
    """jt_region_prefetch: Pass 2 / Pass 4 measure the announced regions in their own tails; the stored samples are the ones an
    on-demand jt_region_measure_pair produces (same kernels, same arithmetic), and other requests are still measured on demand."""
    x = synth.speech_like(30.0, SR, seed=45)
    engine.upload_pcm(x, SR, 1)
    res = H.process_audio(engine)                      # the orchestration announces its regions: these came from the pass tails
    m = res.input
    st = [0.0, 0.0]; du = [0.0, 0.0]
    if m.has_noise_profile and m.noise_profile.duration_ns > 0:
        st[0] = float("%f" % (m.noise_profile.start_ns / 1e9)); du[0] = float("%f" % (m.noise_profile.duration_ns / 1e9))
    if m.has_speech_profile and m.speech_profile.region.duration_ns > 0:
        st[1] = float("%f" % (m.speech_profile.region.start_ns / 1e9)); du[1] = float("%f" % (m.speech_profile.region.duration_ns / 1e9))
    assert du[0] > 0 or du[1] > 0
    stored = {2: (res.filtered_room_tone, res.filtered_speech), 4: (res.final_room_tone, res.final_speech)}
    for stage in (2, 4):
        # a request that differs in the last digit is measured on demand; the single-region call never uses the stored pair
        for k in range(2):
            if du[k] <= 0:
                continue
            one = engine.region_measure(stage, st[k], du[k])
            got = stored[stage][k]
            assert one["frames"] == got.frames and one["frames"] > 0
            for f in ("rms_level", "peak_level", "crest_factor", "momentary", "shortterm", "true_peak", "sample_peak"):
                assert _same(one[f], getattr(got, f)), (stage, k, f)
            for f in L.SPECTRAL_KEYS:
                assert _same(one["spectral"][f], getattr(got.spectral, f)), (stage, k, f)
    # announced, then asked for something else: measured on demand, and the announcement does not leak into a later pass
    p2_before = engine.region_measure_pair(2, [1.0, 5.0], [2.0, 3.0])
    engine.region_prefetch(2, [1.0, 5.0], [2.0, 3.0])
    engine.region_prefetch(2, [1.5, 5.0], [2.0, 3.0])              # the last announcement wins
    res2 = H.process_audio(engine)                                 # (announces its own regions again, replacing ours)
    assert res2.filtered_speech.rms_level == res.filtered_speech.rms_level
    p2_after = engine.region_measure_pair(2, [1.0, 5.0], [2.0, 3.0])
    for a, b in zip(p2_before, p2_after):
        assert all(_same(a[f], b[f]) for f in ("rms_level", "peak_level", "momentary", "true_peak", "frames"))
        assert all(_same(a["spectral"][f], b["spectral"][f]) for f in L.SPECTRAL_KEYS)


def test_announced_regions_without_a_rate_change_and_empty_regions(engine):
    """out_rate == input rate (no resampler: the regions are measured behind the output stage), a region that atrim leaves empty
    (the pass must not fail; the on-demand call still reports it), and an announcement for a stage that is not run."""
    from jivetalking_amd.engine import default_filter_params
    x = synth.speech_like(12.0, SR, seed=46)
    engine.upload_pcm(x, SR, 1)
    p = default_filter_params(); p.out_rate = SR
    st = [2.0, 6.5]; du = [1.5, 3.0]
    engine.region_prefetch(2, st, du)
    engine.pass2(p)
    pair = engine.region_measure_pair(2, st, du)
    for k in range(2):
        one = engine.region_measure(2, st[k], du[k])
        assert one["frames"] == pair[k]["frames"] > 0
        assert all(_same(one[f], pair[k][f]) for f in ("rms_level", "peak_level", "crest_factor", "momentary", "true_peak", "sample_peak"))
        assert all(_same(one["spectral"][f], pair[k]["spectral"][f]) for f in L.SPECTRAL_KEYS)
    # a region past the end of the output: Pass 2 runs, nothing is stored, the explicit request fails as it always did
    engine.region_prefetch(2, [1.0, 500.0], [1.0, 2.0])
    engine.pass2(p)
    with pytest.raises(L.JtError) as ei:
        engine.region_measure_pair(2, [1.0, 500.0], [1.0, 2.0])
    assert ei.value.code == L.JT_E_INVAL
    ok = engine.region_measure_pair(2, [1.0, 0.0], [1.0, 0.0])          # the second region skipped (dur <= 0)
    assert ok[0]["frames"] > 0 and ok[1]["frames"] == 0
    with pytest.raises(L.JtError):
        engine.region_prefetch(3, st, du)                                # only stages 2 and 4 have outputs


def test_early_pass3_measurement_equals_the_explicit_one(engine):
    """Pass 2 queues Pass 3's no-prefix measurement of its output on a stream of its own; jt_pass3 collects it.  Same kernels on
    the same samples: the statistics must equal an explicit measurement bit for bit, a plan WITH a limiter prefix must not use
    it, and replacing the Pass-2 output (jt_upload_s16) must discard it."""
    from jivetalking_amd.engine import default_filter_params
    x = synth.speech_like(16.0, SR, seed=47)
    engine.upload_pcm(x, SR, 1)
    p = default_filter_params()
    keys = ("input_i", "input_tp", "input_lra", "input_thresh")
    engine.pass2(p)
    early = engine.pass3()                                           # collected from the job Pass 2 started
    again = engine.pass3()                                           # nothing pending any more: measured on demand
    with options(engine, no_early_pass3=True):
        engine.pass2(p)
        explicit = engine.pass3()
    for k in keys:
        assert early[k] == explicit[k] == again[k], k
    # a plan with a limiter prefix measures the limited stream, with or without a pending early job
    lim = L.LimiterPlan(1, 6.0, 0.5)
    engine.pass2(p)
    with_pending = engine.pass3(lim)
    without = engine.pass3(lim)
    assert all(with_pending[k] == without[k] for k in keys) and with_pending["input_i"] != early["input_i"]
    # the Pass-2 output replaced after Pass 2: the early job belongs to the old samples
    s16 = engine.download_s16(2)
    engine.pass2(p)
    engine.upload_s16((s16 // 2).astype(np.int16), 44100)
    halved = engine.pass3()
    ref = engine.op_loudnorm_measure_s16((s16 // 2).astype(np.int16), 44100)
    assert all(halved[k] == ref[k] for k in keys) and halved["input_i"] < early["input_i"] - 5.0


def test_pass2_head_prefetch_changes_nothing_but_the_schedule(engine):
    """jt_pass2_prefetch runs the biquad cascade and anlmdn early, on another stream.  Pass 2 must produce the same samples and the
    same analysis whether it continues from that head, finds it stale (different parameters) or never had one; calls that use the
    work buffers in between retire it."""
    from jivetalking_amd.engine import default_filter_params
    x = synth.speech_like(14.0, SR, seed=48)
    engine.upload_pcm(x, SR, 1)
    p = default_filter_params()
    a0 = engine.pass2(p); o0 = engine.download_s16(2).copy()                  # plain
    engine.pass2_prefetch(p)
    a1 = engine.pass2(p); o1 = engine.download_s16(2).copy()                  # continues from the head
    assert np.array_equal(o0, o1) and a0["r128"]["integrated"] == a1["r128"]["integrated"] and a0["astats"]["rms_level"] == a1["astats"]["rms_level"]
    q = default_filter_params(); q.hp_freq = 120.0
    b0 = engine.pass2(q); r0 = engine.download_s16(2).copy()
    engine.pass2_prefetch(p)                                                  # head for 80 Hz, Pass 2 asks for 120 Hz: discarded
    b1 = engine.pass2(q); r1 = engine.download_s16(2).copy()
    assert np.array_equal(r0, r1) and not np.array_equal(r0, o0) and b0["r128"]["integrated"] == b1["r128"]["integrated"]
    engine.pass2_prefetch(p)
    y = engine.op_anlmdn(x[:SR], SR)                                          # an operator call in between retires the head
    engine.pass1(x.size)                                                      # Pass 1 and the band measurements leave a head alone
    engine.pass2_prefetch(p)
    engine.pass1(x.size); engine.band_rms(1.0, 2.0, [1000.0], [3000.0])
    a2 = engine.pass2(p); o2 = engine.download_s16(2)
    assert np.array_equal(o0, o2) and y.size == SR and a2["r128"]["integrated"] == a0["r128"]["integrated"]
    engine.pass2_prefetch_after_pass1(p)                                      # announced: Pass 1 starts it behind its own kernels
    engine.pass1(x.size)
    a3 = engine.pass2(p); o3 = engine.download_s16(2)
    assert np.array_equal(o0, o3) and a3["r128"]["integrated"] == a0["r128"]["integrated"]
    engine.pass2_prefetch_after_pass1(p)                                      # announced but no Pass 1 follows: nothing started, nothing lost
    a4 = engine.pass2(p); o4 = engine.download_s16(2)
    assert np.array_equal(o0, o4) and a4["r128"]["integrated"] == a0["r128"]["integrated"]
    # the orchestration: with and without the early head (and the early Pass-3 measurement) the run is the same run
    r_on = H.process_audio(engine); out_on = engine.download_s16(4).copy()
    with options(engine, no_pass2_prefetch=True, no_early_pass3=True):
        r_off = H.process_audio(engine); out_off = engine.download_s16(4)
    assert np.array_equal(out_on, out_off) and r_on.output_lufs == r_off.output_lufs and r_on.pass2_spec == r_off.pass2_spec


def test_lds_streamed_followers_match_the_tile_staged_ones(engine_ab):
    """k_follow_states_lds (direct-to-LDS loads, longer chunks) against k_follow_states (option follow_tiles, A/B build).  Both restart the
    follower behind an 18-time-constant halo, at different places: the Pass-2 outputs may differ by the halo's 1.5e-8 relative
    state error, i.e. by nothing once rounded to s16 -- at most a stray LSB."""
    from jivetalking_amd.engine import default_filter_params
    engine = engine_ab
    x = synth.speech_like(40.0, SR, seed=49)
    engine.upload_pcm(x, SR, 1)
    p = default_filter_params()
    a = engine.pass2(p); o_lds = engine.download_s16(2).copy()
    with options(engine, follow_tiles=True):
        b = engine.pass2(p); o_tile = engine.download_s16(2)
    d = np.abs(o_lds.astype(np.int32) - o_tile.astype(np.int32))
    assert d.max() <= 1 and np.count_nonzero(d) <= o_lds.size // 10000
    assert abs(a["r128"]["integrated"] - b["r128"]["integrated"]) < 1e-6
    # lengths that are not a multiple of the 16-byte load group (the last group of the signal is read whole, into the buffers'
    # slack), shorter than one chunk, and shorter than the halo
    for n in (x.size - 3, 48000 * 3 + 1, 30011):
        engine.upload_pcm(x[:n], SR, 1)
        engine.pass2(p); u = engine.download_s16(2).copy()
        with options(engine, follow_tiles=True):
            engine.pass2(p); v = engine.download_s16(2)
        dd = np.abs(u.astype(np.int32) - v.astype(np.int32))
        assert u.size == v.size and dd.max() <= 1 and np.count_nonzero(dd) <= max(2, u.size // 10000), n


def test_two_runs_are_bit_identical(engine):
    x = synth.speech_like(20.0, SR, seed=43)
    engine.upload_pcm(x, SR, 1)
    r1 = H.process_audio(engine); o1 = engine.download_s16(4).copy()
    r2 = H.process_audio(engine); o2 = engine.download_s16(4)
    assert np.array_equal(o1, o2) and r1.output_lufs == r2.output_lufs and r1.pass2_spec == r2.pass2_spec


def test_progress_events_follow_the_reference_lifecycle(engine):
    """processor.go:80-158 + normalise.go:737-772: start/end per pass, measurements on the Pass-1 end and Pass-2 events,
    config + diagnostics on the Pass-2 start only, limiter snapshot on the Pass-4 start only."""
    x = synth.speech_like(20.0, SR, seed=44)
    engine.upload_pcm(x, SR, 1)
    seen = []

    def on_update(u):
        seen.append((u.pass_, u.pass_name.decode(), u.progress, bool(u.measurements), bool(u.config), bool(u.diag), u.has_limiter,
                     u.duration, u.measurements.contents.input_i if u.measurements else None))
    res = H.process_audio_with_progress(engine, on_update)
    assert [(p, n, pr) for p, n, pr, *_ in seen] == [
        (1, "Analysing", 0.0), (1, "Analysing", 1.0), (2, "Processing", 0.0), (2, "Processing", 1.0),
        (3, "Measuring", 0.0), (3, "Measuring", 1.0), (4, "Normalising", 0.0), (4, "Normalising", 1.0)]
    assert [s[3] for s in seen] == [False, True, True, True, False, False, False, False]       # measurements
    assert [s[4] and s[5] for s in seen] == [False, False, True, False, False, False, False, False]   # config + diagnostics
    assert [s[6] for s in seen] == [0, 0, 0, 0, 0, 0, 1, 0]                                     # limiter snapshot
    assert seen[1][8] == res.input.input_i and abs(seen[1][7] - 20.0) < 1e-9


# ---------------------------------------------------------------- the less-travelled branches of the path, end to end
def _oracle_pass34(oracle, p2, res, rate=44100):
    """Oracle restatement of Pass 3 / Pass 4 given the Pass-2 s16 output and the host decisions in `res`."""
    x = p2.astype(np.float64) / 32768.0
    lim = res.limiter
    pre = 10 ** (lim.pre_gain_db / 20.0) if (lim.needed and lim.pre_gain_db > 0) else 1.0
    if lim.needed:
        limit = float("%.6f" % (10 ** (lim.ceiling_db / 20.0)))
        y = oracle.alimiter(x * pre, rate, limit, 5.0, 100.0)
        m = oracle.loudnorm_measure(oracle.swr_f64(y, rate, 192000, True), 192000, True)
    else:
        y = x
        m = oracle.loudnorm_measure(oracle.swr_f32(p2.astype(np.float32) / 32768.0, rate, 192000, True).astype(np.float64), 192000, True)
    gain = 10 ** ((res.effective_target_i - res.measure.input_i) / 20.0)
    z = oracle.adeclick(y * gain, rate, 1.7, 55.0, 50.0, method="s")
    z = oracle.alimiter(z, rate, 0.803526, 1.0, 50.0)
    return m, oracle.f64_to_s16(z.astype(np.float32).astype(np.float64))


def test_limiter_prefix_path_loud_peaky_input(engine, oracle):
    """Sparse strong peaks over quiet speech: the projected true peak exceeds the target, so planLimiterForLoudnorm arms the
    alimiter prefix in Pass 3 and Pass 4 (normalise.go:373-561) and the DBLP 192 kHz measurement path is taken."""
    x = synth.speech_like(30.0, SR, seed=51)
    x[::24000] += 0.4 * np.sign(x[::24000] + 1e-9)                  # -8 dBFS ticks over -27 LUFS speech
    x = np.clip(x, -0.98, 0.98).astype(np.float32)
    engine.upload_pcm(x, SR, 1)
    res = H.process_audio(engine)
    p2, p4 = engine.download_s16(2), engine.download_s16(4)
    assert res.limiter.needed == 1
    m, ref = _oracle_pass34(oracle, p2, res)
    assert abs(res.measure.input_i - m["input_i"]) <= 0.011 and abs(res.measure.input_tp - m["input_tp"]) <= 0.011
    d = np.abs(ref.astype(np.int32) - p4.astype(np.int32)); assert d.max() <= 1 and np.count_nonzero(d) <= 2      # (adeclick fast kernel: .5 ties only)
    e = oracle.ebur128(p4.astype(np.float64) / 32768.0, 44100, True, True)
    assert 20 * np.log10(e["true_peak"]) <= -1.0


def test_44k1_input_no_rate_change(engine, oracle):
    """44.1 kHz input: anlmdn K = 265, S = 88 (the hop-pair kernel with dummy end offsets and a short last block), afftdn with A = 551, and no resampler in the
    output format conversion (aformat only converts dbl -> s16)."""
    sr = 44100
    x = synth.speech_like(30.0, sr, seed=52)
    engine.upload_pcm(x, sr, 1)
    res = H.process_audio(engine)
    p2, p4 = engine.download_s16(2), engine.download_s16(4)
    fp = L.FilterParams()
    H.lib().jt_host_filter_params(C.byref(res.effective), C.byref(fp))
    y = x
    if fp.hp_enabled: y = oracle.biquad_f32(y, 0, fp.hp_freq, sr, fp.hp_q)
    if fp.lp_enabled: y = oracle.biquad_f32(y, 1, fp.lp_freq, sr, fp.lp_q)
    if fp.nlm_enabled: y = oracle.anlmdn(y, sr, fp.nlm_strength, fp.nlm_patch_s, fp.nlm_research_s, fp.nlm_smooth)
    if fp.fft_enabled: y = oracle.afftdn(y, sr, fp.fft_nr, fp.fft_nf if fp.fft_nf < 0 else -50.0, list(fp.fft_band_noise) if fp.fft_custom else None)
    yd = y.astype(np.float64)
    if fp.gate_enabled: yd = oracle.agate(yd, sr, fp.gate_threshold, fp.gate_ratio, fp.gate_attack_ms, fp.gate_release_ms, fp.gate_range, fp.gate_knee, fp.gate_makeup)
    if fp.comp_enabled: yd = oracle.acompressor(yd, sr, fp.comp_threshold, fp.comp_ratio, fp.comp_attack_ms, fp.comp_release_ms, fp.comp_makeup, fp.comp_knee, fp.comp_mix)
    ref2 = oracle.f64_to_s16(yd.astype(np.float32).astype(np.float64))
    assert ref2.size == p2.size == x.size
    d = np.abs(ref2.astype(np.int32) - p2.astype(np.int32))
    assert d.max() <= 3 and d.mean() < 0.3
    m, ref4 = _oracle_pass34(oracle, p2, res)
    d = np.abs(ref4.astype(np.int32) - p4.astype(np.int32)); assert d.max() <= 1 and np.count_nonzero(d) <= 2


def test_deesser_enabled_and_short_clip(engine, oracle):
    """De-esser switched on by the caller (off in the default adaptive outcome for this material), and a clip too short for any
    room-tone / speech election (the no-profile fallbacks of AdaptConfig)."""
    base = H.default_config()
    base.deess_enabled = 1; base.deess_intensity = 0.6
    x = synth.speech_like(20.0, SR, seed=53)
    engine.upload_pcm(x, SR, 1)
    res = H.process_audio(engine, base)
    p2 = engine.download_s16(2)
    fp = L.FilterParams()
    H.lib().jt_host_filter_params(C.byref(res.effective), C.byref(fp))
    if fp.deess_enabled and fp.deess_i > 0:
        _, ref = oracle_pass2(oracle, x, fp, SR)
        d = np.abs(ref.astype(np.int32) - p2.astype(np.int32))
        assert d.max() <= 3 and d.mean() < 0.3
    y = synth.speech_like(4.0, SR, seed=54)
    engine.upload_pcm(y, SR, 1)
    r2 = H.process_audio(engine)
    assert r2.input.has_speech_profile == 0 or r2.input.speech_profile.region.duration_ns > 0
    assert engine.download_s16(4).size == int(np.ceil(y.size * 147 / 160))


def test_full_size_60_minute_file_properties(oracle):
    """BASELINE configs[1] at its real size (60 min, 172.8 M samples), through size-independent properties: the output lands on
    target when re-measured by the CPU oracle's ebur128 over the whole file; a second run gives the same bytes; the output
    survives GPU FLAC encode -> GPU FLAC decode bit for bit with the right STREAMINFO MD5; only samples adeclick flagged may
    differ between the limiter input and output path lengths (sample counts: M = ceil(N * 147 / 160))."""
    import hashlib
    from jivetalking_amd import Engine
    sr = 48000
    # the bench talker itself (aperiodic over the hour: 14 063 different 250 ms intervals for the VAD / election / Otsu logic), made on
    # the device by a child process (a tiled minute exercised that logic on 60 near-copies)
    from conftest import bench_talker
    x = np.asarray(bench_talker(3600.0, sr, 1000, 40.0), np.float32)
    n = x.size
    e = Engine(0)
    try:
        e.upload_pcm(x, sr, 1)
        r1 = H.process_audio(e)
        out1 = e.download_s16(4)
        assert out1.size == -(-n * 147 // 160)
        # the host logic at full size: 14 063 intervals (an interval closes on the first 4096-sample decoder frame that brings it to 250 ms:
        # three frames, 256 ms, analyser_metrics.go:165-428), a speech region and a room-tone region elected, the limiter-prefix branch
        assert H.lib(e).jt_host_last_intervals(e.h, None, C.c_int64(0)) == -(-n // (3 * 4096))
        assert r1.input.has_noise_profile == 1 and r1.input.n_candidates >= 1 and r1.limiter.needed == 1 and r1.has_region_samples == 1
        r2 = H.process_audio(e)
        out2 = e.download_s16(4)
        assert np.array_equal(out1, out2) and r1.output_lufs == r2.output_lufs
        # independent re-measurement of the whole file on the CPU
        chk = oracle.ebur128(out1.astype(np.float64) / 32768.0, 44100, True, True)
        assert abs(chk["integrated"] - (-16.0)) <= 0.1, chk["integrated"]
        assert abs(chk["integrated"] - r1.output_lufs) < 0.01
        assert 20 * np.log10(chk["true_peak"]) <= -1.0
        # file legs at full size
        img, info = e.flac_encode(4, md5=True, return_info=True)
        assert info["frames"] == (out1.size + 4095) // 4096 and info["md5"] == hashlib.md5(out1.tobytes()).hexdigest()
        back, _, meta = e.op_decode_audio(img)
        assert np.array_equal(back[:, 0], out1.astype(np.int32)) and meta["flac_frames"] == info["frames"]
    finally:
        e.close()


def test_config0_sixty_second_flac_file_against_the_oracle_chain(engine, oracle, tmp_path):
    """BASELINE configs[0] end to end at its own size: a 60 s 48 kHz mono 16-bit FLAC goes in as a FILE (jt_process_file: decode on the
    GPU, four passes, encode, atomic publish) and the delivered .flac is decoded by the RFC 9639 oracle decoder and compared with the
    pipeline composed from the CPU oracle with the same effective parameters."""
    from jivetalking_amd.engine import Engine
    x16 = np.rint(np.asarray(synth.speech_like(60.0, SR, seed=77), np.float64) * 32767).astype(np.int16)
    src = tmp_path / "episode.flac"
    src.write_bytes(bytes(engine.op_flac_encode(x16, SR)))
    res, out_path, _ = H.process_file(engine, str(src))
    assert out_path.endswith("-processed.flac") and res.loudnorm.normalization_type_dynamic == 0
    rc, dec, info = oracle.flac_decode(open(out_path, "rb").read())
    assert rc == 0 and bytes(info.md5_stored) == bytes(info.md5_decoded)
    p4 = dec[:, 0].astype(np.int16); p2 = engine.download_s16(2)
    assert p4.size == p2.size == int(np.ceil(x16.size * 147 / 160)) and np.array_equal(p4, engine.download_s16(4))
    # north_star: lands on -16 LUFS +/- 0.1, <= -1 dBTP, measured independently on the delivered PCM
    e = oracle.ebur128(p4.astype(np.float64) / 32768.0, 44100, True, True)
    assert abs(e["integrated"] + 16.0) <= 0.1 and 20 * np.log10(e["true_peak"]) <= -1.0
    # Pass 2 against the oracle chain on the decoded input (s16 / 32768 exactly, as libswresample converts it)
    x = x16.astype(np.float32) / np.float32(32768.0)
    fp = L.FilterParams()
    H.lib().jt_host_filter_params(C.byref(res.effective), C.byref(fp))
    _, ref2 = oracle_pass2(oracle, x, fp, SR)
    d2 = np.abs(ref2.astype(np.int32) - p2.astype(np.int32))
    assert ref2.size == p2.size and d2.max() <= 3 and d2.mean() < 0.3, (d2.max(), d2.mean())
    # Pass 3 and Pass 4 on the GPU's own Pass-2 output
    assert res.limiter.needed == 0
    up = oracle.swr_f32(p2.astype(np.float32) / 32768.0, 44100, 192000, True).astype(np.float64)
    m = oracle.loudnorm_measure(up, 192000, True)
    assert abs(res.measure.input_i - m["input_i"]) <= 0.011 and abs(res.measure.input_lra - m["input_lra"]) <= 0.011
    gain = 10 ** ((res.effective_target_i - res.measure.input_i) / 20.0)
    y = oracle.adeclick(p2.astype(np.float64) / 32768.0 * gain, 44100, 1.7, 55.0, 50.0, method="s")
    ref4 = oracle.f64_to_s16(oracle.alimiter(y, 44100, 0.803526, 1.0, 50.0).astype(np.float32).astype(np.float64))
    d4 = np.abs(ref4.astype(np.int32) - p4.astype(np.int32))
    assert d4.max() <= 1 and np.count_nonzero(d4) <= 4, (d4.max(), np.count_nonzero(d4))
