"""Host control logic (C++ mirror of internal/processor) against the reference's own golden tables.

Every expected value / string here is data the reference's tests assert (tests/golden/*.json carry the
file:line cites); the table tests mirror analyser_vad_test.go / adaptive_test.go / normalise_test.go /
filters_test.go one function at a time.  CPU only: the host logic never touches the GPU.
"""
import ctypes as C
import json
import os

import pytest

from jivetalking_amd import hostlogic as H
from jivetalking_amd import _lib as L

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HOP = 250_000_000


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(L.LIB_PATH):
        g.build()
    return H.lib()


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


# ------------------------------------------------------------------ ABI self-check
def test_host_struct_sizes_match_c(lib):
    for which, cls in H.SIZEOF_IDS.items():
        assert lib.jt_host_sizeof(which) == C.sizeof(cls), (which, cls.__name__)


def test_host_symbols_exported(lib):
    for s in H.HOST_SYMBOLS:
        assert hasattr(lib, s), s


# ------------------------------------------------------------------ filter-spec strings
def test_default_pass2_chain_string(lib):
    g = load("filter_specs.json")
    assert H.filter_spec(H.default_config(), 2) == g["default_pass2"]["want"]


def test_pass1_chain_string(lib):
    g = load("filter_specs.json")
    assert H.filter_spec(H.default_config(), 1) == g["pass1"]["want"]


def _cfg_from_json(j):
    c = H.HostConfig()
    for k, v in j.items():
        if k == "cite":
            continue
        if isinstance(v, dict):
            sub = getattr(c, k)
            for kk, vv in v.items():
                setattr(sub, kk, vv)
        else:
            setattr(c, k, v)
    return c


def _meas_from_json(j):
    m = H.Measurements()
    for k in ("input_i", "input_tp", "input_lra", "floor", "voiced_low_percentile", "noise_high_percentile", "gate_separation_db"):
        if k in j:
            setattr(m, k, j[k])
    for k, v in j.get("dynamics", {}).items():
        setattr(m.dynamics, k, v)
    if "noise_profile" in j:
        m.has_noise_profile = 1
        for k, v in j["noise_profile"].items():
            setattr(m.noise_profile, k, v)
    if "speech_profile" in j:
        m.has_speech_profile = 1
        for k, v in j["speech_profile"].items():
            setattr(m.speech_profile.sample, k, v)
    return m


def test_adapt_config_golden_chain_strings(lib):
    g = load("filter_specs.json")
    base = _cfg_from_json(g["test_base_config"])
    for case in g["adapt_cases"]:
        eff, _diag = H.adapt(base, _meas_from_json(case["measurements"]))
        assert H.filter_spec(eff, 2) == case["want"], case["name"]
        # the caller's seed config is never mutated (processor_test.go:442-466)
        assert base.bandlimit_lp.frequency == 16000.0 and base.afftdn_track_noise == 1


def test_pass3_prefix_and_pass4_chain_strings(lib):
    g = load("filter_specs.json")
    cfg = H.default_config()
    for case in g["pass34_cases"]:
        dec = H.LimiterDecision(); plan = L.LimiterPlan()
        lib.jt_host_plan_limiter(C.c_double(case["output_i"]), C.c_double(case["output_tp"]), C.byref(cfg), C.byref(dec), C.byref(plan))
        assert dec.pass3_prefix.decode() == case["want_pass3"], case["name"]
        ms = L.LoudnormStats()
        for k in ("input_i", "input_tp", "input_lra", "input_thresh"):
            setattr(ms, k, case["measurement"][k])
        buf = C.create_string_buffer(4096); ap = L.LoudnormApply()
        lib.jt_host_pass4_spec(C.byref(cfg), C.byref(ms), C.c_double(case["measurement"]["target_offset"]), C.byref(dec),
                               C.c_int(48000), None, buf, C.c_int(4096), C.byref(ap))
        assert buf.value.decode() == case["want_pass4"], case["name"]
        assert abs(ap.brickwall_limit - 0.803526) < 1e-12
        # numeric plan carries the string precision FFmpeg would have parsed
        if case["want_pass3"]:
            assert ("limit=%.6f" % plan.limit) in case["want_pass3"]


# ------------------------------------------------------------------ normalise.go scalar tables
def test_calculate_linear_mode_target_table(lib):
    t = load("normalise_tables.json")["linear_mode_target"]
    for c in t["cases"]:
        eff = C.c_double(); off = C.c_double(); lin = C.c_int()
        lib.jt_host_calculate_linear_mode_target(C.c_double(c["measured_i"]), C.c_double(c["measured_tp"]), C.c_double(c["desired_i"]),
                                                 C.c_double(c["target_tp"]), C.byref(eff), C.byref(off), C.byref(lin))
        assert abs(eff.value - c["want_effective_i"]) < 0.01
        assert abs(off.value - c["want_offset"]) < 0.01
        assert bool(lin.value) == c["want_linear"]


def test_calculate_limiter_ceiling_table(lib):
    t = load("normalise_tables.json")["limiter_ceiling"]
    for c in t["cases"]:
        ce = C.c_double(); nd = C.c_int(); cl = C.c_int()
        lib.jt_host_calculate_limiter_ceiling(C.c_double(c["measured_i"]), C.c_double(c["measured_tp"]), C.c_double(c["target_i"]),
                                              C.c_double(c["target_tp"]), C.byref(ce), C.byref(nd), C.byref(cl))
        assert abs(ce.value - c["want_ceiling"]) < 0.01
        assert bool(nd.value) == c["want_needed"] and bool(cl.value) == c["want_clamped"]


def test_calculate_pre_gain_table(lib):
    t = load("normalise_tables.json")["pre_gain"]
    for c in t["cases"]:
        pg = C.c_double(); rd = C.c_double()
        lib.jt_host_calculate_pre_gain(C.c_double(c["measured_i"]), C.c_double(c["target_i"]), C.c_double(c["target_tp"]), C.byref(pg), C.byref(rd))
        assert abs(pg.value - c["want_pre_gain"]) < 0.01 and abs(rd.value - c["want_rederived"]) < 0.01


def test_internal_tp_makes_linear_cap_inert(lib):
    t = load("normalise_tables.json")["internal_tp_cancellation"]
    for c in t["cases"]:
        itp = lib.jt_host_loudnorm_internal_target_tp(C.c_double(-16.0), C.c_double(c["measured_tp"]), C.c_double(c["measured_i"]))
        eff = C.c_double(); off = C.c_double(); lin = C.c_int()
        lib.jt_host_calculate_linear_mode_target(C.c_double(c["measured_i"]), C.c_double(c["measured_tp"]), C.c_double(-16.0), C.c_double(itp),
                                                 C.byref(eff), C.byref(off), C.byref(lin))
        assert lin.value == 1 and eff.value == -16.0


# ------------------------------------------------------------------ VAD table tests (analyser_vad_test.go)
def vad_interval(idx, level, **kw):
    d = dict(timestamp_ns=idx * HOP, rms_level=level, momentary_lufs=level, centroid=2000.0, entropy=0.40)
    d.update(kw)
    return d


def vad_speech(i): return vad_interval(i, -15)
def vad_quiet(i): return vad_interval(i, -60)
def vad_loud_non_speech(i): return vad_interval(i, -15, centroid=9000.0)
def vad_speech_rich(i, rms=-16.0): return vad_interval(i, rms, peak_level=rms + 12, kurtosis=6.0, rolloff=6000.0, flux=0.004, flatness=0.2)


def runs(lib, rows, split=-30.0, margin=3.0, tol=8):
    iv = H.make_intervals(rows)
    out = (H.Region * 64)()
    n = lib.jt_host_vad_speech_runs(iv, C.c_int64(len(rows)), C.c_double(split), C.c_double(margin), C.c_int(tol), out, C.c_int(64))
    return [(out[i].start_ns, out[i].end_ns) for i in range(n)]


def seq(*parts):
    rows, idx = [], 0
    for n, fn in parts:
        for _ in range(n):
            rows.append(fn(idx)); idx += 1
    return rows


def test_build_speech_runs_table(lib):
    # analyser_vad_test.go:560-690 (TestBuildSpeechRuns): split -30, margin 3, tol = 8 intervals, min run 40
    assert len(runs(lib, seq((50, vad_speech), (7, vad_quiet), (50, vad_speech)))) == 1        # short gap bridges
    assert len(runs(lib, seq((50, vad_speech), (13, vad_quiet), (50, vad_speech)))) == 2       # long gap splits
    assert len(runs(lib, seq((50, vad_speech), (3, lambda i: vad_interval(i, -31)), (50, vad_speech)))) == 1   # neutral zone held
    assert len(runs(lib, seq((50, vad_speech), (1, vad_loud_non_speech), (50, vad_speech)))) == 2   # loud-gap guard
    assert len(runs(lib, seq((50, vad_speech), (1, vad_quiet), (50, vad_speech)))) == 1
    assert len(runs(lib, seq((39, vad_speech), (6, vad_quiet)))) == 0                            # below minimum duration


def test_gap_tolerance_table(lib):
    # analyser_vad_test.go:692-731: interior gaps {4,6,12,30} -> nearest-rank p75 = 12 -> clamp [8,40] = 12
    flags = [1] * 5 + [0] * 4 + [1] * 5 + [0] * 6 + [1] * 5 + [0] * 12 + [1] * 5 + [0] * 30 + [1] * 5 + [0] * 20
    arr = (C.c_int * len(flags))(*flags)
    assert lib.jt_host_vad_gap_tolerance(arr, C.c_int64(len(flags))) == 12
    flags = [1, 1, 1, 0, 0]
    assert lib.jt_host_vad_gap_tolerance((C.c_int * 5)(*flags), C.c_int64(5)) == 8


def gate_stats(lib, rows, split, region=None):
    iv = H.make_intervals(rows)
    v = C.c_double(); n = C.c_double(); s = C.c_double()
    reg = None
    if region is not None:
        reg = C.byref(H.Region(region[0], region[1], region[1] - region[0]))
    lib.jt_host_vad_gate_stats(iv, C.c_int64(len(rows)), C.c_double(split), reg, C.byref(v), C.byref(n), C.byref(s))
    return v.value, n.value, s.value


def test_derive_gate_statistics_table(lib):
    # analyser_vad_test.go:950-1158
    rows = [vad_interval(i, -60 + i) for i in range(20)] + [vad_interval(20 + i, -25 + i) for i in range(21)]
    v, n, s = gate_stats(lib, rows, -30.0, (20 * HOP, 41 * HOP))
    assert abs(v + 23.0) < 1e-3 and abs(n + 42.0) < 1e-3 and abs(s - 19.0) < 1e-3
    rows = [vad_interval(i, -20 + i) for i in range(11)] + [vad_loud_non_speech(11 + i) for i in range(5)]
    v, n, s = gate_stats(lib, rows, -30.0, (0, 16 * HOP))
    assert abs(v + 19.0) < 1e-3                                  # veto failures excluded from the voiced set
    rows = [vad_interval(i, -50 + i) for i in range(11)]
    v, n, s = gate_stats(lib, rows, -45.0, (0, 11 * HOP))
    assert abs(v + 45.0) < 1e-3 and abs(n + 47.0) < 1e-3        # the split governs the partition
    rows = [vad_interval(i, -130) for i in range(10)] + [vad_interval(10 + i, -60 + i) for i in range(20)]
    v, n, s = gate_stats(lib, rows, -30.0, None)
    assert v == 0 and abs(n + 42.0) < 1e-3                       # floored excluded; nil region -> voiced 0


def test_detect_voice_activity_bimodal(lib):
    # analyser_vad_test.go:1160-1222 (TestDetectVoiceActivity): 60 intervals at -55, 80 speech-rich at -16, seed -70
    rows = [vad_interval(i, -55) for i in range(60)] + [vad_speech_rich(60 + i) for i in range(80)]
    iv = H.make_intervals(rows)
    m = H.Measurements()
    assert lib.jt_host_vad_detect(iv, C.c_int64(len(rows)), C.c_double(-70.0), C.byref(m)) == 0
    assert m.has_speech_profile == 1 and m.has_noise_profile == 1 and m.has_room_tone_sample == 1
    assert m.floor_source == 3                                   # "vad_percentile"
    assert -120 < m.floor < -16
    assert m.voiced_low_percentile != 0 and m.noise_high_percentile != 0 and m.gate_separation_db > 0
    v, n, s = gate_stats(lib, rows, m.vad_split, (m.speech_profile.region.start_ns, m.speech_profile.region.end_ns))
    assert (m.voiced_low_percentile, m.noise_high_percentile, m.gate_separation_db) == (v, n, s)
    assert m.voice_activated == 0 and m.floored_fraction == 0.0


def test_detect_voice_activity_no_profile(lib):
    # analyser_vad_test.go:1224-1242: flat low stream -> no profile, voiced percentile stays 0
    rows = [vad_interval(i, -55) for i in range(60)]
    m = H.Measurements()
    lib.jt_host_vad_detect(H.make_intervals(rows), C.c_int64(60), C.c_double(-70.0), C.byref(m))
    assert m.has_speech_profile == 0 and m.voiced_low_percentile == 0


def test_floored_fraction_and_voice_activated(lib):
    # analyser_vad_test.go:190-357,478-516: NaN and <= -115 count as floored; >= 0.20 flags voice-activated
    rows = [vad_interval(i, float("nan")) for i in range(10)] + [vad_interval(10 + i, -120) for i in range(10)] + \
           [vad_speech_rich(20 + i) for i in range(80)]
    iv = H.make_intervals(rows)
    assert abs(lib.jt_host_vad_floored_fraction(iv, C.c_int64(100)) - 0.20) < 1e-12
    m = H.Measurements()
    lib.jt_host_vad_detect(iv, C.c_int64(100), C.c_double(-115.0), C.byref(m))
    assert m.voice_activated == 1
    base = H.default_config()
    eff, diag = H.adapt(base, m)
    assert eff.afftdn_enabled == 0 and diag.afftdn_disabled_voice_activated == 1      # adaptive.go:133-140
    assert "afftdn" not in H.filter_spec(eff, 2)


def seed_interval(i, level, flux):
    # analyser_vad_test.go seedInterval: momentary/RMS level + spectral flux, veto-passing spectrum
    return vad_interval(i, level, flux=flux)


def noise_seed(lib, rows):
    nf = C.c_double(); th = C.c_double()
    ok = lib.jt_host_vad_noise_seed(H.make_intervals(rows), C.c_int64(len(rows)), C.byref(nf), C.byref(th))
    return ok, nf.value, th.value


def test_noise_seed_truncation_picks_lowest_level(lib):
    # analyser_vad_test.go:397-431: 25 tied score-1.0 intervals (-56..-80, descending) + 25 louder high-flux;
    # candidateCount = max(50/5, 8) = 10 -> seeded floor = -80 + 9
    rows = [seed_interval(i, -56 - i, 0.01) for i in range(25)] + [seed_interval(25 + i, -30 + i, 0.50) for i in range(25)]
    ok, nf, th = noise_seed(lib, rows)
    assert ok == 1 and abs(nf - (-71.0)) < 1e-3 and abs(th - (nf + 1.0)) < 1e-12


def test_noise_seed_tie_break_is_order_independent(lib):
    # analyser_vad_test.go:359-395
    import random
    rows = [seed_interval(i, -80 + i, 0.01) for i in range(25)] + [seed_interval(25 + i, -30 + i, 0.50) for i in range(25)]
    a = noise_seed(lib, rows)
    sh = rows[:]; random.Random(7).shuffle(sh)
    for i, r in enumerate(sh):
        r["timestamp_ns"] = i * HOP
    assert noise_seed(lib, sh) == a


def test_noise_seed_excludes_floored_and_handles_all_floored(lib):
    # analyser_vad_test.go:433-476
    rows = [seed_interval(i, -130, 0.01) for i in range(3)] + [seed_interval(3 + i, -70 + i, 0.01) for i in range(40)] + \
           [seed_interval(43 + i, -10 + i, 0.50) for i in range(10)]
    ok, nf, _ = noise_seed(lib, rows)
    assert ok == 1 and nf > -115.0
    ok, _, _ = noise_seed(lib, [seed_interval(i, -130, 0.01) for i in range(15)])
    assert ok == 0


# ------------------------------------------------------------------ adaptive rules (adaptive_test.go tables)
def test_deesser_intensity_mapping(lib):
    # adaptive_deesser.go:45-68 / adaptive_test.go:387: excess < -6 off; -6..-3 -> 0..0.6; -3..0 -> 0.6..0.85; >= 0 -> 0.85
    base = H.default_config()
    for excess, want in [(-8.0, 0.0), (-6.0, 0.0), (-4.5, 0.3), (-3.0, 0.6), (-1.5, 0.725), (0.0, 0.85), (3.0, 0.85)]:
        m = H.Measurements(); m.has_speech_profile = 1
        m.speech_profile.bands_measured = 1; m.speech_profile.body_band_rms = -30.0; m.speech_profile.sib_band_rms = -30.0 + excess
        eff, _ = H.adapt(base, m)
        assert abs(eff.deess_intensity - want) < 1e-12, excess
    m = H.Measurements(); m.has_speech_profile = 1; m.speech_profile.bands_measured = 0
    eff, _ = H.adapt(base, m)
    assert eff.deess_intensity == 0.0


def test_gate_threshold_ratio_range_rules(lib):
    # adaptive_speech_gate.go:88-300 / adaptive_test.go:962-1516
    base = H.default_config()
    m = H.Measurements(); m.has_speech_profile = 1; m.voiced_low_percentile = -34.0; m.gate_separation_db = 21.0; m.input_lra = 12.0
    eff, d = H.adapt(base, m)
    assert abs(eff.gate_threshold - 10 ** (-40 / 20)) < 1e-15 and eff.gate_ratio == 2.0
    assert abs(eff.gate_range - 10 ** (-14 / 20)) < 1e-15 and d.gate_narrow_gap == 0
    m.gate_separation_db = 9.0; m.input_lra = 16.0                                   # narrow gap + wide LRA
    eff, d = H.adapt(base, m)
    assert eff.gate_ratio == 1.5 and abs(eff.gate_range - 10 ** (-8 / 20)) < 1e-15 and d.gate_narrow_gap == 1
    m.voiced_low_percentile = -10.0                                                  # clamp at -25 dB
    eff, _ = H.adapt(base, m)
    assert abs(eff.gate_threshold - 10 ** (-25 / 20)) < 1e-15
    # no profile: max(floor + 12/(1-1/ratio), -40) clamped to [-80,-25]
    m = H.Measurements(); m.floor = -58.0; m.input_i = -42.1; m.input_lra = 6.0
    eff, _ = H.adapt(base, m)
    assert abs(eff.gate_threshold - 10 ** (-34 / 20)) < 1e-12


def test_compressor_threshold_rules(lib):
    # adaptive_levelling_compressor.go:54-99 / adaptive_test.go:1748-2030
    base = H.default_config()
    m = H.Measurements(); m.has_speech_profile = 1; m.speech_profile.sample.rms_level = -24.0; m.dynamics.rms_level = -30.0
    eff, _ = H.adapt(base, m)
    assert eff.comp_threshold_db == -15.0
    m.dynamics.rms_level = -20.0                                                     # full-file RMS above speech RMS wins
    eff, _ = H.adapt(base, m)
    assert eff.comp_threshold_db == -11.0
    m.speech_profile.sample.rms_level = -5.0                                         # clamp at -6
    eff, _ = H.adapt(base, m)
    assert eff.comp_threshold_db == -6.0
    m = H.Measurements(); m.dynamics.peak_level = -10.0                              # no profile: peak - 20
    eff, _ = H.adapt(base, m)
    assert eff.comp_threshold_db == -30.0
    m.dynamics.peak_level = float("nan")
    eff, _ = H.adapt(base, m)
    assert eff.comp_threshold_db == -18.0


def test_afftdn_custom_profile_gating_and_bn_string(lib):
    # adaptive.go:81-170 / adaptive_test.go:1427-1516: custom iff bands measured, separation >= 12, flatness >= 0.45
    base = H.default_config()
    m = H.Measurements(); m.floor = -61.234; m.has_noise_profile = 1
    bands = [-70.0, -71.5, -72.0, -73.0, -74.0, -75.0, -76.0, -77.0, -78.0, -79.0, -80.0, -81.0, -82.0, float("nan"), float("nan")]
    for i, b in enumerate(bands):
        m.noise_profile.band_noise[i] = b
    m.noise_profile.band_noise_n = 15; m.noise_profile.bands_measured = 1
    m.noise_profile.spectral.flatness = 0.5; m.gate_separation_db = 20.0
    eff, d = H.adapt(base, m)
    assert eff.afftdn_custom == 1 and eff.afftdn_track_noise == 0 and d.afftdn_custom == 1
    mean = sum(bands[:13]) / 13
    want = "|".join(["%.1f" % max(-24, min(24, b - mean)) for b in bands[:13]] + ["0.0", "0.0"])
    assert eff.afftdn_band_noise.decode() == want
    spec = H.filter_spec(eff, 2)
    assert f"afftdn=nr=12:nt=custom:bn={want}:tn=0:nf=-61.234" in spec
    m.noise_profile.spectral.flatness = 0.44
    eff, _ = H.adapt(base, m)
    assert eff.afftdn_custom == 0 and "nt=w:tn=0:nf=-61.234" in H.filter_spec(eff, 2)
    m.noise_profile.spectral.flatness = 0.5; m.gate_separation_db = 11.9
    eff, _ = H.adapt(base, m)
    assert eff.afftdn_custom == 0
    m.floor = -95.0                                                                  # nf clamped to [-80,-20]
    eff, _ = H.adapt(base, m)
    assert eff.afftdn_noise_floor == -80.0


def test_afftdn_band_edges(lib):
    # analyser_noise_bands.go:15-51: geometric midpoints, outer edges one half-step out
    import math
    c = [80, 125, 195, 290, 440, 660, 1000, 1500, 2250, 3350, 5000, 7500, 11200, 16000, 24000]
    for i in range(15):
        lo = C.c_double(); hi = C.c_double()
        lib.jt_host_afftdn_band_edges(C.c_int(i), C.byref(lo), C.byref(hi))
        wlo = c[0] / math.sqrt(c[1] / c[0]) if i == 0 else math.sqrt(c[i - 1] * c[i])
        whi = c[14] * math.sqrt(c[14] / c[13]) if i == 14 else math.sqrt(c[i] * c[i + 1])
        assert lo.value == wlo and hi.value == whi


# ---------------------------------------------------------------- output naming (processor_test.go:18-62)
def test_output_path_and_lufs_filename_value():
    cases = [("/tmp/foo.wav", "/tmp/foo-LUFS-16-processed.flac"), ("/tmp/foo.WAV", "/tmp/foo-LUFS-16-processed.flac"),
             ("/tmp/foo.flac", "/tmp/foo-LUFS-16-processed.flac"), ("/tmp/foo.mp3", "/tmp/foo-LUFS-16-processed.flac"),
             ("/tmp/foo", "/tmp/foo-LUFS-16-processed.flac"), ("/tmp/foo.bar.wav", "/tmp/foo.bar-LUFS-16-processed.flac")]
    for inp, want in cases:
        assert H.output_path(inp, 16) == want
    assert H.output_path("clip.wav", 16) == "clip-LUFS-16-processed.flac"
    for lufs, want in [(-16.4, 16), (-16.5, 17), (-16.6, 17), (15.5, 16)]:
        assert H.lufs_filename_value(lufs) == want


# ------------------------------------------------------------------ per-builder cases (filters_test.go:468-989, 1414-1527)
def _spec_for(lib, cfg, pass_no):
    if pass_no != 4:
        return H.filter_spec(cfg, pass_no)
    ms = L.LoudnormStats(); ms.input_i, ms.input_tp, ms.input_lra, ms.input_thresh = -24.0, -5.0, 6.0, -34.0
    dec = H.LimiterDecision(); dec.ceiling_db = -1.0
    buf = C.create_string_buffer(4096); ap = L.LoudnormApply()
    lib.jt_host_pass4_spec(C.byref(cfg), C.byref(ms), C.c_double(-0.5), C.byref(dec), C.c_int(48000), None, buf, C.c_int(4096), C.byref(ap))
    return buf.value.decode()


def _element(spec, prefix):
    for el in spec.split(","):
        if el.startswith(prefix):
            return el
    return ""


def _apply(cfg, sets):
    for k, v in sets.items():
        if isinstance(v, dict):
            sub = getattr(cfg, k)
            for kk, vv in v.items():
                setattr(sub, kk, vv)
        elif isinstance(v, str):
            setattr(cfg, k, v.encode())
        else:
            setattr(cfg, k, v)


def test_filter_builder_cases(lib):
    g = load("filter_builder_cases.json")
    base_j = load("filter_specs.json")["test_base_config"]
    for case in g["cases"]:
        cfg = _cfg_from_json(base_j)
        _apply(cfg, case["set"])
        spec = _spec_for(lib, cfg, case["pass"])
        for s in case.get("contains", []):
            assert s in spec, (case["name"], s, spec)
        for s in case.get("absent", []):
            assert s not in spec, (case["name"], s, spec)
        if "order" in case:
            pos = [spec.index(s) for s in case["order"]]
            assert pos == sorted(pos), (case["name"], spec)
        if "element" in case:
            assert _element(spec, case["element"]["prefix"]) == case["element"]["want"], (case["name"], spec)


def test_default_adeclick_clause_and_shared_analysis_segments(lib):
    g = load("filter_builder_cases.json")
    cfg = H.default_config()
    p4 = _spec_for(lib, cfg, 4)
    assert _element(p4, "adeclick=") == g["default_adeclick"]["want_element"]
    cfg2 = _cfg_from_json(load("filter_specs.json")["test_base_config"])
    cfg2.analysis_enabled = 1; cfg2.target_i = -16.0
    p2 = H.filter_spec(cfg2, 2)
    for pre in ("astats=", "aspectralstats="):
        assert _element(p2, pre) and _element(p2, pre) == _element(p4, pre)
    pref = g["shared_analysis_segments"]["ebur128_prefix"]
    assert _element(p4, "ebur128=") == pref and _element(p2, "ebur128=").startswith(pref)


def test_adeclick_method_spellings_go_into_the_spec_verbatim(lib):
    """AdeclickConfig.Method is written into the filter spec as given (filters.go:958-960) and af_adeclick's option table names each
    method twice: "s" / "save", "a" / "add"; "" leaves the option out.  The long names select the same kernels."""
    import ctypes as C
    want = {0: None, 1: "m=s", 2: "m=a", 3: "m=save", 4: "m=add"}
    for code, text in want.items():
        cfg = H.default_config(); cfg.adeclick_method_s = code
        el = _element(_spec_for(lib, cfg, 4), "adeclick=")
        assert el.startswith("adeclick=") and (":m=" not in el if text is None else el.endswith(":" + text)), (code, el)
