"""Run record + sidecars: jt_host_run_record_json / jt_host_intervals_jsonl / jt_host_candidates_jsonl against the
reference's schema tests (runrecord_tags_test.go:87-470, runrecord_test.go:76-470, runrecord_sidecar_test.go:15-270).  The expected
key lists and shapes are transcribed from those tests (cited per check); CPU only: the record is built from a result struct filled
here with the values the reference's own fixture uses (populatedAudioMeasurements, runrecord_tags_test.go:29-85)."""
import ctypes as C
import json
import math

import numpy as np
import pytest

from jivetalking_amd import hostlogic as H, _lib as L

SPECTRAL = dict(mean=1, variance=2, centroid=2000, spread=400, skewness=1, kurtosis=4, entropy=0.4, flatness=0.6, crest=8, flux=0.04,
                slope=-0.2, decrease=0.12, rolloff=7000)


def set_spectral(s, vals=SPECTRAL):
    for k, v in vals.items():
        setattr(s, k, float(v))


def populated_result():
    r = H.ProcessResult()
    m = r.input
    set_spectral(m.spectral)
    m.momentary, m.shortterm, m.sample_peak = -17, -16.5, -1.2
    m.input_i, m.input_tp, m.input_lra, m.input_thresh, m.target_offset = -18, -1, 7, -28, -2
    d = m.dynamics
    d.dynamic_range, d.rms_level, d.peak_level, d.rms_trough, d.rms_peak = 12, -22, -3, -45, -18
    d.dc_offset, d.flat_factor, d.crest_factor, d.zero_crossings_rate, d.zero_crossings = 0.001, 0, 14, 0.05, 1000
    d.max_difference, d.min_difference, d.mean_difference, d.rms_difference, d.entropy = 0.2, 0, 0.01, 0.02, 0.7
    d.min_level, d.max_level, d.noise_floor_count, d.bit_depth, d.number_of_samples = -90, -3, 500, 16, 480000
    m.floor, m.floor_source, m.floor_prescan, m.floor_astats = -60, 3, -58, -62
    m.room_tone_detect_level, m.voice_activated, m.reduction_headroom, m.duration_s = -59, 0, 38, 10.0
    m.n_candidates = 1
    c = m.candidates[0]
    c.region.start_ns, c.region.end_ns, c.region.duration_ns = 30_000_000_000, 40_000_000_000, 10_000_000_000
    s = c.sample
    s.rms_level, s.peak_level, s.crest_factor, s.momentary_lufs, s.shortterm_lufs, s.true_peak, s.sample_peak = -20, -3, 13, -19, -18, -2, -2.5
    set_spectral(s.spectral)
    c.voicing_density, c.body_band_rms, c.sib_band_rms, c.bands_measured, c.score = 0.8, -25, -30, 1, 7
    m.has_speech_profile = 1
    C.memmove(C.byref(m.speech_profile), C.byref(c), C.sizeof(c))
    m.has_noise_profile = 1
    p = m.noise_profile
    p.start_ns, p.duration_ns, p.measured_noise_floor, p.peak_level, p.crest_factor, p.entropy = 2_000_000_000, 10_000_000_000, -60, -50, 10, 0.5
    set_spectral(p.spectral, dict(mean=1.2, variance=2.4, centroid=1500, spread=350, skewness=0.8, kurtosis=3, entropy=0.55, flatness=0.4, crest=7.5,
                                  flux=0.03, slope=-0.25, decrease=0.11, rolloff=6500))
    m.has_room_tone_sample = 1
    C.memmove(C.byref(m.room_tone_sample), C.byref(c.sample), C.sizeof(c.sample))
    m.voiced_low_percentile, m.noise_high_percentile, m.gate_separation_db = -32, -55, 23
    # processing stages
    for a in (r.filtered, r.final_):
        a.n_frames_meta = 100
        a.r128.integrated, a.r128.lra, a.r128.momentary, a.r128.shortterm, a.r128.true_peak, a.r128.sample_peak = -20.1234, 6.5, -19, -19.5, 0.5, 0.45
        a.astats.crest_factor, a.astats.min_level, a.astats.max_level, a.astats.rms_level = 10.0, -0.5, 0.5, -21
        set_spectral(a.spectral_mean)
    H.lib().jt_host_default_config(C.byref(r.effective))
    r.effective.gate_threshold, r.effective.gate_range, r.effective.loudnorm_enabled, r.effective.target_i = 0.01, 0.05, 1, -16.0
    r.diag.gate_quiet_speech_estimate, r.diag.gate_separation, r.diag.gate_depth_db = -52, 8, 14
    r.measure.input_i, r.measure.input_tp = -20.12, -3.5
    ln = r.loudnorm
    ln.input_i, ln.input_tp, ln.input_lra, ln.input_thresh = -20.12, -3.5, 6.5, -30.4
    ln.output_i, ln.output_tp, ln.output_lra, ln.output_thresh, ln.target_offset = -16.0, 0.62, 6.5, -26.3, 0.0
    r.output_lufs, r.output_tp_db, r.offset, r.within_target, r.effective_target_i, r.linear_possible = -16.02, -1.9, 4.12, 1, -16.0, 1
    r.limiter.needed, r.limiter.ceiling_db, r.limiter.gain_db, r.limiter.filtered_tp = 1, -2.4, 6, -1
    r.limiter.pass3_prefix = b"volume=6.0dB,alimiter=limit=0.500000"
    r.has_region_samples = 1
    for rs in (r.filtered_room_tone, r.filtered_speech, r.final_room_tone, r.final_speech):
        rs.frames, rs.rms_level, rs.peak_level, rs.true_peak, rs.sample_peak, rs.momentary, rs.shortterm = 40, -50, -40, 0.01, 0.009, -48, -49
        set_spectral(rs.spectral)
    r.stage_ms[9] = 1.234
    return r


def keys_of(v, into=None):
    into = set() if into is None else into
    if isinstance(v, dict):
        for k, c in v.items():
            into.add(k); keys_of(c, into)
    elif isinstance(v, list):
        for c in v:
            keys_of(c, into)
    return into


def record(res, **kw):
    text = H.run_record_json(None, res, input_file="ep-LUFS-16-processed.flac", version="1.2.3", executable="/usr/bin/jivetalking",
                             processed_at="2026-01-02T03:04:05Z", sample_rate_hz=48000, channels=1, **kw)
    return text, json.loads(text)


def test_full_shape_and_sorted_keys():
    """runrecord_test.go:76-123 (TestRunRecord_FullShape) + json.MarshalIndent of a map tree: keys sorted at every level."""
    text, rec = record(populated_result())
    assert list(rec.keys()) == sorted(rec.keys())
    assert set(rec) == {"schema_version", "run", "loudness", "dynamics", "spectral", "noise", "regions", "filters", "normalisation"}
    assert rec["schema_version"] == 1 and rec["loudness"]["target_i_lufs"] == -16
    for dom in ("loudness", "dynamics", "spectral"):
        assert list(rec[dom]["stages"]) == ["filtered", "final", "input"]
    assert text.startswith('{\n  "dynamics": {\n    "stages": {\n      "filtered": {') and text.endswith("}")

    def walk(v):
        if isinstance(v, dict):
            assert list(v) == sorted(v)
            for c in v.values():
                walk(c)
        elif isinstance(v, list):
            for c in v:
                walk(c)
    walk(rec)
    assert rec["run"] == {"channels": 1, "duration_s": 10, "executable": "/usr/bin/jivetalking", "input_file": "ep-LUFS-16-processed.flac",
                          "processed_at": "2026-01-02T03:04:05Z", "sample_rate_hz": 48000, "version": "1.2.3"}


def test_canonical_keys_present_and_legacy_absent():
    """runrecord_tags_test.go:87-163,343-470: the §8.4 key surface."""
    _, rec = record(populated_result())
    keys = keys_of(rec)
    for k in ["integrated_lufs", "true_peak_dbtp", "lra_lu", "thresh_lufs", "target_offset_db", "momentary_lufs", "short_term_lufs", "sample_peak_dbfs",
              "rms_level_dbfs", "peak_level_dbfs", "dynamic_range_db", "crest_factor_astats_db", "rms_trough_dbfs", "rms_peak_dbfs", "dc_offset",
              "flat_factor", "zero_crossings_rate", "zero_crossings_count", "min_level_dbfs", "max_level_dbfs", "bit_depth", "number_of_samples",
              "noise_floor_count", "entropy", "floor_dbfs", "floor_source", "floor_prescan_dbfs", "floor_astats_dbfs", "reduction_headroom_db",
              "room_tone_detect_level_dbfs", "voice_activated", "centroid_hz", "spread_hz", "rolloff_hz", "voiced_low_percentile_dbfs",
              "noise_high_percentile_dbfs", "gate_separation_db", "crest_factor_db", "speech_band_body_rms_dbfs", "speech_band_sib_rms_dbfs",
              "measured_floor_dbfs", "spectral_centroid_hz", "spectral_mean", "spectral_variance", "spectral_spread_hz", "spectral_skewness",
              "spectral_entropy", "spectral_crest", "spectral_flux", "spectral_slope", "spectral_decrease", "spectral_rolloff_hz",
              # filters (runrecord_tags_test.go:343-393)
              "rumble_highpass", "bandlimit_lowpass", "noise_reduction", "speech_gate", "levelling_compressor", "deesser", "threshold_db", "ratio",
              "attack_ms", "release_ms", "range_db", "knee", "makeup", "detection", "makeup_db", "frequency_hz", "poles_count", "width", "mix",
              "transform", "strength", "patch_s", "research_s", "smooth", "afftdn_noise_reduction_db", "afftdn_noise_type", "afftdn_track_noise",
              "intensity", "amount", "frequency",
              # diagnostics (:395-431)
              "bandlimit_lowpass_reason", "quiet_speech_estimate_dbfs", "separation_db", "speech_headroom_db", "threshold_unclamped_db",
              "clamp_reason", "speech_gate_depth_db",
              # normalisation (:433-470, with the record's wrapper transforms runrecord_units.go:283-340)
              "input_lufs", "input_dbtp", "output_lufs", "output_dbtp", "gain_applied_db", "within_target", "skipped", "loudnorm_measured",
              "requested_target_lufs", "effective_target_lufs", "linear_mode_forced", "actual_norm_dynamic", "limiter_enabled", "ceiling_dbtp",
              "gain_db", "filtered_dbtp", "pre_gain_db", "limiter_clamped", "pass3_filter_prefix", "region_measurement_s", "normalization_type"]:
        assert k in keys, k
    for k in ["input_i", "input_tp", "input_lra", "input_thresh", "rms_level", "peak_level", "dynamic_range", "crest_factor", "target_offset",
              "momentary_loudness", "short_term_loudness", "sample_peak", "floor", "floor_prescan", "floor_astats", "reduction_headroom",
              "room_tone_detect_level", "min_level", "max_level", "zero_crossings", "spectral_centroid", "spectral_spread", "spectral_rolloff",
              "suggested_gate_threshold", "measured_noise_floor", "downmix", "analysis", "resample", "adeclick", "loudnorm", "filter_order",
              "FilterOrder", "Threshold", "threshold", "attack", "release", "final_measurements", "region_measurement_ns",
              # full series live in the sidecars (runrecord_sidecar_test.go:15-48,96-135)
              "interval_samples", "speech_candidates", "speech_regions", "speech_profile", "noise_profile"]:
        assert k not in keys, k


def test_regions_nested_shape_and_seconds():
    """runrecord_test.go:188-385: regions.room_tone / speech {elected, candidates_summary, samples{input,filtered,final}}, durations as *_s."""
    _, rec = record(populated_result())
    rg = rec["regions"]
    assert set(rg) == {"room_tone", "speech", "gate_statistics"}
    assert set(rg["room_tone"]) == {"elected", "samples"} and set(rg["speech"]) == {"elected", "candidates_summary", "samples"}
    assert set(rg["room_tone"]["samples"]) == set(rg["speech"]["samples"]) == {"input", "filtered", "final"}
    e = rg["room_tone"]["elected"]
    assert e["start_s"] == 2 and e["duration_s"] == 10 and "start" not in e and "duration" not in e
    assert e["spectral_centroid_hz"] == 1500 and e["spectral_kurtosis"] == 3 and e["spectral_flatness"] == 0.4 and e["entropy"] == 0.5
    s = rg["speech"]["elected"]
    assert s["region"] == {"duration_s": 10, "end_s": 40, "start_s": 30} and s["score"] == 7 and s["speech_bands_measured"] is True
    assert "original_start_s" not in s and "was_refined" not in s                       # omitempty
    assert rg["speech"]["candidates_summary"] == {"elected_score": 7, "evaluated_count": 1}
    assert rg["gate_statistics"] == {"gate_separation_db": 23, "noise_high_percentile_dbfs": -55, "voiced_low_percentile_dbfs": -32}
    assert set(rg["speech"]["samples"]["input"]) == {"rms_level_dbfs", "peak_level_dbfs", "crest_factor_db", "spectral", "momentary_lufs",
                                                     "short_term_lufs", "true_peak_dbtp", "sample_peak_dbfs"}     # no election fields (:284-341)


def test_analysis_only_drops_processing_blocks():
    """runrecord_test.go:125-155,307-338."""
    _, rec = record(populated_result(), analysis_only=True)
    assert "filters" not in rec and "normalisation" not in rec
    for dom in ("loudness", "dynamics", "spectral"):
        assert list(rec[dom]["stages"]) == ["input"]
    assert list(rec["regions"]["speech"]["samples"]) == ["input"] and list(rec["regions"]["room_tone"]["samples"]) == ["input"]


def test_non_finite_floats_are_null_and_number_format():
    """runrecord_test.go:157-186 (NaN / Inf -> null) and encoding/json's float format (shortest round-trip digits; exponent form below
    1e-6 and from 1e21; "e-07" printed as "e-7")."""
    r = populated_result()
    r.input.input_lra = math.nan; r.input.dynamics.rms_trough = -math.inf
    r.input.dynamics.dc_offset = 1e-7; r.input.dynamics.max_difference = 0.1; r.input.dynamics.zero_crossings = 1e21
    r.input.dynamics.number_of_samples = 123456789.125; r.input.dynamics.min_difference = 1e-6; r.input.dynamics.mean_difference = -2.5e-10
    text, rec = record(r)
    assert rec["loudness"]["stages"]["input"]["lra_lu"] is None and rec["dynamics"]["stages"]["input"]["rms_trough_dbfs"] is None
    for frag in ('"dc_offset": 1e-7,', '"max_difference": 0.1,', '"zero_crossings_count": 1e+21,', '"number_of_samples": 123456789.125,',
                 '"min_difference": 0.000001,', '"mean_difference": -2.5e-10,', '"integrated_lufs": -18,', '"target_i_lufs": -16'):
        assert frag in text, frag


def test_gate_threshold_is_decibels_and_loudnorm_numeric():
    """runrecord_test.go:442-470 (newFiltersBlock converts the linear gate threshold / range to dB) and :387-440 (loudnorm_measured is
    numeric, parsed from the "%.2f" strings)."""
    _, rec = record(populated_result())
    g = rec["filters"]["speech_gate"]
    assert abs(g["threshold_db"] - (-40.0)) < 1e-9 and abs(g["range_db"] - 20 * math.log10(0.05)) < 1e-12
    lm = rec["normalisation"]["loudnorm_measured"]
    assert lm == {"input_integrated_lufs": -20.12, "input_lra_lu": 6.5, "input_thresh_lufs": -30.4, "input_true_peak_dbtp": -3.5,
                  "normalization_type": "linear", "output_integrated_lufs": -16, "output_lra_lu": 6.5, "output_thresh_lufs": -26.3,
                  "output_true_peak_dbtp": 0.62, "target_offset_db": 0}
    n = rec["normalisation"]
    assert n["region_measurement_s"] == 0.001234 and n["limiter_enabled"] is True and n["pass3_filter_prefix"].startswith("volume=6.0dB")
    # output stages: astats conversions of the Go side (crest ratio -> dB, min / max sample -> dBFS: analyser_metrics_test.go:443-468)
    d = rec["dynamics"]["stages"]["final"]
    assert d["crest_factor_astats_db"] == 20.0 and abs(d["min_level_dbfs"] - 20 * math.log10(0.5)) < 1e-12 and d["max_level_dbfs"] == d["min_level_dbfs"]
    lo = rec["loudness"]["stages"]["filtered"]
    assert lo["integrated_lufs"] == -20.123 and lo["thresh_lufs"] == -30.123 and lo["target_offset_db"] == 0          # "%.3f", I - 10 fallback


def test_candidates_sidecar_lines():
    """runrecord_sidecar_test.go:161-193,248-270: one {"kind":"speech",...} object per candidate, non-finite values nulled."""
    r = populated_result()
    r.input.n_candidates = 2
    C.memmove(C.byref(r.input.candidates[1]), C.byref(r.input.candidates[0]), C.sizeof(r.input.candidates[0]))
    r.input.candidates[1].score = math.nan
    lines = H.candidates_jsonl(r).splitlines()
    assert len(lines) == 2 and all(l.startswith('{"kind":"speech",') for l in lines)
    a, b = (json.loads(l) for l in lines)
    assert a["region"] == {"duration": 10_000_000_000, "end": 40_000_000_000, "start": 30_000_000_000} and a["score"] == 7
    assert b["score"] is None and a["spectral"]["centroid_hz"] == 2000 and a["speech_band_sib_rms_dbfs"] == -30


def test_loudnorm_json_body_round_trips_through_the_reference_parser_fields():
    """normalise_statsfile_test.go:14-51: the ten string fields parseLoudnormStatsFile reads."""
    s = L.LoudnormStats(-23.0, -4.0, 5.0, -33.0, -16.0, -2.0, 5.0, -26.0, 0.0, 0)
    body = json.loads(H.loudnorm_json(s))
    assert body == {"input_i": "-23.00", "input_tp": "-4.00", "input_lra": "5.00", "input_thresh": "-33.00", "output_i": "-16.00",
                    "output_tp": "-2.00", "output_lra": "5.00", "output_thresh": "-26.00", "normalization_type": "linear", "target_offset": "0.00"}
    s.normalization_type_dynamic = 1; s.input_i = -math.inf
    body = json.loads(H.loudnorm_json(s))
    assert body["normalization_type"] == "dynamic" and body["input_i"] == "-inf"


def test_strings_are_escaped_as_encoding_json_does():
    """encoding/json (encode.go appendString): \\b \\f \\n \\r \\t short forms, other control bytes \\u00XX, < > & as \\u003c \\u003e \\u0026, U+2028 /
    U+2029 escaped, every invalid UTF-8 byte replaced by U+FFFD; valid multi-byte text passes through."""
    res = H.ProcessResult()
    name = b"a\x08b\x0cc\x01<d>&\"q\"\\ " + "\u00e9\u2028x\u2029\U0001F3A4".encode() + b" bad:\xff\xc0\xaf\xed\xa0\x80 end.flac"
    import ctypes as C
    pv = H.RunProvenance(name, b"v", b"x", b"t", 1.0, 48000, 1)
    n = H.lib().jt_host_run_record_json(None, C.byref(res), C.byref(pv), C.c_int(0), None, C.c_int64(0))
    buf = C.create_string_buffer(int(n) + 1)
    H.lib().jt_host_run_record_json(None, C.byref(res), C.byref(pv), C.c_int(0), buf, C.c_int64(int(n) + 1))
    raw = buf.value
    want = (b'"a\\bb\\fc\\u0001\\u003cd\\u003e\\u0026\\"q\\"\\\\ ' + "\u00e9".encode() + b"\\u2028x\\u2029" + "\U0001F3A4".encode() +
            b' bad:\\ufffd\\ufffd\\ufffd\\ufffd\\ufffd\\ufffd end.flac"')
    assert want in raw, raw[raw.find(b"input_file"):raw.find(b"input_file") + 200]
    rec = json.loads(raw.decode())                               # and it is valid JSON that decodes to the replaced text
    assert rec["run"]["input_file"] == "a\x08b\x0cc\x01<d>&\"q\"\\ \u00e9\u2028x\u2029\U0001F3A4 bad:" + "\ufffd" * 6 + " end.flac"
