"""Speech election, level variance, interval finalisation and the VU level — the reference's table tests restated over the C ABI
(analyser_candidates_speech_test.go:50-239, analyser_metrics_test.go:499-567, encoder_level_test.go:42-110).  Inputs are built
exactly as the Go helpers build them (cited); CPU only."""
import ctypes as C
import math

import numpy as np
import pytest

from jivetalking_amd import hostlogic as H, _lib as L

HOP = 250_000_000                       # analysisIntervalHop (analyser_vad.go:16)
SEC = 1_000_000_000
ADEQ = 30 * SEC                         # speechDurationAdequacyMinimum (analyser_candidates_speech.go:92)
MIN_SNR = 20.0                          # minSNRMargin (:83)


def score(rms, dur_ns, floor, var):
    return H.lib().jt_host_score_speech_candidate(C.c_double(rms), C.c_int64(dur_ns), C.c_double(floor), C.c_double(var))


def speech_run(start_ns, count, level):
    """speechRunIntervals (analyser_candidates_speech_test.go:115-126)"""
    return [dict(timestamp_ns=start_ns + i * HOP, rms_level=level, momentary_lufs=level, peak_level=level + 12.0) for i in range(count)]


def elect(regions, rows, floor):
    iv = H.make_intervals(rows)
    rg = (H.Region * len(regions))(*[H.Region(a, b, b - a) for a, b in regions])
    best = H.Region(); cands = (H.SpeechCandidate * 8)()
    n = H.lib().jt_host_find_best_speech_region(rg, C.c_int(len(regions)), iv, C.c_int64(len(rows)), C.c_int(1), C.c_double(floor), C.byref(best), cands, C.c_int(8))
    return n, best, cands


def test_snr_monotonicity():
    """TestScoreSpeechCandidateGrounded_SNRMonotonicity (:50-73)"""
    floor, dur = -60.0, 45 * SEC
    assert score(floor + 45.0, dur, floor, 0.0) > score(floor + 25.0, dur, floor, 0.0)
    assert score(floor + (MIN_SNR - 10.0), dur, floor, 0.0) < score(floor + (MIN_SNR + 5.0), dur, floor, 0.0)


def test_duration_adequacy_saturates():
    """TestScoreSpeechCandidateGrounded_DurationAdequacySaturation (:75-96)"""
    floor, rms = -60.0, -20.0
    at_min = score(rms, ADEQ, floor, 0.0)
    assert at_min == score(rms, ADEQ * 3, floor, 0.0)
    assert score(rms, ADEQ // 2, floor, 0.0) < at_min


def test_consistency_tie_break():
    """TestScoreSpeechCandidateGrounded_ConsistencyTieBreak (:98-110)"""
    assert score(-20.0, 45 * SEC, -60.0, 1.0) > score(-20.0, 45 * SEC, -60.0, 9.0)


def test_voice_activated_case_elects_the_sparse_wide_snr_run():
    """TestFindBestSpeechRegion_VoiceActivatedCase (:132-160)"""
    min_iv = ADEQ // HOP
    short = speech_run(0, min_iv + 4, -18.0)
    short_end = short[-1]["timestamp_ns"] + HOP
    long_start = short_end + 5 * SEC
    long_ = speech_run(long_start, (min_iv + 4) * 3, -38.0)
    long_end = long_[-1]["timestamp_ns"] + HOP
    n, best, _ = elect([(0, short_end), (long_start, long_end)], short + long_, -60.0)
    assert n >= 1 and best.start_ns == 0


def test_always_elects_a_lone_sub_floor_run():
    """TestFindBestSpeechRegion_AlwaysElects (:162-191): the fallback path, candidate score under the 0.3 sanity floor"""
    run = speech_run(0, 12, -33.0)
    end = run[-1]["timestamp_ns"] + HOP
    n, best, cands = elect([(0, end)], run, -35.0)
    assert n == 1 and best.start_ns == 0 and cands[0].score < 0.3


def test_all_below_snr_minimum_elects_the_highest():
    """TestFindBestSpeechRegion_AllBelowSNRMinimumElectsHighest (:193-219)"""
    lo = speech_run(0, 74, -49.35); lo_end = lo[-1]["timestamp_ns"] + HOP
    hi_start = lo_end + 5 * SEC
    hi = speech_run(hi_start, 81, -48.46); hi_end = hi[-1]["timestamp_ns"] + HOP
    n, best, _ = elect([(0, lo_end), (hi_start, hi_end)], lo + hi, -60.0)
    assert n == 2 and best.start_ns == hi_start


def test_level_variance():
    """TestLevelVariance (:221-239), axisRMS"""
    flat = H.make_intervals([dict(timestamp_ns=i * HOP, rms_level=-20.0) for i in range(20)])
    spread = H.make_intervals([dict(timestamp_ns=i * HOP, rms_level=-20.0 + (4.0 if i % 2 == 0 else -4.0)) for i in range(20)])
    lv = H.lib().jt_host_level_variance
    assert lv(flat, C.c_int64(20), C.c_int(1)) <= 1e-9
    assert lv(spread, C.c_int64(20), C.c_int(1)) == pytest.approx(16.0) and lv(None, C.c_int64(0), C.c_int(1)) == 0.0


def test_interval_finalize_averages_the_spectral_metrics():
    """TestIntervalAccumulatorFinalize_WritesAveragedSpectralMetrics (analyser_metrics_test.go:499-567): two output frames in one
    interval -> their mean, field by field."""
    a = dict(mean=2.0, variance=4.0, centroid=1000.0, spread=200.0, skewness=1.0, kurtosis=2.0, entropy=0.2, flatness=0.4, crest=6.0, flux=0.02,
             slope=-0.10, decrease=0.06, rolloff=5000.0)
    b = dict(mean=6.0, variance=12.0, centroid=3000.0, spread=600.0, skewness=3.0, kurtosis=6.0, entropy=0.6, flatness=0.8, crest=10.0, flux=0.06,
             slope=-0.30, decrease=0.18, rolloff=9000.0)
    meta = (L.FrameMeta * 2)()
    for m, v in zip(meta, (a, b)):
        m.momentary, m.shortterm, m.true_peak, m.sample_peak = -20.0, -21.0, 0.5, 0.4
        for k, x in v.items():
            setattr(m.spectral, k, x)
    ss = np.full(1, 0.01 * 4096); pk = np.full(1, 0.3)
    out = (H.Interval * 4)()
    n = H.lib().jt_host_build_intervals(C.c_int(48000), C.c_int64(4096), C.c_int(4096), C.c_int(1), ss.ctypes.data_as(C.POINTER(C.c_double)),
                                        pk.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(1), meta, C.c_int64(2), C.c_int(0), out, C.c_int64(4))
    assert n == 1 and out[0].spectral_found == 1
    for k in a:
        assert getattr(out[0].spectral, k) == pytest.approx((a[k] + b[k]) / 2, abs=0.001), k       # spectralTestEpsilon
    assert out[0].rms_level == pytest.approx(20 * math.log10(0.1)) and out[0].peak_level == pytest.approx(20 * math.log10(0.3))


@pytest.mark.parametrize("db,want", [(-65.0, -65.0), (-90.0, -70.0), (None, -70.0), (-59.9, -59.9), (-3.0, -3.0)])
def test_frame_level_clamps_to_the_meter_range(db, want):
    """encoder_level_test.go:42-110 (TestCalculateFrameLevelFloorsAtMeterFloor / ...BelowOldMinus60Floor): a DC frame reads its level,
    below -70 dB and digital silence clamp to the meter floor -70, never -inf.  (s16 frames: +-0.5 dB as in the reference's test.)"""
    amp = 0 if db is None else int(round(32768 * 10 ** (db / 20.0)))
    pcm = np.full(1024, amp, np.int16)
    got = H.lib().jt_host_frame_level_s16(pcm.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int(1024))
    assert math.isfinite(got) and got >= -70.0 and abs(got - want) <= 0.5
    assert H.lib().jt_host_frame_level_s16(None, C.c_int(0)) == -70.0
