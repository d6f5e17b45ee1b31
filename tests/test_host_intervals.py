"""Interval construction (collectAnalysisFrames + intervalAccumulator, analyser.go:571-638, analyser_metrics.go:165-428;
SURVEY App. D) and the metadata print-format quantisation, CPU only."""
import ctypes as C
import math
import random

import numpy as np
import pytest

from jivetalking_amd import hostlogic as H, _lib as L


@pytest.fixture(scope="module")
def lib():
    return H.lib()


def build(lib, sr, n, F, ch, ss, pk, meta, quant):
    nfr = (n + F - 1) // F
    out = (H.Interval * (n // (sr // 5) + 32))()
    k = lib.jt_host_build_intervals(sr, C.c_int64(n), F, ch, ss.ctypes.data_as(C.POINTER(C.c_double)),
                                    pk.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(nfr), meta, C.c_int64(len(meta)), quant,
                                    out, C.c_int64(len(out)))
    return [out[i] for i in range(k)]


def test_interval_cadence_4096_frames_at_48k(lib):
    # App. D: with 4096-sample frames at 48 kHz interval 0 spans frames 0-3 (16384 samples), every later interval
    # 3 frames (12288 samples = 256 ms); each timestamp is the START time of the frame that tripped the 250 ms test.
    sr, F = 48000, 4096
    n = F * 40
    nfr = n // F
    ss = np.full(nfr, F * 0.01, np.float64)        # sum(x^2) per frame for x = 0.1
    pk = np.full(nfr, 0.1, np.float64)
    meta = (L.FrameMeta * 0)()
    iv = build(lib, sr, n, F, 1, ss, pk, meta, 0)
    ts = [i.timestamp_ns for i in iv]
    assert ts[0] == 0
    assert ts[1] == int(3 * F / sr * 1e9)            # frame 3 tripped the first close; its start time stamps interval 1
    assert all(ts[k + 1] - ts[k] in (255999999, 256000000, 256000001) for k in range(1, len(ts) - 1))
    assert abs(iv[0].rms_level + 20.0) < 1e-9 and abs(iv[0].peak_level + 20.0) < 1e-9
    # 40 frames: interval 0 takes 4, then 12 intervals of 3; nothing is left for a trailing partial interval
    assert len(iv) == 13


def test_silent_interval_floor_and_stereo_count(lib):
    sr, F = 48000, 4096
    n = F * 8
    ss = np.zeros(8); pk = np.zeros(8)
    iv = build(lib, sr, n, F, 2, ss, pk, (L.FrameMeta * 0)(), 0)
    assert iv[0].rms_level == -120.0 and iv[0].peak_level == -120.0     # rms < 1e-5 -> -120 (analyser_metrics.go:398-405)


def test_metadata_quantisation_matches_printf(lib):
    rng = random.Random(1)
    sr, blk, F = 48000, 4800, 4096
    n_meta = 1500
    meta = (L.FrameMeta * n_meta)()
    for i in range(n_meta):
        m = meta[i]
        m.momentary = -rng.uniform(10, 70); m.shortterm = -rng.uniform(10, 70)
        m.true_peak = rng.uniform(0, 1); m.sample_peak = rng.uniform(0, 1)
        for k in L.SPECTRAL_KEYS:
            setattr(m.spectral, k, rng.uniform(-1, 1) * 10 ** rng.randint(-6, 5))
    n = n_meta * blk
    nfr = (n + F - 1) // F
    iv = build(lib, sr, n, F, 1, np.ones(nfr), np.full(nfr, 0.5), meta, 1)
    # python restatement with real printf round trips: ebur128 "%.3f", aspectralstats "%g"
    want, cur, start, proc, nxt = [], None, 0, 0, 0

    def fresh():
        return dict(cnt=0, m=0.0, sp={k: 0.0 for k in L.SPECTRAL_KEYS}, tp=None)
    cur = fresh()
    for f in range(nfr):
        nb = min(F, n - f * F)
        t = int(proc / sr * 1e9); proc += nb
        if t - start >= 250_000_000:
            want.append(cur); start = t; cur = fresh()
        seen = (proc // 1024) * 1024
        while nxt < n_meta and (nxt + 1) * blk <= seen:
            m = meta[nxt]
            cur["m"] += float("%.3f" % m.momentary); cur["cnt"] += 1
            for k in L.SPECTRAL_KEYS:
                cur["sp"][k] += float("%g" % getattr(m.spectral, k))
            tpq = float("%.3f" % m.true_peak)
            tp = -120.0 if tpq <= 0 else 20 * math.log10(tpq)
            cur["tp"] = tp if cur["tp"] is None else max(cur["tp"], tp)
            nxt += 1
    for i, w in enumerate(want):
        if not w["cnt"]:
            continue
        assert iv[i].momentary_lufs == w["m"] / w["cnt"]
        for k in L.SPECTRAL_KEYS:
            assert getattr(iv[i].spectral, k) == w["sp"][k] / w["cnt"]
