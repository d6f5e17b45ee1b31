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


# ---------------------------------------------------------------- decoder frames of different lengths (round 6)
def _rule(frame_lens, sr):
    """analyser.go:588-600 on frame lengths: [(interval timestamp, samples in it)]"""
    out = []; start = 0; processed = 0; acc = 0
    for nb in frame_lens:
        t = int(float(processed) / float(sr) * 1e9)
        processed += int(nb); acc += int(nb)
        if t - start >= 250_000_000:
            out.append((start, acc)); start = t; acc = 0
    if acc > 0:
        out.append((start, acc))
    return out


def _build_v(lib, sr, lens, F, ch, ss, pk):
    n = int(np.sum(lens)) if lens is not None else None
    out = (H.Interval * 4096)()
    la = None if lens is None else np.ascontiguousarray(lens, np.int32)
    k = lib.jt_host_build_intervals_v(sr, C.c_int64(n), F, None if la is None else la.ctypes.data_as(C.POINTER(C.c_int32)), ch,
                                      ss.ctypes.data_as(C.POINTER(C.c_double)), pk.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(ss.size),
                                      (L.FrameMeta * 0)(), C.c_int64(0), 0, out, C.c_int64(len(out)))
    return [out[i] for i in range(k)]


@pytest.mark.parametrize("sr,pattern", [(48000, (4096, 2048)), (44100, (1152, 576, 4608)), (96000, (16, 4096, 192)), (44100, (1024,))])
def test_intervals_from_frames_of_different_lengths(lib, sr, pattern):
    """jt_host_build_intervals_v (a variable-blocksize FLAC stream: analyser.go:592 reads every frame's own NbSamples): interval count,
    timestamps and the samples each interval holds follow the rule frame by frame; with all lengths equal it is jt_host_build_intervals."""
    rng = np.random.default_rng(sr + len(pattern))
    lens = []
    while sum(lens) < sr * 20:
        lens.append(int(pattern[len(lens) % len(pattern)]))
    lens[-1] = max(1, lens[-1] - 7)                                    # (a short last frame)
    lens = np.asarray(lens, np.int32)
    amp = rng.uniform(0.01, 0.5, lens.size)
    ss = amp * amp * lens; pk = amp.copy()                             # a constant |x| = amp per frame
    iv = _build_v(lib, sr, lens, int(lens.max()), 1, ss, pk)
    want = _rule(lens.tolist(), sr)
    assert [i.timestamp_ns for i in iv] == [w[0] for w in want]
    pos = 0; f = 0
    for i, (_, cnt) in enumerate(want):
        e = 0.0; n = 0; p = 0.0
        while n < cnt:
            e += ss[f]; n += int(lens[f]); p = max(p, pk[f]); f += 1
        assert n == cnt
        rms = math.sqrt(e / n)
        assert abs(iv[i].rms_level - (-120.0 if rms < 1e-5 else 20 * math.log10(rms))) < 1e-9
        assert abs(iv[i].peak_level - 20 * math.log10(p)) < 1e-9
    if len(pattern) == 1:
        n = int(lens.sum()); F = pattern[0]
        same = build(lib, sr, n, F, 1, ss, pk, (L.FrameMeta * 0)(), 0)
        assert [(a.timestamp_ns, a.rms_level, a.peak_level) for a in same] == [(a.timestamp_ns, a.rms_level, a.peak_level) for a in iv]


def test_loudnorm_histogram_bin_by_table_is_the_bisections(lib):
    # ebur128.c's find_histogram_index is a bisection over 1001 boundaries; the library reads a table keyed by the double's top 20 bits
    # and compares against the boundaries themselves (jt_plan.cpp: hist_index).  Same bin for two million energies spread over the
    # range (and beyond both ends), for energies a few ulps around every boundary (the entry checks those itself), for tiny / huge values.
    lib.jt_host_hist_index_check.restype = C.c_int64
    rng = np.random.default_rng(5)
    e = np.concatenate([10.0 ** rng.uniform(-9, 5, 2_000_000), np.array([0.0, 1e-300, 1.17e-7, 1.18e-7, 2.0 ** -24, 2.0 ** 11, 2.0 ** 12, 1e30, np.inf])])
    assert lib.jt_host_hist_index_check(e.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(e.size)) == 0
