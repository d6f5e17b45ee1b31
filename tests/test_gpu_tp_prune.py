"""ebur128's true peak by branch and bound (k_resample.hip: k_tp_bounds -> k_tp_list_* on the seeds -> k_tp_select -> k_tp_list_* on the
kept units) against the exhaustive kernels (option tp_unpruned) and the CPU oracle.

The reference reads the true peak only as lavfi.r128.true_peak, f_ebur128.c's RUNNING maximum (analyser_metrics.go:224,249,864; the
per-frame peaks are not in the metadata it asks for), so what must be identical is the per-frame running maximum and the file's
value -- bit for bit, since every evaluated output is the same tap sum in the same order."""
import numpy as np
import pytest

from conftest import options

pytestmark = pytest.mark.gpu


def _signals(sr):
    from jivetalking_amd import synth
    rng = np.random.default_rng(sr)
    n = int(sr * 21.7) + 13
    t = np.arange(n) / sr
    sp = synth.speech_like(n / sr + 0.1, sr, seed=7)[:n].astype(np.float32)
    env = np.interp(t, [0, 3, 3.01, 9, 9.01, 15, 21.8], [0.02, 0.02, 0.4, 0.4, 0.08, 0.9, 0.3]).astype(np.float32)
    yield "speech with level steps", sp * env / np.max(np.abs(sp))
    yield "full-scale sine (nothing can be pruned)", (0.5 * np.sin(2 * np.pi * 997.0 * t)).astype(np.float32)
    yield "fs/4 sine at 45 degrees (inter-sample peaks 3 dB over the samples)", (0.5 * np.sin(2 * np.pi * (sr / 4) * t + np.pi / 4)).astype(np.float32)
    z = np.zeros(n, np.float32); yield "silence", z
    a = (0.01 * rng.standard_normal(n)).astype(np.float32)
    a[0] = 0.8; a[1] = -0.8; a[-1] = 0.95; a[n // 2] = -0.9; a[n // 2 + 1] = 0.9
    yield "noise with overs at the first, middle and last samples", a
    d = (0.3 * rng.standard_normal(n)).astype(np.float32) * np.linspace(1.0, 0.001, n).astype(np.float32)
    yield "decaying noise (the first units carry the maximum)", d
    g = (0.3 * rng.standard_normal(n)).astype(np.float32) * np.linspace(0.001, 1.0, n).astype(np.float32)
    yield "growing noise (records all the way)", g


@pytest.mark.parametrize("sr", [48000, 44100, 96000, 88200, 22050])
def test_branch_and_bound_true_peak_equals_the_exhaustive_kernels(engine, oracle, sr):
    """48 / 96 kHz: integer ratios (units of 256 windows, k_tp_list_q4); 44.1 / 88.2 / 22.05 kHz: one polyphase period per unit
    (k_tp_list_period).  tp_prune_min = 0 puts these 22-second signals on the branch-and-bound path (by default only signals above
    2^20 samples take it)."""
    some = 0
    for name, x in _signals(sr):
        with options(engine, tp_unpruned=True):
            b = engine.op_ebur128(x, sr)
        with options(engine, tp_prune_min=0):
            engine.timers()
            a = engine.op_ebur128(x, sr)
            t = engine.timers()
        assert a["true_peak"] == b["true_peak"], (sr, name, a["true_peak"], b["true_peak"])
        assert np.array_equal(a["TP"], b["TP"]), (sr, name, int(np.count_nonzero(a["TP"] != b["TP"])))
        assert a["integrated"] == b["integrated"] and np.array_equal(a["SP"], b["SP"])
        assert t["tp_units_total"] > 0 and 0 < t["tp_units_evaluated"] <= t["tp_units_total"], (sr, name, t)
        print(f"{sr} Hz, {name}: {t['tp_units_evaluated']} of {t['tp_units_total']} units evaluated")
        if name.startswith("speech"):
            some = t["tp_units_evaluated"] / t["tp_units_total"]
            ref = oracle.ebur128(x.astype(np.float64), sr, True, True)
            assert abs(a["true_peak"] - ref["true_peak"]) <= 1e-9 * ref["true_peak"]
            assert np.allclose(a["TP"], ref["TP"][: a["TP"].size], rtol=1e-9, atol=0)
    assert some < 0.6                                                   # (the point of it)


@pytest.mark.parametrize("n", [1 << 20, (1 << 20) + 4799, 3_000_017])
def test_branch_and_bound_true_peak_inside_the_passes(engine, n):
    """Pass 1 on a signal long enough for the default threshold: the frame series the host logic reads (jt_frame_meta.true_peak per
    100 ms frame) and the pass's r128 block, with and without the option."""
    from jivetalking_amd import synth
    sr = 48000
    x = synth.speech_like(n / sr + 0.1, sr, seed=5)[:n].astype(np.float32)
    engine.upload_pcm(x, sr, 1)
    got = []
    for unpruned in (False, True):
        with options(engine, tp_unpruned=unpruned):
            a = engine.pass1(n, 4096, sr)
            got.append((a["r128"]["true_peak"], np.array([m["true_peak"] for m in a["meta"]]), a["r128"]["integrated"]))
    assert got[0][0] == got[1][0] and np.array_equal(got[0][1], got[1][1], equal_nan=True) and got[0][2] == got[1][2]
    t = engine.timers()
    assert t["tp_units_total"] >= n // 256


def test_branch_and_bound_true_peak_through_the_four_passes(engine):
    """The whole job (jt_process_audio) on three minutes of the bench talker with the default true peak and with the exhaustive kernels:
    the limiter plan (it reads Pass 2's true peak), every reported true peak (input, filtered, final, loudnorm's), the regions' and
    the delivered bytes are identical -- Pass 2 and Pass 4 analyse 48 kHz and 44.1 kHz signals above the default threshold (2^20
    samples), so both list kernels run."""
    from conftest import bench_talker
    from jivetalking_amd import hostlogic as H
    x = np.asarray(bench_talker(180.0, 48000, 1021, 40.0), np.float32)
    engine.upload_pcm(x, 48000, 1)
    got = []
    for unpruned in (False, True):
        with options(engine, tp_unpruned=unpruned):
            r = H.process_audio(engine)
            t = engine.timers()
            got.append((engine.download_s16(4).tobytes(), r.input.input_tp, r.filtered.r128.true_peak, r.final_.r128.true_peak, r.output_tp_db,
                        r.output_lufs, int(r.limiter.needed), r.limiter.limit if hasattr(r.limiter, "limit") else 0.0))
            if not unpruned:
                assert 0 < t["tp_units_evaluated"] < t["tp_units_total"]
    assert got[0] == got[1]
