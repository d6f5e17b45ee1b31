"""Round-2 behaviours of the C ABI: sticky cancellation (frame_processor.go:116-118), the publish discipline of
jt_process_file (file_write.go:13-53), the sample-format-dependent arithmetic libavfilter negotiates around the down-mix and the
band graphs, rate pairs whose exact phase count exceeds swresample's 1024-entry bank, and hostile FLAC."""
import glob
import os
import threading
import time

import numpy as np
import pytest

from jivetalking_amd import synth, hostlogic as H, _lib as L
from jivetalking_amd.engine import Engine, default_filter_params
from conftest import options

pytestmark = pytest.mark.gpu
SR = 48000


# ---------------------------------------------------------------- cancellation
def test_cancel_between_passes_is_sticky_and_handle_survives(engine):
    x = synth.speech_like(12.0, SR, seed=41)
    engine.upload_pcm(x, SR, 1)
    engine.pass1(x.size)
    engine.cancel()                                       # lands while the host would be running the VAD
    with pytest.raises(L.JtError) as ei:
        engine.pass2(default_filter_params())
    assert ei.value.code == L.JT_E_CANCELLED
    with pytest.raises(L.JtError) as ei:                  # still cancelled: the passes never clear the flag
        engine.pass1(x.size)
    assert ei.value.code == L.JT_E_CANCELLED
    with pytest.raises(L.JtError) as ei:
        H.process_audio(engine)
    assert ei.value.code == L.JT_E_CANCELLED
    engine.reset_cancel()
    r = H.process_audio(engine)                           # the handle is usable again, same input
    assert abs(r.output_lufs + 16.0) <= 0.1
    engine.cancel()
    engine.upload_pcm(x, SR, 1)                           # a new job clears it as well
    r2 = H.process_audio(engine)
    assert r2.output_lufs == r.output_lufs


def test_cancel_from_second_thread_mid_process(engine):
    x = np.tile(synth.speech_like(60.0, SR, seed=42), 10)  # 10 min: long enough that the call is still running when the cancel lands
    engine.upload_pcm(x, SR, 1)
    H.process_audio(engine)                               # warm (allocations, plans)
    t_full0 = time.perf_counter(); H.process_audio(engine); t_full = time.perf_counter() - t_full0
    hits = 0
    for delay in (0.0005, 0.002, 0.25 * t_full, 0.5 * t_full, 0.8 * t_full):
        engine.reset_cancel()
        th = threading.Timer(delay, engine.cancel)
        t0 = time.perf_counter(); th.start()
        try:
            H.process_audio(engine)
            done = True
        except L.JtError as e:
            assert e.code == L.JT_E_CANCELLED
            done = False
        el = time.perf_counter() - t0
        th.join()
        if not done:
            hits += 1
            assert el < delay + 0.6 * t_full + 0.05, (delay, el, t_full)     # prompt: at most the pass that was in flight
    assert hits >= 3
    engine.reset_cancel()
    assert abs(H.process_audio(engine).output_lufs + 16.0) <= 0.1


# ---------------------------------------------------------------- publish discipline
def _write_flac(tmp_path, engine, seconds=12.0, seed=43, name="talk.flac"):
    x = synth.speech_like(seconds, 44100, seed=seed)
    pcm = np.clip(np.rint(np.asarray(x, np.float64) * 32768), -32768, 32767).astype(np.int16)
    data = engine.op_flac_encode(pcm, 44100)
    p = tmp_path / name
    p.write_bytes(data)
    return str(p), pcm


def test_process_file_publishes_atomically_and_leaves_no_residue(tmp_path, engine):
    src, _ = _write_flac(tmp_path, engine)
    res, out_path, _ = H.process_file(engine, src)
    assert os.path.basename(out_path) == "talk-LUFS-16-processed.flac" and os.path.exists(out_path)
    good = open(out_path, "rb").read()
    assert sorted(os.listdir(tmp_path)) == ["talk-LUFS-16-processed.flac", "talk.flac"]      # no hidden temp left
    # every injected failure: an error, no temp residue, and the previous output untouched (file_write.go, processor.go:130-135)
    for kw in ({"create_temp": 1}, {"write": 1}, {"rename": 1}):
        H.inject_fault(**kw)
        try:
            with pytest.raises(L.JtError):
                H.process_file(engine, src)
        finally:
            H.inject_fault()
        assert sorted(os.listdir(tmp_path)) == ["talk-LUFS-16-processed.flac", "talk.flac"], kw
        assert open(out_path, "rb").read() == good
    # cancelled before the publish: JT_E_CANCELLED, nothing new on disk
    os.remove(out_path)
    th = threading.Timer(0.001, engine.cancel); th.start()
    try:
        H.process_file(engine, src)
        th.join()
    except L.JtError as e:
        th.join()
        assert e.code == L.JT_E_CANCELLED
        assert os.listdir(tmp_path) == ["talk.flac"]
    engine.reset_cancel()
    assert not glob.glob(str(tmp_path / ".processing-*"))


# ---------------------------------------------------------------- down-mix and band graphs per source format
def test_downmix_float_path_matches_oracle_bit_for_bit(engine, oracle):
    a = synth.speech_like(5.0, SR, seed=44); b = synth.speech_like(5.0, SR, seed=45) * 0.7
    st = np.empty(a.size * 2, np.float32); st[0::2] = a; st[1::2] = b
    engine.upload_pcm(st, SR, 2)
    ref = oracle.downmix_stereo(st, 0)
    p1 = engine.pass1(a.size)
    # per-decoder-frame sums come from the RAW interleaved samples (frameSumSquaresAndPeak), the analysis from the down-mix
    e = oracle.ebur128(ref.astype(np.float64), SR, True, True)
    assert abs(p1["r128"]["integrated"] - e["integrated"]) < 1e-6
    assert abs(p1["astats"]["max_level"] - float(ref.max())) == 0.0 and abs(p1["astats"]["min_level"] - float(ref.min())) == 0.0
    # not the integer-format 0.5/0.5 matrix: 3.01 dB apart
    half = (0.5 * a + 0.5 * b).astype(np.float32)
    assert abs(p1["r128"]["integrated"] - oracle.ebur128(half.astype(np.float64), SR, True, True)["integrated"] - 3.0103) < 0.01


@pytest.mark.parametrize("bits,mode", [(16, 1), (24, 2)])
def test_band_rms_integer_source_formats(engine, oracle, bits, mode):
    """A 16-bit source runs the band graphs' biquads in s16p (float state, truncated + clipped per stage, astats / INT16_MAX), a
    24-bit one in s32p (double state); stereo gets the integer-normalised down-mix.  1e-4 dB (halo restarts), quiet band included."""
    q = float(1 << (bits - 1))
    a = np.rint(np.asarray(synth.speech_like(9.0, SR, seed=46), np.float64) * q) / q
    b = np.rint(np.asarray(synth.speech_like(9.0, SR, seed=47), np.float64) * 0.6 * q) / q
    a = a.astype(np.float32); b = b.astype(np.float32)
    lo = [70.0, 1000.0, 6000.0, 13000.0]; hi = [100.0, 3000.0, 9000.0, 19000.0]
    for ch in (1, 2):
        if ch == 1:
            engine.upload_pcm(a, SR, 1); mono = a
        else:
            st = np.empty(a.size * 2, np.float32); st[0::2] = a; st[1::2] = b
            engine.upload_pcm(st, SR, 2); mono = oracle.downmix_stereo(st, mode)
        engine.set_source_format(bits, False)
        got, ok = engine.band_rms(1.0, 6.5, lo, hi)
        seg = mono[SR:SR + int(6.5 * SR)]
        for k in range(len(lo)):
            ref = oracle.band_rms_db_fmt(seg, SR, lo[k], hi[k], mode)
            assert ok[k] == 1 and abs(got[k] - ref) < 1e-4, (bits, ch, k, got[k], ref)
        if bits == 16 and ch == 1:
            # the integer path is audibly different from the float one on a quiet band: that is the point of restating it
            flt = oracle.band_rms_db_fmt(seg, SR, lo[0], hi[0], 0)
            assert abs(flt - oracle.band_rms_db_fmt(seg, SR, lo[0], hi[0], 1)) > 1e-3


def test_load_audio_sets_the_source_format(engine, oracle):
    x = synth.speech_like(6.0, 44100, seed=48)
    pcm = np.clip(np.rint(np.asarray(x, np.float64) * 32768), -32768, 32767).astype(np.int16)
    engine.load_audio(engine.op_flac_encode(pcm, 44100))
    got, _ = engine.band_rms(0.5, 4.0, [80.0], [125.0])
    f = (pcm.astype(np.float32) / 32768.0)[22050:22050 + 4 * 44100]
    assert abs(got[0] - oracle.band_rms_db_fmt(f, 44100, 80.0, 125.0, 1)) < 1e-4
    engine.upload_pcm(pcm.astype(np.float32) / 32768.0, 44100, 1)            # a float upload resets it
    got, _ = engine.band_rms(0.5, 4.0, [80.0], [125.0])
    assert abs(got[0] - oracle.band_rms_db_fmt(f, 44100, 80.0, 125.0, 0)) < 1e-4


# ---------------------------------------------------------------- rate pairs beyond the 1024-phase bank
@pytest.mark.parametrize("sr", [22050, 11025])
def test_true_peak_at_rates_with_more_than_1024_exact_phases(engine, oracle, sr):
    """22050 -> 192000 needs 1280 exact phases, 11025 -> 192000 2560: swresample keeps its 1024-entry bank and truncates the phase
    (resample.c, linear = 0).  Used to divide by zero.
    11025 is not a multiple of 10: f_ebur128.c slides a 4410-sample momentary window in steps of 1102 samples, the engine sums four
    1102-sample blocks (4408 samples) -- a 0.002 LU difference, stated here and in DESIGN.md section 3."""
    x = synth.speech_like(8.0, sr, seed=49)
    got = engine.op_ebur128(x, sr, True)
    ref = oracle.ebur128(x.astype(np.float64), sr, True, True)
    assert abs(got["integrated"] - ref["integrated"]) < (1e-6 if sr % 10 == 0 else 5e-3)
    assert abs(got["true_peak"] / ref["true_peak"] - 1.0) < 1e-9
    assert got["true_peak"] >= got["sample_peak"] * 0.999


def test_22050_hz_file(engine, oracle):
    """The reference's fixed 20.5 kHz band limit is above a 22.05 kHz file's Nyquist: af_biquads refuses to configure (EINVAL) and
    Pass 2 fails (adaptive_bandlimit_lowpass.go:17-20 documents the assumption, no guard).  Same here; analysis-only works, and with
    a band limit below Nyquist the whole pipeline runs (Pass-1 true peak through the truncated-phase resampler)."""
    x = synth.speech_like(30.0, 22050, seed=50)
    engine.upload_pcm(x, 22050, 1)
    a = H.process_audio(engine, analyse_only=True)
    e = oracle.ebur128(x.astype(np.float64), 22050, True, True)
    assert abs(a.input.input_i - e["integrated"]) < 0.002 and abs(a.input.input_tp - 20 * np.log10(e["true_peak"])) < 0.05
    with pytest.raises(L.JtError) as ei:
        H.process_audio(engine)
    assert ei.value.code == L.JT_E_INVAL and "Nyquist" in str(ei.value)
    # the granular passes with the band limit off: Pass 2 at 22.05 kHz (anlmdn K=132 S=44, afftdn 1024-point, 22.05 -> 44.1 kHz 2-phase)
    engine.pass1(x.size, sample_rate=22050)
    a2 = engine.pass2(default_filter_params(lp_enabled=0))
    p2 = engine.download_s16(2)
    assert p2.size == x.size * 2 and np.isfinite(a2["r128"]["integrated"]) and a2["r128"]["true_peak"] >= a2["r128"]["sample_peak"] * 0.999
    f = oracle.ebur128(p2.astype(np.float64) / 32768.0, 44100, True, True)
    assert abs(f["integrated"] - a2["r128"]["integrated"]) < 0.05          # the same signal before / after the 2x resample + s16


# ---------------------------------------------------------------- hostile FLAC
def _crc8(b):
    c = 0
    for v in b:
        c ^= v
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def test_forged_header_with_escaped_partitions_is_rejected_not_faulted(engine, oracle):
    """A frame header is easy to forge (CRC-8 only).  Block size 65535 with a residual partition escaped at 31 bits per sample would
    walk ~250 KB past a short file; the reader must stop at the padded end and report a damaged stream."""
    pcm = (1000 * np.random.default_rng(7).standard_normal(4096 * 3)).astype(np.int16)
    good = bytearray(engine.op_flac_encode(pcm, 44100))
    # STREAMINFO: allow block sizes up to 65535 so that the forged header is "legal"
    good[8 + 2] = 0xFF; good[8 + 3] = 0xFF
    first = 4 + 4 + 34
    # skip the VORBIS_COMMENT block
    pos = first
    blen = int.from_bytes(good[pos + 1:pos + 4], "big"); pos += 4 + blen
    # forged frame: sync, fixed blocksize code 7 (16-bit blocksize-1 follows), rate code from STREAMINFO (0), mono, 16 bit
    hdr = bytearray([0xFF, 0xF8, 0x70, 0x08, 0x00, 0xFF, 0xFE])
    hdr.append(_crc8(hdr))
    # subframe: LPC order 1 (type 32), no wasted bits; warm-up 16 bits; precision 1 -> 4 bits, shift 5 bits, coefficient; then
    # residual coding method 0, partition order 0, Rice parameter 15 = escape, width 31
    bits = "0" + "100000" + "0" + "0" * 16 + "0000" + "00000" + "0" + "00" + "0000" + "1111" + "11111"
    bits += "0" * ((8 - len(bits) % 8) % 8)
    body = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
    evil = bytes(good[:pos]) + bytes(hdr) + body + bytes(64)
    with pytest.raises(L.JtError) as ei:
        engine.op_decode_audio(evil)
    assert ei.value.code == L.JT_E_INVAL
    # the handle survives and still decodes a good stream
    good2 = engine.op_flac_encode(pcm, 44100)
    dec, _, _ = engine.op_decode_audio(good2)
    assert np.array_equal(dec[:, 0], pcm.astype(np.int32))
    # truncations of a valid stream at every region of the last frame
    for cut in (len(good2) - 1, len(good2) - 2, len(good2) - 700, pos + 9):
        with pytest.raises(L.JtError) as ei:
            engine.op_decode_audio(good2[:cut])
        assert ei.value.code == L.JT_E_INVAL


# ---------------------------------------------------------------- frame statistics, progress ticks, run record
def test_frame_stats_match_frameSumSquaresAndPeak(engine):
    """k_frame_stats against the Go loop it replaces (analyser_metrics.go:273-358): per decoder frame, sum(x^2) and max|x| over ALL
    interleaved samples of the frame, in double; mono and stereo, ragged last frame.  Sums 1e-12 relative (tree order), peaks exact."""
    for ch, secs in ((1, 3.3), (2, 2.1)):
        x = synth.speech_like(secs, SR, seed=70 + ch)
        if ch == 2:
            st = np.empty(x.size * 2, np.float32); st[0::2] = x; st[1::2] = 0.5 * np.roll(x, 3); raw = st
        else:
            raw = x
        engine.upload_pcm(raw, SR, ch)
        p1 = engine.pass1(x.size)
        spf = 4096 * ch
        nfr = (x.size + 4095) // 4096
        assert p1["frame_sumsq"].size == nfr
        r64 = raw.astype(np.float64)
        for f in (0, 1, nfr // 2, nfr - 1):
            seg = r64[f * spf:(f + 1) * spf]
            assert p1["frame_sumsq"][f] == pytest.approx(float(np.sum(seg * seg)), rel=1e-12)
            assert p1["frame_peak"][f] == float(np.max(np.abs(seg)))
        assert p1["n_input_frames"] == nfr


def test_progress_ticks_replay_the_reference_cadence(engine):
    """jt_process_audio_ticks: every 100th decoder frame of every pass (analyser.go:602-618, processor.go:320-335,
    normalise.go:292-301,1108-1117) + the 17 band ticks on 0.95..1.0 (analyser_band_runner.go:47-88), between the pass start / end
    events; Level = calculateFrameLevel of the frame at that position."""
    x = np.tile(synth.speech_like(30.0, SR, seed=81), 4)            # 120 s: 1407 decoder frames -> 15 ticks in Pass 1
    engine.upload_pcm(x, SR, 1)
    seen = []
    res = H.process_audio_with_progress(engine, lambda u: seen.append((u.pass_, u.pass_name.decode(), u.progress, u.level)), ticks=True)
    plain = []
    H.process_audio_with_progress(engine, lambda u: plain.append((u.pass_, u.pass_name.decode(), u.progress)))
    assert len(plain) == 8                                                               # start / end per pass, nothing else
    nfr = (x.size + 4095) // 4096
    est = x.size / 4096.0
    p1 = [s for s in seen if s[0] == 1 and s[1] == "Analysing"]
    ticks1 = p1[1:-1]
    assert p1[0][2] == 0.0 and p1[-1][2] == 1.0 and len(ticks1) == (nfr + 99) // 100
    for k, t in enumerate(ticks1):
        assert t[2] == pytest.approx(min(0.95, (100 * k) / est * 0.95)) and -70.0 <= t[3] <= 0.0
    bands = [s for s in seen if s[1] == "Analysing frequency bands"]
    assert len(bands) == 17 and bands[0][2] == pytest.approx(0.95 + 0.05 / 17) and bands[-1][2] == 1.0
    assert [b[2] for b in bands] == sorted(b[2] for b in bands)
    p2 = [s for s in seen if s[0] == 2]
    assert len(p2) == 2 + nfr // 100 and all(0.0 <= s[2] <= 1.0 for s in p2)
    m = engine.download_s16(4).size
    for pno, name in ((3, "Measuring"), (4, "Normalising")):
        pp = [s for s in seen if s[0] == pno]
        assert pp[0][2] == 0.0 and pp[-1][2] == 1.0 and len(pp) == 2 + ((m + 4095) // 4096) // 100
        assert all(s[2] <= 0.99 for s in pp[1:-1]) and [s[2] for s in pp] == sorted(s[2] for s in pp)
    # the Level of a Pass-4 tick is the VU level of that output frame
    out = engine.download_s16(4)
    fr = out[99 * 4096:100 * 4096].astype(np.float64) / 32768.0
    want = max(-70.0, min(0.0, 20 * np.log10(np.sqrt(np.mean(fr * fr)))))
    assert [s for s in seen if s[0] == 4][1][3] == pytest.approx(want, abs=1e-9)
    assert seen.index(bands[-1]) < seen.index(p2[0])                                      # order: Pass-1 ticks, bands, then Pass 2


def test_run_record_of_a_real_run(engine):
    """jt_host_run_record_json on a processed file: every stage block present, values are the result's own, interval_summary from the
    handle's interval series, sidecars one line per interval / candidate (runrecord_write.go:37-45)."""
    import json
    x = synth.speech_like(40.0, SR, seed=91)
    engine.upload_pcm(x, SR, 1)
    res = H.process_audio(engine)
    text = H.run_record_json(engine, res, input_file="talk-LUFS-16-processed.flac", version="dev", processed_at="2026-01-01T00:00:00Z",
                             duration_s=40.0, sample_rate_hz=SR, channels=1)
    rec = json.loads(text)
    assert rec["loudness"]["stages"]["input"]["integrated_lufs"] == res.input.input_i
    assert rec["loudness"]["stages"]["final"]["integrated_lufs"] == res.output_lufs and rec["normalisation"]["output_lufs"] == res.output_lufs
    assert rec["normalisation"]["loudnorm_measured"]["normalization_type"] == "linear" and rec["normalisation"]["within_target"] is True
    assert rec["noise"]["floor_source"] == "vad_percentile" and rec["filters"]["noise_reduction"]["afftdn_enabled"] in (True, False)
    lines = H.intervals_jsonl(engine).splitlines()
    assert rec["interval_summary"]["count"] == len(lines) > 100
    first = json.loads(lines[0])
    assert list(first)[:3] == ["timestamp", "rms_level", "peak_level"] and first["timestamp"] == 0 and "spectral_rolloff" in first
    d = rec["interval_summary"]["rms_distribution"]
    assert d["min_dbfs"] <= d["p10_dbfs"] <= d["p50_dbfs"] <= d["p90_dbfs"] <= d["max_dbfs"] and rec["interval_summary"]["largest_gap_db"] >= 0
    cands = H.candidates_jsonl(res).splitlines()
    assert len(cands) == res.input.n_candidates == rec["regions"]["speech"]["candidates_summary"]["evaluated_count"]
    if res.input.has_noise_profile:
        assert rec["regions"]["room_tone"]["elected"]["start_s"] == res.input.noise_profile.start_ns / 1e9
        assert set(rec["regions"]["room_tone"]["samples"]) == {"input", "filtered", "final"}


def test_granular_call_sequence_equals_the_one_call_mirror(tmp_path):
    """tools/granular_harness.cpp: the call sequence of the cgo shim at the reference's engine seam (upload, pass1, intervals, VAD,
    band RMS x2, AdaptConfig, pass2, regions, limiter plan, pass3, linear target, pass4, regions, download), compiled with g++ against
    include/*.h, must give exactly what jt_process_audio gives: the same s16 samples, statistics and strings."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "granular_harness")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "granular_harness.cpp"),
                           "-L", os.path.join(root, "jivetalking_amd", "lib"), "-ljtgpu", "-Wl,-rpath," + os.path.join(root, "jivetalking_amd", "lib"), "-o", exe])
    x = synth.speech_like(45.0, SR, seed=95)
    p = tmp_path / "in.f32"; x.astype("<f4").tofile(p)
    r = subprocess.run([exe, str(p), str(SR)], capture_output=True, text=True)
    assert r.returncode == 0 and "identical" in r.stdout, (r.returncode, r.stdout, r.stderr)


# ---------------------------------------------------------------------------------------------------------------------
# loudnorm, dynamic mode: the fallback the reference only notices from the stats (normalise.go:687-693)
# ---------------------------------------------------------------------------------------------------------------------
def _stream192(seconds, seed, level=1.0):
    from jivetalking_amd import synth
    return synth.speech_like(seconds, 192000, seed=seed).astype(np.float64) * level


def _ln_close(got, ref, gst, rst):
    # same frame gains and limiter decisions; the input meter's sums run in a different order (chunked K-weighting scan): 1e-9 relative
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), np.max(np.abs(got - ref))
    for k in ("input_i", "input_tp", "input_lra", "input_thresh", "output_i", "output_tp", "output_lra", "output_thresh", "target_offset"):
        assert abs(gst[k] - rst[k]) < 1e-6, (k, gst[k], rst[k])


@pytest.mark.gpu
def test_loudnorm_dynamic_first_pass_matches_oracle(engine, oracle):
    """No measured values (the filter's first pass): gains follow the input's short-term loudness; a hot signal keeps the
    look-ahead limiter busy (attack / sustain / release all exercised: the output peaks sit on the ceiling)."""
    x = _stream192(9.3, 31, 2.5)                       # a partial last frame (9.3 s is not a multiple of 100 ms at 192 kHz + resample lengths)
    x = x[: x.size - 777]
    ref, rst = oracle.loudnorm_dynamic(x, target_tp=-9.0)
    got, gst = engine.op_loudnorm_dynamic(x, target_tp=-9.0)
    assert rst["dynamic"] == 1 and gst["normalization_type_dynamic"] == 1
    assert abs(np.max(np.abs(ref)) - 10 ** (-9 / 20)) < 1e-12          # the limiter worked: peaks sit on the -9 dB ceiling
    _ln_close(got, ref, gst, rst)
    ref, rst = oracle.loudnorm_dynamic(x)                              # and the default ceiling, which this signal stays under
    got, gst = engine.op_loudnorm_dynamic(x)
    _ln_close(got, ref, gst, rst)


@pytest.mark.gpu
def test_loudnorm_dynamic_quiet_start_and_second_pass(engine, oracle):
    """A file that starts below the measured threshold: the gain ramps by 1.0058 per frame until the OUTPUT's short-term loudness
    reaches the target (the one feedback path of the filter); second pass with measured values whose LRA exceeds the target."""
    loud = _stream192(7.0, 32, 1.2)
    x = np.concatenate([loud[: 192000 * 4] * 0.004, loud])
    _, first = oracle.loudnorm_dynamic(x)
    measured = (first["input_i"], 25.0, first["input_tp"], first["input_thresh"])       # LRA above the 20 LU target: not linear
    ref, rst = oracle.loudnorm_dynamic(x, measured=measured, offset=0.37)
    got, gst = engine.op_loudnorm_dynamic(x, measured=measured, offset=0.37)
    assert rst["dynamic"] == 1
    _ln_close(got, ref, gst, rst)
    # exactly 3 s (no inner frame at all) and 3 s + one sample
    for n in (576000, 576001, 576000 + 19200):
        r2, s2 = oracle.loudnorm_dynamic(x[-n:])
        g2, t2 = engine.op_loudnorm_dynamic(x[-n:])
        _ln_close(g2, r2, t2, s2)


@pytest.mark.gpu
def test_loudnorm_dynamic_short_input_is_one_gain(engine, oracle):
    """Shorter than the 3 s the first frame asks for: the filter measures what it got and applies a single gain (and reports
    "linear")."""
    x = _stream192(2.0, 33, 0.8)
    ref, rst = oracle.loudnorm_dynamic(x)
    got, gst = engine.op_loudnorm_dynamic(x)
    assert rst["dynamic"] == 0 and gst["normalization_type_dynamic"] == 0
    _ln_close(got, ref, gst, rst)
    ratio = got[1000:1010] / x[1000:1010]
    assert np.allclose(ratio, ratio[0], rtol=1e-12)


@pytest.mark.gpu
def test_pass4_falls_back_to_dynamic_loudnorm_like_the_filter(engine, oracle):
    """jt_pass4 with second-pass values loudnorm rejects (measured LRA above the LRA target): up to 192 kHz, the dynamic filter, back to
    the source rate, then adeclick, brickwall and s16 as usual -- against the same chain composed from the oracle's pieces."""
    p2 = (synth.speech_like(6.5, 44100, seed=35) * 0.35 * 32767).astype(np.int16)
    engine.upload_s16(p2, 44100)
    first = engine.pass3(None, -16.0, -1.0, 20.0)
    ap = L.LoudnormApply(-16.0, -1.0, 20.0, round(first["input_i"], 2), round(first["input_tp"], 2), 25.0, round(first["input_thresh"], 2), 0.5,
                         1, 1.7, 55.0, 50.0, 1, 0.803526)
    _, st = engine.pass4(None, ap)
    assert st["normalization_type_dynamic"] == 1
    got = engine.download_s16(4)
    up = oracle.swr_f64(p2.astype(np.float64) / 32768.0, 44100, 192000, True)
    y192, ost = oracle.loudnorm_dynamic(up, -16.0, 20.0, -1.0, measured=(ap.measured_i, ap.measured_lra, ap.measured_tp, ap.measured_thresh), offset=0.5)
    assert ost["dynamic"] == 1
    down = oracle.swr_f64(y192, 192000, 44100, True)[: p2.size]
    z = oracle.alimiter(oracle.adeclick(down, 44100, 1.7, 55.0, 50.0, method="s"), 44100, 0.803526, 1.0, 50.0)
    ref = oracle.f64_to_s16(z.astype(np.float32).astype(np.float64))
    assert got.size == ref.size == p2.size
    d = np.abs(ref.astype(np.int32) - got.astype(np.int32))
    assert d.max() <= 1 and np.count_nonzero(d) <= max(4, got.size // 20000), (d.max(), np.count_nonzero(d))
    for k in ("input_i", "input_lra", "input_thresh", "output_i", "output_lra", "output_thresh"):
        assert abs(st[k] - ost[k]) < 1e-3, (k, st[k], ost[k])
    assert abs(st["output_tp"] - ost["output_tp"]) < 1e-6


@pytest.mark.gpu
def test_process_audio_delivers_a_file_whose_lra_prints_as_zero(engine):
    """A steady tone: loudness range 0.00 in the first-pass JSON, which af_loudnorm's init() treats as "not measured".  The reference
    delivers such a file through the dynamic fallback (and says so in its stats); so does jt_process_audio."""
    t = np.arange(int(SR * 12.0)) / SR
    x = (0.05 * np.sin(2 * np.pi * 220.0 * t) * (1.0 + 0.0 * t)).astype(np.float32)
    engine.upload_pcm(x, SR, 1)
    base = H.default_config()
    res = H.process_audio(engine, base)
    assert res.measure.input_lra == 0.0
    assert res.loudnorm.normalization_type_dynamic == 1
    assert engine.download_s16(4).size == int(np.ceil(x.size * 147 / 160))


@pytest.mark.gpu
@pytest.mark.parametrize("sr,secs", [(48000, 23.7), (44100, 9.3), (48000, 0.31), (96000, 11.9), (88200, 2.2)])
def test_afftdn_grouped_kernel_equals_the_frame_at_a_time_kernel(engine_ab, sr, secs):
    """k_afftdn_grp (eight 2048-point or four 4096-point frames per workgroup, one wave per frame's transforms) takes every sum in k_afftdn's order: static floor,
    custom band profile and both sweeps of tn=1 must agree bit for bit (filters.go:563-600 chain; option afftdn_old of the A/B build selects the old kernel)."""
    engine = engine_ab
    r = np.random.default_rng(3)
    n = int(sr * secs)
    x = (0.05 * r.standard_normal(n) * (0.1 + 0.9 * (np.sin(np.arange(n) * 1.3e-4) > 0)) + 0.01 * np.sin(np.arange(n) * 0.02)).astype(np.float32)
    def both(**kw):
        with options(engine, afftdn_old=True):
            a = engine.op_afftdn(x, sr, 12.0, -50.0, **kw)
        return a, engine.op_afftdn(x, sr, 12.0, -50.0, **kw)
    for kw in ({}, {"track": True}, {"band_noise": [-38.0 - 1.5 * i for i in range(15)]}):
        a, b = both(**kw)
        assert np.array_equal(a, b), kw
    assert np.max(np.abs(a - x)) > 1e-4                       # (the filter did something)


@pytest.mark.gpu
@pytest.mark.parametrize("sr,secs", [(44100, 6.0), (88200, 2.5), (22050, 4.0), (48000, 3.0)])
def test_anlmdn_hop_pair_kernel_at_rates_that_do_not_fill_the_lane_layout(engine, oracle, sr, secs):
    """anlmdn=s=0.00001:p=0.006:r=0.002:m=3 (filters.go:95-100) at 44.1 kHz is K = 265 (odd) and S = 88: the wave-per-hop-pair kernel
    runs it on the 96-offset lane layout with dummy end offsets and a short last block.  Against the generic kernel (one thread per
    offset, FFmpeg's sequential sums; option nlm_generic) the outputs agree to f32 round-off of the weighted mean, and both match the oracle."""
    x = synth.speech_like(secs, sr, seed=21).astype(np.float32)
    with options(engine, nlm_generic=True):
        g = engine.op_anlmdn(x, sr)
    f = engine.op_anlmdn(x, sr)
    assert np.max(np.abs(f - g)) < 2e-8
    assert np.max(np.abs(f - x)) > 1e-5                       # (the filter engaged)
    assert np.max(np.abs(f - oracle.anlmdn(x, sr))) < 1e-5


def _bursty_speech(sr, secs, seed):
    """speech-like signal with plosive-like bursts: after the loudnorm gain its peaks would pass the ceiling, so the plan needs the prefix"""
    x = synth.speech_like(secs, sr, seed=seed).astype(np.float32)
    r = np.random.default_rng(seed)
    t = np.arange(960)
    burst = (0.35 * np.hanning(960) * np.sin(2 * np.pi * 180.0 * t / sr)).astype(np.float32)
    for p in r.integers(sr, x.size - sr, size=max(4, int(secs / 60.0 * 60))):
        x[p:p + 960] += burst
    return x


@pytest.mark.gpu
def test_pass4_takes_the_limiter_prefix_pass3_left_behind(engine):
    """Pass 3 (normalise.go:256-264) and Pass 4 (normalise.go:452-497) run the same alimiter prefix on the same Pass-2 output; Pass 4
    reuses Pass 3's result instead of recomputing it.  The delivered samples must not change (option no_lim_keep recomputes)."""
    sr = 48000
    x = _bursty_speech(sr, 75.0, 5)
    outs = []
    for keep in (True, False):
        with options(engine, no_lim_keep=not keep):
            engine.upload_pcm(x, sr, 1)
            r = H.process_audio(engine, H.default_config(), 4096)
            assert r.limiter.needed == 1
            y = np.empty(x.size, np.int16)
            got = engine.download_s16_into(4, y)
            outs.append(y[:got].copy())
    assert outs[0].size == outs[1].size and np.array_equal(outs[0], outs[1])


@pytest.mark.gpu
def test_f64_stream_upsampler_with_eight_waves_equals_the_four_wave_kernel(engine_ab):
    """The limiter-prefix measurement resamples the limited f64 signal to 192 kHz (swr DBLP); k_upsample32_stream8 must give the
    statistics the four-wave kernel gives, to the bit (option ups_no_stream8 of the A/B build selects the latter; p3_unfused: the stand-alone
    upsampler, which the fused Pass-3 sweep otherwise replaces)."""
    engine = engine_ab
    r = np.random.default_rng(8)
    x = np.clip(np.round(0.2 * r.standard_normal(44100 * 21) * 32768.0), -32768, 32767).astype(np.int16)
    lim = L.LimiterPlan(1, 0.25, 1.0)
    with options(engine, p3_unfused=True):
        a = engine.op_loudnorm_measure_s16(x, 44100, limiter=lim)
        with options(engine, ups_no_stream8=True):
            b = engine.op_loudnorm_measure_s16(x, 44100, limiter=lim)
    assert all(np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True) for k in a)


@pytest.mark.gpu
def test_nothing_reads_memory_it_did_not_write():
    """Every operator and the pipeline on a short file after a long, loud one on the same handle equal the fresh-handle results, with every
    device allocation filled with 0xFF bytes beforehand (the process-wide option poison_alloc: the check runs in a process of its own)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stale_memory.py"), "--poison"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "stale-memory check: clean (poisoned allocations)" in r.stdout, r.stdout[-2000:]
