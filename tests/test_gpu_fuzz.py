"""Differential fuzzing and beyond-bench sizes UNDER THE DRIVER'S EYES.

Fixed, seeded, bounded slices of what tools/fuzz_*.py, tools/long_dynamic.py and tools/long_pipeline.py do by hand: random files through
the HIP path (C ABI) against the CPU oracle for every pass and for the decision chain, the FLAC legs on random material, a handle pool on a
random mix, and three sizes beyond the bench's (a 3-hour linear-mode file, a 2-hour dynamic-mode stream, configs[4] at sixty minutes).
Every bar is stated in its test.  The reference behaviour held: processor.go:78-216 (ProcessAudio) on arbitrary inputs.

The oracle is sequential C (one core); ctypes releases the GIL around it, so the oracle side of independent files runs side by side on
the host's cores while the one GPU handle moves on to the next file.  The GPU side is always one call through the C ABI per file.
"""
import ctypes as C
import hashlib
import os
import shutil
import struct
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from jivetalking_amd import synth, hostlogic as H, _lib as L
from conftest import options, bench_talker

pytestmark = pytest.mark.gpu
SR = 48000
_POOL = None


def _pool():
    """Threads for the oracle side (liboracle.so's lazily built histogram tables are touched once before the first job)."""
    global _POOL
    if _POOL is None:
        from oracle import orc
        orc.ebur128(np.zeros(48000, np.float64), 48000, True, True)
        orc.loudnorm_measure(np.zeros(192000, np.float64), 192000, True)
        _POOL = ThreadPoolExecutor(max_workers=max(4, min(48, (os.cpu_count() or 8) - 2)))
    return _POOL


def _parse(spec):
    out = []
    for f in spec.split(","):
        name, _, args = f.partition("=")
        out.append((name, dict(a.partition("=")[::2] for a in (args.split(":") if args else []))))
    return out


def _close(a, b, rel=2e-3, abs_=2e-4):
    try:
        fa, fb = float(a), float(b)
    except ValueError:
        return a == b
    return abs(fa - fb) <= max(abs_, rel * max(abs(fa), abs(fb)))


def _room(rng, x, kind, lo=-3.75, hi=-2.0):
    """kinds 1-3 of the tools' generators: white or low-passed room tone between 10^lo and 10^hi of full scale, (3) long pauses."""
    if kind in (1, 2, 3):
        nz = rng.standard_normal(x.size)
        if kind == 2:
            nz = np.convolve(nz, np.ones(24) / 24, mode="same") * 4
        x += nz * float(10 ** rng.uniform(lo, hi))
    return x


# ------------------------------------------------------------------------------------------------ decision chain (tools/fuzz_decisions.py)
def test_fuzz_decision_chain_24_random_files(engine, oracle):
    """24 random 30-60 s files (12 at 48 kHz, 12 at 44.1 kHz: speech at different levels, white / coloured room tone from -75 to -40
    dBFS, long pauses): the reference's decision chain (analyser.go:571-638 intervals -> analyser_vad.go VAD / elections -> band graphs ->
    adaptive*.go AdaptConfig -> filters.go chain string) on the HIP path's Pass 1 and on the CPU oracle's Pass 1.  Bar: the same
    elections on the same 250 ms intervals, the same switches, the same filter list, every printed parameter within 2e-3 relative
    (afftdn's band profile within its printed 0.1 dB)."""
    import oracle_pass1 as P
    rng = np.random.default_rng(605)
    jobs = []
    for c in range(24):
        sr = (48000, 44100)[c % 2]
        secs = float(rng.uniform(30.0, 60.0))
        x = np.asarray(synth.speech_like(secs, sr, seed=int(rng.integers(1, 10**6))), np.float64) * float(10 ** rng.uniform(-1.0, 0.2))
        kind = int(rng.integers(0, 4))
        x = _room(rng, x, kind)
        if kind == 3:
            for _ in range(int(rng.integers(1, 4))):
                a = int(rng.integers(0, x.size - 6 * sr)); x[a: a + int(rng.uniform(1.5, 5.0) * sr)] *= 0.003
        x = np.clip(x, -1, 1).astype(np.float32)
        engine.upload_pcm(x, sr, 1)
        g = H.process_audio(engine, analyse_only=True)
        jobs.append((c, sr, kind, x, g, _pool().submit(P.oracle_pass1, oracle, x, sr)))
    bad = []
    for c, sr, kind, x, g, fut in jobs:
        m, eff, spec = P.decide(oracle, x, sr, pass1=fut.result())
        gm = g.input
        issues = []
        if (gm.has_speech_profile, gm.has_noise_profile, gm.voice_activated, gm.floor_source, gm.n_candidates, gm.n_speech_regions) != \
           (m.has_speech_profile, m.has_noise_profile, m.voice_activated, m.floor_source, m.n_candidates, m.n_speech_regions):
            issues.append("elections / switches")
        if m.has_speech_profile and gm.has_speech_profile and (gm.speech_profile.region.start_ns, gm.speech_profile.region.duration_ns) != (m.speech_profile.region.start_ns, m.speech_profile.region.duration_ns):
            issues.append("speech region")
        if m.has_noise_profile and gm.has_noise_profile and (gm.noise_profile.start_ns, gm.noise_profile.duration_ns) != (m.noise_profile.start_ns, m.noise_profile.duration_ns):
            issues.append("noise region")
        cg, co = _parse(H.filter_spec(g.effective, 2)), _parse(spec)
        if [f[0] for f in cg] != [f[0] for f in co]:
            issues.append("filter list")
        else:
            for (name, a), (_, b) in zip(cg, co):
                if a.keys() != b.keys():
                    issues.append(name + " keys"); continue
                for k in a:
                    if name == "afftdn" and k == "bn":
                        va, vb = [float(v) for v in a[k].split("|")], [float(v) for v in b[k].split("|")]
                        if len(va) != len(vb) or max(abs(p - q) for p, q in zip(va, vb)) > 0.1001:
                            issues.append("afftdn bn")
                    elif not _close(a[k], b[k]):
                        issues.append(f"{name}.{k} {a[k]} vs {b[k]}")
        if issues:
            bad.append((c, sr, kind, issues))
    assert not bad, bad


# ------------------------------------------------------------------------------------------------ Pass 2 (tools/fuzz_pass2.py)
def _sib_talker(seconds, sr, seed, gain):
    """A talker whose sibilants sit in 6.75-8.25 kHz (synth.speech_like_torch(sib_band=True)): AdaptConfig switches the de-esser on
    (adaptive_deesser.go:45).  Generated by a child process (torch), cached for the session."""
    return np.asarray(bench_talker(seconds, sr, seed, 0.0, gain, sib_band=True), np.float64)


def test_fuzz_pass2_24_random_files_against_the_oracle_chain(engine, oracle):
    """24 random files (12-24 s; the sibilant talkers 40-50 s so that a speech profile is elected) -- 48 / 44.1 / 96 kHz, mono and stereo
    with L != R (the float rematrix first, filters.go:607-615), room tone of different levels and colours, a pause, and six talkers whose
    sibilance sits within 6 dB of the body band so that AdaptConfig switches the de-esser on -- through jt_process_audio;
    the delivered Pass-2 s16 against the reference's Pass-2 chain (processor.go:255-373: highpass, lowpass, anlmdn, afftdn, agate,
    acompressor, deesser, dbl -> flt, aresample 44.1 kHz s16) composed from the CPU oracle with the parameters the host logic printed.
    Bar (the suite's): <= 3 LSB of s16 anywhere, < 0.3 LSB on average (afftdn's f32 transform schedule); the de-esser must be on in >= 2 files."""
    from test_gpu_pipeline import oracle_pass2
    rng = np.random.default_rng(609)
    jobs = []
    rates = [48000, 44100, 96000, 48000, 44100, 48000]
    for c in range(24):
        sr = rates[c % 6]
        ch = 2 if c % 3 == 2 else 1
        secs = float(rng.uniform(12.0, 24.0)) * (0.6 if sr == 96000 else 1.0)
        kind = 4 if (c % 6 == 3 or c % 12 == 4) else int(rng.integers(0, 4))
        seed = int(rng.integers(1, 10**6))
        if kind == 4:
            base = _sib_talker(60.0, sr, 4000 + sr // 1000, 1.4 if sr == 48000 else 2.0)
            secs = float(rng.uniform(40.0, 50.0))
            a = int(rng.integers(0, base.size - int(secs * sr)))
            x = base[a: a + int(secs * sr)].copy()
        else:
            x = np.asarray(synth.speech_like(secs, sr, seed=seed, speech_dbfs=float(rng.uniform(-36, -24)), room_dbfs=float(rng.uniform(-75, -50))), np.float64)
        x *= float(10 ** rng.uniform(-1.0, 0.2))
        x = _room(rng, x, kind, -3.75, -2.2)
        if kind == 3:
            a = int(rng.integers(0, x.size - 5 * sr)); x[a: a + int(rng.uniform(1.5, 4.0) * sr)] *= 0.003
        x = np.clip(x, -1, 1).astype(np.float32)
        if ch == 2:
            other = (np.roll(x, int(rng.integers(1, 200))) * np.float32(rng.uniform(0.3, 1.0))).astype(np.float32)
            st = np.empty(x.size * 2, np.float32); st[0::2] = x; st[1::2] = other
            engine.upload_pcm(st, sr, 2)
            x = oracle.downmix_stereo(st, 0)
        else:
            engine.upload_pcm(x, sr, 1)
        res = H.process_audio(engine)
        p2 = engine.download_s16(2).copy()
        fp = L.FilterParams(); H.lib().jt_host_filter_params(C.byref(res.effective), C.byref(fp))
        jobs.append((c, sr, ch, kind, int(fp.deess_enabled), p2, _pool().submit(oracle_pass2, oracle, x, fp, sr)))
    bad = []; worst = 0; wmean = 0.0; deess = 0
    for c, sr, ch, kind, de, p2, fut in jobs:
        _, ref = fut.result()
        deess += de
        if ref.size != p2.size:
            bad.append((c, sr, ch, kind, "length", ref.size, p2.size)); continue
        d = np.abs(ref.astype(np.int32) - p2.astype(np.int32))
        worst = max(worst, int(d.max())); wmean = max(wmean, float(d.mean()))
        if d.max() > 3 or d.mean() >= 0.3:
            bad.append((c, sr, ch, kind, int(d.max()), float(d.mean())))
    print(f"pass-2 fuzz: worst max {worst} LSB, worst mean {wmean:.4f} LSB, de-esser on in {deess} of 24")
    assert not bad, bad
    assert deess >= 2, deess


# ------------------------------------------------------------------------------------------------ Pass 3 + Pass 4 (tools/fuzz_pass4_chain.py)
def test_fuzz_pass4_24_random_files_against_the_oracle_chain(engine, oracle):
    """24 random 18-32 s files whose levels, plosive bursts, hiss bursts and quiet lead-ins are drawn so that the plain linear branch, the
    limiter prefix (normalise.go:373-465) and af_loudnorm's dynamic mode (normalise.go:683-693) all occur: Pass 3 + Pass 4 on the GPU
    against the reference's Pass-4 graph composed from the CPU oracle (oracle/chain.py: volume, alimiter, loudnorm linear or dynamic at
    192 kHz, adeclick, brickwall alimiter, s16) on the GPU run's Pass-2 output and the spec string the host logic printed.
    Bars: the oracle takes the same branch; linear-mode files (with or without the prefix): ZERO s16 samples differ; dynamic-mode files:
    <= 8 samples differ, each by 1 LSB (f64 envelope products at the .5 tie of the s16 rounding); the landing equal at the printed 0.01.
    At least one file of every branch must have occurred."""
    from oracle import chain
    rng = np.random.default_rng(603)
    jobs = []
    for c in range(24):
        secs = float(rng.uniform(18.0, 32.0))
        x = np.asarray(synth.speech_like(secs, SR, seed=int(rng.integers(1, 10**6))), np.float64) * float(10 ** rng.uniform(-1.2, 0.3))
        kind = c % 4
        if kind == 1:                                            # plosive bursts: the limiter prefix
            w = int(0.02 * SR); b = float(rng.uniform(0.2, 0.6)) * np.hanning(w) * np.sin(2 * np.pi * 180.0 * np.arange(w) / SR)
            for pos in range(SR, x.size - SR, int(rng.uniform(0.7, 2.5) * SR)):
                x[pos:pos + w] += b
        elif kind == 2:                                          # hiss bursts far above the speech: linear mode impossible
            for pos in range(2 * SR, x.size - 2 * SR, int(rng.uniform(3, 8) * SR)):
                n = int(rng.uniform(0.05, 0.4) * SR); x[pos:pos + n] += rng.standard_normal(n) * float(rng.uniform(0.1, 0.5))
        elif kind == 3:                                          # a quiet lead-in
            x[: int(rng.uniform(2, 8) * SR)] *= 10 ** rng.uniform(-3, -1.5)
        x = np.clip(x, -1.0, 1.0)
        engine.upload_pcm(x.astype(np.float32), SR, 1)
        res = H.process_audio(engine)
        p2, p4 = engine.download_s16(2).copy(), engine.download_s16(4).copy()
        spec = bytes(res.pass4_spec).split(b"\0")[0]
        branch = ("dynamic" if res.loudnorm.normalization_type_dynamic else "linear") + ("+prefix" if res.limiter.needed else "")

        def job(p2=p2, spec=spec):
            ref = chain.pass4(p2, 44100, spec)
            return ref["s16"], ref["dynamic"], chain.landing(ref["s16"], 44100)
        jobs.append((c, kind, branch, p4, float(res.output_lufs), int(res.loudnorm.normalization_type_dynamic), _pool().submit(job)))
    bad = []; tally = {}
    for c, kind, branch, p4, lufs, dyn, fut in jobs:
        ref, rdyn, land = fut.result()
        tally[branch] = tally.get(branch, 0) + 1
        if ref.size != p4.size or rdyn != dyn:
            bad.append((c, kind, branch, "branch / length", rdyn, dyn)); continue
        d = np.abs(ref.astype(np.int32) - p4.astype(np.int32))
        nd = int(np.count_nonzero(d))
        ok = (nd == 0) if not dyn else (nd <= 8 and d.max() <= 1)
        if not ok or abs(land["output_lufs"] - lufs) > 0.011:
            bad.append((c, kind, branch, nd, int(d.max()), land["output_lufs"], lufs))
    print("pass-4 fuzz branches:", tally)
    assert not bad, bad
    assert any(k.startswith("dynamic") for k in tally) and "linear" in tally and "linear+prefix" in tally, tally


# ------------------------------------------------------------------------------------------------ dynamic mode, stream path (tools/fuzz_dynamic_stream.py)
def test_fuzz_dynamic_loudnorm_stream_path_64_random_streams(engine):
    """af_loudnorm's dynamic mode on 64 random 192 kHz streams of 8-24 s (level, ceiling, offset, quiet stretches, isolated spikes around
    the ring-end corner, clipped plateaus, ragged lengths, measured values that open the above_threshold phase): the stream path (gains,
    peak list, one wave walking the limiter machine, segments applied by the whole GPU) against the one-workgroup kernel that walks the
    filter's own state machine sample by sample (held against the oracle at 1e-9 in tests/test_gpu_round2.py / round4 / round5).
    Bar: every output sample and every statistic IDENTICAL; the stream path must have carried frames in >= 32 of the 64."""
    rng = np.random.default_rng(601)
    base = [synth.speech_like(24.0, 192000, seed=s).astype(np.float64) for s in (101, 102, 103)]
    bad = []; carried = 0
    for c in range(64):
        x = base[c % 3][: int(192000 * rng.uniform(8.0, 24.0)) - int(rng.integers(0, 19200))].copy() * float(10 ** rng.uniform(-0.5, 1.0))
        kind = int(rng.integers(0, 5))
        if kind == 1:
            for _ in range(int(rng.integers(1, 4))):
                a = int(rng.integers(0, x.size - 192000)); x[a: a + int(rng.integers(19200, 4 * 192000))] *= 10 ** rng.uniform(-4, -1)
        elif kind == 2:
            x *= 0.01
            for t in rng.integers(600000, x.size - 50000, size=int(rng.integers(3, 40))):
                x[t] = rng.uniform(0.3, 0.9) * rng.choice([-1, 1])
                if rng.random() < 0.5:
                    x[t + int(rng.integers(19150, 19250))] = rng.uniform(0.3, 0.9)
        elif kind == 3:
            x = np.clip(x, -0.5, 0.5)
        tp = float(rng.uniform(-20.0, -0.5)); off = float(rng.choice([0.0, rng.uniform(-6, 12)]))
        meas = None if rng.random() < 0.6 else (float(rng.uniform(-30, -14)), 7.0, -2.0, float(rng.uniform(-45, -25)))
        with options(engine, ln_no_stream=True):
            want, wst = engine.op_loudnorm_dynamic(x, target_tp=tp, offset=off, measured=meas)
        got, gst = engine.op_loudnorm_dynamic(x, target_tp=tp, offset=off, measured=meas)
        carried += int(engine.timers()["ln_stream_frames"]) > 0
        if not (np.array_equal(got, want) and gst == wst):
            bad.append((c, kind, tp, off, meas, x.size, int(np.count_nonzero(got != want))))
    assert not bad, bad
    assert carried >= 32, carried


# ------------------------------------------------------------------------------------------------ FLAC legs (tools/fuzz_flac.py)
def test_fuzz_flac_200_random_cases(engine, oracle):
    """(1) 100 random mono s16 signals (speech, white noise, a silence run with one event, full-scale squares, DC, sparse impulses, a
    random walk; 1 .. 300 000 samples; five rates) through the GPU encoder (encoder.go:54-110's format): the image decodes to the same
    samples in the oracle's RFC 9639 decoder (every CRC checked) and in the GPU decoder, and the STREAMINFO MD5 is the decoder's.
    (2) 100 streams from the oracle's coverage encoder (random predictor / residual coding / block-size modes incl. 1152 and 4608,
    1-2 channels, all four stereo decorrelations, 8-24 bits, wasted bits) decode to the same samples on the GPU, through the one-walk
    mono kernel and through the two-walk path.  Bar: bit-exact, 200 of 200."""
    rng = np.random.default_rng(602)
    bad = []
    for c in range(100):
        n = int(rng.integers(1, 300000)); kind = c % 7; rate = int(rng.choice([44100, 48000, 8000, 96000, 22050]))
        if kind == 0: x = (np.asarray(synth.speech_like(n / 44100 + 0.1, 44100, seed=int(rng.integers(1, 10**6))))[:n] * 32768 * 10 ** rng.uniform(-1, 0.5)).clip(-32768, 32767)
        elif kind == 1: x = rng.integers(-32768, 32768, n)
        elif kind == 2: x = np.zeros(n); a = int(rng.integers(0, n)); x[a: a + int(rng.integers(1, 5000))] = rng.integers(-3000, 3000)
        elif kind == 3: x = np.where((np.arange(n) // int(rng.integers(1, 300))) % 2, 32767, -32768)
        elif kind == 4: x = np.full(n, int(rng.integers(-32768, 32768)))
        elif kind == 5: x = np.zeros(n); x[rng.integers(0, n, size=max(1, n // 5000))] = rng.integers(-32768, 32768, size=max(1, n // 5000))
        else: x = (rng.standard_normal(n).cumsum() * 50).clip(-32768, 32767)
        x = np.asarray(x, np.int16)
        data = engine.op_flac_encode(x, rate, md5=True)
        rc, dec, info = oracle.flac_decode(data)
        g, _, meta = engine.op_decode_audio(data)
        if not (rc == 0 and np.array_equal(dec[:, 0], x) and np.array_equal(g[:, 0], x.astype(np.int32)) and bytes(info.md5_stored) == bytes(info.md5_decoded)):
            bad.append(("encode", c, kind, n, rate, rc))
    for c in range(100):
        ch = int(rng.choice([1, 1, 2])); bps, order = [(8, 3), (16, 8), (24, 32), (16, 12), (20, 5), (12, 2)][c % 6]
        mode = int(rng.choice([0, 1, 2])) | (8 if rng.random() < 0.3 else 0) | (16 if rng.random() < 0.3 else 0) | (32 if rng.random() < 0.4 else 0)
        if ch == 2:
            mode |= int(rng.choice([0, 64, 128, 192]))
        n = int(rng.integers(100, 60000)); bs = int(rng.choice([16, 192, 576, 1152, 4096, 4608]))
        x = (rng.standard_normal((n, ch)).cumsum(0) * (1 << (bps - 6)) / 30).clip(-(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int32)
        if mode & 8:
            x[: min(n, 512)] &= ~7
        data = oracle.flac_encode(x, 44100, bps, bs, mode, order)
        g, _, _ = engine.op_decode_audio(data)
        with options(engine, flac_no_ahead=True):
            g2, _, _ = engine.op_decode_audio(data)
        if not (np.array_equal(g, x) and np.array_equal(g2, x)):
            bad.append(("decode", c, ch, bps, order, mode, bs, n))
    assert not bad, bad


# ------------------------------------------------------------------------------------------------ cadence and layouts (tools/fuzz_cadence.py)
def test_fuzz_cadence_and_layouts_48_random_files(engine, oracle):
    """Round 6's input leg on random material (a fixed slice of tools/fuzz_cadence.py): 48 files of random container (FLAC with block sizes
    192 .. 4608, fixed and variable; WAV u8 / s16 / s24 / f32 / f64, plain and WAVE_FORMAT_EXTENSIBLE), rate (22.05 .. 96 kHz), 1 .. 8
    channels in random layouts, 3 .. 40 s.  Per file: the decoder-frame cadence the library reports = the rule's (FLAC: the stream's
    frames; WAV: wavdec's 4096-byte packets of whole sample blocks); the interval series of jt_analyse_only(frame_samples = 0) =
    analyser.go:588-600 on those frames (count, timestamps, per-interval RMS to 1e-9 dB); astats' Min / Max level of the down-mix
    BIT-identical to the oracle's libswresample matrix for the layout; and for every fourth file the whole job twice (identical bytes) with
    the landing re-measured by the oracle's ebur128 (0.011 LU).  Bar: 48 of 48."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_cadence
    bad, failures = fuzz_cadence.run(engine, 48, 606, verbose=False)
    assert bad == 0, failures


# ------------------------------------------------------------------------------------------------ handle pool on a random mix (tools/fuzz_pool.py)
def _wav16(x16, rate, ch):
    payload = x16.astype("<i2").tobytes(); align = 2 * ch
    fmt = struct.pack("<HHIIHH", 1, ch, rate, rate * align, align, 16)
    body = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body


def test_fuzz_pool_random_mix_of_files(engine, oracle):
    """runBoundedPool's contract (cmd/jivetalking/pool.go:122-228: one file's failure does not stop the others; every file's outcome is
    its own) on a random mix through a handle pool: 20 files -- mono and stereo FLAC, 16 / 24 bit, WAV, 1 s .. 4 min, a talker that takes
    the dynamic mode, silence, a corrupted frame, a truncated file, a file that is not audio.  Bar: every file's outcome -- the bytes of
    its output file, or its error code -- is what jt_process_file gives for that file ALONE on a fresh handle, in both batches of a pool
    of eight one-stream handles and of a pool of three; no temp-file residue."""
    rng = np.random.default_rng(604)
    d = tempfile.mkdtemp(prefix="jtfz", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        paths = []
        long_talk = {r: np.asarray(bench_talker(240.0, r, 2200 + r // 1000, 40.0), np.float64) for r in (48000, 44100)}
        hot = np.asarray(bench_talker(120.0, 48000, 2300, 40.0, 4.0), np.float64)
        for k in range(20):
            kind = k % 10; rate = (48000, 44100)[int(rng.integers(0, 2))]
            secs = float(np.exp(rng.uniform(np.log(1.0), np.log(240.0))))
            if kind == 1:
                x, rate = hot[: int(min(secs, 120.0) * 48000)], 48000
            else:
                x = long_talk[rate][: int(secs * rate)]
            x = x * float(10 ** rng.uniform(-0.7, 0.2))
            x16 = np.clip(np.rint(x * 32768), -32768, 32767).astype(np.int16)
            name = os.path.join(d, f"f{k:03d}")
            if kind == 2:
                st = np.stack([x16, (x16 * 0.6).astype(np.int16)], axis=1).astype(np.int32)[: 20 * rate]
                data, name = oracle.flac_encode(st, rate, 16, 4096, 2 | 64, 8), name + ".flac"
            elif kind == 3:
                data, name = oracle.flac_encode((x16.astype(np.int32) << 8)[: 20 * rate, None], rate, 24, 4096, 2, 8), name + ".flac"
            elif kind == 4: data, name = _wav16(x16, rate, 1), name + ".wav"
            elif kind == 5: data, name = engine.op_flac_encode(np.zeros(int(rate * min(secs, 30)), np.int16), rate, md5=True), name + ".flac"
            elif kind == 6:
                data = bytearray(engine.op_flac_encode(x16, rate, md5=True)); data[len(data) // 2] ^= 0x20; data, name = bytes(data), name + ".flac"
            elif kind == 7:
                data = engine.op_flac_encode(x16, rate, md5=True); data, name = data[: len(data) * 2 // 3], name + ".flac"
            elif kind == 8: data, name = b"this is not audio" * 50, name + ".flac"
            else: data, name = engine.op_flac_encode(x16, rate, md5=True), name + ".flac"
            with open(name, "wb") as f:
                f.write(data)
            paths.append(name)
        from jivetalking_amd import Engine
        want = []
        with Engine(0) as e1:
            for p in paths:
                try:
                    res, out, _ = H.process_file(e1, p, md5=True)
                    want.append((0, hashlib.md5(open(out, "rb").read()).hexdigest())); os.unlink(out)
                except L.JtError as ex:
                    want.append((ex.code, None))
        assert sum(1 for w in want if w[0] == 0) >= 10 and sum(1 for w in want if w[0] != 0) >= 4, want
        bad = []
        for K in (8, 3):
            with H.Pool((0,), K) as P:
                for b in range(2):
                    order = rng.permutation(len(paths))
                    failed, fr, _ = P.process_files([paths[i] for i in order], md5=True)
                    for j, i in enumerate(order):
                        r = fr[j]
                        got = (0, hashlib.md5(open(r.output_path.decode(), "rb").read()).hexdigest()) if r.rc == 0 else (r.rc, None)
                        if r.rc == 0:
                            os.unlink(r.output_path.decode())
                        if got != want[i]:
                            bad.append((K, b, os.path.basename(paths[i]), got, want[i], r.error))
        left = [q for q in os.listdir(d) if q.startswith(".processing-")]
        assert not bad and not left, (bad, left)
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ beyond the bench's sizes
def _windows_vs_oracle_pass2(oracle, x, p2, fp, sr, a0, a1, step, in_per, out_per, lead_s=10):
    """The delivered Pass-2 s16 between input samples [a0, a1) against the oracle chain, in windows of `step` input samples that each run
    behind `lead_s` seconds of lead-in (the chain's memories -- biquads, afftdn's priors, the followers' 200 ms release -- are long
    converged there); windows start on polyphase periods (`in_per` inputs -> `out_per` outputs).  Returns (max, mean, n)."""
    from test_gpu_pipeline import oracle_pass2
    pad = lead_s * sr // in_per * in_per
    futs = []
    for s in range(a0, a1, step):
        e = min(s + step, a1)
        last = e >= x.size
        # a window that is not the file's last carries a tail it does not compare (the resampler's and anlmdn's look-ahead)
        tail = 0 if last else sr // in_per * in_per
        seg = np.ascontiguousarray(x[s - pad: e + tail])
        futs.append((s, e, _pool().submit(oracle_pass2, oracle, seg, fp, sr)))
    mx = 0; tot = 0.0; cnt = 0
    for s, e, f in futs:
        _, ref = f.result()
        o0 = (s - pad) // in_per * out_per
        skip = pad // in_per * out_per
        n = ((e - s) // in_per * out_per) if e < x.size else ref.size - skip
        got = p2[o0 + skip: o0 + skip + n]
        d = np.abs(ref[skip: skip + n].astype(np.int32) - got.astype(np.int32))
        assert d.size == n and n > 0
        mx = max(mx, int(d.max())); tot += float(d.sum()); cnt += n
    return mx, tot / cnt, cnt


def _landing_parallel(oracle, s16, rate=44100, chunk_s=600):
    """chain.landing of a long delivered file with the host's cores: the integrated loudness from one sequential ebur128 without the true
    peak (K-weighting + gating are one recurrence), the true peak as the maximum over ten-minute chunks that overlap by a second (the
    32-tap polyphase sum of a sample sees 16 neighbours either side; a chunk's own first and last half second are the neighbour's)."""
    xf = np.asarray(s16, np.int16).astype(np.float64) / 32768.0
    whole = _pool().submit(oracle.ebur128, xf, rate, True, False)
    step = chunk_s * rate
    tps = [_pool().submit(lambda a=a: float(oracle.ebur128(xf[max(0, a - rate): a + step + rate], rate, True, True)["true_peak"])) for a in range(0, xf.size, step)]
    tp = max(f.result() for f in tps)
    return {"output_lufs": float(whole.result()["integrated"]), "output_dbtp": float(20 * np.log10(tp)) if tp > 0 else float("-inf")}


def test_three_hour_file_linear_mode(engine, oracle):
    """A THREE-HOUR 48 kHz mono file (518.4 M samples, three times configs[1]'s) through jt_process_audio: properties of the whole job
    (sample counts ceil(N * 147 / 160); two runs deliver identical bytes; linear mode with the limiter prefix; the CPU oracle's ebur128
    of the delivered s16 lands where the result says, inside -16 +/- 0.1 LUFS and <= -1 dBTP -- processor.go:199-208, normalise.go:897),
    and the LAST TEN MINUTES of the Pass-2 output, end-of-stream flush included, against the oracle chain (five two-minute windows side by
    side, each behind ten seconds of lead-in): <= 3 LSB anywhere, < 0.3 LSB on average; and the last two minutes of Pass 4 against the
    oracle's Pass-4 graph on the GPU's own Pass-2 output, started 20 s earlier on adeclick's window grid: <= 2 samples by <= 1 LSB."""
    from oracle import chain
    secs = 3 * 3600.0
    x = bench_talker(secs, SR, 1000, 40.0)
    n = x.size
    engine.upload_pcm(np.asarray(x, np.float32), SR, 1)
    t0 = time.perf_counter(); res = H.process_audio(engine); dt = time.perf_counter() - t0
    p2 = engine.download_s16(2).copy(); p4 = engine.download_s16(4).copy()
    h1 = hashlib.md5(p4.tobytes()).hexdigest()
    res2 = H.process_audio(engine)
    assert hashlib.md5(engine.download_s16(4).tobytes()).hexdigest() == h1 and res2.output_lufs == res.output_lufs
    m = -(-n * 147 // 160)
    assert p2.size == m and p4.size == m
    assert res.loudnorm.normalization_type_dynamic == 0 and res.limiter.needed == 1 and res.within_target == 1
    fp = L.FilterParams(); H.lib().jt_host_filter_params(C.byref(res.effective), C.byref(fp))
    a0 = (n - 600 * SR) // 160 * 160
    mx, mean, cnt = _windows_vs_oracle_pass2(oracle, x, p2, fp, SR, a0, n, 120 * SR, 160, 147)
    # Pass 4's tail: the reference's graph on the GPU's Pass-2 output from a start on adeclick's hop grid (w=55 ms, o=50 % at 44.1 kHz:
    # hop 1212), 140 s before the end; the first 20 s are lead-in (limiter gains at rest between the talker's bursts, adeclick is
    # window-local), the last 120 s are compared
    spec = bytes(res.pass4_spec).split(b"\0")[0]
    hop = 1212
    s0 = (m - 140 * 44100) // hop * hop
    ref = chain.pass4(p2[s0:], 44100, spec)
    cmp0 = 20 * 44100
    d4 = np.abs(ref["s16"][cmp0:].astype(np.int32) - p4[s0 + cmp0:].astype(np.int32))
    lo = _landing_parallel(oracle, p4)
    print(f"3-hour file: {dt * 1e3:.0f} ms = {secs / dt:.0f} xRT; pass 2 last 10 min vs oracle: max {mx} LSB, mean {mean:.4f} over {cnt}; pass 4 last 2 min: "
          f"{int(np.count_nonzero(d4))} differ (max {int(d4.max())}); lands {res.output_lufs:.2f} LUFS / {res.output_tp_db:.2f} dBTP, oracle {lo['output_lufs']:.2f} / {lo['output_dbtp']:.2f}")
    assert mx <= 3 and mean < 0.3
    assert ref["dynamic"] == 0 and np.count_nonzero(d4) <= 2 and d4.max() <= 1
    assert abs(lo["output_lufs"] - res.output_lufs) <= 0.011 and abs(lo["output_lufs"] + 16.0) <= 0.1 and lo["output_dbtp"] <= -1.0


def test_two_hour_dynamic_mode_stream(engine):
    """af_loudnorm's dynamic mode on a 2.1-HOUR 192 kHz stream (1.45e9 samples = 0.68 of 2^31; 75 600 frames, past the launch count at
    which round 5's shift overflow faulted): the stream path against the one-workgroup kernel.  Bar: every sample and every statistic
    identical; the stream path carried > 99 % of the frames."""
    hours = 2.1
    unit = synth.speech_like(60.0, 192000, seed=77).astype(np.float64) * 2.5
    n = int(hours * 3600 * 192000)
    x = np.tile(unit, n // unit.size + 1)[:n]
    x[:: 192000 * 97] *= 1.7
    del unit
    got, gst = engine.op_loudnorm_dynamic(x, target_tp=-9.0)
    frames = int(engine.timers()["ln_stream_frames"])
    with options(engine, ln_no_stream=True):
        want, wst = engine.op_loudnorm_dynamic(x, target_tp=-9.0)
    del x
    assert gst == wst and gst["normalization_type_dynamic"] == 1
    assert frames > 0.99 * (n // 19200), frames
    same = True
    for a in range(0, n, 1 << 26):
        same = same and np.array_equal(got[a: a + (1 << 26)], want[a: a + (1 << 26)])
    assert same
    assert float(np.max(np.abs(got[-10**7:]))) <= 10 ** (-9.0 / 20.0) + 1e-12


def test_configs4_at_sixty_minutes(engine, oracle):
    """BASELINE configs[4] at SIXTY minutes: 96 kHz stereo (L != R), 345.6 M frames -- rematrix down-mix, Pass 1 / 2 at 96 kHz (anlmdn
    K = 576, afftdn's 4096-point instance), 96 -> 44.1 kHz polyphase, true peak via 96 -> 192 kHz.  Properties of the whole job (sample
    count ceil(N * 147 / 320); two runs identical; Pass 1's integrated loudness equal to the oracle's ebur128 of the oracle's down-mix at the
    printed precision; the oracle's ebur128 of the delivered s16 lands where the result says, -16 +/- 0.1 LUFS, <= -1 dBTP), and a
    FIVE-MINUTE window (minutes 30-35) of the Pass-2 output against the oracle chain at 96 kHz (ten 30-second windows side by side, each
    behind ten seconds of lead-in): <= 3 LSB anywhere, < 0.3 LSB on average."""
    from oracle import chain
    sr = 96000; secs = 3600.0
    a = np.asarray(bench_talker(secs, sr, 1096, 40.0), np.float32)
    st = np.empty(a.size * 2, np.float32)
    st[0::2] = a
    st[1::2] = np.roll(a, 37) * np.float32(0.8) + np.roll(a, 7 * sr + 13) * np.float32(0.1)      # L != R: an early echo and a late one
    n = a.size
    del a
    engine.upload_pcm(st, sr, 2)
    t0 = time.perf_counter(); res = H.process_audio(engine); dt = time.perf_counter() - t0
    p2 = engine.download_s16(2).copy(); p4 = engine.download_s16(4).copy()
    h1 = hashlib.md5(p4.tobytes()).hexdigest()
    H.process_audio(engine)
    assert hashlib.md5(engine.download_s16(4).tobytes()).hexdigest() == h1
    m = -(-n * 147 // 320)
    assert p2.size == m and p4.size == m
    mono = oracle.downmix_stereo(st, 0)
    del st
    e1 = _pool().submit(oracle.ebur128, mono.astype(np.float64), sr, True, False)
    fp = L.FilterParams(); H.lib().jt_host_filter_params(C.byref(res.effective), C.byref(fp))
    a0 = 30 * 60 * sr
    mx, mean, cnt = _windows_vs_oracle_pass2(oracle, mono, p2, fp, sr, a0, a0 + 300 * sr, 30 * sr, 320, 147)
    lo = _landing_parallel(oracle, p4); ei = e1.result()
    print(f"configs[4] at 60 min: {dt * 1e3:.0f} ms = {secs / dt:.0f} xRT; pass 2 minutes 30-35 vs oracle: max {mx} LSB, mean {mean:.4f} over {cnt}; "
          f"input I {res.input.input_i:.3f} (oracle {ei['integrated']:.3f}); lands {res.output_lufs:.2f} / {res.output_tp_db:.2f}, oracle {lo['output_lufs']:.2f} / {lo['output_dbtp']:.2f}")
    assert mx <= 3 and mean < 0.3
    assert abs(res.input.input_i - ei["integrated"]) < 0.002
    assert abs(lo["output_lufs"] - res.output_lufs) <= 0.011
    if not res.loudnorm.normalization_type_dynamic:
        assert abs(lo["output_lufs"] + 16.0) <= 0.1 and lo["output_dbtp"] <= -1.0
