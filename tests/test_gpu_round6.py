"""Round 6: the decoder's frame cadence comes from the FILE (missing #3 / next #2)."""
import ctypes as C
import struct

import numpy as np
import pytest

from jivetalking_amd import synth, hostlogic as H, _lib as L

pytestmark = pytest.mark.gpu


def _wav(x, rate, ch, kind):
    """RIFF/WAVE image of interleaved samples in [-1, 1): kind 's16' | 's24' | 'f32'."""
    if kind == "f32":
        payload, tag, bits = np.asarray(x, "<f4").tobytes(), 3, 32
    elif kind == "s16":
        payload, tag, bits = np.clip(np.rint(np.asarray(x, np.float64) * 32768), -32768, 32767).astype("<i2").tobytes(), 1, 16
    else:
        v = np.clip(np.rint(np.asarray(x, np.float64) * 8388608), -8388608, 8388607).astype("<i4")
        payload, tag, bits = v.view(np.uint8).reshape(-1, 4)[:, :3].tobytes(), 1, 24
    align = ch * bits // 8
    fmt = struct.pack("<HHIIHH", tag, ch, rate, rate * align, align, bits)
    body = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body


def _reference_interval_starts(frame_lens, sr):
    """analyser.go:588-600 as written, on frame lengths alone: for every decoder frame t = samples so far / rate as a time.Duration
    (truncated nanoseconds), the frame is counted into the open interval, THEN the interval is closed (with the open interval's start
    as its timestamp) if t - start >= 250 ms and the next one starts at t; a trailing interval that saw samples is appended
    (analyser.go:635-638).  Returns [(timestamp_ns, samples in the interval)]."""
    out = []; start = 0; processed = 0; acc = 0
    for nb in frame_lens:
        t = int(float(processed) / float(sr) * 1e9)
        processed += int(nb); acc += int(nb)
        if t - start >= 250_000_000:
            out.append((start, acc)); start = t; acc = 0
    if acc > 0:
        out.append((start, acc))
    return out


CASES = [
    # name, rate, channels, container / sample format, the cadence the reference's decoder would deliver
    ("wav_s16_44k1_mono", 44100, 1, "s16", 2048),          # wavdec.c: 4096-byte packets = 2048 s16 samples
    ("wav_f32_44k1_mono", 44100, 1, "f32", 1024),          # = 1024 f32 samples
    ("wav_s24_48k_mono", 48000, 1, "s24", 1365),           # 4096 // 3 * 3 = 4095 bytes
    ("wav_s16_44k1_stereo", 44100, 2, "s16", 1024),        # block_align 4
    ("flac_1152_44k1", 44100, 1, ("flac", 1152, 2), 1152),
    ("flac_4608_48k", 48000, 1, ("flac", 4608, 2), 4608),
    ("flac_variable_48k", 48000, 1, ("flac", 4096, 2 | 32), None),      # frames alternate 4096 / 2048: per-frame lengths
    ("flac_576_stereo_44k1", 44100, 2, ("flac", 576, 2 | 192), 576),
]


@pytest.mark.parametrize("name,sr,ch,fmt,cadence", CASES, ids=[c[0] for c in CASES])
def test_decoder_frame_cadence_comes_from_the_file(engine, oracle, name, sr, ch, fmt, cadence):
    """The reference closes its 250 ms analysis intervals on DECODER-FRAME boundaries (analyser.go:588-600; the frames are whatever
    Reader.ReadFrame delivers, reader.go:129): the FLAC stream's own block size(s), the WAV demuxer's 4096-byte packets.  With
    frame_samples = 0 the library takes that cadence from the file jt_load_audio decoded.  Held here, per file:
      * jt_audio_meta / jt_input_frame_layout report the cadence the rule gives;
      * the interval series (jt_host_last_intervals) has exactly the reference rule's count and timestamps, and every interval's RMS
        level is the RMS of exactly the samples the rule puts in it (1e-9 dB);
      * the decision chain (VAD, elections, band graphs in the source's format, AdaptConfig, chain string) equals the CPU oracle's
        with the same framing: same elections on the same intervals, same switches, printed parameters within 2e-3;
      * and a 4096 cadence (what the library assumed for every file until round 5) gives DIFFERENT interval boundaries for this file --
        the test would not notice a regression to the constant otherwise."""
    import oracle_pass1 as P
    from test_gpu_fuzz import _parse, _close
    secs = 42.0
    a = np.asarray(synth.speech_like(secs, sr, seed=661), np.float64)
    b = np.roll(a, 23) * 0.7
    if isinstance(fmt, tuple):
        _, bs, mode = fmt
        q = 32768.0
        pcm = np.clip(np.rint((a if ch == 1 else np.stack([a, b], axis=1)) * q), -32768, 32767).astype(np.int32)
        data = oracle.flac_encode(pcm, sr, 16, bs, mode, 8)
        raw = (pcm.astype(np.float32) / np.float32(q)).reshape(-1)
        band_mode = 1
    else:
        inter = a if ch == 1 else np.stack([a, b], axis=1).reshape(-1)
        data = _wav(inter, sr, ch, fmt)
        if fmt == "f32":
            raw, band_mode = np.asarray(inter, np.float32), 0
        elif fmt == "s16":
            raw, band_mode = (np.clip(np.rint(inter * 32768), -32768, 32767) / 32768.0).astype(np.float32), 1
        else:
            raw, band_mode = (np.clip(np.rint(inter * 8388608), -8388608, 8388607) / 8388608.0).astype(np.float32), 2
    meta = engine.load_audio(data)
    n = meta["frames"]
    fs, var, nfr, lens = engine.input_frame_layout()
    if cadence is not None:
        assert (meta["decoder_frame_samples"], meta["decoder_frames_variable"]) == (cadence, 0) and (fs, var) == (cadence, False)
        assert nfr == meta["decoder_frames"] == -(-n // cadence)
        frame_lens = [cadence] * (n // cadence) + ([n % cadence] if n % cadence else [])
    else:
        assert meta["decoder_frames_variable"] == 1 and var and lens is not None and int(lens.sum()) == n
        assert set(lens[:-1].tolist()) == {4096, 2048} and meta["decoder_frame_samples"] == 4096 and nfr == lens.size == meta["decoder_frames"]
        frame_lens = lens.tolist()
    g = H.process_audio(engine, frame_samples=0, analyse_only=True)
    iv = (H.Interval * 4096)()
    niv = H.lib().jt_host_last_intervals(engine.h, iv, C.c_int64(4096))
    want = _reference_interval_starts(frame_lens, sr)
    assert niv == len(want)
    assert [iv[i].timestamp_ns for i in range(niv)] == [w[0] for w in want]
    mono = oracle.downmix_stereo(raw, 0) if ch == 2 else raw
    pos = 0
    r64 = raw.astype(np.float64)
    for i, (_, cnt) in enumerate(want):
        seg = r64[pos * ch:(pos + cnt) * ch]; pos += cnt
        rms = float(np.sqrt(np.mean(seg * seg)))
        ref_db = -120.0 if rms < 1e-5 else 20 * np.log10(rms)
        assert abs(iv[i].rms_level - ref_db) < 1e-9, (i, iv[i].rms_level, ref_db)
    assert pos == n
    # a 4096 cadence (what the library assumed for every file until round 5) closes OTHER intervals for most of these files; where the
    # frames are 2048 long, or alternate 4096 / 2048, the frame that trips the 250 ms test starts on a multiple of 4096 either way
    w4096 = _reference_interval_starts([4096] * (n // 4096) + ([n % 4096] if n % 4096 else []), sr)
    assert ([w[0] for w in w4096] != [w[0] for w in want]) == (name not in ("wav_s16_44k1_mono", "flac_variable_48k"))
    # the decision chain against the oracle's with the same framing
    kw = dict(frame_lens=np.asarray(frame_lens, np.int32)) if cadence is None else dict(frame_samples=cadence)
    band_x = oracle.downmix_stereo(raw, band_mode) if ch == 2 and band_mode else None       # integer sources: the band graphs' integer matrix
    m, eff, spec = P.decide(oracle, mono, sr, raw=raw if ch == 2 else None, band_mode=band_mode, band_x=band_x, **kw)
    gm = g.input
    assert (gm.has_speech_profile, gm.has_noise_profile, gm.voice_activated, gm.floor_source, gm.n_candidates, gm.n_speech_regions) == \
           (m.has_speech_profile, m.has_noise_profile, m.voice_activated, m.floor_source, m.n_candidates, m.n_speech_regions)
    if m.has_speech_profile:
        assert (gm.speech_profile.region.start_ns, gm.speech_profile.region.duration_ns) == (m.speech_profile.region.start_ns, m.speech_profile.region.duration_ns)
    if m.has_noise_profile:
        assert (gm.noise_profile.start_ns, gm.noise_profile.duration_ns) == (m.noise_profile.start_ns, m.noise_profile.duration_ns)
    cg, co = _parse(H.filter_spec(g.effective, 2)), _parse(spec)
    assert [f[0] for f in cg] == [f[0] for f in co]
    for (fname, pa), (_, pb) in zip(cg, co):
        assert pa.keys() == pb.keys(), fname
        for k in pa:
            if fname == "afftdn" and k == "bn":
                va, vb = [float(v) for v in pa[k].split("|")], [float(v) for v in pb[k].split("|")]
                assert len(va) == len(vb) and max(abs(p - q) for p, q in zip(va, vb)) <= 0.1001, (pa[k], pb[k])
            else:
                assert _close(pa[k], pb[k]), (fname, k, pa[k], pb[k])


def test_frame_samples_zero_after_an_upload_is_4096_and_an_override_wins(engine):
    """PCM that was uploaded has no file behind it: frame_samples = 0 means 4096 then (the reference's FLAC encoder's frame, what
    every earlier round assumed), and a positive frame_samples always overrides the file's cadence (a caller that decodes the file
    itself knows its frames)."""
    sr = 44100
    x = synth.speech_like(20.0, sr, seed=662)
    engine.upload_pcm(x, sr, 1)
    assert engine.input_frame_layout()[:3] == (4096, False, -(-x.size // 4096))
    H.process_audio(engine, frame_samples=0, analyse_only=True)
    a = H.intervals_jsonl(engine)
    H.process_audio(engine, frame_samples=4096, analyse_only=True)
    assert H.intervals_jsonl(engine) == a
    engine.load_audio(_wav(x, sr, 1, "f32"))
    H.process_audio(engine, frame_samples=0, analyse_only=True)
    b = H.intervals_jsonl(engine)
    assert b != a
    H.process_audio(engine, frame_samples=4096, analyse_only=True)
    assert H.intervals_jsonl(engine) == a
    with pytest.raises(L.JtError) as ei:
        H.process_audio(engine, frame_samples=-1, analyse_only=True)
    assert ei.value.code == L.JT_E_INVAL


def test_progress_ticks_follow_the_files_cadence(engine):
    """The reference ticks every 100th decoder frame (analyser.go:602-618, processor.go:320-335) with Progress = frames / (duration x
    rate / 4096): a 1024-sample cadence gives four times the Pass-1 ticks of a 4096 one, and their progress values climb four times
    slower per tick (capped at 0.95)."""
    sr = 48000
    x = synth.speech_like(40.0, sr, seed=663)
    counts = {}
    for kind, want in (("f32", 1024), ("s16", 2048)):
        engine.load_audio(_wav(x, sr, 1, kind))
        ticks = []
        H.process_audio_with_progress(engine, lambda u: ticks.append((u.pass_, (u.pass_name or b"").decode(), u.progress)), frame_samples=0, ticks=True)
        p1 = [t for t in ticks if t[0] == 1 and t[1] == "Analysing" and 0.0 < t[2] < 0.95]
        nfr = -(-x.size // want)
        est = x.size / 4096.0
        counts[kind] = len([t for t in ticks if t[0] == 1 and t[1] == "Analysing"])
        # every 100th frame: fc = 100, 200, ... < nfr with progress fc / est * 0.95 (those below the cap)
        exp = [min(0.95, fc / est * 0.95) for fc in range(100, nfr, 100)]
        exp = [e for e in exp if e < 0.95]
        got = sorted(t[2] for t in p1)
        assert len(got) == len(exp) and all(abs(a - b) < 1e-12 for a, b in zip(got, exp)), (kind, len(got), len(exp), got[:5], exp[:5])
    assert counts["f32"] > counts["s16"]


# ---------------------------------------------------------------------------------------------- surround layouts
LAYOUTS = [(3, 0x7, "3.0"), (3, 0, "3 channels without a layout: 2.1"), (4, 0x107, "4.0"), (4, 0x33, "quad"), (5, 0x607, "5.0(side)"),
           (5, 0x37, "5.0"), (6, 0x60F, "5.1(side)"), (6, 0x3F, "5.1"), (6, 0, "6 channels without a layout: 5.1"), (7, 0x70F, "6.1"),
           (8, 0x63F, "7.1"), (8, 0xFF, "7.1(wide)"), (2, 0x104, "FC + BC"), (1, 0x4, "mono"), (1, 0x8, "LFE alone")]


def _multichannel(ch, n, sr, seed):
    """ch different signals, interleaved: speech-like at different levels / delays plus a little noise of their own"""
    rng = np.random.default_rng(seed)
    base = np.asarray(synth.speech_like(n / sr + 0.2, sr, seed=seed), np.float64)
    x = np.empty((n, ch), np.float32)
    for c in range(ch):
        x[:, c] = (np.roll(base, 53 * c)[:n] * (0.9 - 0.08 * c) + rng.standard_normal(n) * 0.003 * (c + 1)).astype(np.float32)
    return x.reshape(-1)


@pytest.mark.parametrize("ch,mask,name", LAYOUTS, ids=[l[2] for l in LAYOUTS])
def test_surround_downmix_is_libswresamples_default_matrix(engine, oracle, ch, mask, name):
    """aformat=channel_layouts=mono (filters.go:607-615) of a source with any layout of FL FR FC LFE BL BR FLC FRC BC SL SR: the mono
    signal the passes see must be libswresample's default rematrix row for a FRONT_CENTER output as the oracle restates it
    (oracle/orc_basic.c: orc_downmix_layout -- coefficients of swr_build_matrix2, float products and sums in channel order).
    Float graphs (Pass 1): astats' Min / Max level of the down-mix BIT-identical to the oracle's extremes, RMS 1e-9 relative, integrated
    loudness 1e-6 LU.  Band graphs of 16- and 24-bit sources (integer-normalised matrix, S16P integer products / FLTP + s32
    conversion): the band RMS within 1e-4 dB of the oracle's band graph on the oracle's down-mix."""
    sr = 48000; n = 6 * sr
    x = _multichannel(ch, n, sr, 670 + ch)
    engine.upload_pcm(x, sr, ch, channel_mask=mask)
    ref = oracle.downmix_layout(x, ch, mask, 0)
    p1 = engine.pass1(n)
    if np.any(ref != 0):
        assert p1["astats"]["max_level"] == float(ref.max()) and p1["astats"]["min_level"] == float(ref.min())
        rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
        assert abs(10 ** (p1["astats"]["rms_level"] / 20) / rms - 1.0) < 1e-9
        e = oracle.ebur128(ref.astype(np.float64), sr, True, True)
        assert abs(p1["r128"]["integrated"] - e["integrated"]) < 1e-6
    else:
        assert p1["astats"]["max_level"] == 0.0 and p1["astats"]["min_level"] == 0.0          # LFE alone: lfe_mix_level = 0
        return
    if ch == 1:
        return
    lo, hi = [200.0, 1000.0, 6000.0], [400.0, 3000.0, 9000.0]
    for bits, mode in ((16, 1), (24, 2)):
        q = float(1 << (bits - 1))
        xi = (np.rint(np.asarray(x, np.float64) * q) / q).astype(np.float32)
        engine.upload_pcm(xi, sr, ch, channel_mask=mask)
        engine.set_source_format(bits, False)
        got, ok = engine.band_rms(1.0, 4.0, lo, hi)
        seg = oracle.downmix_layout(xi, ch, mask, mode)[sr:sr + 4 * sr]
        for k in range(3):
            want = oracle.band_rms_db_fmt(seg, sr, lo[k], hi[k], mode)
            assert ok[k] == 1 and abs(got[k] - want) < 1e-4, (name, bits, k, got[k], want)


def test_surround_files_through_the_whole_path(engine, oracle):
    """A 5.1 FLAC (libavcodec's layout for six channels: 5.1(side)), a WAVE_FORMAT_EXTENSIBLE 5.1(back) file and a plain three-channel
    WAV (no layout: swr_init assumes 2.1) through jt_load_audio: the reported layout, Pass 1 on the oracle's down-mix, and for the FLAC
    the whole job -- Pass-2 s16 against the oracle chain on the oracle's down-mix (<= 3 LSB, mean < 0.3), landing -16 +/- 0.1 LUFS."""
    import ctypes as C
    from test_gpu_pipeline import oracle_pass2
    sr = 48000; n = 40 * sr
    x6 = _multichannel(6, n, sr, 681)
    pcm = np.clip(np.rint(x6.astype(np.float64) * 32768), -32768, 32767).astype(np.int32).reshape(-1, 6)
    meta = engine.load_audio(oracle.flac_encode(pcm, sr, 16, 4096, 2, 8))
    assert (meta["channels"], meta["channel_mask"], meta["decoder_frame_samples"]) == (6, 0x60F, 4096)
    raw = (pcm.astype(np.float32) / np.float32(32768.0)).reshape(-1)
    mono = oracle.downmix_layout(raw, 6, 0x60F, 0)
    res = H.process_audio(engine, frame_samples=0)
    e = oracle.ebur128(mono.astype(np.float64), sr, True, True)
    assert abs(res.input.input_i - e["integrated"]) < 0.002
    fp = L.FilterParams(); H.lib().jt_host_filter_params(C.byref(res.effective), C.byref(fp))
    _, ref2 = oracle_pass2(oracle, mono, fp, sr)
    p2 = engine.download_s16(2)
    d = np.abs(ref2.astype(np.int32) - p2.astype(np.int32))
    assert ref2.size == p2.size and d.max() <= 3 and d.mean() < 0.3
    f = oracle.ebur128(engine.download_s16(4).astype(np.float64) / 32768.0, 44100, True, True)
    assert abs(f["integrated"] + 16.0) <= 0.1 and 20 * np.log10(f["true_peak"]) <= -1.0
    # WAVE_FORMAT_EXTENSIBLE, f32, 5.1(back)
    payload = x6.astype("<f4").tobytes()
    fmt = struct.pack("<HHIIHHHHIH", 0xFFFE, 6, sr, sr * 24, 24, 32, 22, 32, 0x3F, 3) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    body = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
    meta = engine.load_audio(b"RIFF" + struct.pack("<I", 4 + len(body)) + b"WAVE" + body)
    assert (meta["channels"], meta["channel_mask"], meta["is_float"], meta["decoder_frame_samples"]) == (6, 0x3F, 1, 4096 // 24)
    p1 = engine.pass1(n)
    ref = oracle.downmix_layout(x6, 6, 0x3F, 0)
    assert p1["astats"]["max_level"] == float(ref.max()) and p1["astats"]["min_level"] == float(ref.min())
    # plain PCM WAV with three channels: no layout in the file
    x3 = _multichannel(3, n, sr, 682)
    meta = engine.load_audio(_wav(x3, sr, 3, "f32"))
    assert (meta["channels"], meta["channel_mask"]) == (3, 0xB)
    p1 = engine.pass1(n)
    ref = oracle.downmix_layout(x3, 3, 0, 0)
    assert p1["astats"]["max_level"] == float(ref.max()) and p1["astats"]["min_level"] == float(ref.min())


def test_pool_numa_binding_changes_no_byte_and_restores_the_callers_affinity(engine, oracle):
    """VERDICT r5 next #6b: a handle pool binds its worker threads and finisher jobs to the NUMA node its GPU hangs off (option pool_numa,
    default on).  The first worker of a batch runs on the CALLING thread: its affinity must be what it was afterwards; the outputs must be
    the same bytes with the binding on and off; and the node the library reports is what sysfs says for the device's PCI address."""
    import ctypes as C, hashlib, os, shutil, tempfile
    lib = H.lib()
    ncpu = C.c_int()
    node = lib.jt_host_device_numa_node(C.c_int(0), C.byref(ncpu))
    assert node >= -1 and (ncpu.value > 0) == (node >= 0)
    if node >= 0:
        cpus = open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
        assert cpus and ncpu.value <= (os.cpu_count() or 1)
    d = tempfile.mkdtemp(prefix="jtnm", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        paths = []
        for k in range(6):
            x = synth.speech_like(20.0 + 3 * k, 48000, seed=690 + k)
            pcm = np.clip(np.rint(np.asarray(x, np.float64) * 32768), -32768, 32767).astype(np.int16)
            p = os.path.join(d, f"n{k}.flac")
            with open(p, "wb") as f:
                f.write(engine.op_flac_encode(pcm, 48000, md5=True))
            paths.append(p)
        before = os.sched_getaffinity(0)
        hashes = []
        for numa in (1, 0, 1):
            assert lib.jt_set_option(None, b"pool_numa", str(numa).encode()) == 0
            with H.Pool((0,), 3) as P:
                failed, fr, _ = P.process_files(paths, md5=True)
                assert failed == 0
                hs = hashlib.md5()
                for r in fr:
                    hs.update(open(r.output_path.decode(), "rb").read()); os.unlink(r.output_path.decode())
                hashes.append(hs.hexdigest())
            assert os.sched_getaffinity(0) == before
        assert hashes[0] == hashes[1] == hashes[2]
    finally:
        lib.jt_set_option(None, b"pool_numa", b"1")
        shutil.rmtree(d, ignore_errors=True)


def test_noise_floor_branch_and_bound_equals_the_exhaustive_sweep(engine, oracle):
    """astats' Noise_floor / Noise_floor_count (af_astats.c: the minimum over all 50 ms windows of the window's peak, and how many windows
    reach it) by branch and bound, round 6: an upper bound of the minimum from block maxima alone (k_nf_runmax), then k_nf_main skips every
    start block whose windows all contain a whole block louder than the bound.  Exact by construction -- held here against the exhaustive
    sweep (option nf_unpruned) and the oracle on speech with pauses, digital silence in the middle (ties: thousands of windows at 0),
    a constant signal (every window ties), a quiet tail, full-scale noise, and lengths around the window and block sizes."""
    from conftest import options
    rng = np.random.default_rng(665)
    sr = 48000
    sp = np.asarray(synth.speech_like(30.0, sr, seed=665), np.float32)
    z = sp.copy(); z[7 * sr: 9 * sr] = 0.0
    tail = sp.copy(); tail[-3 * sr:] *= 1e-3
    cases = [("speech", sp), ("digital silence inside", z), ("constant", np.full(5 * sr, 0.25, np.float32)), ("quiet tail", tail),
             ("noise", rng.uniform(-1, 1, 4 * sr).astype(np.float32)), ("one window", sp[: 2400]), ("window + 1", sp[: 2401]),
             ("3 blocks", sp[: 2400 + 129]), ("44.1 kHz", np.asarray(synth.speech_like(12.0, 44100, seed=666), np.float32))]
    for name, x in cases:
        rate = 44100 if name == "44.1 kHz" else sr
        a = engine.op_astats(x, rate)
        with options(engine, nf_unpruned=True):
            b = engine.op_astats(x, rate)
        assert (a["noise_floor"], a["noise_floor_count"]) == (b["noise_floor"], b["noise_floor_count"]), (name, a["noise_floor"], b["noise_floor"], a["noise_floor_count"], b["noise_floor_count"])
        ref = oracle.astats(x.astype(np.float64), rate)
        assert a["noise_floor_count"] == ref["noise_floor_count"], (name, a["noise_floor_count"], ref["noise_floor_count"])
        if np.isfinite(ref["noise_floor_db"]):
            assert abs(a["noise_floor"] - ref["noise_floor_db"]) <= 1e-9 * max(1.0, abs(ref["noise_floor_db"])), (name, a["noise_floor"], ref["noise_floor_db"])


def test_adeclick_solver_lists_sorted_or_not_repair_the_same_samples(engine):
    """The split adeclick's solver lists are sorted by the windows' flagged-sample counts (k_dk_sort_scan / _scatter: a wave's two windows
    match, the longest start first); option dk_unsorted keeps the order the front kernels appended them in.  A window's result does not
    depend on its partner or its place: the same doubles and the same repaired count either way -- on a signal whose windows alternate
    between a few clicks and hundreds (what makes the sorted pairs differ from the appended ones), at 44.1 and 48 kHz, both methods."""
    from conftest import options
    rng = np.random.default_rng(6161)
    for sr, method in ((44100, "s"), (48000, "s"), (44100, "a")):
        n = 40 * sr
        x = 0.2 * np.sin(2 * np.pi * 180.0 * np.arange(n) / sr) + 0.02 * rng.standard_normal(n)
        # sparse clicks everywhere, dense click bursts in every third 55 ms window
        idx = rng.choice(n, 4000, replace=False); x[idx] += rng.choice([-1.0, 1.0], idx.size) * rng.uniform(0.3, 0.9, idx.size)
        w = int(sr * 0.055)
        for s0 in range(0, n - w, 3 * w):
            k = rng.choice(w, 160, replace=False); x[s0 + k] += rng.choice([-1.0, 1.0], k.size) * rng.uniform(0.2, 0.8, k.size)
        x = np.clip(x, -1.0, 1.0)
        ya, ca = engine.op_adeclick(x, sr, t=2.0, w=55.0, o=75.0, method=method, return_count=True)
        with options(engine, dk_unsorted=True):
            yb, cb = engine.op_adeclick(x, sr, t=2.0, w=55.0, o=75.0, method=method, return_count=True)
        assert ca == cb and ca > 10000, (sr, method, ca, cb)
        assert np.array_equal(ya.view(np.uint64), yb.view(np.uint64)), (sr, method, int(np.count_nonzero(ya != yb)))
