"""Round-3 behaviours: BASELINE configs[3]'s per-GPU share (ten-minute files queued over one GPU, several in flight) gives the
bytes the one-at-a-time path gives; a device that cannot be opened does not take files away from the devices that can
(cmd/jivetalking/pool.go:122-153: one failure never stops the others)."""
import os

import numpy as np
import pytest

from jivetalking_amd import synth, hostlogic as H, _lib as L
from jivetalking_amd.engine import Engine
from conftest import options

pytestmark = pytest.mark.gpu
SR = 48000


def _ten_minute_flacs(engine, d, count, minutes=10.0):
    """`count` different files: a seeded 60 s talker repeated to length (numpy: torch's HIP runtime cannot be initialised after the
    library's in one process), every second file with a plosive-like burst every 1.5 s so that its plan needs the limiter prefix."""
    paths = []
    for k in range(count):
        base = np.asarray(synth.speech_like(min(60.0, minutes * 60.0), SR, seed=300 + k), np.float64)
        x = np.tile(base, int(np.ceil(minutes * 60.0 / (base.size / SR))))[: int(minutes * 60.0 * SR)].copy()
        if k % 2:
            w = int(0.02 * SR)
            burst = 0.35 * np.hanning(w) * np.sin(2 * np.pi * 180.0 * np.arange(w) / SR)
            for pos in range(SR, x.size - SR, int(1.5 * SR)):
                x[pos:pos + w] += burst
        pcm = np.clip(np.rint(x * 32768), -32768, 32767).astype(np.int16)
        p = os.path.join(str(d), f"ten{k}.flac")
        with open(p, "wb") as f:
            f.write(engine.op_flac_encode(pcm, SR, md5=True))
        paths.append(p)
    return paths


def test_ten_minute_files_in_flight_equal_one_at_a_time(engine, tmp_path):
    """8 ten-minute files with 4 in flight (configs[3], one GPU's share in small) vs the same files one at a time on one handle: the
    output files are byte-identical and every result field that is a measurement is equal."""
    paths = _ten_minute_flacs(engine, tmp_path, 8)
    want = []
    for p in paths:
        res, out_path, _ = H.process_file(engine, p)
        want.append((res.output_lufs, res.output_tp_db, res.input_lufs, int(res.limiter.needed), open(out_path, "rb").read()))
        os.unlink(out_path)
    assert {w[3] for w in want} == {0, 1}                      # both plans occur: with and without the limiter prefix
    failed, res, dev = H.process_files_multi(paths, devices=(0,), in_flight_per_device=4)
    assert failed == 0 and all(d == 0 for d in dev)
    for k in range(8):
        r = res[k]
        assert r.rc == 0 and r.wall_ms > 0
        assert (r.result.output_lufs, r.result.output_tp_db, r.result.input_lufs, int(r.result.limiter.needed)) == want[k][:4]
        assert open(r.output_path.decode(), "rb").read() == want[k][4], f"file {k}: bytes differ with 4 in flight"
        assert abs(r.result.output_lufs + 16.0) <= 0.1 and r.result.output_tp_db <= -1.0
    assert not [q for q in os.listdir(str(tmp_path)) if q.startswith(".processing-")]


def test_every_visible_device_serves_the_shared_queue(engine, tmp_path):
    """BASELINE configs[2] / configs[3] in small, on whatever the box has: jt_process_files_multi over ALL visible devices (HIP's own
    count, jt_device_count) with two workers each.  Every device serves at least one file when there are at least as many files as
    devices (pool.go:122-153: whichever worker is free takes the next file), the outputs are byte-identical to the same files one at a
    time on one handle, and a caller-owned pool (jt_handle_pool_*) gives the same bytes batch after batch with its handles kept open.
    With one visible device this is the single-device test with device_of_file checked; on an 8-GPU node it drives all eight."""
    ndev = L.device_count()
    assert ndev >= 1
    devices = tuple(range(ndev))
    nfiles = max(6, 2 * ndev)
    paths = _ten_minute_flacs(engine, tmp_path, nfiles, minutes=1.0)
    want = []
    for p in paths:
        res, out_path, _ = H.process_file(engine, p)
        want.append((res.output_lufs, res.output_tp_db, int(res.limiter.needed), open(out_path, "rb").read()))
        os.unlink(out_path)

    def check(failed, res, dev, what):
        assert failed == 0, what
        assert set(dev) == set(devices), f"{what}: devices that served files {sorted(set(dev))}, visible {devices}"
        for k in range(nfiles):
            r = res[k]
            assert r.rc == 0 and dev[k] in devices
            assert (r.result.output_lufs, r.result.output_tp_db, int(r.result.limiter.needed)) == want[k][:3], (what, k)
            assert open(r.output_path.decode(), "rb").read() == want[k][3], f"{what}: file {k} (device {dev[k]}) differs from the one-at-a-time bytes"
            os.unlink(r.output_path.decode())
    check(*H.process_files_multi(paths, devices=devices, in_flight_per_device=2), "jt_process_files_multi")
    with H.Pool(devices=devices, in_flight_per_device=2) as pool:
        assert sorted(set(pool.workers())) == list(devices) and len(pool.workers()) == 2 * ndev
        for batch in range(2):                                  # the same handles, warm the second time
            check(*pool.process_files(paths), f"pool batch {batch}")
        failed, res, dev = pool.process_files(paths[:1])        # fewer files than workers: still served, by some device
        assert failed == 0 and dev[0] in devices
        os.unlink(res[0].output_path.decode())
    assert not [q for q in os.listdir(str(tmp_path)) if q.startswith(".processing-")]


def test_a_pool_without_any_device_fails_every_file_of_every_batch(engine, tmp_path):
    paths = _ten_minute_flacs(engine, tmp_path, 2, minutes=0.2)
    with H.Pool(devices=(99,), in_flight_per_device=2) as pool:
        assert pool.workers() == []
        for _ in range(2):
            failed, res, dev = pool.process_files(paths)
            assert failed == 2 and dev == [-1, -1] and all(res[k].rc != 0 and b"jt_open(99)" in res[k].error for k in range(2))
    with H.Pool(devices=(99, 0), in_flight_per_device=1) as pool:      # the worker of the dead device finds nothing else to serve
        assert pool.workers() == [0]
        failed, res, dev = pool.process_files(paths)
        assert failed == 0 and dev == [0, 0], [(r.rc, r.error) for r in res]


def test_a_device_that_cannot_be_opened_takes_no_files(engine, tmp_path):
    """a worker whose jt_open failed used to pop files and fail them in microseconds.  devices = {0, 99}: every file is
    served by device 0; devices = {99}: every file fails with the open error, none is left in its initial state."""
    paths = _ten_minute_flacs(engine, tmp_path, 5, minutes=0.25)
    failed, res, dev = H.process_files_multi(paths, devices=(0, 99), in_flight_per_device=1)
    assert failed == 0 and all(d == 0 for d in dev) and all(res[k].rc == 0 for k in range(5))
    failed, res, dev = H.process_files_multi(paths[:1], devices=(99, 0), in_flight_per_device=1)   # fewer files than devices: the
    assert failed == 0 and dev == [0]                                                                # worker moves on to a spare one
    failed, res, dev = H.process_files_multi(paths, devices=(99,), in_flight_per_device=2)
    assert failed == 5 and all(res[k].rc != 0 and b"jt_open(99)" in res[k].error for k in range(5)) and dev == [-1] * 5


def test_begin_job_keeps_a_cancel_that_arrives_before_the_first_call(engine, tmp_path):
    """The Go shim arms context.AfterFunc before it calls jt_process_file; a ctx cancelled in between used to be wiped by the call's
    own "new job" reset.  Inside a jt_begin_job bracket the first call observes it; after jt_end_job the handle behaves as before."""
    paths = _ten_minute_flacs(engine, tmp_path, 1, minutes=0.2)
    engine.begin_job()
    engine.cancel()
    with pytest.raises(L.JtError) as ei:
        H.process_file(engine, paths[0])
    assert ei.value.code == L.JT_E_CANCELLED
    assert not [q for q in os.listdir(str(tmp_path)) if q.startswith(".processing-") or q.endswith("-processed.flac")]
    engine.end_job()
    engine.cancel()                                            # outside a bracket a new job clears the flag, as before
    res, out_path, _ = H.process_file(engine, paths[0])
    assert os.path.exists(out_path) and abs(res.output_lufs + 16.0) <= 0.1
    engine.begin_job()                                         # a bracket with no cancel is an ordinary job
    res2, out2, _ = H.process_file(engine, paths[0])
    engine.end_job()
    assert res2.output_lufs == res.output_lufs


def test_pass1_completes_when_the_announced_pass2_head_cannot_be_built(engine):
    """with the fixed 20.5 kHz low-pass announced as the early Pass-2 head, jt_pass1 of a 22.05 kHz file used to abort with
    EINVAL mid-pass.  The reference completes Pass 1 and fails in Pass 2 (af_biquads refuses the corner): the head is dropped, Pass 1
    returns its analysis, jt_pass2 raises."""
    from jivetalking_amd.engine import default_filter_params
    x = synth.speech_like(8.0, 22050, seed=51)
    engine.upload_pcm(x, 22050, 1)
    engine.pass2_prefetch_after_pass1(default_filter_params())
    a = engine.pass1(x.size, sample_rate=22050)
    assert a is not None
    with pytest.raises(L.JtError) as ei:
        engine.pass2(default_filter_params())
    assert ei.value.code == L.JT_E_INVAL and "Nyquist" in str(ei.value)


def test_adeclick_split_pipeline_equals_the_one_kernel_version(engine_ab):
    """adeclick runs as a front kernel (AR fit, detector, index list, right-hand side) plus register-resident solver kernels (two windows
    per wave for bands up to 31 rows, one for bands up to 48).  Same operations on the same values as round 2's single kernel
    (option adeclick_fused of the A/B build): the outputs are bit-identical, on speech, on speech with real clicks, and on a signal loud enough that most
    windows take the wide-band solver."""
    engine = engine_ab
    rng = np.random.default_rng(5)
    for seed, gain, clicks in ((21, 1.0, 0), (22, 4.0, 150), (23, 16.0, 400)):
        x = np.asarray(synth.speech_like(20.0, 44100, seed=seed), np.float64) * gain
        if clicks:
            pos = rng.integers(1000, x.size - 1000, clicks)
            x[pos] += rng.uniform(-0.5, 0.5, clicks)
        with options(engine, adeclick_fused=True):
            a = engine.op_adeclick(x, 44100)
        b = engine.op_adeclick(x, 44100)
        assert np.array_equal(a, b), f"seed {seed}: {int((a != b).sum())} samples differ, max {np.abs(a - b).max():.3g}"
        assert int((a != x).sum()) > 1000                          # the filter did repair samples
    # another rate: 32 kHz (W = 1760, AR order 35: Levinson-Durbin stays inside the front kernel, the solvers run with a shorter band)
    x = np.asarray(synth.speech_like(15.0, 32000, seed=24), np.float64) * 3.0
    with options(engine, adeclick_fused=True):
        a = engine.op_adeclick(x, 32000)
    b = engine.op_adeclick(x, 32000)
    assert np.array_equal(a, b) and int((a != x).sum()) > 500


def test_pass3_started_inside_pass2_with_the_planned_prefix_equals_the_explicit_one(engine):
    """jt_pass3_plan_hook: a plan that needs the limiter prefix starts Pass 3's measurement inside Pass 2, as soon as Pass 2's loudness
    and true peak exist.  A schedule change only - every number and every output sample equals the run without it
    (option no_early_plan), on a talker whose plosives need the prefix and on one that does not."""
    w = int(0.02 * SR)
    burst = 0.35 * np.hanning(w) * np.sin(2 * np.pi * 180.0 * np.arange(w) / SR)
    for seed, plosives in ((61, True), (62, False)):
        x = np.asarray(synth.speech_like(45.0, SR, seed=seed), np.float64)
        if plosives:
            for pos in range(SR, x.size - SR, int(1.5 * SR)):
                x[pos:pos + w] += burst
        x = x.astype(np.float32)
        res = []
        for early in (False, True):
            engine.set_option("no_early_plan", not early)        # (ends on early = True: the default)
            engine.upload_pcm(x, SR, 1)
            r = H.process_audio(engine)
            res.append((int(r.limiter.needed), r.measure.input_i, r.measure.input_tp, r.measure.input_lra, r.measure.input_thresh,
                        r.output_lufs, r.output_tp_db, r.filtered.r128.integrated, r.final_.r128.true_peak, engine.download_s16(4).tobytes(),
                        engine.timers()["pass3_ms"]))
        assert res[0][0] == (1 if plosives else 0)
        assert res[0][:10] == res[1][:10], f"seed {seed}: the early start changed a result"
        if plosives:
            assert res[1][10] < res[0][10]                       # and Pass 3 itself only collects


def _parse_chain(spec):
    out = []
    for f in spec.split(","):
        name, _, args = f.partition("=")
        kv = {}
        for a in args.split(":") if args else []:
            k, _, v = a.partition("=")
            kv[k] = v
        out.append((name, kv))
    return out


def _close(a, b, rel=2e-3, abs_=2e-4):
    try:
        fa, fb = float(a), float(b)
    except ValueError:
        return a == b
    return abs(fa - fb) <= max(abs_, rel * max(abs(fa), abs(fb)))


@pytest.mark.parametrize("sr,seed,seconds", [(48000, 21, 45.0), (48000, 71, 60.0), (44100, 72, 40.0)])
def test_decision_chain_on_gpu_pass1_equals_the_chain_on_oracle_pass1(engine, oracle, sr, seed, seconds):
    """the end-to-end tests handed the GPU run's OWN effective parameters to the oracle chain, which pins the
    kernels but not the decisions.  Here the reference's decision chain (interval builder -> VAD / elections -> band graphs ->
    AdaptConfig -> chain string) runs twice: on the Pass-1 measurements of the HIP path (jt_analyse_only) and on the same measurements
    taken by the CPU oracle (tests/oracle_pass1.py).  Elections land on the same 250 ms intervals, every filter is switched the same
    way, and every printed parameter agrees to the measurement tolerances (2e-3 relative: aspectralstats' f32 FFT; bn to 0.1 dB)."""
    import oracle_pass1 as P
    x = synth.speech_like(seconds, sr, seed=seed)
    engine.upload_pcm(x, sr, 1)
    g = H.process_audio(engine, analyse_only=True)
    m, eff, spec = P.decide(oracle, x, sr)
    gm = g.input
    # elections and switches: exact
    assert (gm.has_speech_profile, gm.has_noise_profile, gm.voice_activated, gm.floor_source, gm.n_candidates, gm.n_speech_regions) == \
           (m.has_speech_profile, m.has_noise_profile, m.voice_activated, m.floor_source, m.n_candidates, m.n_speech_regions)
    if m.has_speech_profile:
        assert (gm.speech_profile.region.start_ns, gm.speech_profile.region.duration_ns) == (m.speech_profile.region.start_ns, m.speech_profile.region.duration_ns)
    if m.has_noise_profile:
        assert (gm.noise_profile.start_ns, gm.noise_profile.duration_ns) == (m.noise_profile.start_ns, m.noise_profile.duration_ns)
    assert abs(gm.input_i - m.input_i) <= 0.002 and abs(gm.floor - m.floor) <= 1e-3 and abs(gm.vad_split - m.vad_split) <= 1e-9
    # the chain string: same filters in the same order, same switches, parameters to tolerance
    cg, co = _parse_chain(H.filter_spec(g.effective, 2)), _parse_chain(spec)
    assert [f[0] for f in cg] == [f[0] for f in co]
    for (name, a), (_, b) in zip(cg, co):
        assert a.keys() == b.keys(), name
        for k in a:
            if name == "afftdn" and k == "bn":
                va, vb = [float(v) for v in a[k].split("|")], [float(v) for v in b[k].split("|")]
                assert len(va) == len(vb) and max(abs(p - q) for p, q in zip(va, vb)) <= 0.1001, (a[k], b[k])
            else:
                assert _close(a[k], b[k]), (name, k, a[k], b[k])


def test_limiter_prefix_branch_matches_the_oracle_chain(engine, oracle):
    """The branch the bench's `value` is measured on: a talker with plosives whose plan needs the limiter prefix.
    Pass 3 (volume -> alimiter on the s16 output, swr in double to 192 kHz, loudnorm's first-pass statistics) and Pass 4 (the limited
    signal x linear gain -> adeclick -> brickwall alimiter -> s16) against the same chain composed from the CPU oracle, given the GPU
    run's Pass-2 output: statistics equal at the JSON's %.2f, final PCM within 1 LSB at rounding ties, landing -16 LUFS / <= -1 dBTP."""
    w = int(0.02 * SR)
    burst = 0.35 * np.hanning(w) * np.sin(2 * np.pi * 180.0 * np.arange(w) / SR)
    x = np.asarray(synth.speech_like(30.0, SR, seed=91), np.float64)
    for pos in range(SR, x.size - SR, int(1.5 * SR)):
        x[pos:pos + w] += burst
    engine.upload_pcm(x.astype(np.float32), SR, 1)
    res = H.process_audio(engine)
    p2, p4 = engine.download_s16(2), engine.download_s16(4)
    assert res.limiter.needed == 1
    pre_db = float("%.1f" % max(res.limiter.pre_gain_db, 0.0))
    limit = float("%.6f" % (10 ** (res.limiter.ceiling_db / 20.0)))
    src = p2.astype(np.float64) / 32768.0
    if pre_db > 0:
        src = (src.astype(np.float32) * np.float32(10 ** (pre_db / 20.0))).astype(np.float64)      # af_volume, precision=float
    lim = oracle.alimiter(src, 44100, limit, 5.0, 100.0)
    m = oracle.loudnorm_measure(oracle.swr_f64(lim, 44100, 192000, True), 192000, True)
    assert abs(res.measure.input_i - m["input_i"]) <= 0.011 and abs(res.measure.input_tp - m["input_tp"]) <= 0.011
    assert abs(res.measure.input_lra - m["input_lra"]) <= 0.011 and abs(res.measure.input_thresh - m["input_thresh"]) <= 0.011
    gain = 10 ** ((res.effective_target_i - res.measure.input_i) / 20.0)
    y = oracle.adeclick(lim * gain, 44100, 1.7, 55.0, 50.0, method="s")
    z = oracle.alimiter(y, 44100, 0.803526, 1.0, 50.0)
    ref = oracle.f64_to_s16(z.astype(np.float32).astype(np.float64))
    d = np.abs(ref.astype(np.int32) - p4.astype(np.int32))
    assert d.max() <= 1 and np.count_nonzero(d) <= 4, (int(d.max()), int(np.count_nonzero(d)))
    e = oracle.ebur128(p4.astype(np.float64) / 32768.0, 44100, True, True)
    assert abs(e["integrated"] + 16.0) <= 0.1 and 20 * np.log10(e["true_peak"]) <= -1.0


def test_round3_paths_equal_their_switched_off_versions_on_a_long_file(engine_ab):
    """Twenty minutes (BASELINE configs[1] in small; tools/ab_pipeline_fused.py does the same on the 60-min bench file): the split adeclick
    pipeline, Levinson-Durbin one lane per window and Pass 3 started inside Pass 2 are kernel / schedule changes only - the delivered PCM
    and every reported number equal the run with each of them switched off (A/B build: it holds the one-kernel adeclick)."""
    import hashlib
    engine = engine_ab
    w = int(0.02 * SR)
    burst = 0.35 * np.hanning(w) * np.sin(2 * np.pi * 180.0 * np.arange(w) / SR)
    base = np.asarray(synth.speech_like(60.0, SR, seed=95), np.float64)
    for pos in range(SR, base.size - SR, int(1.5 * SR)):
        base[pos:pos + w] += burst
    x = np.tile(base.astype(np.float32), 20)

    def run():
        engine.upload_pcm(x, SR, 1)
        r = H.process_audio(engine)
        return (hashlib.md5(engine.download_s16(4).tobytes()).hexdigest(), r.output_lufs, r.output_tp_db, r.measure.input_i, r.measure.input_tp,
                int(r.limiter.needed), r.final_.r128.true_peak, engine.timers()["declick_repaired"])
    ref = run()
    assert ref[5] == 1 and abs(ref[1] + 16.0) <= 0.1 and ref[2] <= -1.0
    for var in ("adeclick_fused", "no_early_plan", "dk_levinson_in_kernel"):
        with options(engine, **{var: True}):
            got = run()
        assert got == ref, var


@pytest.mark.gpu
def test_dynamic_loudnorm_workgroup_kernel_equals_the_one_wave_kernel(engine_ab):
    """The dynamic-mode fallback runs as a workgroup with the limiter's frame in LDS (k_loudnorm_dynamic_wg); option dyn_one_wave (A/B build) keeps the
    one-wave kernel that walks the ring in global memory.  Same arithmetic on the same values: the 192 kHz outputs are bit-identical --
    a stream the limiter holds down continuously (the reference's +13 dB offset= into a -1 dBTP ceiling), one it touches now and
    then (-9 dBTP), a quiet start (the output-meter phase), and a length that ends in a partial frame."""
    from jivetalking_amd import synth
    x = synth.speech_like(21.0, 192000, seed=41).astype(np.float64) * 2.5
    x = np.concatenate([x[: 192000 * 3] * 0.004, x])[: -4321]
    outs = {}
    for mode in ("wg", "one"):
        with options(engine_ab, dyn_one_wave=(mode == "one")):
            outs[mode] = [engine_ab.op_loudnorm_dynamic(x, target_tp=tp, offset=off)[0] for tp, off in ((-9.0, 0.0), (-1.0, 13.0), (-1.0, 0.0))]
    for a, b in zip(outs["wg"], outs["one"]):
        assert np.array_equal(a, b)
    assert abs(np.max(np.abs(outs["wg"][0])) - 10 ** (-9 / 20)) < 1e-12       # the limiter worked


@pytest.mark.gpu
def test_k_weighting_one_sweep_agrees_with_two_sweeps(engine_ab):
    """K-weighting as one sweep (zero-state energies + cross terms with the homogeneous-response table, the carried state added per
    chunk afterwards) against the two-sweep form it replaces (option kw_two_sweeps, A/B build): the momentary / short-term series and the gated
    figures agree to 1e-7 LU on every block above the -70 LUFS gate (1e-9 on speech) -- speech, and a signal with a DC offset and a
    30 Hz rumble: the carried state of the RLB high-pass is large there and the three terms cancel most of each other, which is where
    the one-sweep form loses digits (tools/kw_probe.py: 4e-9 LU at a DC offset of 0.3 under -34 LUFS of programme, 4e-5 LU on blocks
    50 dB under the gate); lengths that end inside a chunk, 44.1 / 48 / 96 kHz."""
    rng = np.random.default_rng(7)
    for sr, secs in ((48000, 12.3), (44100, 7.77), (96000, 5.01)):
        t = np.arange(int(sr * secs)) / sr
        x = synth.speech_like(secs, sr, seed=51).astype(np.float32)[: t.size]
        y = (0.3 + 0.2 * np.sin(2 * np.pi * 30.0 * t) + 0.01 * rng.standard_normal(t.size)).astype(np.float32)
        for sig in (x, y):
            a = engine_ab.op_ebur128(sig, sr)
            with options(engine_ab, kw_two_sweeps=True):
                b = engine_ab.op_ebur128(sig, sr)
            tol = 1e-9 if sig is x else 1e-7
            assert abs(a["integrated"] - b["integrated"]) < tol and abs(a["lra"] - b["lra"]) < tol
            assert np.array_equal(np.isfinite(a["M"]), np.isfinite(b["M"]))
            for key in ("M", "S"):
                live = np.isfinite(b[key]) & (b[key] > -70.0)
                assert live.any() and np.max(np.abs(a[key][live] - b[key][live])) < tol, (sr, key)


@pytest.mark.gpu
def test_host_arithmetic_staged_behind_the_analysis_chains_reports_the_same_numbers(engine):
    """The host side of a pass's analysis (gating, LRA, astats merges, the per-frame assembly) runs chain by chain while the later chains
    are still on the GPU, and the announced output regions are measured on a stream of their own: schedule changes only.  Every
    measurement of the run record (input, filtered, final, the four region samples, both loudnorm records) equals the run that waits for
    the whole analysis first (option no_staged_finish) with the regions behind the full chains (region_rot)."""
    import ctypes as C
    w = int(0.02 * SR)
    burst = 0.35 * np.hanning(w) * np.sin(2 * np.pi * 180.0 * np.arange(w) / SR)
    base = np.asarray(synth.speech_like(60.0, SR, seed=97), np.float64)
    for pos in range(SR, base.size - SR, int(1.5 * SR)):        # (plosive-like bursts: the limiter-prefix branch, Pass 3 inside Pass 2)
        base[pos:pos + w] += burst
    x = np.tile(base.astype(np.float32), 3)

    def run():
        engine.upload_pcm(x, SR, 1)
        r = H.process_audio(engine)
        parts = (r.input, r.filtered, r.measure, r.final_, r.loudnorm, r.filtered_room_tone, r.filtered_speech, r.final_room_tone, r.final_speech)
        return [C.string_at(C.addressof(p), C.sizeof(p)) for p in parts] + [engine.download_s16(4).tobytes()]
    ref = run()
    # (region_full_astats: the regions' astats with all its chains instead of the one sweep their three reported fields need)
    # (round 6: the early head's biquads beside Pass 1's analysis, astats' exponential-average chain behind its reduce chain, the noise
    #  floor by branch and bound -- schedule / pruning only: every measurement and every delivered byte the same without them)
    for env in ({"no_staged_finish": "1"}, {"region_rot": "0"}, {"region_rot": "3", "no_staged_finish": "1"}, {"region_full_astats": "1"},
                {"no_early_biquad": "1"}, {"as_avg_behind_spec": "1"}, {"nf_unpruned": "1"}, {"no_early_biquad": "1", "as_avg_behind_spec": "1", "nf_unpruned": "1"},
                {"dk_unsorted": "1"},       # (adeclick's solver lists in the order the front kernels appended them: pairing and queue order only)
                {"p2_device_join": "1", "dk_device_join": "1"}):      # (the two joins a step waits on inside the queue instead of on the host thread)
        with options(engine, **env):
            got = run()
        assert got == ref, env

