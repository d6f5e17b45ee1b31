"""Generator behind tools/gen_ffmpeg_golden.sh: ffmpeg CLI -> tests/golden/ffmpeg/{manifest.json, *.npz}.

One manifest entry per (fixture, filter chain).  Every entry names the oracle / engine operator that restates the chain
(`op` + `args`), so tests/test_ffmpeg_golden.py needs no knowledge of filter strings.  Inputs are produced by the repository's own
deterministic generators (jivetalking_amd/synth.py: the reference's hermetic fixture, testutil_test.go:28-135, and the speech-like
signal of the bench), written as WAV in the sample format under test, and stored next to the outputs so that the test reads exactly
what ffmpeg read.  numpy only; no GPU, no oracle.
"""
import argparse
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jivetalking_amd import synth  # noqa: E402


def wav_bytes(x, sr, ch, kind):
    """kind: 's16' | 's24' | 'f32' | 'f64'; x interleaved float in [-1, 1) (already quantised for integer kinds)."""
    x = np.asarray(x)
    if kind == "s16":
        data = np.clip(np.rint(x * 32768.0), -32768, 32767).astype("<i2").tobytes(); tag, bits = 1, 16
    elif kind == "s24":
        v = np.clip(np.rint(x * 8388608.0), -8388608, 8388607).astype("<i4")
        data = b"".join(struct.pack("<i", int(s))[:3] for s in v); tag, bits = 1, 24
    elif kind == "f32":
        data = x.astype("<f4").tobytes(); tag, bits = 3, 32
    else:
        data = x.astype("<f8").tobytes(); tag, bits = 3, 64
    ba = ch * bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, tag, ch, sr, sr * ba, ba, bits)
    return hdr + b"data" + struct.pack("<I", len(data)) + data


def probe_signal(sr, seconds, impulse_at=None, step_at=None, kind="f32", dither_dbfs=-80.0):
    """Alignment probe: dither at `dither_dbfs` (the LCG of testutil_test.go, so that no filter sees digital silence) plus ONE unit
    event - an impulse of 0.5 at sample `impulse_at`, or a step from 0 to 0.25 at sample `step_at`.  Where the event lands in a
    filter's output IS the filter's latency compensation / priming / flush behaviour."""
    n = int(sr * seconds)
    lcg = np.empty(n, np.float64)
    v = 12345
    for i in range(n):
        v = (v * 1664525 + 1013904223) & 0xFFFFFFFF
        lcg[i] = v / float(0xFFFFFFFF) * 2.0 - 1.0
    x = lcg * 10.0 ** (dither_dbfs / 20.0)
    if impulse_at is not None:
        x[int(impulse_at)] += 0.5
    if step_at is not None:
        x[int(step_at):] += 0.25
    if kind == "s16":
        return np.rint(x * 32768) / 32768
    return x.astype(np.float32).astype(np.float64)


def alignment_probes(seconds48=1.5, seconds44=1.5):
    """One entry per alignment / flush assumption of DESIGN.md section 3 (numbered as there), each with an impulse at n = 0, one
    mid-file and one as far before EOF as the filter's own reach (K+S for anlmdn, W = 3A for afftdn, W for adeclick, the look-ahead
    for alimiter, the filter length for swr), and a step.  Returns (fixtures, entries):
      fixtures[name] = (signal, rate, channels, kind)
      entries = [(entry name, fixture, -af string, raw format, op, args, output rate or None, note)]"""
    fx, ents = {}, []
    n48, n44 = int(48000 * seconds48), int(44100 * seconds44)

    def add(prefix, sr, kind, n, reach, af, fmt, op, args, rate, note):
        events = {"imp_head": dict(impulse_at=0), "imp_mid": dict(impulse_at=n // 2), "imp_tail": dict(impulse_at=n - 1 - reach),
                  "step_mid": dict(step_at=n // 2)}
        for ev, kw in events.items():
            fname = "probe_%dk_%s_%s_%d" % (sr // 1000, kind, ev, reach)
            if fname not in fx:
                fx[fname] = (probe_signal(sr, n / float(sr), kind=kind, **kw), sr, 1, kind)
            ents.append(("%s_%s" % (prefix, ev), fname, af, fmt, op, args, rate, note))

    add("a01_anlmdn_alignment", 48000, "f32", n48, 288 + 96, "anlmdn=s=0.00001:p=0.0060:r=0.0020:m=3", "f32le", "anlmdn",
        {"s": 0.00001, "p": 0.0060, "r": 0.0020, "m": 3.0}, None, "DESIGN s3 (1): output n <-> input n, zero-padded tail of K+S")
    add("a02_afftdn_alignment", 48000, "f32", n48, 1800, "afftdn=nr=12:nt=w:tn=0:nf=-55", "f32le", "afftdn", {"nr": 12.0, "nf": -55.0, "track": False}, None,
        "DESIGN s3 (2): frame t covers [tA-(W-A), tA+A), zero history, zero-flushed tail")
    add("a03_adeclick_alignment", 44100, "s16", n44, 2425, "volume=0.0dB,adeclick=t=1.7:w=55:o=50:m=s", "f64le", "adeclick",
        {"pre_gain_db": 0.0, "t": 1.7, "w": 55.0, "o": 50.0, "m": "s"}, None, "DESIGN s3 (3): FIFO primed with (W-hop)/2 zeros, zero-padded last windows")
    add("a04_alimiter_latency1_alignment", 44100, "s16", n44, 220,
        "volume=6.0dB,alimiter=limit=0.250000:attack=5:release=100:level_in=1:level_out=1:level=0:latency=1:asc=1:asc_level=0.8", "f64le", "alimiter",
        {"pre_gain_db": 6.0, "limit": 0.25, "attack": 5.0, "release": 100.0}, None, "DESIGN s3 (4): latency=1 trims lookahead-1 outputs, zero-flushed tail")
    add("a06_swr_48k_to_44k1_alignment", 48000, "f32", n48, 32, "aformat=sample_rates=44100:channel_layouts=mono:sample_fmts=s16", "s16le", "resample_s16",
        {"out_rate": 44100}, None, "DESIGN s3 (6): output m centred on input m*in/out, first output at m = 0, mirrored flush, length ceil(N*out/in)")
    add("a06_swr_44k1_to_192k_alignment", 44100, "s16", n44, 32, "aresample=192000,aformat=sample_fmts=flt", "f32le", "swr_up",
        {"out_rate": 192000}, None, "DESIGN s3 (6): the 192 kHz loudnorm / true-peak stream")
    return fx, ents


def run(ffmpeg, args):
    p = subprocess.run([ffmpeg, "-hide_banner", "-nostdin", "-y"] + args, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("ffmpeg failed: %s\n%s" % (" ".join(args), p.stderr[-2000:]))
    return p.stderr


def parse_ametadata(path):
    """ametadata=mode=print file -> list of {key: float} per frame."""
    frames, cur = [], None
    for line in open(path):
        line = line.strip()
        if line.startswith("frame:"):
            cur = {}; frames.append(cur)
        elif "=" in line and cur is not None:
            k, v = line.split("=", 1)
            try:
                cur[k] = float(v)
            except ValueError:
                cur[k] = float("nan") if v.lower() == "nan" else (float("-inf") if v.startswith("-inf") else float("inf"))
    return frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ffmpeg", default="ffmpeg"); ap.add_argument("--version", default=""); ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="jtgolden")
    manifest = {"ffmpeg": a.version, "entries": []}

    # ---- fixtures
    fx = {}
    fix3 = np.asarray(synth.reference_fixture(3.0, 48000, noise_dbfs=-50.0), np.float64) / 32768.0     # testutil_test.go generator (int16 grid)
    fx["ref3_48k_s16"] = (fix3, 48000, 1, "s16")
    sp = np.asarray(synth.speech_like(8.0, 48000, seed=101), np.float64)
    fx["speech_48k_f32"] = (sp.astype(np.float32).astype(np.float64), 48000, 1, "f32")
    fx["speech_48k_s16"] = (np.rint(sp * 32768) / 32768, 48000, 1, "s16")
    l = np.asarray(synth.speech_like(6.0, 96000, seed=102), np.float64); r = 0.8 * np.roll(l, 37) + 0.1 * np.asarray(synth.speech_like(6.0, 96000, seed=103), np.float64)
    st = np.empty(l.size * 2); st[0::2] = l; st[1::2] = r
    fx["speech_96k_stereo_f32"] = (st.astype(np.float32).astype(np.float64), 96000, 2, "f32")
    st16 = np.rint(st * 32768) / 32768
    fx["speech_96k_stereo_s16"] = (st16, 96000, 2, "s16")
    s441 = np.asarray(synth.speech_like(8.0, 44100, seed=104), np.float64)
    fx["speech_44k1_s16"] = (np.rint(s441 * 32768) / 32768, 44100, 1, "s16")
    fx["speech_44k1_s24"] = (np.rint(s441 * 8388608) / 8388608, 44100, 1, "s24")
    fx["speech_44k1_s16_2s5"] = (fx["speech_44k1_s16"][0][: int(44100 * 2.5)].copy(), 44100, 1, "s16")
    s30 = np.asarray(synth.speech_like(30.0, 44100, seed=105), np.float64)
    fx["speech_44k1_s16_30s"] = (np.rint(s30 * 32768) / 32768, 44100, 1, "s16")
    fx["speech_44k1_s16_3s5"] = (fx["speech_44k1_s16"][0][: int(44100 * 3.5)].copy(), 44100, 1, "s16")
    probe_fx, probe_entries = alignment_probes()
    fx.update(probe_fx)
    paths = {}
    for name, (x, sr, ch, kind) in fx.items():
        paths[name] = os.path.join(tmp, name + ".wav")
        open(paths[name], "wb").write(wav_bytes(x, sr, ch, kind))

    def raw_out(name, fixture, af, fmt, op, args, rate=None, note=""):
        """one filter chain -> raw PCM in `fmt` (f32le / f64le / s16le)"""
        out = os.path.join(tmp, name + ".raw")
        codec = {"f32le": "pcm_f32le", "f64le": "pcm_f64le", "s16le": "pcm_s16le", "s32le": "pcm_s32le"}[fmt]
        run(a.ffmpeg, ["-i", paths[fixture], "-af", af, "-f", fmt, "-c:a", codec] + (["-ar", str(rate)] if rate else []) + [out])
        dt = {"f32le": "<f4", "f64le": "<f8", "s16le": "<i2", "s32le": "<i4"}[fmt]
        y = np.fromfile(out, dt)
        x, sr, ch, kind = fx[fixture]
        np.savez_compressed(os.path.join(a.out, name + ".npz"), x=x.astype(np.float64 if kind == "f64" else np.float32) if kind.startswith("f") else x, y=y)
        manifest["entries"].append({"name": name, "kind": "pcm", "fixture": fixture, "rate": sr, "channels": ch, "source_format": kind,
                                    "af": af, "out_format": fmt, "op": op, "args": args, "note": note})

    def meta_out(name, fixture, af, op, args, note=""):
        """analysis chain + ametadata print -> per-frame dictionaries"""
        log = os.path.join(tmp, name + ".log")
        run(a.ffmpeg, ["-i", paths[fixture], "-af", af + ",ametadata=mode=print:file=" + log.replace(":", "\\:"), "-f", "null", "-"])
        frames = parse_ametadata(log)
        x, sr, ch, kind = fx[fixture]
        np.savez_compressed(os.path.join(a.out, name + ".npz"), x=x)
        json.dump(frames, open(os.path.join(a.out, name + ".frames.json"), "w"))
        manifest["entries"].append({"name": name, "kind": "metadata", "fixture": fixture, "rate": sr, "channels": ch, "source_format": kind,
                                    "af": af, "op": op, "args": args, "note": note})

    def loudnorm_json(name, fixture, af, op, args):
        err = run(a.ffmpeg, ["-i", paths[fixture], "-af", af, "-f", "null", "-"])
        m = re.search(r"\{[^{}]*\"input_i\"[^{}]*\}", err, re.S)
        if not m:
            raise RuntimeError("no loudnorm JSON in ffmpeg output for " + name)
        x, sr, ch, kind = fx[fixture]
        np.savez_compressed(os.path.join(a.out, name + ".npz"), x=x)
        json.dump(json.loads(m.group(0)), open(os.path.join(a.out, name + ".loudnorm.json"), "w"))
        manifest["entries"].append({"name": name, "kind": "loudnorm", "fixture": fixture, "rate": sr, "channels": ch, "source_format": kind,
                                    "af": af, "op": op, "args": args})

    F = "speech_48k_f32"
    # ---- Pass-2 filters one by one on the f32 fixture (fltp / dblp as libavfilter negotiates them), then the golden chain
    raw_out("hp_lp_tdii", F, "highpass=f=80:poles=2:width_type=q:width=0.707:normalize=1:a=tdii,lowpass=f=20500:poles=2:width_type=q:width=0.707:normalize=1:a=tdii",
            "f32le", "biquad_hp_lp", {"hp": 80.0, "lp": 20500.0, "q": 0.707})
    raw_out("anlmdn", F, "anlmdn=s=0.00001:p=0.0060:r=0.0020:m=3", "f32le", "anlmdn", {"s": 0.00001, "p": 0.0060, "r": 0.0020, "m": 3.0})
    raw_out("afftdn_nf", F, "afftdn=nr=12:nt=w:tn=0:nf=-55", "f32le", "afftdn", {"nr": 12.0, "nf": -55.0, "track": False})
    raw_out("afftdn_tn1", F, "afftdn=nr=12:nt=w:tn=1", "f32le", "afftdn", {"nr": 12.0, "nf": -50.0, "track": True})
    bn = [6, 5, 4, 3, 2, 1, 0, -1, -2, -3, -4, -5, -6, -6, -6]
    raw_out("afftdn_custom", F, "afftdn=nr=12:nt=custom:bn=" + "|".join("%.1f" % v for v in bn) + ":tn=0:nf=-48", "f32le", "afftdn",
            {"nr": 12.0, "nf": -48.0, "track": False, "band_noise": bn})
    raw_out("agate", F, "agate=threshold=0.010000:ratio=2.0:attack=5.00:release=200:range=0.1995:knee=3.0:detection=rms:makeup=1.0", "f64le", "agate",
            {"threshold": 0.01, "ratio": 2.0, "attack": 5.0, "release": 200.0, "range": 0.1995, "knee": 3.0, "makeup": 1.0})
    raw_out("acompressor", F, "acompressor=threshold=0.125893:ratio=3.0:attack=10:release=200:makeup=1.00:knee=4.0:detection=rms:mix=1.00", "f64le", "acompressor",
            {"threshold": 0.125893, "ratio": 3.0, "attack": 10.0, "release": 200.0, "makeup": 1.0, "knee": 4.0, "mix": 1.0})
    raw_out("deesser", F, "deesser=i=0.60:m=0.50:f=0.80", "f64le", "deesser", {"i": 0.6, "m": 0.5, "f": 0.8})
    raw_out("resample_s16", F, "aformat=sample_rates=44100:channel_layouts=mono:sample_fmts=s16", "s16le", "resample_s16", {"out_rate": 44100})
    raw_out("golden_pass2_chain", F,
            "aformat=channel_layouts=mono,highpass=f=80:poles=2:width_type=q:width=0.707:normalize=1:a=tdii,lowpass=f=20500:poles=2:width_type=q:width=0.707:normalize=1:a=tdii,"
            "anlmdn=s=0.00001:p=0.0060:r=0.0020:m=3,afftdn=nr=12:nt=w:tn=1,agate=threshold=0.010000:ratio=2.0:attack=5.00:release=200:range=0.1995:knee=3.0:detection=rms:makeup=1.0,"
            "acompressor=threshold=0.125893:ratio=3.0:attack=10:release=200:makeup=1.00:knee=4.0:detection=rms:mix=1.00,"
            "aformat=sample_rates=44100:channel_layouts=mono:sample_fmts=s16,asetnsamples=n=4096", "s16le", "pass2_default_chain", {}, note="filters_test.go:298-311")
    # ---- down-mix and the band graph per source format
    raw_out("downmix_f32", "speech_96k_stereo_f32", "aformat=channel_layouts=mono,anull,aformat=sample_fmts=fltp", "f32le", "downmix", {"mode": 0},
            note="a float-only consumer follows: internal + output FLTP, 1/sqrt2 each")
    raw_out("downmix_s16_to_fltp", "speech_96k_stereo_s16", "aformat=channel_layouts=mono:sample_fmts=fltp", "f32le", "downmix", {"mode": 0},
            note="s16 source, fltp consumer (Pass 1 / Pass 2): still 1/sqrt2 in float")
    raw_out("downmix_s16_int", "speech_96k_stereo_s16", "aformat=channel_layouts=mono:sample_fmts=s16p", "s16le", "downmix", {"mode": 1},
            note="s16 source, s16p consumer (band graphs): integer matrix 0.5 / 0.5")
    for nm, fxn in (("band_s16", "speech_48k_s16"), ("band_f32", F), ("band_s24", "speech_44k1_s24")):
        meta_out(nm, fxn, "aformat=channel_layouts=mono,atrim=start=1.000000:duration=5.000000,asetpts=PTS-STARTPTS,highpass=f=80.000000:p=2,lowpass=f=125.000000:p=2,astats=metadata=1:measure_perchannel=0",
                 "band_rms", {"start": 1.0, "duration": 5.0, "lo": 80.0, "hi": 125.0}, note="analyser_bands.go:33")
    # ---- analysis graphs
    for nm, fxn in (("pass1_ref3", "ref3_48k_s16"), ("pass1_speech", F), ("pass1_96k_stereo", "speech_96k_stereo_f32")):
        meta_out(nm, fxn, "aformat=channel_layouts=mono,astats=metadata=1:measure_perchannel=all,aspectralstats=win_size=2048:win_func=hann:measure=all,ebur128=metadata=1:peak=sample+true:dualmono=true:target=-16",
                 "pass1_analysis", {}, note="filters.go:42-45,624-626,684-689")
    meta_out("region_44k1", "speech_44k1_s16", "atrim=start=1.500000:duration=4.000000,asetpts=PTS-STARTPTS,astats=metadata=1:measure_perchannel=0,aspectralstats=measure=all,ebur128=metadata=1:peak=sample+true",
             "region_measure", {"start": 1.5, "duration": 4.0}, note="analyser_output.go:18")
    # ---- Pass 3 / Pass 4 on the 44.1 kHz s16 fixture
    P = "speech_44k1_s16"
    loudnorm_json("pass3_measure", P, "loudnorm=I=-16.0:TP=-1.0:LRA=20.0:dual_mono=true:print_format=json", "loudnorm_measure", {"prefix": None})
    loudnorm_json("pass3_measure_limited", P, "volume=6.0dB,alimiter=limit=0.500000:attack=5:release=100:level_in=1:level_out=1:level=0:latency=1:asc=1:asc_level=0.8,"
                  "loudnorm=I=-16.0:TP=-1.0:LRA=20.0:dual_mono=true:print_format=json", "loudnorm_measure", {"prefix": {"pre_gain_db": 6.0, "limit": 0.5}})
    # the dynamic fallback of Pass 4 (second-pass values whose LRA exceeds the target) followed by the reference's own aresample, and the
    # first pass on clips of 2.5 s (shorter than loudnorm's first frame: one gain), 4 s and 30 s: those pin how the filter's input meter
    # treats its flush frame
    raw_out("loudnorm_dynamic", P, "loudnorm=I=-16.00:TP=-1.00:LRA=20.0:measured_I=-24.00:measured_TP=-6.00:measured_LRA=25.00:measured_thresh=-34.50:offset=0.50:"
            "dual_mono=true:linear=true:print_format=json,aresample=44100", "f64le", "loudnorm_dynamic",
            {"target_i": -16.0, "target_tp": -1.0, "target_lra": 20.0, "measured": [-24.0, 25.0, -6.0, -34.5], "offset": 0.5}, rate=44100)
    raw_out("loudnorm_dynamic_first_pass", P, "loudnorm=I=-16.0:TP=-9.0:LRA=20.0:dual_mono=true:print_format=json,aresample=44100", "f64le", "loudnorm_dynamic",
            {"target_i": -16.0, "target_tp": -9.0, "target_lra": 20.0, "measured": None, "offset": 0.0}, rate=44100, note="a -9 dB ceiling keeps the limiter busy")
    for nm, fxn in (("pass3_measure_2s5", "speech_44k1_s16_2s5"), ("a12a_loudnorm_flush_3s5", "speech_44k1_s16_3s5"), ("pass3_measure_30s", "speech_44k1_s16_30s")):
        loudnorm_json(nm, fxn, "loudnorm=I=-16.0:TP=-1.0:LRA=20.0:dual_mono=true:print_format=json", "loudnorm_measure", {"prefix": None})
    raw_out("alimiter_level", P, "volume=12.0dB,alimiter=limit=0.500000:attack=5:release=100:level_in=1:level_out=1:level=0:latency=1:asc=1:asc_level=0.8", "f64le", "alimiter",
            {"pre_gain_db": 12.0, "limit": 0.5, "attack": 5.0, "release": 100.0})
    raw_out("alimiter_brickwall", P, "volume=14.0dB,alimiter=limit=0.803526:attack=1:release=50:level_in=1:level_out=1:level=0:latency=1:asc=1:asc_level=0.8", "f64le", "alimiter",
            {"pre_gain_db": 14.0, "limit": 0.803526, "attack": 1.0, "release": 50.0})
    raw_out("adeclick", P, "volume=8.0dB,adeclick=t=1.7:w=55:o=50:m=s", "f64le", "adeclick", {"pre_gain_db": 8.0, "t": 1.7, "w": 55.0, "o": 50.0, "m": "s"})
    # ---- one impulse / step probe per alignment assumption of DESIGN.md section 3
    for (nm, fxn, af, fmt, op, args, rate, note) in probe_entries:
        raw_out(nm, fxn, af, fmt, op, args, rate=rate, note=note)
    json.dump(manifest, open(os.path.join(a.out, "manifest.json"), "w"), indent=1)
    print("wrote %d entries to %s" % (len(manifest["entries"]), a.out))


if __name__ == "__main__":
    main()
