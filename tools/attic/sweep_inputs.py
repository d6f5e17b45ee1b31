"""Robustness sweep of jt_process_audio over sample rates, durations, channel counts and degenerate signals: every case must either
deliver an output or return a JT_E_* code -- never fault.  python tools/sweep_inputs.py"""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
from jivetalking_amd import Engine, synth, hostlogic as H, _lib as L
e = Engine(0)
base = H.default_config()
ok = bad = 0
def run(tag, x, sr, ch):
    global ok, bad
    try:
        e.upload_pcm(x, sr, ch)
        r = H.process_audio(e, base)
        out = e.download_s16(4)
        print(f"{tag:42s} ok   I={r.output_lufs:7.2f} TP={r.output_tp_db:6.2f} dyn={r.loudnorm.normalization_type_dynamic} n_out={out.size}")
        ok += 1
    except L.JtError as ex:
        print(f"{tag:42s} code {ex.code}: {str(ex)[:90]}")
        bad += 1
for sr in (8000, 16000, 32000, 44100, 48000, 88200, 96000, 192000):
    for dur in (0.3, 1.0, 2.9, 3.0, 3.05, 5.0, 12.0):
        x = synth.speech_like(dur, sr, seed=int(sr + dur * 10))
        run(f"speech {sr} Hz {dur} s mono", x, sr, 1)
for dur in (2.0, 6.0):
    sr = 48000
    l = synth.speech_like(dur, sr, seed=7); st = np.empty(l.size * 2, np.float32); st[0::2] = l; st[1::2] = 0.5 * l
    run(f"speech 48000 Hz {dur} s stereo", st, sr, 2)
sr = 48000; t = np.arange(sr * 8) / sr
run("dc offset 0.2", np.full(sr * 5, 0.2, np.float32), sr, 1)
run("square wave full scale", np.sign(np.sin(2 * np.pi * 100 * t)).astype(np.float32), sr, 1)
run("tone -80 dBFS", (1e-4 * np.sin(2 * np.pi * 440 * t)).astype(np.float32), sr, 1)
run("white noise -6 dBFS", (0.5 * np.random.default_rng(1).standard_normal(sr * 8)).clip(-1, 1).astype(np.float32), sr, 1)
run("impulses", np.where(np.arange(sr * 6) % 4800 == 0, 0.9, 0.0).astype(np.float32), sr, 1)
run("NaN in input", np.concatenate([synth.speech_like(4.0, sr, seed=3), [np.nan], synth.speech_like(1.0, sr, seed=4)]).astype(np.float32), sr, 1)
print("delivered", ok, "refused", bad)
