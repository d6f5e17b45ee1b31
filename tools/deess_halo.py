"""How much warm-up the de-esser's chunks need: output with a halo of H samples against a halo of 262144, on signals with sustained
sibilance at several intensities (JT_DEESS_HALO / JT_DEESS_CHUNK are read by the launch).  python tools/deess_halo.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from jivetalking_amd import Engine, synth
from jivetalking_amd.engine import default_filter_params
e = Engine(0)
sr = 48000
def sig(kind, secs, seed):
    r = np.random.default_rng(seed); n = int(secs * sr); t = np.arange(n) / sr
    if kind == "speech": return synth.speech_like(secs, sr, seed=seed).astype(np.float32)
    if kind == "hiss": return (0.2 * r.standard_normal(n) * (0.2 + 0.8 * (np.sin(2 * np.pi * 0.7 * t) > 0))).astype(np.float32)          # long sibilant stretches
    if kind == "tone8k": return (0.5 * np.sin(2 * np.pi * 8000 * t) * (0.3 + 0.7 * (np.sin(2 * np.pi * 0.3 * t) > 0)) + 0.01 * r.standard_normal(n)).astype(np.float32)
    if kind == "loudhiss": return np.clip(0.7 * r.standard_normal(n), -1, 1).astype(np.float32)
def run(x, inten, halo, chunk=3072):
    os.environ["JT_DEESS_HALO"] = str(halo); os.environ["JT_DEESS_CHUNK"] = str(chunk)
    p = default_filter_params(); p.gate_enabled = 0; p.comp_enabled = 0; p.deess_enabled, p.deess_i = 1, inten
    return e.op_dynamics(x, sr, p).astype(np.float64)
for kind in ("speech", "hiss", "tone8k", "loudhiss"):
    x = sig(kind, 12.0, 4)
    for inten in (0.2, 0.5, 1.0):
        ref = run(x, inten, 262144)
        print(kind, inten, " ".join("halo %d: %.2e" % (h, np.max(np.abs(run(x, inten, h) - ref))) for h in (2048, 4096, 8192, 16384, 32768)), " (output rms %.3g, max |y - x| %.3g)" % (np.sqrt(np.mean(ref ** 2)), np.max(np.abs(ref - x))))
