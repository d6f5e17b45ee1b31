"""anlmdn: hop-pair kernel against the generic kernel over sample rates (8-96 kHz: every patch length parity, radii far below the lane layout) and
lengths around the hop size; prints mismatches above f32 round-off.  python tools/nlm_edge.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from jivetalking_amd import Engine, synth
e = Engine(0)
def run(generic, x, sr):
    if generic: os.environ["JT_NLM_GENERIC"] = "1"
    else: os.environ.pop("JT_NLM_GENERIC", None)
    return e.op_anlmdn(x, sr)
for sr in (8000, 11025, 16000, 24000, 44100, 48000, 64000, 88200, 96000):
    for n in (1, 7, 100, 531, 532, 1063, 5000, 30011):
        r = np.random.default_rng(n + sr)
        x = (1e-3 * r.standard_normal(n) + 0.05 * np.sin(np.arange(n) * 0.03)).astype(np.float32)
        a, b = run(True, x, sr), run(False, x, sr)
        d = float(np.max(np.abs(a - b)))
        if d > 2e-8 or not np.all(np.isfinite(b)): print("MISMATCH", sr, n, d)
print("edge sweep done")
