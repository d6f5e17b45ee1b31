"""anlmdn: the wave-per-hop-pair kernel (any patch length, search radius up to the lane layout's) against the generic kernel (JT_NLM_GENERIC=1)
at the rates whose defaults do not fill the layout (44.1 / 88.2 / 22.05 kHz) and at 48 / 96 kHz; max |difference| and timing.
python tools/ab_nlm_generic.py [check|time <rate>]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from jivetalking_amd import Engine, synth
what = sys.argv[1] if len(sys.argv) > 1 else "check"
e = Engine(0)
def run(generic, x, sr):
    if generic: os.environ["JT_NLM_GENERIC"] = "1"
    else: os.environ.pop("JT_NLM_GENERIC", None)
    return e.op_anlmdn(x, sr)
if what == "check":
    for sr, secs in ((44100, 20.0), (44100, 0.7), (88200, 6.0), (22050, 9.0), (32000, 5.0), (48000, 8.0), (96000, 4.0)):
        x = synth.speech_like(secs, sr, seed=3).astype(np.float32)
        a, b = run(True, x, sr), run(False, x, sr)
        d = np.abs(a - b); chg = np.abs(b - x) > 0
        print(sr, secs, "max |fast - generic| %.3g  (changed samples %d of %d, max change %.3g)" % (d.max(), int(chg.sum()), x.size, np.abs(b - x).max()))
else:
    import torch
    sr = int(sys.argv[2]) if len(sys.argv) > 2 else 44100
    x = synth.speech_like_torch(1800.0, sr, seed=1000, device="cuda:0").cpu().numpy()
    for g in (True, False):
        for _ in range(3): run(g, x, sr)
