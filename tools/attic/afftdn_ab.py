"""afftdn: the grouped kernel (k_afftdn_grp) against the frame-at-a-time kernel (JT_AFFTDN_OLD=1): outputs must be bit-identical
(static floor, custom profile, tn=1), `time`: launches of both on a 60-min file for rocprofv3 --stats.  python tools/afftdn_ab.py check|time"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from jivetalking_amd import Engine, synth
what = sys.argv[1] if len(sys.argv) > 1 else "check"
e = Engine(0)
def run(old, x, sr, **kw):
    if old: os.environ["JT_AFFTDN_OLD"] = "1"
    else: os.environ.pop("JT_AFFTDN_OLD", None)
    return e.op_afftdn(x, sr, 12.0, -50.0, **kw)
for sr, secs in () if what != "check" else ((48000, 47.3), (44100, 31.0), (48000, 0.5), (96000, 21.0), (88200, 7.7), (96000, 0.3), (48000, 1800.0), (96000, 600.0)):
    x = synth.speech_like_torch(secs, sr, seed=7, device="cuda:0").cpu().numpy()
    for kw in ({}, {"track": True}, {"band_noise": [-40.0 - i for i in range(15)]}):
        a, b = run(True, x, sr, **kw), run(False, x, sr, **kw)
        print(sr, secs, kw if "band_noise" not in kw else "custom", "identical" if np.array_equal(a, b) else "DIFF max %.3g at %d of %d" % (np.max(np.abs(a - b)), int(np.argmax(np.abs(a - b))), a.size))
if what == "time":        # run under rocprofv3 --kernel-trace --stats: 3 launches of each kernel and mode on a 60-min file
    tsr = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
    x = synth.speech_like_torch(3600.0 if tsr <= 48000 else 1800.0, tsr, seed=1000, device="cuda:0").cpu().numpy()
    for old in (True, False):
        for kw in ({}, {"track": True}):
            for _ in range(3): run(old, x, tsr, **kw)
