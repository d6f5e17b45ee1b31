"""adeclick on a long signal: run-to-run determinism of k_adeclick_fast, and its distance to the sequential-order k_adeclick
(JT_ADECLICK_EXACT=1): samples that differ, the largest difference, repaired-sample counts.  python tools/declick_det.py [minutes]"""
import sys, os, numpy as np
sys.path.insert(0, '.')
import torch
from jivetalking_amd import Engine, synth
mins = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
SR = 44100
x = synth.speech_like_torch(mins * 60.0, SR, seed=1000, device="cuda:0").cpu().numpy().astype(np.float64)
x *= 10 ** (11.0 / 20.0)                      # the level Pass 4 hands to adeclick (after the loudnorm gain)
rng = np.random.default_rng(5)
for p in rng.integers(0, x.size, int(mins * 40)):          # sparse clicks so that the interpolator runs
    x[p] += rng.choice([-1.0, 1.0]) * rng.uniform(0.2, 0.6)
e = Engine(0)
a, ca = e.op_adeclick(x, SR, return_count=True)
b, cb = e.op_adeclick(x, SR, return_count=True)
print("fast run-to-run: differing samples", int(np.sum(a != b)), "max", float(np.max(np.abs(a - b))), "repaired", ca, cb)
os.environ["JT_ADECLICK_EXACT"] = "1"
r, cr = e.op_adeclick(x, SR, return_count=True)
d = np.abs(a - r)
big = d > 1e-9
print("fast vs sequential-order: repaired", ca, cr, " samples differing at all", int(np.sum(a != r)), " > 1e-9:", int(np.sum(big)),
      " max abs diff", float(d.max()), " windows touched (of %d)" % (x.size // 1212), len(set((np.nonzero(big)[0] // 1212).tolist())))
