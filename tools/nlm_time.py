"""anlmdn alone on a 60-min synthetic file: event-timed launch duration.  python tools/nlm_time.py [minutes]"""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
from jivetalking_amd import Engine, synth
mins = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
SR = 48000
x = synth.speech_like_torch(mins * 60.0, SR, seed=1000, device="cuda:0").cpu().numpy()
e = Engine(0)
import hashlib
ts = []
for _ in range(6):
    y = e.op_anlmdn(x, SR)
    ts.append(e.timers()["nlm_ms"])          # the last launch, HIP events on the handle's stream
flops = x.size * 192 * 6
print("output md5", hashlib.md5(np.asarray(y).tobytes()).hexdigest())
print("anlmdn launch ms:", [round(t, 3) for t in ts], " best", round(min(ts), 3), " non-FMA VALU frac", round(flops / (min(ts) * 1e-3) / 78.6e12, 4))
