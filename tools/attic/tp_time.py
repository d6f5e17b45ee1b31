"""ebur128 op alone (30-min synthetic file): run under rocprofv3 --kernel-trace --stats to read the true-peak kernel in isolation.
python tools/tp_time.py [rate]"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from jivetalking_amd import Engine, synth
SR = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
x = synth.speech_like_torch(1800.0, SR, seed=1000, device="cuda:0").cpu().numpy()
e = Engine(0)
for _ in range(4):
    r = e.op_ebur128(x, SR)
print({k: (round(float(v), 4) if np.isscalar(v) else None) for k, v in r.items()} if isinstance(r, dict) else r)
