import sys, numpy as np, os
sys.path.insert(0, '.')
import torch
from jivetalking_amd import Engine, synth
SR=48000
x = synth.speech_like_torch(120.0, SR, seed=1000, device="cuda:0").cpu().numpy()
e = Engine(0)
a = e.op_anlmdn(x, SR); b = e.op_anlmdn(x, SR); c = e.op_anlmdn(x, SR)
print("run-to-run differing samples:", int(np.sum(a != b)), int(np.sum(a != c)))
d = np.nonzero(a != b)[0]
if d.size: print(d[:20], (d[:20] - 0) % 577, a[d[:5]], b[d[:5]])
os.environ["JT_NLM_OLD"] = "1"
o = e.op_anlmdn(x, SR); o2 = e.op_anlmdn(x, SR)
print("old run-to-run:", int(np.sum(o != o2)), " new vs old max diff:", float(np.max(np.abs(a - o))), "n diff", int(np.sum(a != o)))
