import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth
SR=48000
x = synth.speech_like_torch(120.0, SR, seed=1000, device="cuda:0").cpu().numpy()
e=Engine(0)
got=e.op_anlmdn(x,SR)
ch = got!=x
print("changed frac", ch.mean())
# per-second rms and changed fraction
for s in range(0,40):
    seg=slice(s*SR,(s+1)*SR)
    print(s, "rms dB %.1f" % (20*np.log10(np.sqrt(np.mean(x[seg]**2))+1e-12)), "changed %.3f" % ch[seg].mean())
