import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth
sr=44100
x=(synth.speech_like_torch(1200.0, sr, seed=3, device="cuda:0").cpu().numpy()*3.0).astype(np.float64)
e=Engine(0)
e.op_adeclick(x[:sr*10],sr)
t=time.time(); y,c=e.op_adeclick(x,sr,return_count=True); print("wall",time.time()-t,"repaired/window",c/(x.size/1212))
