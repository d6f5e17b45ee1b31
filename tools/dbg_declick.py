import sys, time, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth
sr=44100
x=(synth.speech_like(120.0, sr, seed=3)*3.0).astype(np.float64)
e=Engine(0)
for _ in range(2):
    t=time.time(); y,c=e.op_adeclick(x,sr,return_count=True); print("wall",time.time()-t,"repaired",c, c/(x.size/1212))
