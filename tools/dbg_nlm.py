import sys, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine
from oracle import orc
orc.lib()
SR=48000
rng=np.random.default_rng(3)
x=(rng.standard_normal(SR*2)*10**(-66/20)).astype(np.float32)
e=Engine(0)
ref=orc.anlmdn(x,SR); got=e.op_anlmdn(x,SR)
d=np.abs(got-ref)
print("max err", d.max(), "scale", np.abs(x).max(), "n bad", (d>1e-5*np.abs(x).max()).sum(), "of", x.size)
bad=np.nonzero(d>1e-5*np.abs(x).max())[0]
print(bad[:20], bad[-5:] if bad.size else None)
if bad.size:
    H=577
    print("hop idx of bad", np.unique((bad+ (288+96))//H)[:20])
    i=bad[0]; print(x[i], ref[i], got[i])
print("frac changed ref", (ref!=x).mean(), "got", (got!=x).mean())
