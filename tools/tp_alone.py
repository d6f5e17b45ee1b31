import sys; sys.path.insert(0, '.')
import numpy as np
from jivetalking_amd import Engine, synth
e = Engine(0)
x = np.tile(synth.speech_like(60.0, 48000, seed=3).astype(np.float32), 60)
for _ in range(3): r = e.op_ebur128(x, 48000)
print(r["integrated"], r["true_peak"])
