"""Per-sample cost of the alimiter kernel's state machine on one continuously hot segment (one lane does all the work)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine
e = Engine(0)
rng = np.random.default_rng(1)
for name, x in (("noise, every sample near the limit", rng.uniform(-1, 1, 88200)), ("220 Hz tone above the limit", 0.9 * np.sin(2 * np.pi * 220 * np.arange(88200) / 44100))):
    for att, rel in ((5.0, 100.0), (1.0, 50.0)):
        e.op_alimiter(x[:4410], 44100, 0.5, att, rel)
        t0 = time.time(); y = e.op_alimiter(x, 44100, 0.5, att, rel); dt = time.time() - t0
        print(f"{name}: attack {att} ms release {rel} ms: {dt * 1e3:.1f} ms for {x.size} samples = {dt / x.size * 1e9:.0f} ns per sample")
