"""The split adeclick pipeline (front + register-resident solvers) against the one-kernel version of round 2 (JT_ADECLICK_FUSED=1):
same operations on the same values, so the outputs must be bit-identical.  python tools/ab_declick_split.py [seconds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jivetalking_amd import Engine, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
e = Engine(0)
rc = 0
for seed, gain in ((11, 4.0), (12, 1.0), (13, 12.0)):
    x = np.asarray(synth.speech_like(min(secs, 60.0), 44100, seed=seed), np.float64)
    x = np.tile(x, int(np.ceil(secs / 60.0)))[: int(secs * 44100)] * gain
    rng = np.random.default_rng(seed)
    pos = rng.integers(1000, x.size - 1000, 200)
    x[pos] += rng.uniform(-0.5, 0.5, pos.size)                      # real clicks as well
    os.environ["JT_ADECLICK_FUSED"] = "1"
    a = e.op_adeclick(x, 44100)
    del os.environ["JT_ADECLICK_FUSED"]
    b = e.op_adeclick(x, 44100)
    same = np.array_equal(a, b)
    d = np.abs(a - b)
    print(f"seed {seed} gain {gain}: identical={same} differing={int((d > 0).sum())} max={d.max():.3g} changed_by_filter={int((a != x).sum())}")
    rc |= 0 if same else 1
sys.exit(rc)
