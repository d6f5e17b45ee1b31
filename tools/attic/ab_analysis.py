"""Digest of the analysis operators' outputs (astats, aspectralstats, ebur128) on fixed inputs: run before and after a kernel change that
must not move a bit.  python tools/ab_analysis.py"""
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from jivetalking_amd import Engine, synth
e = Engine(0)
def dig(o):
    h = hashlib.sha256()
    def walk(v):
        if isinstance(v, dict):
            for k in sorted(v): h.update(str(k).encode()); walk(v[k])
        elif isinstance(v, (list, tuple)):
            for x in v: walk(x)
        elif v is None: h.update(b"none")
        else: h.update(np.ascontiguousarray(np.asarray(v)).tobytes())
    walk(o); return h.hexdigest()[:16]
for sr, secs in ((48000, 61.3), (44100, 33.0), (48000, 0.9), (96000, 20.0), (48000, 900.0)):
    r = np.random.default_rng(11); nn = int(secs * sr)                      # (numpy input: the torch generator is not bit-repeatable)
    x = (np.cumsum(r.standard_normal(nn)) * 1e-3 % 0.4 - 0.2 + 0.05 * r.standard_normal(nn)).astype(np.float32) * (0.2 + 0.8 * (np.sin(np.arange(nn) * 2e-4) > 0))
    x = x.astype(np.float32)
    for name, fn in (("astats", e.op_astats), ("spectral", e.op_aspectralstats), ("ebur128", e.op_ebur128)):
        o = fn(x, sr)
        if isinstance(o, dict): print(sr, secs, name, " ".join("%s=%s" % (k, dig(o[k])[:6]) for k in sorted(o)))
        else: print(sr, secs, name, dig(o))
