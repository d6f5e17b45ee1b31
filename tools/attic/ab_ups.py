"""Digest of the Pass-3 measurement (s16 -> 192 kHz f32 stream -> K-weighting -> loudnorm statistics) and of the dynamic-mode loudnorm
(f64 stream) on fixed inputs: run before and after a change of the stream upsampler that must not move a bit.
`time`: a few launches on a 60-min 44.1 kHz stream for rocprofv3 --stats.  python tools/ab_ups.py check|time"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from jivetalking_amd import Engine
what = sys.argv[1] if len(sys.argv) > 1 else "check"
e = Engine(0)
def sig(sr, secs, seed):
    r = np.random.default_rng(seed); n = int(sr * secs)
    x = 0.2 * r.standard_normal(n) * (0.1 + 0.9 * (np.sin(np.arange(n) * 9e-5) > 0))
    return np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)
import hashlib
def dig(o):
    h = hashlib.sha256()
    if isinstance(o, dict):
        for k in sorted(o): h.update(k.encode()); h.update(np.ascontiguousarray(np.asarray(o[k])).tobytes())
    elif isinstance(o, (tuple, list)):
        for v in o: h.update(dig(v).encode())
    else: h.update(np.ascontiguousarray(np.asarray(o)).tobytes())
    return h.hexdigest()[:16]
if what == "check":
    for sr, secs in ((44100, 47.3), (44100, 3.01), (44100, 0.4), (22050, 12.0), (88200, 9.0), (48000, 7.0), (44100, 901.0)):
        x = sig(sr, secs, 5)
        print(sr, secs, dig(e.op_loudnorm_measure_s16(x, sr)))
    from jivetalking_amd import _lib as L
    print("limiter-prefix measure (f64 stream)", dig(e.op_loudnorm_measure_s16(sig(44100, 30.0, 6), 44100, limiter=L.LimiterPlan(1, 0.25, 1.0))))
else:
    x = sig(44100, 3600.0, 9)
    for _ in range(4): e.op_loudnorm_measure_s16(x, 44100)
