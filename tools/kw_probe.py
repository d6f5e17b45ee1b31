"""one-sweep vs two-sweep K-weighting: difference of the momentary series against the DC offset of the signal (cancellation)"""
import os, sys, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine
e = Engine(0); rng = np.random.default_rng(7); sr = 48000
t = np.arange(sr * 8) / sr
for dc in (0.0, 0.03, 0.3, 0.9):
    for nz in (0.01, 0.0001):
        y = (dc + nz * rng.standard_normal(t.size)).astype(np.float32)
        os.environ.pop("JT_KW_TWO_SWEEPS", None); a = e.op_ebur128(y, sr)
        os.environ["JT_KW_TWO_SWEEPS"] = "1"; b = e.op_ebur128(y, sr); os.environ.pop("JT_KW_TWO_SWEEPS", None)
        f = np.isfinite(a["M"]) & np.isfinite(b["M"])
        print(f"dc {dc:4.2f} noise {nz:6.4f}: I {b['integrated']:8.3f} LUFS, max |dM| {np.max(np.abs(a['M'][f] - b['M'][f])):.3g} LU, dI {abs(a['integrated'] - b['integrated']):.3g}")
