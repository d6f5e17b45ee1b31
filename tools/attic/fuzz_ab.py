"""Randomised A/B of this round's kernel rewrites against the kernels they replaced (each has an environment switch): random rates, lengths
and seeds; bit-identity where the arithmetic order was kept, f32 round-off where it was not.  python tools/fuzz_ab.py [cases]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from jivetalking_amd import Engine, synth, _lib as L
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
e = Engine(0)
rng = np.random.default_rng(2024)
def with_env(name, fn):
    os.environ[name] = "1"
    try: return fn()
    finally: os.environ.pop(name, None)
bad = 0
for c in range(ncases):
    sr = int(rng.choice([44100, 48000, 88200, 96000, 32000, 22050]))
    secs = float(rng.choice([0.02, 0.11, 0.6, 1.7, 3.1, 7.7, 19.3]))
    seed = int(rng.integers(1, 1 << 30))
    n = max(1, int(sr * secs))
    r = np.random.default_rng(seed)
    kind = c % 3
    if kind == 0: x = synth.speech_like(max(secs, 0.05), sr, seed=seed % 1000)[:n].astype(np.float32)
    elif kind == 1: x = (0.3 * r.standard_normal(n) * (0.05 + 0.95 * (np.sin(np.arange(n) * 2e-4 * (1 + c)) > 0))).astype(np.float32)
    else: x = (0.02 * r.standard_normal(n) + 0.2 * np.sin(np.arange(n) * 0.01 * (1 + c % 7))).astype(np.float32)
    # anlmdn: hop-pair kernel vs the generic one (sums in a different order: f32 round-off of the weighted mean)
    a = with_env("JT_NLM_GENERIC", lambda: e.op_anlmdn(x, sr)); b = e.op_anlmdn(x, sr)
    if np.max(np.abs(a - b)) > 3e-8 * max(1.0, float(np.max(np.abs(x)))): bad += 1; print("anlmdn", sr, secs, seed, np.max(np.abs(a - b)))
    # afftdn: grouped vs frame-at-a-time kernel, bit for bit, three modes
    for kw in ({}, {"track": True}, {"band_noise": [-35.0 - i for i in range(15)]}):
        a = with_env("JT_AFFTDN_OLD", lambda: e.op_afftdn(x, sr, 12.0, -50.0, **kw)); b = e.op_afftdn(x, sr, 12.0, -50.0, **kw)
        if not np.array_equal(a, b): bad += 1; print("afftdn", sr, secs, seed, list(kw), np.max(np.abs(a - b)))
    # f64 stream upsampler: eight waves vs four, via the limiter-prefix measurement
    s16 = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)
    for rate in (44100, 48000):
        lim = L.LimiterPlan(1, 0.25, 0.6)
        a = with_env("JT_UPS_NO_STREAM8", lambda: e.op_loudnorm_measure_s16(s16, rate, limiter=lim)); b = e.op_loudnorm_measure_s16(s16, rate, limiter=lim)
        if not all(np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True) for k in a): bad += 1; print("stream8", rate, secs, seed)
print("fuzz A/B:", ncases, "cases,", "clean" if bad == 0 else "%d mismatches" % bad)
