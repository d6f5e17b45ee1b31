"""CPU timing of the Pass-1 host stage (interval construction + VAD / speech election) on an hour-sized synthetic metadata set."""
import ctypes as C, sys, time, random
import numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import hostlogic as H, _lib as L
lib = H.lib()
sr, F, blk = 48000, 4096, 4800
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n = int(minutes * 60 * sr)
nfr = (n + F - 1) // F
n_meta = n // blk
rng = np.random.default_rng(5)
# speech-like level track: alternating talk / pause segments
lvl = np.empty(n_meta)
i = 0
while i < n_meta:
    talk = rng.integers(20, 120); pause = rng.integers(3, 25)
    lvl[i:i + talk] = -18 + rng.normal(0, 3, min(talk, n_meta - i)); i += talk
    if i < n_meta:
        lvl[i:i + pause] = -55 + rng.normal(0, 2, min(pause, n_meta - i)); i += pause
meta = (L.FrameMeta * n_meta)()
for k in range(n_meta):
    m = meta[k]
    m.momentary = lvl[k]; m.shortterm = lvl[k] - 1; m.true_peak = 10 ** ((lvl[k] + 12) / 20); m.sample_peak = 10 ** ((lvl[k] + 11) / 20)
    for j, key in enumerate(L.SPECTRAL_KEYS):
        setattr(m.spectral, key, float(rng.uniform(0.1, 1) * (1000 if key in ("centroid", "spread", "rolloff") else 1)))
flv = np.interp(np.arange(nfr) * F / blk, np.arange(n_meta), lvl)
ss = (10 ** (flv / 20)) ** 2 * F
pk = 10 ** ((flv + 10) / 20)
out = (H.Interval * (n // (sr // 5) + 32))()
a1 = L.Analysis()
a1.r128.integrated = -20.0; a1.r128.true_peak = 0.5; a1.r128.sample_peak = 0.45; a1.r128.lra = 8.0
a1.astats.rms_trough = -60.0; a1.astats.rms_level = -25.0
m = H.Measurements()
for rep in range(5):
    t0 = time.perf_counter()
    k = lib.jt_host_build_intervals(sr, C.c_int64(n), F, 1, ss.ctypes.data_as(C.POINTER(C.c_double)), pk.ctypes.data_as(C.POINTER(C.c_double)),
                                    C.c_int64(nfr), meta, C.c_int64(n_meta), 1, out, C.c_int64(len(out)))
    t1 = time.perf_counter()
    rc = lib.jt_host_detect(C.byref(a1), out, C.c_int64(k), C.c_double(n / sr), C.c_double(-16.0), 1, C.byref(m))
    t2 = time.perf_counter()
    print(f"intervals {k}: build {1e3 * (t1 - t0):.3f} ms, detect {1e3 * (t2 - t1):.3f} ms (rc {rc}, speech regions {m.n_speech_regions})")
