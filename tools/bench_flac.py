"""FLAC output leg on a 60-minute mono 44.1 kHz s16 signal: GPU encode time, MD5 time, oracle decode check."""
import sys, time, hashlib
sys.path.insert(0, "/root/repo")
import numpy as np
from jivetalking_amd.engine import Engine
from jivetalking_amd import synth

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
check = "--check" in sys.argv
e = Engine()
sr = 44100
base = np.asarray(synth.speech_like(60.0, sr, seed=3, speech_dbfs=-20.0), np.float64)
x = np.tile(base, int(np.ceil(minutes)))[: int(minutes * 60 * sr)]
pcm = np.clip(x * 32768, -32768, 32767).astype(np.int16)
for md5 in (False, True):
    for rep in range(3):
        t0 = time.perf_counter()
        f, info = e.op_flac_encode(pcm, sr, md5=md5, return_info=True)
        dt = (time.perf_counter() - t0) * 1e3
    print(f"md5={md5}: {len(f)} bytes ratio {len(f)/(2*pcm.size):.3f} gpu {info['gpu_ms']:.2f} ms md5 {info['md5_ms']:.1f} ms "
          f"total {info['total_ms']:.1f} ms (call incl. H2D {dt:.1f} ms) frames {info['frames']}")
if check:
    from oracle import orc
    t0 = time.perf_counter()
    rc, y, oi = orc.flac_decode(f)
    print("oracle decode rc", rc, "equal", np.array_equal(y[:, 0], pcm), "md5", bytes(oi.md5_stored) == bytes(oi.md5_decoded),
          f"{time.perf_counter()-t0:.1f}s")
