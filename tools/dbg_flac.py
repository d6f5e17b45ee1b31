"""Debug driver: GPU FLAC encoder vs the oracle decoder on assorted inputs."""
import sys, time, hashlib
sys.path.insert(0, "/root/repo")
import numpy as np
from jivetalking_amd.engine import Engine
from oracle import orc

e = Engine()
rng = np.random.default_rng(7)

def check(name, x, rate=44100):
    x = np.ascontiguousarray(x, np.int16)
    f, info = e.op_flac_encode(x, rate, md5=True, return_info=True)
    rc, y, oi = orc.flac_decode(f)
    ok = rc == 0 and y is not None and y.shape[0] == x.size and np.array_equal(y[:, 0], x.astype(np.int32))
    md5ok = bytes(oi.md5_stored) == bytes(oi.md5_decoded) == hashlib.md5(x.tobytes()).digest()
    si = (oi.sample_rate == rate and oi.channels == 1 and oi.bps == 16 and oi.total_samples == x.size and
          oi.min_framesize == oi.obs_min_framesize and oi.max_framesize == oi.obs_max_framesize and oi.min_blocksize == 4096)
    print(f"{name:28s} n={x.size:9d} bytes={len(f):9d} ratio={len(f)/(2*x.size):.3f} rc={rc} pcm={ok} md5={md5ok} si={si} "
          f"gpu={info['gpu_ms']:.2f}ms md5={info['md5_ms']:.1f}ms total={info['total_ms']:.1f}ms")
    return ok and md5ok and si

t = np.arange(44100 * 20) / 44100.0
speechy = (3000 * np.sin(2 * np.pi * 140 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 400 * rng.standard_normal(t.size)).astype(np.int16)
allok = True
allok &= check("speechy 20s", speechy)
allok &= check("silence", np.zeros(50000, np.int16))
allok &= check("dc", np.full(10000, -1234, np.int16))
allok &= check("white full-scale", rng.integers(-32768, 32768, 30000).astype(np.int16))
allok &= check("single sample", np.array([-32768], np.int16))
for n in (2, 9, 16, 17, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 8191, 8192, 8193 + 300):
    allok &= check(f"noise n={n}", (1000 * rng.standard_normal(n)).astype(np.int16))
allok &= check("alternating extremes", np.tile(np.array([32767, -32768], np.int16), 6000))
imp = np.zeros(20000, np.int16); imp[::997] = 32767; imp[5::1013] = -32768
allok &= check("sparse impulses", imp)
allok &= check("low noise +-1", rng.integers(-1, 2, 40000).astype(np.int16))
allok &= check("sine 1k", (20000 * np.sin(2 * np.pi * 1000 * t[:100000])).astype(np.int16))
allok &= check("48k rate", speechy[:30000], 48000)
allok &= check("odd rate 12345", speechy[:30000], 12345)
allok &= check("rate 11000", speechy[:30000], 11000)
allok &= check("rate 352800", speechy[:30000], 352800)
big = np.tile(speechy, 12)[: 44100 * 60 * 4]
allok &= check("4 min (utf8 3-byte frame no)", big)
print("ALL OK" if allok else "FAILURES")
