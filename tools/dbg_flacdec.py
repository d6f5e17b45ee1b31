"""Debug driver: GPU FLAC / WAV decoder vs the oracle on coverage streams."""
import sys, time, struct
sys.path.insert(0, "/root/repo")
import numpy as np
from jivetalking_amd.engine import Engine
from oracle import orc

e = Engine()
rng = np.random.default_rng(1)
allok = True
def sig(n, ch, bps, kind=0):
    if kind == 0:
        x = (rng.standard_normal((n, ch)).cumsum(0) * (1 << (bps - 6)) / 30)
    else:
        x = rng.standard_normal((n, ch)) * (1 << (bps - 3))
    return x.clip(-(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int32)

for mode in (0, 1, 2, 2 | 8, 2 | 16, 1 | 32, 2 | 32, 2 | 64, 2 | 128, 2 | 192, 1 | 16 | 8):
    for ch in (1, 2, 3):
        if ch == 3 and mode >= 64: continue
        for bps, order in ((8, 3), (16, 8), (24, 32), (16, 12), (20, 5)):
            x = sig(20000 + 77, ch, bps)
            if mode & 8: x[:512] &= ~7
            f = orc.flac_encode(x, 44100, bps, 1152 if mode & 32 else 4096, mode, order)
            try:
                i32, f32, m = e.op_decode_audio(f)
                ok = np.array_equal(i32, x) and np.array_equal(f32, (x.astype(np.float64) / (1 << (bps - 1))).astype(np.float32))
            except Exception as ex:
                ok = False; print("EXC", ex)
            if not ok:
                allok = False
                print("FAIL mode", mode, "ch", ch, "bps", bps, "order", order)
print("coverage streams", "OK" if allok else "FAILED")
# GPU encoder -> GPU decoder, 10 minutes
pcm = (np.tile(sig(44100 * 10, 1, 16)[:, 0], 60)).astype(np.int16)
f = e.op_flac_encode(pcm, 44100)
t0 = time.perf_counter(); i32, f32, m = e.op_decode_audio(f); dt = time.perf_counter() - t0
print("10 min roundtrip", np.array_equal(i32[:, 0], pcm.astype(np.int32)), m, f"{dt*1e3:.1f} ms wall (2 decodes + D2H)")
