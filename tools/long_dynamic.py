"""The dynamic mode's stream path on long streams: a 192 kHz stream of `hours` (default 2.9: 2.0e9 samples; 3.5 h = 2.4e9 is past 2^31 and
past one attempt's 100 000 frames) against the one-workgroup kernel.  Needs ~35 GB of host memory per stream.  usage: long_dynamic.py [hours]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 2.9
unit = synth.speech_like(60.0, 192000, seed=77).astype(np.float64) * 2.5
n = int(hours * 3600 * 192000)
x = np.tile(unit, n // unit.size + 1)[:n]
x[:: 192000 * 97] *= 1.7                                   # (not exactly periodic)
print(f"{n} samples ({n / 2**31:.3f} of 2^31), {x.nbytes / 2**30:.1f} GiB", flush=True)
e = Engine(0)
import os
for kv in filter(None, os.environ.get('JT_OPTS', '').split(',')):
    e.set_option(*kv.split('=', 1))
out = {}
for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("stream", "wg")):
    e.set_option("ln_no_stream", mode == "wg")
    t0 = time.time(); y, st = e.op_loudnorm_dynamic(x, target_tp=-9.0); dt = time.time() - t0
    t = e.timers()
    print(f"{mode}: {dt:.1f} s incl. transfers, stream frames {t['ln_stream_frames']} of {n // 19200}, reasons {t['ln_stream_why']}, dynamic {st['normalization_type_dynamic']}, out peak {np.max(np.abs(y[-10**7:])):.6f}", flush=True)
    out[mode] = (y, st)
if len(out) == 2:
    same = np.array_equal(out["stream"][0], out["wg"][0]) and out["stream"][1] == out["wg"][1]
    print("identical" if same else "DIFFERENT")
