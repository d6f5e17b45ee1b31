"""The workgroup dynamic-loudnorm kernel against the one-wave kernel (JT_DYN_ONE_WAVE=1): identical output, time.
usage: ab_dynamic_wg.py [seconds] [level]"""
import os, sys, time, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
level = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
x = synth.speech_like(secs, 192000, seed=41).astype(np.float64) * level
e = Engine(0)
res = {}
for mode in ("wg", "one"):
    if mode == "one": os.environ["JT_DYN_ONE_WAVE"] = "1"
    else: os.environ.pop("JT_DYN_ONE_WAVE", None)
    for tp, off in ((-9.0, 0.0), (-1.0, 13.0)):
        e.op_loudnorm_dynamic(x[: 192000 * 4], target_tp=tp, offset=off)
        t0 = time.time(); y, st = e.op_loudnorm_dynamic(x, target_tp=tp, offset=off); dt = time.time() - t0
        res[(mode, tp)] = y
        print(f"{mode:3s} tp {tp:5.1f} offset {off:4.1f}: {dt:.3f} s for {secs:g} s ({secs / dt:.0f} xRT incl. transfers), out peak {np.max(np.abs(y)):.6f}, dynamic {st['normalization_type_dynamic']}")
for tp in (-9.0, -1.0):
    print("identical" if np.array_equal(res[("wg", tp)], res[("one", tp)]) else f"DIFFERENT: max {np.max(np.abs(res[('wg', tp)] - res[('one', tp)])):g}", "at tp", tp)
