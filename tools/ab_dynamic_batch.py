"""Dynamic-mode loudnorm: the batched SUSTAIN step against the per-peak walk (option ln_no_batch): identical output, time.
usage: ab_dynamic_batch.py [seconds] [level]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
level = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
x = synth.speech_like(secs, 192000, seed=41).astype(np.float64) * level
e = Engine(0)
res = {}
for mode in ("batch", "per-peak", "batch"):
    e.set_option("ln_no_batch", mode == "per-peak")
    for tp, off in ((-9.0, 0.0), (-1.0, 13.0), (-20.0, 0.0)):
        e.op_loudnorm_dynamic(x[: 192000 * 4], target_tp=tp, offset=off)
        t0 = time.time(); y, st = e.op_loudnorm_dynamic(x, target_tp=tp, offset=off); dt = time.time() - t0
        res[(mode, tp)] = y
        print(f"{mode:8s} tp {tp:5.1f} offset {off:4.1f}: {dt:.3f} s for {secs:g} s ({secs / dt:.0f} xRT incl. transfers), out peak {np.max(np.abs(y)):.6f}, dynamic {st['normalization_type_dynamic']}", flush=True)
for tp in (-9.0, -1.0, -20.0):
    a, b = res[("batch", tp)], res[("per-peak", tp)]
    print("identical" if np.array_equal(a, b) else f"DIFFERENT: max {np.max(np.abs(a - b)):g} at {int(np.argmax(np.abs(a - b)))}", "at tp", tp)
