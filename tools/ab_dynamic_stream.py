"""Dynamic-mode loudnorm: the stream path (data-parallel sweeps around a list-walking state machine) against the one-workgroup kernel
(option ln_no_stream) and the per-peak walk (ln_no_batch): identical output, frames covered, time.
usage: ab_dynamic_stream.py [seconds]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
e = Engine(0)
cases = [(41, 2.5, -9.0, 0.0, 0.0, 0), (7, 2.5, -20.0, 0.0, 0.0, 333), (8, 4.0, -6.0, 3.0, 1.5, 19199), (9, 1.2, -12.0, 0.0, 0.0, 777),
         (10, 8.0, -1.0, 13.0, 2.9, 1), (11, 3.0, -15.0, 6.0, 9.0, 9600), (12, 6.0, -3.0, 0.0, 0.0, 0)]
bad = 0
for seed, level, tp, off, quiet_s, cut in cases:
    x = synth.speech_like(secs, 192000, seed=seed).astype(np.float64) * level
    if quiet_s > 0:
        x = np.concatenate([x[: int(192000 * quiet_s)] * 0.004, x])
    if cut:
        x = x[: x.size - cut]
    out = {}
    for mode in ("stream", "wg", "stop37", "stream"):
        e.set_option("ln_no_stream", mode == "wg"); e.set_option("ln_stream_stop", 37 if mode == "stop37" else 0)
        t0 = time.time(); y, st = e.op_loudnorm_dynamic(x, target_tp=tp, offset=off); dt = time.time() - t0
        tm = e.timers()
        out[mode] = (y, st)
        print(f"seed {seed} level {level} tp {tp} off {off} quiet {quiet_s}: {mode:6s} {dt:.3f} s, stream frames {tm['ln_stream_frames']} of {int(x.size / 19200)}, why {tm['ln_stream_why']}, dynamic {st['normalization_type_dynamic']}", flush=True)
    for other in ("stream", "stop37"):
        a, b = out[other][0], out["wg"][0]
        same = np.array_equal(a, b) and out[other][1] == out["wg"][1]
        if not same:
            bad += 1
            d = np.abs(a - b); i = int(np.argmax(d > 0))
            print(f"   {other} DIFFERENT: {int(np.count_nonzero(d))} samples, first at {i} (frame {i // 19200}), max {d.max():g}")
        else:
            print(f"   {other} identical")
print("FAILED" if bad else "all identical")
