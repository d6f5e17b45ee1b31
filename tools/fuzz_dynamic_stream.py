"""Dynamic-mode loudnorm, stream path against the one-workgroup kernel on random streams: level, ceiling, offset, quiet stretches, spikes,
ragged lengths, measured values that open the above_threshold phase.  usage: fuzz_dynamic_stream.py [cases] [seed]"""
import sys, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
e = Engine(0)
base = [synth.speech_like(36.0, 192000, seed=s).astype(np.float64) for s in (101, 102, 103)]
bad = 0; reasons = {}
for c in range(cases):
    x = base[c % 3][: int(192000 * rng.uniform(12.0, 36.0)) - int(rng.integers(0, 19200))].copy() * float(10 ** rng.uniform(-0.5, 1.0))
    kind = int(rng.integers(0, 5))
    if kind == 1:                                                  # quiet stretches
        for _ in range(int(rng.integers(1, 4))):
            a = int(rng.integers(0, x.size - 192000)); x[a: a + int(rng.integers(19200, 4 * 192000))] *= 10 ** rng.uniform(-4, -1)
    elif kind == 2:                                                # isolated spikes (long gaps between peaks: releases, the ring-end corner's neighbourhood)
        x *= 0.01
        for t in rng.integers(600000, x.size - 50000, size=int(rng.integers(3, 40))):
            x[t] = rng.uniform(0.3, 0.9) * rng.choice([-1, 1])
            if rng.random() < 0.5: x[t + int(rng.integers(19150, 19250))] = rng.uniform(0.3, 0.9)
    elif kind == 3:                                                # clipped: plateaus of equal samples
        x = np.clip(x, -0.5, 0.5)
    tp = float(rng.uniform(-20.0, -0.5)); off = float(rng.choice([0.0, rng.uniform(-6, 12)]))
    meas = None if rng.random() < 0.6 else (float(rng.uniform(-30, -14)), 7.0, -2.0, float(rng.uniform(-45, -25)))
    e.set_option("ln_no_stream", True); want, wst = e.op_loudnorm_dynamic(x, target_tp=tp, offset=off, measured=meas)
    e.set_option("ln_no_stream", False); got, gst = e.op_loudnorm_dynamic(x, target_tp=tp, offset=off, measured=meas)
    t = e.timers(); reasons[int(t["ln_stream_why"])] = reasons.get(int(t["ln_stream_why"]), 0) + 1
    ok = np.array_equal(got, want) and gst == wst
    if not ok:
        bad += 1
        d = np.abs(got - want); i = int(np.argmax(d > 0))
        print(f"case {c} kind {kind} tp {tp:.2f} off {off:.2f} meas {meas} n {x.size}: DIFFERENT {int(np.count_nonzero(d))} samples, first at {i} (frame {i // 19200}), max {d.max():g}; frames {t['ln_stream_frames']} why {t['ln_stream_why']}", flush=True)
print(f"{cases} cases, {bad} different; reason masks seen: {reasons}")
