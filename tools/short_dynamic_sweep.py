"""Dynamic-mode loudnorm on 134 stream lengths from 2 to 13.5 s (around the 3 s minimum, the stream path's 49-frame threshold, whole and ragged frames):
stream path against the one-workgroup kernel."""
import sys, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth
e = Engine(0)
base = synth.speech_like(14.0, 192000, seed=9).astype(np.float64) * 3.0
bad = 0; n_cases = 0
for secs in np.arange(2.0, 13.5, 0.173):
    for cut in (0, 7777):
        x = base[: int(secs * 192000) - cut]
        e.set_option("ln_no_stream", True); w, ws = e.op_loudnorm_dynamic(x, target_tp=-9.0)
        e.set_option("ln_no_stream", False); g, gs = e.op_loudnorm_dynamic(x, target_tp=-9.0)
        n_cases += 1
        if not (np.array_equal(w, g) and ws == gs):
            bad += 1; print("DIFFERENT at", secs, cut, e.timers()["ln_stream_frames"])
print(n_cases, "lengths,", bad, "different")
