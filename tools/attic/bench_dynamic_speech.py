"""Pass 4 through loudnorm's dynamic fallback on speech (second-pass values with an LRA above the target).  python tools/bench_dynamic_speech.py [minutes]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from jivetalking_amd import Engine, synth, hostlogic as H, _lib as L
mins = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
SR = 48000
x = synth.speech_like_torch(mins * 60.0, SR, seed=1000, device="cuda:0").cpu().numpy()
e = Engine(0)
e.upload_pcm(x, SR, 1)
res = H.process_audio(e)                    # normal run: leaves the Pass-2 output on the device
m = res.measure
for lra in (res.measure.input_lra, 25.0):
    ap = L.LoudnormApply(-16.0, -1.0, 20.0, m.input_i, m.input_tp, lra, m.input_thresh, res.offset, 1, 1.7, 55.0, 50.0, 1, 0.803526)
    t0 = time.time(); a, st = e.pass4(None, ap); dt = time.time() - t0
    print(f"{mins:g} min, measured_LRA={lra:.2f}: pass 4 {dt * 1e3:.1f} ms, dynamic={st['normalization_type_dynamic']}, output I {st['output_i']:.2f} TP {st['output_tp']:.2f}")
