"""Pass 4 through loudnorm's dynamic fallback on a long file: a steady signal (loudness range prints as 0.00, which af_loudnorm's init()
reads as "not measured").  python tools/bench_dynamic.py [minutes] [level]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from jivetalking_amd import Engine, hostlogic as H
mins = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
level = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
SR = 48000
rng = np.random.default_rng(1)
n = int(mins * 60 * SR)
t = np.arange(n) / SR
x = (level * (np.sin(2 * np.pi * 220.0 * t) + 0.5 * np.sin(2 * np.pi * 447.0 * t + 1.0))).astype(np.float32)
e = Engine(0)
e.upload_pcm(x, SR, 1)
for _ in range(2):
    t0 = time.time(); res = H.process_audio(e); dt = time.time() - t0
    print(f"{mins:g} min: {dt * 1e3:.1f} ms per file ({mins * 60 / dt:.0f} xRT), pass4 {res.pass_ms[3]:.1f} ms, dynamic={res.loudnorm.normalization_type_dynamic}, "
          f"input LRA {res.measure.input_lra:.2f}, output {res.output_lufs:.2f} LUFS / {res.output_tp_db:.2f} dBTP")
