"""Run one operator on synthetic speech (for rocprofv3 counter passes): python tools/run_op.py anlmdn|afftdn|... [seconds]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth
op = sys.argv[1]; secs = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
SR = 48000
x = synth.speech_like_torch(secs, SR, seed=1000, device="cuda:0").cpu().numpy()
e = Engine(0)
for _ in range(2):
    if op == "anlmdn": e.op_anlmdn(x, SR)
    elif op == "afftdn": e.op_afftdn(x, SR, 12.0, -55.0)
    elif op == "resample": e.op_resample_s16(x, SR, 44100)
    elif op == "loudnorm": e.op_loudnorm_measure_s16((x[: int(x.size * 0.91875)] * 32767).astype(np.int16), 44100)
print("done", op, x.size)
