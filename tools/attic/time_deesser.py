"""The gate -> compressor -> de-esser chain with the de-esser on (AdaptConfig enables it on sibilant voices; the bench voice leaves it off):
run under rocprofv3 --kernel-trace --stats.  python tools/time_deesser.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from jivetalking_amd import Engine, synth
from jivetalking_amd.engine import default_filter_params
e = Engine(0)
x = synth.speech_like_torch(3600.0, 48000, seed=1000, device="cuda:0").cpu().numpy()
p = default_filter_params()
p.deess_enabled, p.deess_i = 1, 0.4
for _ in range(3): e.op_dynamics(x, 48000, p)
