#!/bin/bash
# Phase clocks of k_afftdn_grp as wave $AF_WAVE sees them (-DJT_AF_PROFILE=<wave> build, run on the GPU box): cycles per workgroup in each phase, modes 0/2/1
cd "$GRAFT_REPO_ROOT"
touch jivetalking_amd/csrc/k_fft.hip
make -s -C jivetalking_amd/csrc EXTRA="-DJT_AF_PROFILE=${AF_WAVE:-0}" 2>&1 | grep -E " error"
python - <<PY 2>&1 | grep -E "clocks|ms"
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from jivetalking_amd import Engine, synth
e = Engine(0)
x = synth.speech_like_torch(3600.0, 48000, seed=1000, device="cuda:0").cpu().numpy()
e.op_afftdn(x, 48000, 12.0, -50.0)
e.op_afftdn(x, 48000, 12.0, -50.0, track=True)
PY
touch jivetalking_amd/csrc/k_fft.hip
make -s -C jivetalking_amd/csrc 2>&1 | grep -E " error"
