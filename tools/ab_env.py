"""A/B of an environment switch inside ONE process (alternating runs, so clocks and box are shared): declick / step times.
usage: ab_env.py JT_DK_NO_XCD [runs] [plosives per minute: 40 = the bench talker (limiter prefix branch)]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic
var = sys.argv[1]; n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sr = 48000
pl = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
x = synth.speech_like_torch(3600.0, sr, seed=1000, device="cuda:0", plosives_per_min=pl)
e = Engine(0)
e.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x)
base = hostlogic.default_config()
import time
res = {0: [], 1: []}
for i in range(2 * n_runs + 2):
    on = i & 1
    if on: os.environ[var] = "1"
    else: os.environ.pop(var, None)
    t0 = time.perf_counter(); hostlogic.process_audio(e, base, 4096); dt = (time.perf_counter() - t0) * 1e3
    if i >= 2: res[on].append((e.timers()["declick_ms"], dt))
for on in (0, 1):
    a = np.array(res[on])
    print(f"{var}={'1' if on else 'unset'}: declick_ms median {np.median(a[:,0]):.2f} min {a[:,0].min():.2f}; step median {np.median(a[:,1]):.2f} min {a[:,1].min():.2f}")
