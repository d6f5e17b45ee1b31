"""A/B of an environment switch inside ONE process (alternating runs, so clocks and box are shared): declick / step times.
usage: ab_env.py JT_DK_NO_XCD|VAR=a,b,- [runs] [plosives per minute: 40 = the bench talker (limiter prefix branch)]"""
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
if "=" in var:      # VAR=a,b,c: cycle through the values ("-" = unset)
    var, vals = var.split("=", 1); vals = vals.split(",")
else:
    vals = ["-", "1"]
res = {v: [] for v in vals}
for i in range(len(vals) * (n_runs + 1)):
    v = vals[i % len(vals)]
    if v == "-": os.environ.pop(var, None)
    else: os.environ[var] = v
    t0 = time.perf_counter(); hostlogic.process_audio(e, base, 4096); dt = (time.perf_counter() - t0) * 1e3
    if i >= len(vals): res[v].append((e.timers()["declick_ms"], dt))
for v in vals:
    a = np.array(res[v])
    print(f"{var}={'unset' if v == '-' else v}: declick_ms median {np.median(a[:,0]):.2f} min {a[:,0].min():.2f}; step median {np.median(a[:,1]):.2f} min {a[:,1].min():.2f}")
