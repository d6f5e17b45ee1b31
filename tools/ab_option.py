"""A/B of a jt_set_option key inside ONE process on the bench file (alternating runs, so clocks and box are shared): step / adeclick times.
usage: ab_option.py KEY=a,b,c [runs] [ab: 1 = the A/B build (keys of JT_OPT_AB_*)] [minutes]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic
key, vals = sys.argv[1].split("=", 1); vals = vals.split(",")
n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ab = len(sys.argv) > 3 and sys.argv[3] == "1"
minutes = float(sys.argv[4]) if len(sys.argv) > 4 else 60.0
sr = 48000
x = synth.speech_like_torch(minutes * 60.0, sr, seed=1000, device="cuda:0", plosives_per_min=40.0)
e = Engine(0, ab=ab)
e.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x)
base = hostlogic.default_config()
res = {v: [] for v in vals}
for i in range(len(vals) * (n_runs + 1)):
    v = vals[i % len(vals)]
    e.set_option(key, v)
    t0 = time.perf_counter(); hostlogic.process_audio(e, base, 4096); dt = (time.perf_counter() - t0) * 1e3
    t = e.timers()
    if i >= len(vals): res[v].append((t["declick_ms"], dt, t["pass1_ms"], t["pass2_ms"], t["pass4_ms"]))
for v in vals:
    a = np.array(res[v])
    print(f"{key}={v}: step median {np.median(a[:,1]):.2f} min {a[:,1].min():.2f} ms; pass 1 / 2 / 4 medians {np.median(a[:,2]):.2f} / {np.median(a[:,3]):.2f} / {np.median(a[:,4]):.2f}; adeclick {np.median(a[:,0]):.2f}")
