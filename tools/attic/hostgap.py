import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic
sr = 48000
x = synth.speech_like_torch(3600.0, sr, seed=1000, device="cuda:0"); torch.cuda.synchronize()
eng = Engine(0); eng.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x)
base = hostlogic.default_config()
for _ in range(3): r = hostlogic.process_audio(eng, base, 4096)
acc = np.zeros(10); pm = np.zeros(4); tot = 0
for _ in range(6):
    t0 = time.perf_counter(); r = hostlogic.process_audio(eng, base, 4096); tot += time.perf_counter() - t0
    acc += np.array(list(r.stage_ms)[:10]); pm += np.array(list(r.pass_ms)[:4])
names = ["pass1", "vad", "bands", "adapt", "pass2", "regions2", "plan", "pass3", "pass4", "regions4"]
print("step wall %.2f ms; stage wall:" % (tot / 6 * 1e3), " ".join("%s=%.2f" % (n, v / 6) for n, v in zip(names, acc)), "| sum %.2f" % (acc.sum() / 6))
print("GPU pass ms (events):", " ".join("%.2f" % (v / 6) for v in pm))
