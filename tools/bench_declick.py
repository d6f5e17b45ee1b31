"""adeclick launch time inside Pass 4 on the bench workload (HIP events): min / median over N runs.  For A/B builds."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sr = 48000
x = synth.speech_like_torch(3600.0, sr, seed=1000, device="cuda:0", plosives_per_min=40.0)
e = Engine(0)
e.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x)
base = hostlogic.default_config()
ms = []
for i in range(n_runs + 1):
    hostlogic.process_audio(e, base, 4096)
    if i: ms.append(e.timers()["declick_ms"])
ms = np.array(ms)
print(os.environ.get("JT_LIB_PATH", "default lib"), f"declick_ms min {ms.min():.2f} median {np.median(ms):.2f} max {ms.max():.2f}")
