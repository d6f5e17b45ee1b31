"""Host wall clock of jt_process_audio's stages on the bench file (jt_process_result.stage_ms: pass 1, intervals + VAD, bands, adapt, pass 2,
regions(2), plan, pass 3, pass 4, regions(4)) beside the passes' GPU times (HIP events): what the host adds between the passes."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch  # noqa: F401
from jivetalking_amd import Engine, synth, hostlogic
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
sr = 48000
x = synth.speech_like_torch(minutes * 60.0, sr, seed=1000, device="cuda:0", plosives_per_min=40.0)
e = Engine(0)
e.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x)
base = hostlogic.default_config()
rows = []
import time
for i in range(10):
    t0 = time.perf_counter(); r = hostlogic.process_audio(e, base, 4096); dt = (time.perf_counter() - t0) * 1e3
    if i >= 2: rows.append(list(r.stage_ms) + list(r.pass_ms) + [dt])
m = np.median(np.array(rows), axis=0)
names = ["pass1", "intervals+VAD", "bands", "adapt", "pass2", "regions(2)", "plan", "pass3", "pass4", "regions(4)"]
print("host wall ms: " + ", ".join(f"{n} {v:.3f}" for n, v in zip(names, m[:10])) + f"; sum {m[:10].sum():.3f}")
print("GPU event ms: pass 1 / 2 / 3 / 4 " + " / ".join(f"{v:.3f}" for v in m[10:14]) + f"; wall of the call {m[14]:.3f} ms (outside the stages: {m[14] - m[:10].sum():.3f})")
