"""One file alone on a ONE-stream handle (every kernel serialised, nothing shares the GPU): python tools/solo_serial.py minutes [steps]
Under rocprofv3 --kernel-trace this gives every kernel's uncontended duration for that file length."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import torch
from jivetalking_amd import Engine, synth, hostlogic
mins = float(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x = synth.speech_like_torch(mins * 60.0, 48000, seed=1000, device="cuda:0", plosives_per_min=40.0); torch.cuda.synchronize()
e = Engine(0, streams=int(os.environ.get("JT_PROBE_STREAMS", "1")))
e.attach_device_pcm(x.data_ptr(), x.numel(), 48000, 1, keepalive=x)
hostlogic.process_audio(e)
t0 = time.perf_counter()
for _ in range(steps): hostlogic.process_audio(e)
print(f"SOLO {mins:g} min: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms per file")
