"""K contexts on one GPU, each running jt_process_audio on its own copy of a file: per-stage wall times of every worker (which stage
stretches when contexts share the GPU).  python tools/inflight_probe.py [K] [minutes] [steps]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mins = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sr = 48000
x = synth.speech_like_torch(mins * 60.0, sr, seed=1000, device="cuda:0"); torch.cuda.synchronize()
base = hostlogic.default_config()
if os.environ.get("JT_PROBE_LIKE_BENCH"):
    # the order bench.py --in-flight uses: one context works alone first, the others are created afterwards
    e0 = Engine(0); e0.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x)
    for _ in range(10): hostlogic.process_audio(e0, base, 4096)
    torch.cuda.synchronize()
    engs = [e0] + [Engine(0) for _ in range(K - 1)]
    for e in engs[1:]:
        e.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x); hostlogic.process_audio(e, base, 4096)
    torch.cuda.synchronize()
else:
    S = int(os.environ.get("JT_PROBE_STREAMS", "0"))
    engs = [Engine(0, streams=S, blocking_sync=bool(S)) for _ in range(K)]
    for e in engs:
        e.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x); hostlogic.process_audio(e, base, 4096)
names = ["pass1", "vad", "bands", "adapt", "pass2", "regions2", "plan", "pass3", "pass4", "regions4"]
acc = [np.zeros(10) for _ in engs]; wall = [0.0] * K
def worker(i):
    t0 = time.perf_counter()
    for _ in range(steps):
        r = hostlogic.process_audio(engs[i], base, 4096)
        acc[i] += np.array(list(r.stage_ms)[:10])
    wall[i] = (time.perf_counter() - t0) / steps * 1e3
t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(i,)) for i in range(K)]
for t in th: t.start()
for t in th: t.join()
tot = time.perf_counter() - t0
print(f"K={K}: {tot / (K * steps) * 1e3:.2f} ms per file; per worker step wall {[round(w, 1) for w in wall]}")
for i in range(K): print("  worker", i, " ".join(f"{n}={v / steps:.1f}" for n, v in zip(names, acc[i])))
