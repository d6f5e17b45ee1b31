"""What the progress callbacks cost a step: jt_process_audio, jt_process_audio_cb (pass starts / ends) and jt_process_audio_ticks (the reference's
every-100-frames and band ticks, which need the per-frame levels of each stage output).  python tools/time_ticks.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic
sr = 48000
x = synth.speech_like_torch(3600.0, sr, seed=1000, device="cuda:0"); torch.cuda.synchronize()
eng = Engine(0); eng.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x)
base = hostlogic.default_config()
cnt = [0]
def cb(u): cnt[0] += 1
for mode in ("plain", "cb", "ticks"):
    ts = []
    for _ in range(6):
        cnt[0] = 0
        t0 = time.perf_counter()
        if mode == "plain": hostlogic.process_audio(eng, base, 4096)
        else: hostlogic.process_audio_with_progress(eng, cb, base, 4096, ticks=(mode == "ticks"))
        ts.append((time.perf_counter() - t0) * 1e3)
    print(mode, "ms per step %.2f (min %.2f), callbacks %d" % (np.mean(ts[2:]), np.min(ts[2:]), cnt[0]))
