import sys, hashlib, numpy as np, torch
sys.path.insert(0, "/root/repo")
from jivetalking_amd import Engine, synth, hostlogic
x = synth.speech_like_torch(600.0, 48000, seed=1000, device="cuda:0", plosives_per_min=40.0)
xs = x.cpu().numpy()
print("talker", hashlib.sha256(xs.tobytes()).hexdigest()[:16])
e = Engine(0)
e.upload_pcm(xs, 48000, 1)
for i in range(2):
    r = hostlogic.process_audio(e, hostlogic.default_config(), 4096)
    print("p2", hashlib.sha256(e.download_s16(2).tobytes()).hexdigest()[:16], "p4", hashlib.sha256(e.download_s16(4).tobytes()).hexdigest()[:16], e.timers()["declick_repaired"], r.output_lufs)
