"""The whole pipeline at sizes beyond the bench's: a 3-hour file (518 M samples) through jt_process_audio, and a 35-minute file that takes
loudnorm's dynamic mode (the stream path inside Pass 4, then the aresample back); landings, and the delivered s16 against a second run.
usage: long_pipeline.py [hours] [dynamic minutes]"""
import os, sys, time, hashlib
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
dmin = float(sys.argv[2]) if len(sys.argv) > 2 else 35.0
e = Engine(0)
for name, secs, kw in (("linear", hours * 3600.0, dict(plosives_per_min=40.0)), ("dynamic", dmin * 60.0, dict(plosives_per_min=40.0, sib_gain=4.0))):
    x = synth.speech_like_torch(secs, 48000, seed=1000, device="cuda:0", **kw); torch.cuda.synchronize()
    e.attach_device_pcm(x.data_ptr(), x.numel(), 48000, 1, keepalive=x)
    hs = []
    for it in range(2):
        t0 = time.perf_counter(); r = H.process_audio(e); dt = time.perf_counter() - t0
        hs.append(hashlib.md5(e.download_s16(4).tobytes()).hexdigest())
    print(f"{name}: {secs / 60:.0f} min in {dt * 1e3:.1f} ms = {secs / dt:.0f} xRT; dynamic {int(r.loudnorm.normalization_type_dynamic)}, prefix {int(r.limiter.needed)}, lands {r.output_lufs:.2f} LUFS / {r.output_tp_db:.2f} dBTP, "
          f"stream frames {e.timers()['ln_stream_frames']}; two runs {'identical' if hs[0] == hs[1] else 'DIFFERENT'}", flush=True)
    del x
