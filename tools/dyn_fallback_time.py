"""The bench's dynamic_fallback file (10 min of the talker with hiss bursts 24 dB up: loudnorm's linear mode is not possible) through
jt_process_audio, with the limiter's batched SUSTAIN step and with the per-peak walk (option ln_no_batch): time, landing, identical bytes."""
import os, sys, time, hashlib
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic
sr = 48000
yd = synth.speech_like_torch(600.0, sr, seed=1000, device="cuda:0", plosives_per_min=40.0, sib_gain=4.0); torch.cuda.synchronize()
e = Engine(0)
e.attach_device_pcm(yd.data_ptr(), yd.numel(), sr, 1, keepalive=yd)
out = {}
for mode in os.environ.get("JT_DYN_MODES", "stream,batch,per-peak,stream").split(","):
    e.set_option("ln_no_batch", mode == "per-peak"); e.set_option("ln_no_stream", mode == "batch")
    ts = []
    for it in range(3):
        t0 = time.perf_counter(); r = hostlogic.process_audio(e); ts.append(time.perf_counter() - t0)
    out[mode] = hashlib.md5(e.download_s16(4).tobytes()).hexdigest()
    print(f"{mode:8s}: {min(ts) * 1e3:7.1f} ms per ten minutes = {600 / min(ts):6.0f} xRT; dynamic {int(r.loudnorm.normalization_type_dynamic)}, lands {r.output_lufs:.2f} LUFS / {r.output_tp_db:.2f} dBTP, pass4 {e.timers()['pass4_ms']:.1f} ms, stream frames {e.timers()['ln_stream_frames']} (why {e.timers()['ln_stream_why']})", flush=True)
print("delivered s16:", "identical" if len(set(out.values())) == 1 else "DIFFERENT")
e.set_option("ln_no_batch", False); e.set_option("ln_no_stream", False); e.set_option("swr_untiled", True)
hostlogic.process_audio(e)
print("untiled aresample:", "identical" if hashlib.md5(e.download_s16(4).tobytes()).hexdigest() == out["stream"] else "DIFFERENT", f"pass4 {e.timers()['pass4_ms']:.1f} ms")
e.set_option("swr_untiled", False)
if os.environ.get("JT_DYN_DIAG"):
    e.set_option("ln_no_batch", False); e.set_option("ln_no_stream", False); e.set_option("host_timing", True)
    hostlogic.process_audio(e)
