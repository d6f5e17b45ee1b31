"""The step when the limiter prefix is needed (a file whose peaks would pass the ceiling after the loudnorm gain -- the usual case for real
speech, not for the bench's synthetic talker): plosive-like bursts are added to the bench signal.  Prints the pass times and the plan.
python tools/bench_limiter_prefix.py [minutes]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic
mins = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
sr = 48000
x = synth.speech_like_torch(mins * 60.0, sr, seed=1000, device="cuda:0")
if os.environ.get("JT_NUMPY_INPUT"):      # bit-repeatable input for digests (the torch generator is not)
    x = torch.from_numpy(synth.speech_like(mins * 60.0, sr, seed=1000).astype(np.float32)).to("cuda:0")
n = x.numel()
g = torch.Generator(device="cuda:0").manual_seed(7)
pos = torch.randint(sr, n - sr, (int(mins * float(os.environ.get("JT_BURSTS_PER_MIN", "40"))),), device="cuda:0", generator=g)    # bursts a minute (default 40)
env = torch.hann_window(960, device="cuda:0")
t = torch.arange(960, device="cuda:0")
burst = (0.35 * env * torch.sin(2 * np.pi * 180.0 * t / sr)).float()
for k in range(0, pos.numel()):
    p = int(pos[k]); x[p:p + 960] += burst
torch.cuda.synchronize()
eng = Engine(0); eng.attach_device_pcm(x.data_ptr(), n, sr, 1, keepalive=x)
base = hostlogic.default_config()
for _ in range(2): r = hostlogic.process_audio(eng, base, 4096)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); r = hostlogic.process_audio(eng, base, 4096); ts.append((time.perf_counter() - t0) * 1e3)
tm = eng.timers()
print("ms per step %.2f (min %.2f)  pass ms %.2f %.2f %.2f %.2f" % (np.mean(ts), np.min(ts), tm["pass1_ms"], tm["pass2_ms"], tm["pass3_ms"], tm["pass4_ms"]))
print("limiter needed %d ceiling %.3f  output %.2f LUFS %.2f dBTP  mode %s" % (r.limiter.needed if hasattr(r.limiter, "needed") else -1, getattr(r.limiter, "ceiling_db", float("nan")), r.output_lufs, r.output_tp_db, "linear" if r.linear_possible else "dynamic"))
import hashlib
y = np.empty(int(n * 147 // 160) + 64, np.int16); got = eng.download_s16_into(4, y)
print("output digest", hashlib.sha256(y[:got].tobytes()).hexdigest()[:16], "samples", got)
