"""Digest of everything jt_process_audio delivers for a seeded synthetic file (the measurement structs and the final s16): two builds
or two option sets that must be bit-identical print the same line.  usage: out_digest.py [minutes] [rate] [KEY=VALUE ...]"""
import os, sys, hashlib, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401
from jivetalking_amd import Engine, synth, hostlogic
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
sr = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
x = synth.speech_like_torch(minutes * 60.0, sr, seed=1000, device="cuda:0", plosives_per_min=40.0)
e = Engine(0)
for kv in sys.argv[3:]:
    k, _, v = kv.partition("="); e.set_option(k, v)
e.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x)
r = hostlogic.process_audio(e, hostlogic.default_config(), 4096)
h = hashlib.sha256()
for p in (r.input, r.filtered, r.measure, r.final_, r.loudnorm, r.filtered_room_tone, r.filtered_speech, r.final_room_tone, r.final_speech):
    h.update(C.string_at(C.addressof(p), C.sizeof(p)))
h.update(e.download_s16(4).tobytes())
print(f"digest {h.hexdigest()[:32]}  ({minutes:g} min at {sr} Hz; output {r.output_lufs:.3f} LUFS)")
