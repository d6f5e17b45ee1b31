"""The dynamic-mode file of the bench in fresh processes: the delivered s16 and its FLAC image hash the same every time."""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, ".")
from jivetalking_amd import Engine, synth, hostlogic
x = synth.speech_like_torch(600.0, 48000, seed=1000, device="cuda:0", plosives_per_min=40.0, sib_gain=4.0)
xs = x.cpu().numpy()
e = Engine(0); e.upload_pcm(xs, 48000, 1)
r = hostlogic.process_audio(e, hostlogic.default_config(), 4096)
pcm = e.download_s16(4)
print("dyn", int(r.loudnorm.normalization_type_dynamic), "p4", hashlib.sha256(pcm.tobytes()).hexdigest()[:16], "flac", hashlib.sha256(e.flac_encode(4)).hexdigest()[:16], r.output_lufs)
