"""I/O legs on a 60-minute 44.1 kHz mono s16 stream: GPU FLAC encode, GPU FLAC decode (for rocprofv3 --kernel-trace --stats)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from jivetalking_amd.engine import Engine
from jivetalking_amd import synth

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
reps = 3
sr = 44100
base = np.asarray(synth.speech_like(60.0, sr, seed=3, speech_dbfs=-20.0), np.float64)
pcm = np.clip(np.tile(base, int(np.ceil(minutes)))[: int(minutes * 60 * sr)] * 32768, -32768, 32767).astype(np.int16)
e = Engine()
for _ in range(reps):
    img, info = e.op_flac_encode(pcm, sr, md5=False, return_info=True)
print("encode", {k: info[k] for k in ("gpu_ms", "total_ms", "bytes", "frames")})
for _ in range(reps):
    m = e.load_audio(img)
print("decode", {k: m[k] for k in ("gpu_ms", "total_ms", "flac_frames", "flac_candidates")})
