"""24 ten-minute files, K in flight, no MD5, one repetition (for a kernel trace).  usage: sat_trace.py [K]"""
import os, sys, time, tempfile, shutil
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
d = tempfile.mkdtemp(prefix="jtst", dir="/dev/shm")
e = Engine(0); paths = []
for k in range(24):
    x = synth.speech_like_torch(600.0, 48000, seed=5000 + k, device="cuda:0", plosives_per_min=40.0 if k % 2 else 0.0)
    pcm = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy()
    pk = os.path.join(d, f"f{k:02d}.flac"); open(pk, "wb").write(e.op_flac_encode(pcm, 48000, md5=False)); paths.append(pk)
e.close()
torch.cuda.synchronize(); time.sleep(0.2)
t0 = time.time(); failed, fr, _ = H.process_files_multi(paths, devices=(0,), in_flight_per_device=K, md5=False); w = time.time() - t0
print(f"BATCH {w:.4f} s, {w / 24 * 1e3:.1f} ms per file, failed {failed}")
shutil.rmtree(d, ignore_errors=True)
