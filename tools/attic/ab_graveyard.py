"""Mixed-length batch through jt_process_files_multi: files of 3..12 minutes in random order, 6 in flight, three repetitions.
Run with JT_GRAVEYARD_GB=0 (every superseded buffer freed at once: a device-wide wait each time, the behaviour before round 3) and without."""
import os, sys, time, tempfile, shutil
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
rng = np.random.default_rng(5)
d = tempfile.mkdtemp(prefix="jtgy", dir="/dev/shm")
e = Engine(0); paths = []; total = 0.0
for k in range(24):
    minutes = float(rng.uniform(3.0, 12.0)); total += minutes * 60.0
    x = synth.speech_like_torch(minutes * 60.0, 48000, seed=4000 + k, device="cuda:0", plosives_per_min=40.0 if k % 2 else 0.0)
    pcm = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy()
    pk = os.path.join(d, f"f{k:02d}.flac"); open(pk, "wb").write(e.op_flac_encode(pcm, 48000, md5=False)); paths.append(pk)
e.close()
ws = []
for rep in range(3):
    t0 = time.time(); failed, fr, _ = H.process_files_multi(paths, devices=(0,), in_flight_per_device=6, md5=False); ws.append(time.time() - t0)
    for q in os.listdir(d):
        if q.endswith("-processed.flac"): os.unlink(os.path.join(d, q))
print(f"JT_GRAVEYARD_GB={os.environ.get('JT_GRAVEYARD_GB', 'default')}: 24 files, {total / 60:.0f} min, 6 in flight: wall " + ", ".join(f"{w:.3f}" for w in ws) + f" s; best {total / min(ws):.0f} xRT, failed {failed}")
shutil.rmtree(d, ignore_errors=True)
