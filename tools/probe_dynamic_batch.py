"""A batch of files that take the dynamic-loudnorm fallback through jt_process_files_multi: do the workers overlap?
usage: probe_dynamic_batch.py [files] [minutes]"""
import os, sys, time, tempfile, shutil
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
files = int(sys.argv[1]) if len(sys.argv) > 1 else 4
minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
d = tempfile.mkdtemp(prefix="jtdyn", dir="/dev/shm")
e = Engine(0)
paths = []
for k in range(files):
    x = synth.speech_like_torch(minutes * 60.0, 48000, seed=3000 + k, device="cuda:0", plosives_per_min=40.0, sib_gain=4.0)
    pcm = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy()
    pk = os.path.join(d, f"dyn{k:02d}.flac"); open(pk, "wb").write(e.op_flac_encode(pcm, 48000, md5=True)); paths.append(pk)
e.close()
for nfl in (1, files):
    t0 = time.time(); failed, fr, _ = H.process_files_multi(paths[:nfl], devices=(0,), in_flight_per_device=nfl, md5=False); w = time.time() - t0
    print(f"{nfl} file(s) in flight: wall {w:.2f} s, failed {failed}, per-file wall {[round(fr[i].wall_ms / 1e3, 2) for i in range(nfl)]}, dynamic {[int(fr[i].result.loudnorm.normalization_type_dynamic) for i in range(nfl)]}")
    for q in os.listdir(d):
        if q.endswith("-processed.flac"): os.unlink(os.path.join(d, q))
shutil.rmtree(d, ignore_errors=True)
