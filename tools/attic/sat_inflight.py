"""File-to-file throughput of ten-minute files on one GPU against the number of workers in flight, with and without the STREAMINFO MD5."""
import os, sys, time, tempfile, shutil
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
d = tempfile.mkdtemp(prefix="jtsi", dir="/dev/shm")
e = Engine(0); paths = []
nf = 24
for k in range(nf):
    x = synth.speech_like_torch(600.0, 48000, seed=5000 + k, device="cuda:0", plosives_per_min=40.0 if k % 2 else 0.0)
    pcm = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy()
    pk = os.path.join(d, f"f{k:02d}.flac"); open(pk, "wb").write(e.op_flac_encode(pcm, 48000, md5=False)); paths.append(pk)
e.close()
for md5 in (False, True):
    for k in (1, 2, 3, 4, 6, 8):
        best = 1e9
        for rep in range(2):
            t0 = time.time(); failed, fr, _ = H.process_files_multi(paths, devices=(0,), in_flight_per_device=k, md5=md5); best = min(best, time.time() - t0)
            for q in os.listdir(d):
                if q.endswith("-processed.flac"): os.unlink(os.path.join(d, q))
        print(f"md5={md5} in flight {k}: {best / nf * 1e3:.1f} ms per file ({nf * 600 / best:.0f} xRT), failed {failed}")
shutil.rmtree(d, ignore_errors=True)
