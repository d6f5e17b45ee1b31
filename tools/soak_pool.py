"""Soak: the same 32 ten-minute files (mono FLAC, every fourth one a file that takes loudnorm's dynamic mode) through a handle pool,
batch after batch, STREAMINFO MD5 on; every batch's outputs must be the first batch's, byte for byte.  usage: soak_pool.py [batches] [in flight]"""
import os, sys, time, tempfile, shutil, hashlib
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 10
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = tempfile.mkdtemp(prefix="jtsoak", dir="/dev/shm")
e = Engine(0); paths = []
for k in range(32):
    dyn = k % 4 == 3
    x = synth.speech_like_torch(600.0 - 7.0 * k, 48000, seed=7000 + k, device="cuda:0", plosives_per_min=40.0 if k % 2 else 0.0, sib_gain=4.0 if dyn else 1.0)
    pcm = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy()
    pk = os.path.join(d, f"f{k:02d}.flac"); open(pk, "wb").write(e.op_flac_encode(pcm, 48000, md5=True)); paths.append(pk)
e.close()
ref = None; bad = 0
with H.Pool((0,), K) as P:
    for b in range(NB):
        t0 = time.time(); failed, fr, _ = P.process_files(paths, md5=True); w = time.time() - t0
        hs = [hashlib.md5(open(r.output_path.decode(), "rb").read()).hexdigest() if r.rc == 0 else "FAILED" for r in fr]
        ndyn = sum(1 for r in fr if r.rc == 0 and r.result.loudnorm.normalization_type_dynamic)
        for r in fr:
            if r.rc == 0: os.unlink(r.output_path.decode())
        if ref is None: ref = hs
        diff = [i for i in range(32) if hs[i] != ref[i]]
        bad += len(diff) + int(failed)
        print(f"batch {b}: {w * 1e3 / 32:.2f} ms/file, failed {failed}, dynamic {ndyn}, differing from batch 0: {diff}", flush=True)
left = [q for q in os.listdir(d) if q.startswith(".processing-")]
print("residue:", left, " RESULT:", "FAILED" if bad or left else "ok")
shutil.rmtree(d, ignore_errors=True)
