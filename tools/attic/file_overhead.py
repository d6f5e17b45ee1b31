"""Where does a ten-minute file's wall time go on one handle (steady state)?  wall, the four passes' GPU time, the I/O legs."""
import os, sys, time, tempfile, shutil
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import numpy as np, torch
from jivetalking_amd import Engine, synth, hostlogic as H
d = tempfile.mkdtemp(prefix="jtfo", dir="/dev/shm")
e = Engine(0); paths = []
for k in range(6):
    x = synth.speech_like_torch(600.0, 48000, seed=6000 + k, device="cuda:0", plosives_per_min=40.0)
    pcm = (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16).cpu().numpy()
    pk = os.path.join(d, f"f{k:02d}.flac"); open(pk, "wb").write(e.op_flac_encode(pcm, 48000, md5=False)); paths.append(pk)
for md5 in (False,):
    for pk in paths:
        t0 = time.time(); res, outp, io = H.process_file(e, pk, md5=md5); w = (time.time() - t0) * 1e3
        t = e.timers()
        gp = t["pass1_ms"] + t["pass2_ms"] + t["pass3_ms"] + t["pass4_ms"]
        print(f"wall {w:.1f} ms; passes (GPU events) {gp:.1f}; io read {io[0]:.1f} decode {io[1]:.1f} encode {io[2]:.1f} write {io[3]:.1f}; stage_wall sum {sum(res.stage_ms):.1f}")
        os.unlink(outp)
shutil.rmtree(d, ignore_errors=True)
