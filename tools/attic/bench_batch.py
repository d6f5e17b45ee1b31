"""BASELINE configs[3] shape on one GPU: a queue of 10-minute 48 kHz mono FLAC files through jt_process_files with K in flight."""
import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from jivetalking_amd.engine import Engine
from jivetalking_amd import synth, hostlogic

nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 16
minutes = 10.0
sr = 48000
e = Engine()
paths = []
for k in range(nfiles):
    base = np.asarray(synth.speech_like(60.0, sr, seed=100 + k), np.float64)
    pcm = np.clip(np.rint(np.tile(base, int(minutes))[: int(minutes * 60 * sr)] * 32768), -32768, 32767).astype(np.int16)
    p = f"/dev/shm/jt_batch_{k}.flac"
    open(p, "wb").write(e.op_flac_encode(pcm, sr, md5=False))
    paths.append(p)
e.close()
for k in (1, 2, 3, 4):
    t0 = time.perf_counter()
    failed, res = hostlogic.process_files(paths, in_flight=k, md5=False)
    dt = time.perf_counter() - t0
    print(f"in flight {k}: {nfiles} x {minutes:g} min in {dt*1e3:.0f} ms = {nfiles*minutes*60/dt:.0f} xRT disk to disk, failed {failed}")
for p in paths:
    os.remove(p)
for r in res:
    if r.rc == 0: os.remove(r.output_path.decode())
