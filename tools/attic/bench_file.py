"""File in -> file out (jt_process_file) on a 60-minute 48 kHz mono 16-bit FLAC in /dev/shm: the reference's ProcessAudio(inputPath)."""
import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from jivetalking_amd.engine import Engine
from jivetalking_amd import synth, hostlogic

minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
sr = 48000
base = np.asarray(synth.speech_like(60.0, sr, seed=3), np.float64)
pcm = np.clip(np.rint(np.tile(base, int(np.ceil(minutes)))[: int(minutes * 60 * sr)] * 32768), -32768, 32767).astype(np.int16)
e = Engine()
src = "/dev/shm/jt_bench_in.flac"
open(src, "wb").write(e.op_flac_encode(pcm, sr, md5=True))
print("input", os.path.getsize(src), "bytes")
for md5 in (False, True):
    for rep in range(3):
        t0 = time.perf_counter()
        res, out, io = hostlogic.process_file(e, src, md5=md5)
        dt = (time.perf_counter() - t0) * 1e3
    print(f"md5={md5}: wall {dt:.1f} ms = {minutes*60/dt*1e3:.0f} xRT; read {io[0]:.1f} decode {io[1]:.1f} passes {sum(res.stage_ms):.1f} "
          f"encode {io[2]:.1f} write {io[3]:.1f} ms; out {os.path.getsize(out)} bytes {os.path.basename(out)} LUFS {res.output_lufs:.2f}")
os.remove(src); os.remove(out)
