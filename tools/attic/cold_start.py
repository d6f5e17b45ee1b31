"""What a one-shot CLI run pays: context creation, the first file (allocations, pinned staging, kernel loads) and a second file on the
warm handle.  python tools/cold_start.py [minutes]"""
import os, sys, time
t00 = time.perf_counter()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from jivetalking_amd import Engine, hostlogic
mins = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
sr = 48000
t0 = time.perf_counter()
eng = Engine(0)
t1 = time.perf_counter()
r = np.random.default_rng(5); n = int(mins * 60 * sr)
x = (0.1 * r.standard_normal(n) * (0.15 + 0.85 * (np.sin(np.arange(n) * 3e-5) > 0))).astype(np.float32)
pcm = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)
d = "/dev/shm/jtcold"; os.makedirs(d, exist_ok=True)
src = os.path.join(d, "a.flac"); open(src, "wb").write(eng.op_flac_encode(pcm, sr, md5=True))
del eng
base = hostlogic.default_config()
t2 = time.perf_counter()
eng = Engine(0)
t3 = time.perf_counter()
_, outp, io1 = hostlogic.process_file(eng, src, base, 4096, md5=False)
t4 = time.perf_counter()
_, outp, io2 = hostlogic.process_file(eng, src, base, 4096, md5=False)
t5 = time.perf_counter()
print(f"import {t0 - t00:.2f} s; first context {t1 - t0:.3f} s; fresh context {t3 - t2:.3f} s; first file {t4 - t3:.3f} s (io {[round(v, 1) for v in io1]}); second file {t5 - t4:.3f} s (io {[round(v, 1) for v in io2]})")
import shutil; shutil.rmtree(d, ignore_errors=True)
