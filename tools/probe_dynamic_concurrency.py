"""Do the dynamic-loudnorm kernels of several handles run side by side?  N threads, one Engine each, the same 60 s stream."""
import sys, time, threading, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine
rng = np.random.default_rng(3)
n = 60 * 192000
x = (0.05 * rng.standard_normal(n)).astype(np.float64)
def run(e, out, i):
    t0 = time.time(); e.op_loudnorm_dynamic(x, offset=13.0); out[i] = time.time() - t0
for N in (1, 2, 4, 8):
    es = [Engine(0) for _ in range(N)]
    for e in es: e.op_loudnorm_dynamic(x[:192000 * 5])          # warm-up (allocations)
    out = [0.0] * N
    th = [threading.Thread(target=run, args=(es[i], out, i)) for i in range(N)]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    print(f"{N} handles: wall {time.time() - t0:.2f} s, per-thread {min(out):.2f} .. {max(out):.2f} s")
    for e in es: e.close()
