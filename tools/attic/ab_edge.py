"""48 k -> 44.1 k output stage: the one-wave-per-phase edge kernel against k_polyphase on the edge blocks (JT_EDGE_POLYPHASE), lengths around
the block size.  python tools/ab_edge.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from jivetalking_amd import Engine
e = Engine(0)
bad = 0
r = np.random.default_rng(1)
for n in (1, 5, 159, 160, 161, 10239, 10240, 10241, 20479, 20481, 30720, 48000, 99999, 480000, 1234567):
    x = (0.5 * r.standard_normal(n)).astype(np.float32)
    os.environ["JT_EDGE_POLYPHASE"] = "1"; a = e.op_resample_s16(x, 48000, 44100); os.environ.pop("JT_EDGE_POLYPHASE")
    b = e.op_resample_s16(x, 48000, 44100)
    if not np.array_equal(a, b): bad += 1; print("DIFF n", n, a.size, b.size, np.max(np.abs(a.astype(int) - b.astype(int))))
print("edge kernel A/B:", "identical" if bad == 0 else bad)
