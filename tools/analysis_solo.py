"""Every kernel of a pass's full-length analysis ALONE (one stream, one after another): astats, aspectralstats (selected frames, as in a
pass: through pass1), ebur128, on twenty minutes of the bench talker at 48 kHz.  Run under rocprofv3 --kernel-trace --stats."""
import os, sys, numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
from conftest import bench_talker
from jivetalking_amd import Engine
x = np.asarray(bench_talker(1200.0, 48000, 1000, 40.0), np.float32)
e = Engine(0)
for it in range(3):
    e.op_astats(x, 48000)
    e.op_ebur128(x, 48000)
    e.op_aspectralstats(x, 48000)
print("done")
