"""True-peak kernels nearly alone (only the K-weighting job beside them): k_tp_stream against k_upsample32<MODE 0, QL 1> (A/B build, option
tp_old) on twenty minutes at 44.1 kHz.  Run under rocprofv3 --kernel-trace --stats (tools/tp_solo.sh prints the two kernels' durations)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from jivetalking_amd import Engine
rng = np.random.default_rng(1)
x = (0.1 * rng.standard_normal(44100 * 1200)).astype(np.float32)
e = Engine(0, ab=True)
for old in (False, True, False, True):
    e.set_option("tp_old", old)
    r = e.op_ebur128(x, 44100)
print("true peak", r["true_peak"])
