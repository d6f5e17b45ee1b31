"""Branch-and-bound true peak nearly alone (only the K-weighting job beside it) against the exhaustive kernels (option tp_unpruned) on
twenty minutes of the bench talker at 48 kHz and, resampled by the library, at 44.1 kHz.  Run under rocprofv3 --kernel-trace --stats
(tools/tp_prune_solo.sh); prints the evaluated fraction of the units."""
import os, sys, numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
from conftest import bench_talker
from jivetalking_amd import Engine
x = np.asarray(bench_talker(1200.0, 48000, 1000, 40.0), np.float32)
e = Engine(0)
y = (e.op_resample_s16(x, 48000, 44100).astype(np.float32) / 32768.0)
for sr, sig in ((48000, x), (44100, y)):
    for unpruned in (False, True, False, True):
        e.set_option("tp_unpruned", unpruned)
        r = e.op_ebur128(sig, sr)
        t = e.timers()
        print(sr, "unpruned" if unpruned else "pruned", "true peak", r["true_peak"], "units", t["tp_units_evaluated"], "of", t["tp_units_total"], flush=True)
