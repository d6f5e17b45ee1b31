"""adeclick's default kernel (summation order relaxed) against the sequential-order kernel (option adeclick_exact, bit-exact to the
oracle) INSIDE the four-pass job on a long file: repaired-sample counts, delivered s16 samples that differ and by how much; and on the
Pass-4 input itself (Pass-2 output x the loudnorm gain): detector decisions that flip, largest f64 difference.
python tools/declick_fast_vs_exact.py [minutes] [plosives_per_min]"""
import json, sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from jivetalking_amd import Engine, synth, hostlogic as H
mins = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
plos = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
SR = 48000
x = synth.speech_like_torch(mins * 60.0, SR, seed=1000, device="cuda:0", plosives_per_min=plos)
n = x.numel(); torch.cuda.synchronize()
e = Engine(0)
e.attach_device_pcm(x.data_ptr(), n, SR, 1, keepalive=x)
out = {}
res = {}
for mode in ("fast", "exact"):
    e.set_option("adeclick_exact", mode == "exact")
    t0 = time.perf_counter(); r = H.process_audio(e); dt = time.perf_counter() - t0
    res[mode] = (e.download_s16(4).copy(), int(e.timers()["declick_repaired"]), r.output_lufs, r.output_tp_db, dt)
e.set_option("adeclick_exact", False)
a, b = res["fast"][0].astype(np.int32), res["exact"][0].astype(np.int32)
d = np.abs(a - b)
out["job"] = {"minutes": mins, "repaired_fast": res["fast"][1], "repaired_exact": res["exact"][1], "s16_samples": int(a.size),
              "s16_differing": int(np.count_nonzero(d)), "s16_max_abs_diff": int(d.max()), "s16_diff_hist": {str(k): int(np.count_nonzero(d == k)) for k in range(1, 6)},
              "output_lufs": [res["fast"][2], res["exact"][2]], "output_dbtp": [res["fast"][3], res["exact"][3]], "ms": [round(res["fast"][4] * 1e3, 1), round(res["exact"][4] * 1e3, 1)]}
# the operator pair on the Pass-4 input (no limiter prefix in this reconstruction: the detector sees the same kind of signal)
p2 = e.download_s16(2).astype(np.float64) / 32768.0
gain = 10 ** ((r.effective_target_i - r.measure.input_i) / 20.0)
sig = p2 * gain
fa, ca = e.op_adeclick(sig, 44100, return_count=True)
e.set_option("adeclick_exact", True)
fb, cb = e.op_adeclick(sig, 44100, return_count=True)
e.set_option("adeclick_exact", False)
flips = int(np.count_nonzero((fa != sig) != (fb != sig)))
dd = np.abs(fa - fb)
out["operator"] = {"repaired_fast": ca, "repaired_exact": cb, "flipped_decisions": flips, "samples": int(sig.size), "max_abs_diff": float(dd.max()),
                   "differing_gt_1e-9": int(np.count_nonzero(dd > 1e-9)), "differing_gt_half_lsb": int(np.count_nonzero(dd > 0.5 / 32768))}
print(json.dumps(out))
