"""Full-size property check of the round-3 schedule / kernel changes: the 60-min bench file through jt_process_audio with every new path on
(split adeclick, planned Pass 3) and with each switched off (JT_ADECLICK_FUSED, JT_NO_EARLY_PLAN, JT_DK_LEVINSON_IN_KERNEL): the final s16
output and every reported number must be identical.  python tools/ab_pipeline_fused.py [minutes]"""
import hashlib, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from jivetalking_amd import Engine, synth, hostlogic
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
sr = 48000
x = synth.speech_like_torch(minutes * 60.0, sr, seed=1000, device="cuda:0", plosives_per_min=40.0)
e = Engine(0)
e.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x)
base = hostlogic.default_config()
def run(env):
    for k in ("JT_ADECLICK_FUSED", "JT_NO_EARLY_PLAN", "JT_DK_LEVINSON_IN_KERNEL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    r = hostlogic.process_audio(e, base, 4096)
    out = e.download_s16(4)
    return (hashlib.md5(out.tobytes()).hexdigest(), r.output_lufs, r.output_tp_db, r.measure.input_i, r.measure.input_tp, int(r.limiter.needed),
            r.final_.r128.true_peak, r.final_.astats.rms_level, e.timers()["declick_repaired"])
ref = run({})
print("default        ", ref)
rc = 0
for name, env in (("fused adeclick ", {"JT_ADECLICK_FUSED": "1"}), ("no early plan  ", {"JT_NO_EARLY_PLAN": "1"}), ("levinson inside", {"JT_DK_LEVINSON_IN_KERNEL": "1"})):
    got = run(env)
    same = got == ref
    print(name, "identical" if same else f"DIFFERENT {got}")
    rc |= 0 if same else 1
sys.exit(rc)
