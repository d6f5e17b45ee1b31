import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from jivetalking_amd import Engine, synth, hostlogic
e = Engine(0)
base = hostlogic.default_config()
for sg, pl in ((1.2, 40), (1.4, 40), (1.7, 40), (2.0, 40)):
    x = synth.speech_like_torch(600.0, 48000, seed=1000, device="cuda:0", plosives_per_min=pl, sib_gain=sg, sib_band=True)
    e.attach_device_pcm(x.data_ptr(), x.numel(), 48000, 1, keepalive=x)
    r = hostlogic.process_audio(e, base, 4096, analyse_only=True)
    sp = r.input.speech_profile
    spec = hostlogic.filter_spec(r.effective, 2)
    print(f"sib_gain {sg} plosives {pl}: body {sp.body_band_rms:.2f} sib {sp.sib_band_rms:.2f} excess {sp.sib_band_rms - sp.body_band_rms:.2f} measured {sp.bands_measured} deesser {'deesser' in spec}")
