"""adeclick solver phase clocks on the bench file (A/B build compiled with -DJT_DK_PROFILE; JT_LIB_PATH_AB = that library).
Prints the split's window counts and the <32> solver's wave clocks per phase (prologue, factorisation, back substitution, output)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from jivetalking_amd import _lib
if os.environ.get("JT_LIB_PATH_AB"): _lib.LIB_PATH_AB = os.environ["JT_LIB_PATH_AB"]
import torch  # noqa: F401
from jivetalking_amd import Engine, synth, hostlogic
sr = 48000
x = synth.speech_like_torch(3600.0, sr, seed=1000, device="cuda:0", plosives_per_min=40.0)
e = Engine(0, ab=True)
e.attach_device_pcm(x.data_ptr(), x.numel(), sr, 1, keepalive=x)
base = hostlogic.default_config()
hostlogic.process_audio(e, base, 4096)
e.set_option("dk_profile", "1")
hostlogic.process_audio(e, base, 4096)
print(e.timers())
