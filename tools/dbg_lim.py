import sys, numpy as np, ctypes as C
sys.path.insert(0,'.')
from jivetalking_amd import Engine, synth, hostlogic as H, _lib as L
SR=48000
x = synth.speech_like(30.0, SR, seed=51)
x[::24000] += 0.4 * np.sign(x[::24000] + 1e-9)
x = np.clip(x, -0.98, 0.98).astype(np.float32)
e=Engine(0); e.upload_pcm(x, SR, 1)
l=H.lib(); base=H.default_config(); res=H.ProcessResult()
rc=l.jt_process_audio(e.h, C.byref(base), 4096, C.byref(res))
print("rc",rc, l.jt_last_error(e.h).decode())
print("input I", res.input.input_i, "tp", res.input.input_tp, "lra", res.input.input_lra)
print("filtered I", res.filtered.r128.integrated, "tp lin", res.filtered.r128.true_peak, "lra", res.filtered.r128.lra)
print("limiter needed", res.limiter.needed, "ceil", res.limiter.ceiling_db, "pregain", res.limiter.pre_gain_db, "gain_db", res.limiter.gain_db, "clamped", res.limiter.clamped)
print("measure", res.measure.input_i, res.measure.input_tp, res.measure.input_lra, res.measure.input_thresh)
print("eff target", res.effective_target_i, "offset", res.offset, "linear_possible", res.linear_possible)
print(res.pass4_spec.decode()[:330])
