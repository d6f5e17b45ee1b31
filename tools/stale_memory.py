"""Does anything read memory it did not write?  (1) every operator and the whole pipeline on a short file after a long, loud one on the same
handle must equal the same call on a fresh handle; (2) run with --poison (the process-wide option poison_alloc: allocations filled with 0xFF)
the results must not move either.  python tools/stale_memory.py [--poison]"""
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from jivetalking_amd import Engine, synth, hostlogic as H, _lib as L
from jivetalking_amd.engine import default_filter_params
POISON = "--poison" in sys.argv
if POISON: L.set_global_option("poison_alloc", "1")
def dig(o):
    h = hashlib.sha256()
    def walk(v):
        if isinstance(v, dict):
            for k in sorted(v): h.update(str(k).encode()); walk(v[k])
        elif isinstance(v, (list, tuple)):
            for x in v: walk(x)
        elif v is None: h.update(b"none")
        else: h.update(np.ascontiguousarray(np.asarray(v)).tobytes())
    walk(o); return h.hexdigest()[:12]
def loud(sr, secs, seed):
    r = np.random.default_rng(seed); n = int(sr * secs)
    return (0.9 * np.sign(r.standard_normal(n)) * r.random(n)).astype(np.float32)             # full-scale junk to leave behind
def ops(e, x, sr):
    p = default_filter_params()
    s16 = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)
    out = {}
    out["biquad"] = dig(e.op_biquad(x, sr)); out["anlmdn"] = dig(e.op_anlmdn(x, sr)); out["afftdn"] = dig(e.op_afftdn(x, sr, 12.0, -50.0))
    out["afftdn_tn1"] = dig(e.op_afftdn(x, sr, 12.0, -50.0, track=True)); out["dynamics"] = dig(e.op_dynamics(x, sr, p))
    out["adeclick"] = dig(e.op_adeclick(x.astype(np.float64), 44100)); out["alimiter"] = dig(e.op_alimiter(x.astype(np.float64), sr, 0.3))
    out["resample"] = dig(e.op_resample_s16(x, sr, 44100)) if sr != 44100 else "-"
    out["ebur128"] = dig(e.op_ebur128(x, sr)); out["astats"] = dig(e.op_astats(x, sr)); out["spectral"] = dig(e.op_aspectralstats(x, sr))
    out["ln_measure"] = dig(e.op_loudnorm_measure_s16(s16, 44100)); out["ln_measure_lim"] = dig(e.op_loudnorm_measure_s16(s16, 44100, limiter=L.LimiterPlan(1, 0.25, 1.0)))
    img = e.op_flac_encode(s16, 44100, md5=True)
    out["flac"] = dig(img)
    out["flac_decode"] = dig(e.op_decode_audio(img)[0])                                          # mono: one walk per candidate, rows gathered
    # dynamic-mode loudnorm at 192 kHz (the stream path behind the first launch): the signal stretched to 192 kHz by repetition, driven into the ceiling
    x192 = np.repeat(x.astype(np.float64), 192000 // sr + 1)[: int(192000 * (x.size / sr) * 1.9)] * 3.0
    y192, st192 = e.op_loudnorm_dynamic(x192, target_tp=-9.0)
    out["ln_dynamic"] = dig(y192) + dig(st192)
    e.upload_pcm(x, sr, 1); r = H.process_audio(e, H.default_config(), 4096)
    y = np.empty(x.size, np.int16); got = e.download_s16_into(4, y); out["pipeline"] = dig(y[:got])
    return out
bad = 0
for sr in (48000, 44100):
    short = synth.speech_like(7.3, sr, seed=5).astype(np.float32)
    fresh = ops(Engine(0), short, sr)
    e = Engine(0); ops(e, loud(sr, 31.7, 9), sr)
    used = ops(e, short, sr)
    for k in fresh:
        if fresh[k] != used[k]: bad += 1; print("DIFFERS after a longer file:", sr, k, fresh[k], used[k])
print("stale-memory check:", "clean" if bad == 0 else "%d differences" % bad, "(poisoned allocations)" if POISON else "")
