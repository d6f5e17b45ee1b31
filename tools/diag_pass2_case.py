"""One case of tools/fuzz_pass2.py stage by stage: every Pass-2 filter on the GPU against the oracle's, each fed the ORACLE's output of
the stage before (so a difference belongs to the stage it shows up in).  usage: diag_pass2_case.py <seed> <case>"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from jivetalking_amd import Engine, synth, hostlogic as H, _lib as L
from oracle import orc
rng = np.random.default_rng(int(sys.argv[1])); want = int(sys.argv[2])
e = Engine(0)
for c in range(want + 1):
    sr = int(rng.choice([48000, 48000, 44100])); secs = float(rng.uniform(16.0, 30.0)); kind = int(rng.integers(0, 5)); seed = int(rng.integers(1, 10**6))
    if kind == 4:
        x = synth.speech_like_torch(secs, sr, seed=seed, device="cuda:0", sib_gain=float(rng.uniform(0.4, 1.5)), sib_band=True).cpu().numpy().astype(np.float64)
    else:
        x = np.asarray(synth.speech_like(secs, sr, seed=seed, speech_dbfs=float(rng.uniform(-36, -24)), room_dbfs=float(rng.uniform(-75, -50))), np.float64)
    x *= float(10 ** rng.uniform(-1.0, 0.2))
    if kind in (1, 2, 3):
        nz = rng.standard_normal(x.size)
        if kind == 2:
            nz = np.convolve(nz, np.ones(24) / 24, mode="same") * 4
        x += nz * float(10 ** rng.uniform(-3.75, -2.2))
    if kind == 3:
        a = int(rng.integers(0, x.size - 5 * sr)); x[a: a + int(rng.uniform(1.5, 4.0) * sr)] *= 0.003
    x = np.clip(x, -1, 1).astype(np.float32)
print("case", want, sr, "Hz", secs, "s kind", kind, "n", x.size)
e.upload_pcm(x, sr, 1); res = H.process_audio(e); p2 = e.download_s16(2)
fp = L.FilterParams(); H.lib().jt_host_filter_params(C.byref(res.effective), C.byref(fp))
print("spec:", bytes(res.pass2_spec).split(b"\0")[0].decode()[:600])
def rep(name, g, o, scale=32768.0):
    d = np.abs(np.asarray(g, np.float64) - np.asarray(o, np.float64)) * scale
    i = int(np.argmax(d))
    print(f"  {name:12s}: max {d.max():.4f} LSB-equivalents at {i} ({i / sr:.2f} s), mean {d.mean():.5f}; |oracle| there {abs(float(o[i])):.6f}", flush=True)
y = x
o1 = orc.biquad_f32(y, 0, fp.hp_freq, sr, fp.hp_q) if fp.hp_enabled else y
o1 = orc.biquad_f32(o1, 1, fp.lp_freq, sr, fp.lp_q) if fp.lp_enabled else o1
g1 = e.op_biquad(y, sr, (int(fp.hp_enabled), fp.hp_freq, fp.hp_q), (int(fp.lp_enabled), fp.lp_freq, fp.lp_q)); rep("biquads", g1, o1)
o2 = orc.anlmdn(o1, sr, fp.nlm_strength, fp.nlm_patch_s, fp.nlm_research_s, fp.nlm_smooth)
g2 = e.op_anlmdn(o1, sr, fp.nlm_strength, fp.nlm_patch_s, fp.nlm_research_s, fp.nlm_smooth); rep("anlmdn", g2, o2)
bn = list(fp.fft_band_noise) if fp.fft_custom else None
nf = fp.fft_nf if fp.fft_nf < 0 else -50.0
o3 = orc.afftdn(o2, sr, fp.fft_nr, nf, bn)
g3 = e.op_afftdn(o2, sr, fp.fft_nr, nf, bn); rep("afftdn", g3, o3)
od = o3.astype(np.float64)
od = orc.agate(od, sr, fp.gate_threshold, fp.gate_ratio, fp.gate_attack_ms, fp.gate_release_ms, fp.gate_range, fp.gate_knee, fp.gate_makeup)
od = orc.acompressor(od, sr, fp.comp_threshold, fp.comp_ratio, fp.comp_attack_ms, fp.comp_release_ms, fp.comp_makeup, fp.comp_knee, fp.comp_mix)
g4 = e.op_dynamics(o3, sr, fp); rep("gate + comp", g4, od.astype(np.float32))
ref = orc.f64_to_s16(od.astype(np.float32).astype(np.float64) if sr == 44100 else orc.swr_f64(od.astype(np.float32).astype(np.float64), sr, 44100, True))
d = np.abs(ref.astype(np.int32) - p2.astype(np.int32)); i = int(np.argmax(d))
print(f"  whole pass  : max {d.max()} LSB at {i} ({i / 44100:.2f} s), mean {d.mean():.4f}; oracle s16 there {ref[i]}, gpu {p2[i]}")
# where do the large differences sit?
big = np.nonzero(d >= 3)[0]
print("  samples with >= 3 LSB:", big.size, "first", big[:10], "time span", (big.min() / 44100, big.max() / 44100) if big.size else None)
print("biquads apart:")
for nm, hp, lp in (("highpass", (1, fp.hp_freq, fp.hp_q), (0, 20500.0, 0.707)), ("lowpass", (0, 80.0, 0.707), (1, fp.lp_freq, fp.lp_q))):
    g = e.op_biquad(x, sr, hp, lp)
    o = orc.biquad_f32(x, 0, hp[1], sr, hp[2]) if hp[0] else orc.biquad_f32(x, 1, lp[1], sr, lp[2])
    rep(nm, g, o)
    dd = np.abs(g.astype(np.float64) - o.astype(np.float64)); rel = dd / np.maximum(np.abs(o.astype(np.float64)), 1e-6)
    print("     exact samples:", int(np.count_nonzero(dd == 0)), "of", dd.size, " median |d| / ulp(o):", float(np.median(dd / np.maximum(np.spacing(np.abs(o)), 1e-30))))
for rate in (48000, 44100, 32000, 22050):
    g = e.op_biquad(x, rate, (0, 80.0, 0.707), (1, min(20500.0, rate * 0.46), 0.707)); o = orc.biquad_f32(x, 1, min(20500.0, rate * 0.46), rate, 0.707)
    print("  lowpass at rate", rate, "f", min(20500.0, rate * 0.46), ": max |d|", float(np.abs(g - o).max()), "exact", int(np.count_nonzero(g == o)), "of", g.size)
print("highpass 80 Hz alone at several rates, same samples; and on other signals at 44.1 kHz:")
for rate in (48000, 44100, 32000, 96000):
    g = e.op_biquad(x, rate, (1, 80.0, 0.707), (0, 20500.0, 0.707)); o = orc.biquad_f32(x, 0, 80.0, rate, 0.707)
    print("  rate", rate, ": exact", int(np.count_nonzero(g == o)), "of", g.size, " max |d|", float(np.abs(g - o).max()))
for nm, sig in (("speech only", np.asarray(synth.speech_like(21.0, 44100, seed=5), np.float32)),
                ("white noise -50 dB", (np.random.default_rng(1).standard_normal(928670) * 10 ** -2.5).astype(np.float32)),
                ("speech + dc 0.01", np.asarray(synth.speech_like(21.0, 44100, seed=5), np.float32) + np.float32(0.01))):
    g = e.op_biquad(sig, 44100, (1, 80.0, 0.707), (0, 20500.0, 0.707)); o = orc.biquad_f32(sig, 0, 80.0, 44100, 0.707)
    print("  ", nm, ": exact", int(np.count_nonzero(g == o)), "of", g.size, " max |d|", float(np.abs(g - o).max()))
print("the ORACLE chain's own sensitivity: the GPU's biquad output and the oracle's biquad output, both through the oracle's later stages")
def later(y1):
    y2 = orc.anlmdn(y1, sr, fp.nlm_strength, fp.nlm_patch_s, fp.nlm_research_s, fp.nlm_smooth)
    y3 = orc.afftdn(y2, sr, fp.fft_nr, nf, bn)
    yd = orc.agate(y3.astype(np.float64), sr, fp.gate_threshold, fp.gate_ratio, fp.gate_attack_ms, fp.gate_release_ms, fp.gate_range, fp.gate_knee, fp.gate_makeup)
    y5 = orc.acompressor(yd, sr, fp.comp_threshold, fp.comp_ratio, fp.comp_attack_ms, fp.comp_release_ms, fp.comp_makeup, fp.comp_knee, fp.comp_mix)
    return y2, y3, yd, y5
A = later(g1); Bq = later(o1)
for nm, a, b in zip(("anlmdn", "afftdn", "agate", "acompressor"), A, Bq):
    rep("after " + nm, a, b)
print("the GPU's own operators composed (each fed the GPU's previous output) against the pipeline's Pass 2:")
c2 = e.op_anlmdn(g1, sr, fp.nlm_strength, fp.nlm_patch_s, fp.nlm_research_s, fp.nlm_smooth)
c3 = e.op_afftdn(c2, sr, fp.fft_nr, nf, bn)
c4 = e.op_dynamics(c3, sr, fp)
comp16 = orc.f64_to_s16(c4.astype(np.float64) if sr == 44100 else orc.swr_f64(c4.astype(np.float64), sr, 44100, True))
for nm, a in (("composition vs pipeline", comp16), ("composition vs oracle chain", None)):
    b = p2 if a is not None else ref
    a = comp16
    dd = np.abs(a.astype(np.int32) - b.astype(np.int32)); print(f"  {nm}: max {dd.max()} LSB, mean {dd.mean():.4f}, differing {int(np.count_nonzero(dd))}")
print("the pipeline again with schedule switches:")
for opt in ("no_pass2_prefetch", "no_early_pass3", "no_early_plan", "no_staged_finish", "no_r128_first", "no_spec_direct", "nlm_generic", "adeclick_exact"):
    e.set_option(opt, True)
    e.upload_pcm(x, sr, 1); r2 = H.process_audio(e); q2 = e.download_s16(2)
    e.set_option(opt, False)
    dd = np.abs(q2.astype(np.int32) - comp16.astype(np.int32))
    print(f"  {opt:20s}: vs composition max {dd.max()} LSB, mean {dd.mean():.4f}; vs default pipeline {'same' if np.array_equal(q2, p2) else 'DIFFERENT'}")
print("jt_pass2 called alone with the parameters jt_host_filter_params reports:")
e.upload_pcm(x, sr, 1); e.pass2(fp); q = e.download_s16(2)
for nm, b in (("pipeline's Pass 2", p2), ("composition", comp16), ("oracle chain", ref)):
    dd = np.abs(q.astype(np.int32) - b.astype(np.int32)); print(f"  jt_pass2 alone vs {nm}: max {dd.max()} LSB, mean {dd.mean():.4f}")
for name in ("hp_enabled", "hp_freq", "hp_q", "lp_enabled", "lp_freq", "nlm_enabled", "nlm_strength", "fft_enabled", "fft_nr", "fft_nf", "fft_custom", "fft_track_noise",
             "gate_enabled", "gate_threshold", "gate_ratio", "gate_attack_ms", "gate_release_ms", "gate_range", "gate_knee", "gate_makeup",
             "comp_enabled", "comp_threshold", "comp_ratio", "comp_attack_ms", "comp_release_ms", "comp_makeup", "comp_knee", "comp_mix", "deess_enabled", "out_rate"):
    if hasattr(fp, name): print("   ", name, getattr(fp, name))
