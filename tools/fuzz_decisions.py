"""The reference's decision chain (intervals -> VAD / elections -> band graphs -> AdaptConfig -> chain string) on the Pass-1 measurements
of the HIP path against the same chain on the CPU oracle's Pass 1 (tests/oracle_pass1.py), for random files: room tone of different
levels and colours, speech level, pauses, sibilance.  Elections must land on the same 250 ms intervals, every filter must be switched the
same way, printed parameters within the measurement tolerances.  usage: fuzz_decisions.py [cases] [seed]"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from jivetalking_amd import Engine, synth, hostlogic as H
from oracle import orc
import oracle_pass1 as P
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
e = Engine(0)


def parse(spec):
    out = []
    for f in spec.split(","):
        name, _, args = f.partition("=")
        out.append((name, dict(a.partition("=")[::2] for a in (args.split(":") if args else []))))
    return out


def close(a, b, rel=2e-3, abs_=2e-4):
    try:
        fa, fb = float(a), float(b)
    except ValueError:
        return a == b
    return abs(fa - fb) <= max(abs_, rel * max(abs(fa), abs(fb)))


bad = 0
for c in range(cases):
    sr = int(rng.choice([48000, 44100]))
    secs = float(rng.uniform(30.0, 70.0))
    x = np.asarray(synth.speech_like(secs, sr, seed=int(rng.integers(1, 10**6))), np.float64) * float(10 ** rng.uniform(-1.0, 0.2))
    kind = int(rng.integers(0, 4))
    if kind >= 1:                                            # room tone: white or low-passed, -75 .. -40 dBFS
        nz = rng.standard_normal(x.size)
        if kind == 2:
            nz = np.convolve(nz, np.ones(24) / 24, mode="same") * 4
        x += nz * float(10 ** rng.uniform(-3.75, -2.0))
    if kind == 3:                                            # long pauses
        for _ in range(int(rng.integers(1, 4))):
            a = int(rng.integers(0, x.size - 6 * sr)); x[a: a + int(rng.uniform(1.5, 5.0) * sr)] *= 0.003
    x = np.clip(x, -1, 1).astype(np.float32)
    e.upload_pcm(x, sr, 1)
    g = H.process_audio(e, analyse_only=True)
    t0 = time.time()
    m, eff, spec = P.decide(orc, x, sr)
    gm = g.input
    issues = []
    if (gm.has_speech_profile, gm.has_noise_profile, gm.voice_activated, gm.floor_source, gm.n_candidates, gm.n_speech_regions) != \
       (m.has_speech_profile, m.has_noise_profile, m.voice_activated, m.floor_source, m.n_candidates, m.n_speech_regions):
        issues.append("elections / switches")
    if m.has_speech_profile and gm.has_speech_profile and (gm.speech_profile.region.start_ns, gm.speech_profile.region.duration_ns) != (m.speech_profile.region.start_ns, m.speech_profile.region.duration_ns):
        issues.append("speech region")
    if m.has_noise_profile and gm.has_noise_profile and (gm.noise_profile.start_ns, gm.noise_profile.duration_ns) != (m.noise_profile.start_ns, m.noise_profile.duration_ns):
        issues.append("noise region")
    cg, co = parse(H.filter_spec(g.effective, 2)), parse(spec)
    if [f[0] for f in cg] != [f[0] for f in co]:
        issues.append("filter list")
    else:
        for (name, a), (_, b) in zip(cg, co):
            if a.keys() != b.keys():
                issues.append(name + " keys"); continue
            for k in a:
                if name == "afftdn" and k == "bn":
                    va, vb = [float(v) for v in a[k].split("|")], [float(v) for v in b[k].split("|")]
                    if len(va) != len(vb) or max(abs(p - q) for p, q in zip(va, vb)) > 0.1001:
                        issues.append("afftdn bn")
                elif not close(a[k], b[k]):
                    issues.append(f"{name}.{k} {a[k]} vs {b[k]}")
    bad += bool(issues)
    print(f"case {c:2d} {sr} Hz {secs:5.1f} s kind {kind}: speech {gm.has_speech_profile} noise {gm.has_noise_profile} candidates {gm.n_candidates} regions {gm.n_speech_regions} "
          f"I {gm.input_i:.2f} floor {gm.floor:.2f} (oracle {time.time() - t0:.1f} s): {'same' if not issues else 'DIFFERENT: ' + '; '.join(issues)}", flush=True)
print(f"{cases} cases, {bad} with differences")
