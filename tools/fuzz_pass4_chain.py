"""Pass 3 / Pass 4 of random short files on the GPU against the reference's chain composed from the CPU oracle (oracle/chain.py), given
the GPU run's Pass-2 output and the spec string the host logic printed for Pass 4: delivered s16 sample by sample, the landing.
Levels, bursts, hiss and quiet stretches are drawn so that the plain linear branch, the limiter prefix and loudnorm's dynamic mode all
occur.  usage: fuzz_pass4_chain.py [cases] [seed]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth, hostlogic as H
from oracle import chain, orc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
SR = 48000
e = Engine(0)
worst = 0; tally = {}
for c in range(cases):
    secs = float(rng.uniform(18.0, 40.0))
    x = np.asarray(synth.speech_like(secs, SR, seed=int(rng.integers(1, 10**6))), np.float64) * float(10 ** rng.uniform(-1.2, 0.3))
    kind = int(rng.integers(0, 4))
    if kind == 1:                                            # plosive bursts: the limiter prefix
        w = int(0.02 * SR); b = float(rng.uniform(0.2, 0.6)) * np.hanning(w) * np.sin(2 * np.pi * 180.0 * np.arange(w) / SR)
        for pos in range(SR, x.size - SR, int(rng.uniform(0.7, 2.5) * SR)):
            x[pos:pos + w] += b
    elif kind == 2:                                          # hiss bursts far above the speech: linear mode impossible
        for pos in range(2 * SR, x.size - 2 * SR, int(rng.uniform(3, 8) * SR)):
            n = int(rng.uniform(0.05, 0.4) * SR); x[pos:pos + n] += rng.standard_normal(n) * float(rng.uniform(0.1, 0.5))
    elif kind == 3:                                          # a quiet lead-in
        x[: int(rng.uniform(2, 8) * SR)] *= 10 ** rng.uniform(-3, -1.5)
    x = np.clip(x, -1.0, 1.0)
    e.upload_pcm(x.astype(np.float32), SR, 1)
    res = H.process_audio(e)
    p2, p4 = e.download_s16(2), e.download_s16(4)
    t0 = time.time()
    ref = chain.pass4(p2, 44100, bytes(res.pass4_spec).split(b"\0")[0])
    d = np.abs(ref["s16"].astype(np.int32) - p4.astype(np.int32))
    land = chain.landing(ref["s16"], 44100)
    branch = ("dynamic" if res.loudnorm.normalization_type_dynamic else "linear") + ("+prefix" if res.limiter.needed else "")
    tally[branch] = tally.get(branch, 0) + 1
    worst = max(worst, int(d.max()))
    flag = "" if (d.max() <= 1 and ref["dynamic"] == int(res.loudnorm.normalization_type_dynamic) and abs(land["output_lufs"] - res.output_lufs) <= 0.011) else "   <-- LOOK"
    print(f"case {c:2d} kind {kind} {secs:5.1f} s {branch:15s}: max |d| {int(d.max())} LSB, {int(np.count_nonzero(d))} of {d.size} differ; lands {res.output_lufs:.2f} LUFS / {res.output_tp_db:.2f} dBTP, "
          f"oracle {land['output_lufs']:.2f} / {land['output_dbtp']:.2f} (oracle {time.time() - t0:.1f} s){flag}", flush=True)
print("branches:", tally, " worst difference:", worst, "LSB")
