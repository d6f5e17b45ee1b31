"""FLAC legs on random material.  (1) The GPU encoder's image of random mono s16 signals (speech, noise, silence runs, full-scale squares,
DC, sparse impulses, ragged lengths) decodes to the same samples in the oracle's RFC 9639 decoder and in the GPU decoder, and the
STREAMINFO MD5 is the decoder's.  (2) Streams from the oracle's coverage encoder with random predictor / residual-coding / block-size
modes, 1-2 channels, 8-24 bits decode to the same samples on the GPU (mono: the one-walk kernel; and again with flac_no_ahead).
usage: fuzz_flac.py [cases] [seed]"""
import sys, numpy as np
sys.path.insert(0, '.')
from jivetalking_amd import Engine, synth
from oracle import orc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
e = Engine(0)
bad = 0
for c in range(cases):
    n = int(rng.integers(1, 400000)); kind = int(rng.integers(0, 7)); rate = int(rng.choice([44100, 48000, 8000, 96000, 22050]))
    if kind == 0: x = (np.asarray(synth.speech_like(n / 44100 + 0.1, 44100, seed=int(rng.integers(1, 10**6))))[:n] * 32768 * 10 ** rng.uniform(-1, 0.5)).clip(-32768, 32767)
    elif kind == 1: x = rng.integers(-32768, 32768, n)
    elif kind == 2: x = np.zeros(n); a = int(rng.integers(0, n)); x[a: a + int(rng.integers(1, 5000))] = rng.integers(-3000, 3000)
    elif kind == 3: x = np.where((np.arange(n) // int(rng.integers(1, 300))) % 2, 32767, -32768)
    elif kind == 4: x = np.full(n, int(rng.integers(-32768, 32768)))
    elif kind == 5: x = np.zeros(n); x[rng.integers(0, n, size=max(1, n // 5000))] = rng.integers(-32768, 32768, size=max(1, n // 5000))
    else: x = (rng.standard_normal(n).cumsum() * 50).clip(-32768, 32767)
    x = np.asarray(x, np.int16)
    data = e.op_flac_encode(x, rate, md5=True)
    rc, dec, info = orc.flac_decode(data)
    g, _, meta = e.op_decode_audio(data)
    ok = rc == 0 and np.array_equal(dec[:, 0], x) and np.array_equal(g[:, 0], x.astype(np.int32)) and bytes(info.md5_stored) == bytes(info.md5_decoded)
    if not ok:
        bad += 1; print(f"(1) case {c} kind {kind} n {n} rate {rate}: FAILED (oracle rc {rc})", flush=True)
for c in range(cases):
    ch = int(rng.choice([1, 1, 2])); bps, order = [(8, 3), (16, 8), (24, 32), (16, 12), (20, 5), (12, 2)][int(rng.integers(0, 6))]
    mode = int(rng.choice([0, 1, 2])) | (8 if rng.random() < 0.3 else 0) | (16 if rng.random() < 0.3 else 0) | (32 if rng.random() < 0.4 else 0)
    if ch == 2: mode |= int(rng.choice([0, 64, 128, 192]))
    n = int(rng.integers(100, 60000)); bs = int(rng.choice([16, 192, 576, 1152, 4096, 4608]))
    x = (rng.standard_normal((n, ch)).cumsum(0) * (1 << (bps - 6)) / 30).clip(-(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int32)
    if mode & 8: x[: min(n, 512)] &= ~7
    data = orc.flac_encode(x, 44100, bps, bs, mode, order)
    g, _, _ = e.op_decode_audio(data)
    e.set_option("flac_no_ahead", True); g2, _, _ = e.op_decode_audio(data); e.set_option("flac_no_ahead", False)
    if not (np.array_equal(g, x) and np.array_equal(g2, x)):
        bad += 1; print(f"(2) case {c} ch {ch} bps {bps} order {order} mode {mode} bs {bs} n {n}: FAILED", flush=True)
print(f"{2 * cases} cases, {bad} failed")
