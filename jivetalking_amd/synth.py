"""Deterministic synthetic inputs for tests and bench (no network, no datasets).

* reference_fixture(): the reference's own hermetic test signal, bit-reproducible — int16 mono, sine + LCG white
  noise (rngState=12345, x = x*1664525 + 1013904223) + optional gap (internal/processor/testutil_test.go:28-135;
  benchmark variant benchmark_test.go:94-109).
* speech_like(): seeded speech-shaped signal: glottal pulse train through three formant resonators,
  3-6 Hz syllable envelope, phrases 2-12 s separated by pauses 0.3-3 s, sibilant bursts 6-9 kHz, pink-ish room tone.
* speech_like_torch(): the same recipe generated directly in HBM (bench; inputs resident before the timed region).
"""
import numpy as np


def reference_fixture(duration_s=5.0, sample_rate=44100, tone_hz=440.0, tone_dbfs=-23.0, noise_dbfs=0.0,
                      gap_start=0.0, gap_dur=0.0):
    n = int(duration_s * sample_rate)
    tone_amp = 10.0 ** (tone_dbfs / 20.0) if (tone_hz > 0 and tone_dbfs < 0) else 0.0
    noise_amp = 10.0 ** (noise_dbfs / 20.0) if noise_dbfs < 0 else 0.0
    s0 = int(gap_start * sample_rate)
    s1 = int((gap_start + gap_dur) * sample_rate)
    idx = np.arange(n)
    in_gap = (idx >= s0) & (idx < s1) & (gap_dur > 0)
    sample = np.zeros(n, np.float64)
    if tone_amp > 0:
        t = idx.astype(np.float64) / float(sample_rate)
        sample += np.where(in_gap, 0.0, tone_amp * np.sin(2.0 * np.pi * tone_hz * t))
    if noise_amp > 0:
        # LCG stream: one draw per sample (gap samples draw too when a noise floor is configured).
        # Jump-ahead by doubling: x_k = A[k]*x0 + C[k] (mod 2^32), A[m+j] = A[m]*A[j], C[m+j] = A[j]*C[m] + C[j].
        M = np.uint64(0xFFFFFFFF)
        A = np.empty(n + 1, np.uint64); Cc = np.empty(n + 1, np.uint64)
        A[0], Cc[0] = 1, 0
        if n >= 1:
            A[1], Cc[1] = 1664525, 1013904223
        m = 1
        while m < n:
            j = min(m, n - m)
            A[m + 1:m + j + 1] = (A[m] * A[1:j + 1]) & M
            Cc[m + 1:m + j + 1] = (A[1:j + 1] * Cc[m] + Cc[1:j + 1]) & M
            m += j
        buf = ((A[1:] * np.uint64(12345) + Cc[1:]) & M).astype(np.float64)
        rnd = (buf / float(0xFFFFFFFF)) * 2.0 - 1.0
        sample += noise_amp * rnd
    else:
        sample = np.where(in_gap, 0.0, sample)
    sample = np.clip(sample, -1.0, 1.0)
    return np.trunc(sample * 32767.0).astype(np.int16)


def _resonator(x, f, bw, sr):
    """two-pole resonator (formant) — scipy.signal.lfilter for speed"""
    from scipy.signal import lfilter
    r = np.exp(-np.pi * bw / sr)
    a1, a2 = -2 * r * np.cos(2 * np.pi * f / sr), r * r
    g = 1 - r
    return lfilter([g], [1.0, a1, a2], x)


def speech_like(duration_s, sample_rate=48000, seed=0, speech_dbfs=-30.0, room_dbfs=-62.0):
    """Seeded speech-shaped f32 mono signal in [-1, 1]."""
    from scipy.signal import lfilter
    rng = np.random.default_rng(seed)
    sr = sample_rate
    n = int(duration_s * sr)
    t = np.arange(n) / sr
    # phrase / pause gating
    gate = np.zeros(n)
    pos = int(rng.uniform(0.5, 1.5) * sr)
    while pos < n:
        plen = int(rng.uniform(2.0, 12.0) * sr)
        gate[pos:pos + plen] = 1.0
        pos += plen + int(rng.uniform(0.3, 3.0) * sr)
    # long pauses so >= 8-10 s room-tone runs exist for the VAD
    k = max(1, int(duration_s // 60))
    for _ in range(k):
        s = int(rng.uniform(0.05, 0.9) * n)
        gate[s:s + int(rng.uniform(9.0, 12.0) * sr)] = 0.0
    ramp = int(0.02 * sr)
    gate = lfilter(np.ones(ramp) / ramp, [1.0], gate)
    # F0 contour and glottal pulse train
    f0 = 150 + 50 * np.sin(2 * np.pi * 0.23 * t + rng.uniform(0, 6)) + 15 * np.sin(2 * np.pi * 1.7 * t)
    phase = np.cumsum(f0 / sr)
    pulses = (np.diff(np.floor(phase), prepend=0.0) > 0).astype(np.float64)
    src = lfilter([1.0], [1.0, -0.97], pulses) - 0.02
    v = _resonator(src, 650, 90, sr) + 0.6 * _resonator(src, 1450, 120, sr) + 0.3 * _resonator(src, 2700, 160, sr)
    syl = 0.55 + 0.45 * np.sin(2 * np.pi * (4.2 + 0.8 * np.sin(2 * np.pi * 0.11 * t)) * t)
    v = v * np.clip(syl, 0, None)
    # sibilants: band-limited noise bursts
    nz = rng.standard_normal(n)
    sib = _resonator(nz, 7500, 2500, sr)
    sib_env = (np.sin(2 * np.pi * 1.3 * t + 1.0) > 0.93).astype(np.float64)
    sib_env = lfilter(np.ones(ramp) / ramp, [1.0], sib_env)
    speech = (v / (np.sqrt(np.mean(v ** 2)) + 1e-12) + 0.25 * sib * sib_env / (np.std(sib) + 1e-12)) * gate
    sp_rms = np.sqrt(np.mean(speech[gate > 0.5] ** 2)) if np.any(gate > 0.5) else 1.0
    speech *= 10 ** (speech_dbfs / 20.0) / (sp_rms + 1e-12)
    # room tone: pink-ish noise
    room = lfilter([0.05], [1.0, -0.95], rng.standard_normal(n))
    room *= 10 ** (room_dbfs / 20.0) / (np.std(room) + 1e-12)
    x = speech + room
    peak = np.max(np.abs(x))
    lim = 10 ** (-3.0 / 20.0)
    if peak > lim:
        x *= lim / peak
    return x.astype(np.float32)


def speech_like_torch(duration_s, sample_rate=48000, seed=0, device="cuda", speech_dbfs=-30.0, room_dbfs=-62.0,
                      plosives_per_min=0.0, sib_gain=0.25, sib_band=False):
    """Speech-shaped f32 mono signal generated directly on the GPU (torch is plumbing for device memory only).

    Harmonic synthesis instead of recursive resonators (recursions do not vectorise): voiced phrases = sum of
    F0 harmonics weighted by a three-formant envelope, syllable AM, sibilant noise bursts, low-passed room tone.
    `plosives_per_min` > 0 adds that many 20 ms plosive-like bursts a minute inside the phrases (peaks ~-9 dBFS over a -30 dBFS voice:
    the ~20 dB crest factor of a real close-miked talker, which makes the loudnorm plan need the limiter prefix, normalise.go:452-497);
    `sib_gain` scales the sibilant bursts (0.25 = the default voice); `sib_band` concentrates them in 6.75-8.25 kHz (smoothed noise on a
    7.5 kHz carrier instead of differenced white noise): with sib_gain ~0.5 the 6-9 kHz band then sits within 6 dB of the 1-3 kHz body
    band, which is what makes AdaptConfig switch the de-esser on (adaptive_deesser.go:45)."""
    import torch
    g = torch.Generator(device=device).manual_seed(int(seed))
    sr = sample_rate
    n = int(duration_s * sr)
    t = torch.arange(n, device=device, dtype=torch.float64) / sr
    # phrase gate on a 10 ms grid (host-side small loop), upsampled
    rng = np.random.default_rng(seed)
    grid = int(duration_s * 100) + 1
    gate = np.zeros(grid, np.float32)
    pos = int(rng.uniform(0.5, 1.5) * 100)
    while pos < grid:
        plen = int(rng.uniform(2.0, 12.0) * 100)
        gate[pos:pos + plen] = 1.0
        pos += plen + int(rng.uniform(0.3, 3.0) * 100)
    for _ in range(max(1, int(duration_s // 60))):
        s = int(rng.uniform(0.05, 0.9) * grid)
        gate[s:s + int(rng.uniform(9.0, 12.0) * 100)] = 0.0
    gate_t = torch.from_numpy(gate).to(device)
    idx = torch.clamp((t * 100).long(), max=grid - 1)
    frac = (t * 100 - idx.double()).float()
    g0 = gate_t[idx]
    g1 = gate_t[torch.clamp(idx + 1, max=grid - 1)]
    gate_s = g0 + (g1 - g0) * frac
    f0 = 150 + 50 * torch.sin(2 * np.pi * 0.23 * t + 1.1) + 15 * torch.sin(2 * np.pi * 1.7 * t)
    # the phase is the integral of f0 in closed form: torch's device-side cumsum (a decoupled look-back scan) sums 10^8 doubles in an order
    # that varies from run to run, and with it the last bits of every sample of the talker -- the bench file was not the same file twice
    del f0
    w1, w2 = 2 * np.pi * 0.23, 2 * np.pi * 1.7
    phase = (150 * t - (50 / w1) * torch.cos(w1 * t + 1.1) - (15 / w2) * torch.cos(w2 * t)) * (2 * np.pi)
    v = torch.zeros(n, device=device, dtype=torch.float32)

    def formant(f):
        return (np.exp(-0.5 * ((f - 650) / 180) ** 2) + 0.6 * np.exp(-0.5 * ((f - 1450) / 250) ** 2)
                + 0.3 * np.exp(-0.5 * ((f - 2700) / 350) ** 2) + 0.02)
    for k in range(1, 25):
        v += float(formant(150.0 * k) / k ** 0.5) * torch.sin(phase * k).float()
    syl = 0.55 + 0.45 * torch.sin(2 * np.pi * (4.2 + 0.8 * torch.sin(2 * np.pi * 0.11 * t)) * t)
    v = v * torch.clamp(syl, min=0).float()
    nz = torch.randn(n, device=device, dtype=torch.float32, generator=g)
    if sib_band:
        lp = torch.nn.functional.avg_pool1d(nz[None, None, :], 32, stride=1, padding=16)[0, 0, :n]      # ~750 Hz of noise bandwidth
        hp = lp * torch.cos(2 * np.pi * 7500.0 * t).float()
    else:
        hp = nz - torch.roll(nz, 1)               # crude high-pass -> sibilant-band emphasis
    sib_env = (torch.sin(2 * np.pi * 1.3 * t + 1.0) > 0.93).float()
    speech = (v / (v.std() + 1e-12) + float(sib_gain) * hp / (hp.std() + 1e-12) * sib_env) * gate_s
    act = gate_s > 0.5
    sp_rms = speech[act].pow(2).mean().sqrt() if bool(act.any()) else torch.tensor(1.0, device=device)
    speech = speech * (10 ** (speech_dbfs / 20.0) / (sp_rms + 1e-12))
    rn = torch.randn(n, device=device, dtype=torch.float32, generator=g)
    room = (rn + torch.roll(rn, 1) + torch.roll(rn, 2) + torch.roll(rn, 3)) * 0.25
    room = room * (10 ** (room_dbfs / 20.0) / (room.std() + 1e-12))
    x = speech + room
    if plosives_per_min > 0 and n > 2 * sr:
        cnt = max(1, int(duration_s / 60.0 * plosives_per_min))
        gp = torch.Generator(device=device).manual_seed(int(seed) + 7)
        pos = torch.randint(sr, n - sr, (cnt,), device=device, generator=gp)
        w = int(0.02 * sr)
        tt = torch.arange(w, device=device)
        burst = (0.35 * torch.hann_window(w, device=device) * torch.sin(2 * np.pi * 180.0 * tt / sr)).float()
        keep = gate_s[pos] > 0.5                                   # bursts belong to phrases, not to the room tone
        pos = pos[keep] if bool(keep.any()) else pos
        # bursts that would overlap an earlier one are dropped: index_add_ adds with atomics, and two bursts on the same samples were summed in
        # an order that varied from run to run (a last-bit difference that the peak normalisation below then spread over the whole file)
        pos, _ = torch.sort(pos)
        if pos.numel() > 1:
            ok = torch.ones_like(pos, dtype=torch.bool)
            ok[1:] = (pos[1:] - pos[:-1]) >= w          # (a run of three close bursts may lose its third for nothing: fine)
            pos = pos[ok]
        x.index_add_(0, (pos[:, None] + tt[None, :]).reshape(-1), burst.repeat(pos.numel()))
    peak = x.abs().max()
    lim = 10 ** (-3.0 / 20.0)
    if float(peak) > lim:
        x = x * (lim / peak)
    return x.contiguous()
