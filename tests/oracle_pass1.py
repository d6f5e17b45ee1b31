"""Pass 1 of the reference's analysis graph composed from the CPU ORACLE (test infrastructure): what jt_pass1 hands the host logic -
the whole-file jt_analysis, one jt_frame_meta per 100 ms ebur128 output frame and the per-decoder-frame sum(x^2) / peak - built from
orc.astats / orc.aspectralstats / orc.ebur128 the way the metadata travels through the filter graph (the GPU side's
assemble_analysis in jt_api.cpp), and the host decision chain run on it: intervals -> VAD / elections -> band measurements (oracle
band graph) -> AdaptConfig -> chain string.  The decision-chain parity test compares that with what jt_process_audio decides from
the GPU's own Pass 1."""
import ctypes as C

import numpy as np

from jivetalking_amd import hostlogic as H
from jivetalking_amd import _lib as L

ASTATS_MAP = {"peak_level": "peak_level_db", "rms_level": "rms_level_db", "rms_peak": "rms_peak_db", "rms_trough": "rms_trough_db",
              "noise_floor": "noise_floor_db", "bit_depth": "abs_peak_count"}


def oracle_pass1(orc, x, sr, frame_samples=4096, frame_lens=None, raw=None):
    """`frame_lens`: per-frame lengths of a stream whose decoder frames differ (a variable-blocksize FLAC); `raw`: the interleaved
    source of a stereo file (channels = raw.size // x.size) -- the per-frame sum(x^2) / peak are taken on the RAW samples of all
    channels (analyser_metrics.go:273-358), everything else on the down-mix x."""
    x = np.ascontiguousarray(x, np.float32)
    n = x.size
    blk = sr // 10
    nfull = n // blk
    nframes = nfull + (1 if n % blk else 0)
    e = orc.ebur128(x.astype(np.float64), sr, True, True)
    hops = orc.aspectralstats(x, sr)                                   # one record per 1024-sample hop
    nhops = (n + 1023) // 1024
    a = L.Analysis()
    st = orc.astats(x.astype(np.float64), sr)
    for k in L.ASTATS_KEYS:
        setattr(a.astats, k, st[ASTATS_MAP.get(k, k)])
    meta = (L.FrameMeta * (nframes + 2))()
    mean = np.zeros(13)
    for k in range(nframes):
        h = min((k * blk) // 1024, nhops - 1)                          # the hop containing the frame's first sample
        rec = hops[h]
        mean += rec
        for j, key in enumerate(L.SPECTRAL_KEYS):
            setattr(meta[k].spectral, key, float(rec[j]))
        if k < nfull:
            meta[k].momentary, meta[k].shortterm = float(e["M"][k]), float(e["S"][k])
            meta[k].true_peak, meta[k].sample_peak = float(e["TP"][k]), float(e["SP"][k])
        else:
            meta[k].momentary = meta[k].shortterm = meta[k].true_peak = meta[k].sample_peak = float("nan")
    if nframes:
        mean /= nframes
    for j, key in enumerate(L.SPECTRAL_KEYS):
        setattr(a.spectral_mean, key, float(mean[j]))
    a.n_frames_meta = nframes
    r = a.r128
    r.integrated, r.lra, r.lra_low, r.lra_high = e["integrated"], e["lra"], e["lra_low"], e["lra_high"]
    r.momentary = float(e["M"][nfull - 1]) if nfull else float("nan")
    r.shortterm = float(e["S"][nfull - 1]) if nfull else float("nan")
    r.true_peak = float(e["TP"][nfull - 1]) if nfull else 0.0
    r.sample_peak = float(e["SP"][nfull - 1]) if nfull else 0.0
    r.target_threshold = e["target_threshold"]
    src = np.asarray(x if raw is None else raw, np.float64)
    ch = src.size // n
    if frame_lens is not None:
        off = np.concatenate([[0], np.cumsum(np.asarray(frame_lens, np.int64))]) * ch
        assert off[-1] == src.size
        fss = np.ascontiguousarray(np.add.reduceat(src * src, off[:-1]))
        fpk = np.ascontiguousarray(np.maximum.reduceat(np.abs(src), off[:-1]))
        return a, meta, nframes, fss, fpk
    nfr = (n + frame_samples - 1) // frame_samples
    pad = np.zeros(nfr * frame_samples * ch, np.float64); pad[:src.size] = src
    fr = pad.reshape(nfr, frame_samples * ch)
    fss = np.ascontiguousarray((fr * fr).sum(axis=1)); fpk = np.ascontiguousarray(np.abs(fr).max(axis=1))
    return a, meta, nframes, fss, fpk


def decide(orc, x, sr, base=None, frame_samples=4096, pass1=None, frame_lens=None, raw=None, band_mode=0, band_x=None, want_intervals=False):
    """The reference's Pass-1 decision chain on oracle measurements.  Returns (measurements, effective config, Pass-2 chain string).
    `pass1` = a precomputed oracle_pass1(orc, x, sr, frame_samples) (the fuzz tests measure many files side by side on the host's cores)."""
    l = H.lib()
    base = base or H.default_config()
    a, meta, nframes, fss, fpk = pass1 if pass1 is not None else oracle_pass1(orc, x, sr, frame_samples, frame_lens, raw)
    n = x.size
    ch = 1 if raw is None else raw.size // n
    iv = (H.Interval * (n // (sr // 5) + 16))()
    fl = None if frame_lens is None else np.ascontiguousarray(frame_lens, np.int32)
    niv = l.jt_host_build_intervals_v(C.c_int(sr), C.c_int64(n), C.c_int(frame_samples), None if fl is None else fl.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(ch),
                                      fss.ctypes.data_as(C.POINTER(C.c_double)), fpk.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(fss.size),
                                      meta, C.c_int64(nframes), C.c_int(1), iv, C.c_int64(len(iv)))
    m = H.Measurements()
    rc = l.jt_host_detect(C.byref(a), iv, C.c_int64(min(niv, len(iv))), C.c_double(n / sr), C.c_double(base.target_i), C.c_int(1), C.byref(m))
    assert rc == 0, rc
    q = lambda v: float("%f" % v)

    def band(start_ns, dur_ns, lo, hi):
        bx = x if band_x is None else band_x          # (an integer stereo source: the band graphs' own integer down-mix)
        s0 = int(round(start_ns * 1e-9 * sr)); seg = bx[s0:s0 + int(round(dur_ns * 1e-9 * sr))]
        return orc.band_rms_db_fmt(np.ascontiguousarray(seg, np.float32), sr, lo, hi, band_mode)

    if m.has_speech_profile and m.speech_profile.region.duration_ns > 0:
        rg = m.speech_profile.region
        b, s = band(rg.start_ns, rg.duration_ns, 1000.0, 3000.0), band(rg.start_ns, rg.duration_ns, 6000.0, 9000.0)
        m.speech_profile.body_band_rms, m.speech_profile.sib_band_rms = q(b), q(s)
        m.speech_profile.bands_measured = 1 if np.isfinite(b) and np.isfinite(s) else 0
    if m.has_noise_profile and m.noise_profile.duration_ns > 0:
        fin = 0
        for i in range(15):
            lo, hi = C.c_double(), C.c_double()
            l.jt_host_afftdn_band_edges(C.c_int(i), C.byref(lo), C.byref(hi))
            # a corner at or above Nyquist (the 24 kHz band of a 48 kHz file): astats reports a non-finite RMS "as a matter of
            # course" (analyser_noise_bands.go:97-103) and the value is stored as it is
            below = hi.value < sr / 2 and lo.value < sr / 2
            v = band(m.noise_profile.start_ns, m.noise_profile.duration_ns, lo.value, hi.value) if below else float("nan")
            m.noise_profile.band_noise[i] = q(v) if np.isfinite(v) else v
            fin += 1 if np.isfinite(v) else 0
        m.noise_profile.band_noise_n = 15
        m.noise_profile.bands_measured = 1 if fin >= 10 else 0
    l.jt_host_finish_measurements(C.byref(m))
    eff, diag = H.adapt(base, m)
    if want_intervals:
        return m, eff, H.filter_spec(eff, 2), [iv[i] for i in range(min(niv, len(iv)))]
    return m, eff, H.filter_spec(eff, 2)
