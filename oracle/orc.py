"""ctypes binding for the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package (jivetalking_amd/).  See jt_oracle.h.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
        if not os.path.exists(path) or any(os.path.getmtime(f) > os.path.getmtime(path) for f in srcs):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


f32p, f64p, i16p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int16)


def biquad_coeffs(kind, freq, q, sr):
    b = (C.c_double * 3)()
    a = (C.c_double * 3)()
    lib().orc_biquad_coeffs(C.c_int(kind), C.c_double(freq), C.c_double(q), C.c_int(sr), b, a)
    return np.array(b[:]), np.array(a[:])


def biquad_f32(x, kind, freq, sr, q=0.707):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    b = (C.c_double * 3)()
    a = (C.c_double * 3)()
    lib().orc_biquad_coeffs(C.c_int(kind), C.c_double(freq), C.c_double(q), C.c_int(sr), b, a)
    lib().orc_biquad_tdii_f32(_p(x, C.c_float), _p(y, C.c_float), C.c_int64(x.size), b, a)
    return y


def band_rms_db_fmt(x, sr, lo_hz, hi_hz, mode):
    """The band graph in the sample format libavfilter negotiates for the source: mode 0 fltp, 1 s16p, 2 s32p."""
    x = np.ascontiguousarray(x, np.float32)
    f = lib().orc_band_rms_db_fmt
    f.restype = C.c_double
    return float(f(_p(x, C.c_float), C.c_int64(x.size), C.c_int(sr), C.c_double(lo_hz), C.c_double(hi_hz), C.c_int(mode)))


def downmix_stereo(interleaved, mode=0):
    """aformat=channel_layouts=mono of a stereo source (libswresample rematrix): mode 0 = float 1/sqrt2 (Pass 1 / Pass 2),
    1 = s16 integer matrix, 2 = s32 via float 0.5 (the band graphs of integer sources)."""
    x = np.ascontiguousarray(interleaved, np.float32)
    out = np.empty(x.size // 2, np.float32)
    lib().orc_downmix_stereo(_p(x, C.c_float), C.c_int64(x.size // 2), C.c_int(mode), _p(out, C.c_float))
    return out


def downmix_layout(interleaved, channels, mask=0, mode=0):
    """aformat=channel_layouts=mono of any layout up to 8 channels (swr_build_matrix2's default matrix; mask 0 = the default layout of
    the channel count); modes as downmix_stereo.  Raises ValueError for a layout the restatement does not cover."""
    x = np.ascontiguousarray(interleaved, np.float32)
    out = np.empty(x.size // channels, np.float32)
    rc = lib().orc_downmix_layout(_p(x, C.c_float), C.c_int64(x.size // channels), C.c_int(channels), C.c_uint64(mask), C.c_int(mode), _p(out, C.c_float))
    if rc != 0:
        raise ValueError("layout not covered")
    return out


def downmix_coeffs(channels, mask=0, normalise=False):
    c = (C.c_double * 8)()
    if lib().orc_downmix_coeffs(C.c_int(channels), C.c_uint64(mask), C.c_int(int(normalise)), c) != 0:
        raise ValueError("layout not covered")
    return np.array(c[:channels])


def band_rms_db(x, sr, lo_hz, hi_hz):
    """Band-limited Overall RMS of a region in dB (analyser_bands.go:33: highpass, lowpass, astats)."""
    x = np.ascontiguousarray(x, np.float32)
    f = lib().orc_band_rms_db
    f.restype = C.c_double
    return float(f(_p(x, C.c_float), C.c_int64(x.size), C.c_int(sr), C.c_double(lo_hz), C.c_double(hi_hz)))


def biquad_f64(x, kind, freq, sr, q=0.707):
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    b = (C.c_double * 3)()
    a = (C.c_double * 3)()
    lib().orc_biquad_coeffs(C.c_int(kind), C.c_double(freq), C.c_double(q), C.c_int(sr), b, a)
    lib().orc_biquad_tdii_f64(_p(x, C.c_double), _p(y, C.c_double), C.c_int64(x.size), b, a)
    return y


def anlmdn(x, sr, s=0.00001, p=0.006, r=0.002, m=3.0):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x)
    lib().orc_anlmdn_f32(_p(x, C.c_float), _p(y, C.c_float), C.c_int64(x.size), C.c_int(sr),
                         C.c_double(s), C.c_double(p), C.c_double(r), C.c_double(m))
    return y


def afftdn(x, sr, nr=12.0, nf=-50.0, band_noise=None, track=False, return_floor=False):
    """afftdn=nr:nf[:nt=custom:bn=...][:tn=1].  return_floor: also the noise floor (dB) after every frame."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros_like(x)
    bn = None
    if band_noise is not None:
        bn_arr = np.ascontiguousarray(band_noise, np.float64)
        assert bn_arr.size == 15
        bn = _p(bn_arr, C.c_double)
    adv = sr // 80
    nfr = (x.size + adv - 1) // adv + 2
    fl = np.zeros(nfr, np.float64)
    lib().orc_afftdn_tn_f32(_p(x, C.c_float), _p(y, C.c_float), C.c_int64(x.size), C.c_int(sr),
                            C.c_double(nr), C.c_double(nf), bn, C.c_int(1 if track else 0), _p(fl, C.c_double), C.c_int64(nfr))
    return (y, fl) if return_floor else y


class GateParams(C.Structure):
    _fields_ = [("threshold", C.c_double), ("ratio", C.c_double), ("attack_ms", C.c_double),
                ("release_ms", C.c_double), ("range", C.c_double), ("knee", C.c_double),
                ("makeup", C.c_double), ("detection_rms", C.c_int)]


class CompParams(C.Structure):
    _fields_ = [("threshold", C.c_double), ("ratio", C.c_double), ("attack_ms", C.c_double),
                ("release_ms", C.c_double), ("makeup", C.c_double), ("knee", C.c_double),
                ("mix", C.c_double), ("detection_rms", C.c_int)]


def agate(x, sr, threshold=0.01, ratio=2.0, attack=5.0, release=200.0, range_=0.1995, knee=3.0, makeup=1.0):
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    p = GateParams(threshold, ratio, attack, release, range_, knee, makeup, 1)
    lib().orc_agate_f64(_p(x, C.c_double), _p(y, C.c_double), C.c_int64(x.size), C.c_int(sr), C.byref(p))
    return y


def acompressor(x, sr, threshold=0.125893, ratio=3.0, attack=10.0, release=200.0, makeup=1.0, knee=4.0, mix=1.0):
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    p = CompParams(threshold, ratio, attack, release, makeup, knee, mix, 1)
    lib().orc_acompressor_f64(_p(x, C.c_double), _p(y, C.c_double), C.c_int64(x.size), C.c_int(sr), C.byref(p))
    return y


def deesser(x, sr, i=0.5, m=0.5, f=0.8):
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    lib().orc_deesser_f64(_p(x, C.c_double), _p(y, C.c_double), C.c_int64(x.size), C.c_int(sr),
                          C.c_double(i), C.c_double(m), C.c_double(f))
    return y


def alimiter(x, sr, limit, attack=5.0, release=100.0, asc_level=0.8):
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    lib().orc_alimiter_f64(_p(x, C.c_double), _p(y, C.c_double), C.c_int64(x.size), C.c_int(sr),
                           C.c_double(limit), C.c_double(attack), C.c_double(release), C.c_double(asc_level))
    return y


def adeclick(x, sr, t=1.7, w=55.0, o=50.0, a=2.0, b=2.0, method="s", return_count=False):
    """af_adeclick.c; method 'a' = overlap-add, 's' = overlap-save."""
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    cnt = C.c_int64(0)
    l = lib(); l.orc_adeclick_f64.restype = C.c_int
    rc = l.orc_adeclick_f64(_p(x, C.c_double), _p(y, C.c_double), C.c_int64(x.size), C.c_int(sr), C.c_double(t), C.c_double(w),
                            C.c_double(o), C.c_double(a), C.c_double(b), C.c_int(1 if method == "s" else 0), C.byref(cnt))
    if rc != 0:
        raise RuntimeError("adeclick: singular interpolation matrix")
    return (y, cnt.value) if return_count else y


def swr_f64(x, in_rate, out_rate, flush=True):
    x = np.ascontiguousarray(x, np.float64)
    cap = int(np.ceil(x.size * out_rate / in_rate)) + 4
    y = np.empty(cap, np.float64)
    fn = lib().orc_swr_resample_f64
    fn.restype = C.c_int64
    m = fn(_p(x, C.c_double), C.c_int64(x.size), C.c_int(in_rate), C.c_int(out_rate),
           _p(y, C.c_double), C.c_int64(cap), C.c_int(1 if flush else 0))
    return y[:m].copy()


def swr_f32(x, in_rate, out_rate, flush=True):
    x = np.ascontiguousarray(x, np.float32)
    cap = int(np.ceil(x.size * out_rate / in_rate)) + 4
    y = np.empty(cap, np.float32)
    fn = lib().orc_swr_resample_f32
    fn.restype = C.c_int64
    m = fn(_p(x, C.c_float), C.c_int64(x.size), C.c_int(in_rate), C.c_int(out_rate),
           _p(y, C.c_float), C.c_int64(cap), C.c_int(1 if flush else 0))
    return y[:m].copy()


def f64_to_s16(x):
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty(x.size, np.int16)
    lib().orc_f64_to_s16(_p(x, C.c_double), _p(y, C.c_int16), C.c_int64(x.size))
    return y


class Ebur128Out(C.Structure):
    _fields_ = [("integrated", C.c_double), ("lra", C.c_double), ("lra_low", C.c_double),
                ("lra_high", C.c_double), ("momentary_last", C.c_double), ("shortterm_last", C.c_double),
                ("sample_peak", C.c_double), ("true_peak", C.c_double), ("target_threshold", C.c_double),
                ("nblocks", C.c_int64)]


def ebur128(x, sr, dualmono=True, true_peak=True):
    x = np.ascontiguousarray(x, np.float64)
    cap = x.size // (sr // 10) + 2
    m = np.empty(cap); s = np.empty(cap); tp = np.empty(cap); sp = np.empty(cap)
    o = Ebur128Out()
    lib().orc_ebur128_mono(_p(x, C.c_double), C.c_int64(x.size), C.c_int(sr), C.c_int(int(dualmono)),
                           C.c_int(int(true_peak)), C.byref(o), _p(m, C.c_double), _p(s, C.c_double),
                           _p(tp, C.c_double), _p(sp, C.c_double), C.c_int64(cap))
    nb = o.nblocks
    d = {k: getattr(o, k) for k, _ in Ebur128Out._fields_}
    d.update(M=m[:nb].copy(), S=s[:nb].copy(), TP=tp[:nb].copy(), SP=sp[:nb].copy())
    return d


class LoudnormIn(C.Structure):
    _fields_ = [("input_i", C.c_double), ("input_tp", C.c_double), ("input_lra", C.c_double),
                ("input_thresh", C.c_double)]


class LoudnormParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("target_i", "target_lra", "target_tp", "measured_i", "measured_lra", "measured_tp", "measured_thresh", "offset")] + \
               [("linear", C.c_int), ("dual_mono", C.c_int)]


class LoudnormStats(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("input_i", "input_tp", "input_lra", "input_thresh", "output_i", "output_tp", "output_lra", "output_thresh",
                                          "target_offset")] + [("dynamic", C.c_int)]


def loudnorm_dynamic(x192, target_i=-16.0, target_lra=20.0, target_tp=-1.0, measured=None, offset=0.0, linear=True, dual_mono=True, rate=192000):
    """af_loudnorm on a mono stream already at 192 kHz.  measured = (I, LRA, TP, thresh) or None (the filter's defaults: first pass).
    Returns (output stream, stats dict)."""
    x = np.ascontiguousarray(x192, np.float64)
    y = np.zeros_like(x)
    mi, ml, mt, mth = measured if measured is not None else (0.0, 0.0, 99.0, -70.0)
    p = LoudnormParams(target_i, target_lra, target_tp, mi, ml, mt, mth, offset, int(linear), int(dual_mono))
    st = LoudnormStats()
    lib().orc_loudnorm_dynamic_mono.restype = C.c_int64
    n = lib().orc_loudnorm_dynamic_mono(_p(x, C.c_double), C.c_int64(x.size), C.c_int(rate), C.byref(p), _p(y, C.c_double), C.byref(st))
    assert n == x.size, (n, x.size)
    return y, {k: getattr(st, k) for k, _ in LoudnormStats._fields_}


def loudnorm_measure(x, sr, dual_mono=True):
    x = np.ascontiguousarray(x, np.float64)
    o = LoudnormIn()
    lib().orc_loudnorm_measure_mono(_p(x, C.c_double), C.c_int64(x.size), C.c_int(sr), C.c_int(int(dual_mono)), C.byref(o))
    return {k: getattr(o, k) for k, _ in LoudnormIn._fields_}


ASTATS_FIELDS = ["dc_offset", "min_level", "max_level", "min_difference", "max_difference", "mean_difference",
                 "rms_difference", "peak_level_db", "rms_level_db", "rms_peak_db", "rms_trough_db", "crest_factor",
                 "flat_factor", "peak_count", "noise_floor_db", "noise_floor_count", "entropy", "dynamic_range",
                 "zero_crossings", "zero_crossings_rate", "number_of_samples", "abs_peak_count"]


class AstatsOut(C.Structure):
    _fields_ = [(k, C.c_double) for k in ASTATS_FIELDS]


def astats(x, sr):
    x = np.ascontiguousarray(x, np.float64)
    o = AstatsOut()
    lib().orc_astats_mono(_p(x, C.c_double), C.c_int64(x.size), C.c_int(sr), C.byref(o))
    return {k: getattr(o, k) for k in ASTATS_FIELDS}


SPECTRAL_KEYS = ["mean", "variance", "centroid", "spread", "skewness", "kurtosis", "entropy",
                 "flatness", "crest", "flux", "slope", "decrease", "rolloff"]


def aspectralstats(x, sr, win_size=2048):
    x = np.ascontiguousarray(x, np.float32)
    hop = win_size // 2
    cap = (x.size + hop - 1) // hop + 1
    st = np.zeros((cap, 13), np.float64)
    fn = lib().orc_aspectralstats_mono
    fn.restype = C.c_int64
    nh = fn(_p(x, C.c_float), C.c_int64(x.size), C.c_int(sr), C.c_int(win_size), _p(st, C.c_double), C.c_int64(cap))
    return st[:nh].copy()


# ---------------------------------------------------------------- FLAC (RFC 9639), orc_flac.c
class FlacInfo(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("sample_rate", "channels", "bps", "min_blocksize", "max_blocksize", "min_framesize",
                                       "max_framesize")] + \
               [(k, C.c_int64) for k in ("total_samples", "decoded_samples", "frames", "audio_offset")] + \
               [(k, C.c_int) for k in ("metadata_blocks", "variable_blocksize", "crc8_errors", "crc16_errors",
                                       "obs_min_framesize", "obs_max_framesize", "obs_max_blocksize", "last_blocksize")] + \
               [("md5_stored", C.c_uint8 * 16), ("md5_decoded", C.c_uint8 * 16)]


def flac_decode(data, want_pcm=True):
    """Sequential RFC 9639 decode with CRC-8/CRC-16 checks.  Returns (rc, pcm[int32, frames x channels] or None, info);
    rc == 0 means every frame parsed and every CRC matched.  info.md5_decoded is the MD5 of the decoded PCM."""
    buf = np.frombuffer(bytes(data), np.uint8)
    info = FlacInfo()
    fn = lib().orc_flac_decode
    fn.restype = C.c_int
    rc = fn(_p(buf, C.c_uint8), C.c_int64(buf.size), None, C.c_int64(0), C.byref(info))
    pcm = None
    if want_pcm and info.channels > 0 and info.decoded_samples > 0:
        pcm = np.zeros((info.decoded_samples, info.channels), np.int32)
        rc = fn(_p(buf, C.c_uint8), C.c_int64(buf.size), _p(pcm, C.c_int32), C.c_int64(info.decoded_samples), C.byref(info))
    return rc, pcm, info


def flac_encode(pcm, sample_rate, bps=16, blocksize=4096, mode=2, lpc_order=8):
    """Format-coverage encoder for decoder tests (see orc_flac.c for the mode bits).  pcm: int array, frames x channels."""
    a = np.ascontiguousarray(pcm, np.int32)
    if a.ndim == 1:
        a = a[:, None]
    n, ch = a.shape
    fn = lib().orc_flac_encode
    fn.restype = C.c_int64
    args = (_p(a, C.c_int32), C.c_int64(n), C.c_int(ch), C.c_int(bps), C.c_int(sample_rate), C.c_int(blocksize), C.c_int(mode),
            C.c_int(lpc_order))
    size = fn(*args, None, C.c_int64(0))
    out = np.zeros(size, np.uint8)
    fn(*args, _p(out, C.c_uint8), C.c_int64(size))
    return out.tobytes()
