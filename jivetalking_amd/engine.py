"""Engine — one handle per worker / GPU stream over the C ABI (include/jtgpu.h).  numpy in, numpy/dicts out."""
import ctypes as C
import numpy as np
from . import _lib as L


def _st2dict(s):
    out = {}
    for name, _t in s._fields_:
        v = getattr(s, name)
        if isinstance(v, C.Structure):
            out[name] = _st2dict(v)
        elif isinstance(v, C.Array):
            out[name] = list(v)
        else:
            out[name] = v
    return out


def default_filter_params(**over):
    """DefaultFilterConfig() (filters.go:353-532) as numeric jt_filter_params, at the emitted string precision."""
    p = L.FilterParams()
    p.hp_enabled, p.hp_freq, p.hp_q = 1, 80.0, 0.707
    p.lp_enabled, p.lp_freq, p.lp_q = 1, 20500.0, 0.707
    p.nlm_enabled, p.nlm_strength, p.nlm_patch_s, p.nlm_research_s, p.nlm_smooth = 1, 0.00001, 0.0060, 0.0020, 3.0
    p.fft_enabled, p.fft_nr, p.fft_nf, p.fft_custom, p.fft_track_noise = 1, 12.0, -50.0, 0, 0
    p.gate_enabled, p.gate_threshold, p.gate_ratio = 1, 0.010000, 2.0
    p.gate_attack_ms, p.gate_release_ms, p.gate_range, p.gate_knee, p.gate_makeup = 5.00, 200.0, 0.1995, 3.0, 1.0
    p.comp_enabled, p.comp_threshold, p.comp_ratio = 1, 0.125893, 3.0
    p.comp_attack_ms, p.comp_release_ms, p.comp_makeup, p.comp_knee, p.comp_mix = 10.0, 200.0, 1.00, 4.0, 1.00
    p.deess_enabled, p.deess_i, p.deess_m, p.deess_f = 0, 0.0, 0.50, 0.80
    p.out_rate, p.out_frame_samples = 44100, 4096
    for k, v in over.items():
        if k == "fft_band_noise":
            for i, x in enumerate(v):
                p.fft_band_noise[i] = float(x)
        else:
            setattr(p, k, v)
    return p


def _flac_info(info):
    d = {k: getattr(info, k) for k, _ in info._fields_ if k != "md5"}
    d["md5"] = bytes(info.md5).hex()
    return d


class Engine:
    def __init__(self, device=0, ab=None, streams=0, blocking_sync=False, **options):
        """ab=True: the A/B build of the library (superseded kernel generations, tuning knobs; it also imports JT_<KEY> variables at
        jt_open, which is how tools/ switch kernels: JT_USE_AB_LIB=1 JT_NLM_OLD=1 python tools/...).  options: jt_set_option pairs."""
        if ab is None:
            import os
            ab = bool(os.environ.get("JT_USE_AB_LIB"))
        self.lib = L.load(ab)
        self.ab = bool(ab)
        self.h = C.c_void_p()
        rc = self.lib.jt_open_ex(C.c_int(device), C.c_int(int(streams)), C.c_int(1 if blocking_sync else 0), C.byref(self.h))      # (0, 0) = jt_open
        if rc != 0:
            raise L.JtError(rc, "jt_open failed (no MI355X visible?)")
        self._keep = None
        for k, v in options.items():
            self.set_option(k, v)

    def set_option(self, key, value=True):
        """jt_set_option: the switches that used to be JT_* environment variables (include/jtgpu.h lists the keys)."""
        if isinstance(value, bool):
            value = "1" if value else "0"
        self._ck(self.lib.jt_set_option(self.h, str(key).encode(), str(value).encode()))

    def close(self):
        if self.h:
            self.lib.jt_close(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            raise L.JtError(rc, self.lib.jt_last_error(self.h).decode())

    # ---- input
    def upload_pcm(self, pcm, sample_rate, channels=1, channel_mask=0):
        """jt_upload_pcm_layout: channel_mask = the source's layout (0: the default layout of the channel count)."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        frames = pcm.size // channels
        self._ck(self.lib.jt_upload_pcm_layout(self.h, pcm.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(frames),
                                               C.c_int(sample_rate), C.c_int(channels), C.c_uint64(channel_mask)))

    def set_source_format(self, bits_per_sample, is_float=False):
        """The decoder's native sample format of the uploaded PCM (jt_set_source_format): selects the band graphs' arithmetic."""
        self._ck(self.lib.jt_set_source_format(self.h, C.c_int(bits_per_sample), C.c_int(1 if is_float else 0)))

    def cancel(self):
        self.lib.jt_cancel(self.h)

    def reset_cancel(self):
        self.lib.jt_reset_cancel(self.h)

    def begin_job(self):
        """jt_begin_job: clear the cancel flag once; nothing clears it again until end_job (a caller that arms its cancel source first)."""
        self.lib.jt_begin_job(self.h)

    def end_job(self):
        self.lib.jt_end_job(self.h)

    def attach_device_pcm(self, dev_ptr, frames, sample_rate, channels=1, keepalive=None):
        self._keep = keepalive
        self._ck(self.lib.jt_attach_device_pcm(self.h, C.c_void_p(dev_ptr), C.c_int64(frames), C.c_int(sample_rate), C.c_int(channels)))

    def upload_s16(self, pcm, sample_rate):
        pcm = np.ascontiguousarray(pcm, np.int16)
        self._ck(self.lib.jt_upload_s16(self.h, pcm.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int64(pcm.size), C.c_int(sample_rate)))

    def load_audio(self, data):
        """FLAC / WAV file image -> the handle's input (include/jtgpu.h: jt_load_audio).  Returns the metadata dict."""
        buf = np.frombuffer(bytes(data), np.uint8); m = L.AudioMeta()
        self._ck(self.lib.jt_load_audio(self.h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(buf.size), C.byref(m)))
        return _st2dict(m)

    def input_frame_layout(self):
        """jt_input_frame_layout: (frame_samples, variable, n_frames, per-frame lengths or None) of the handle's current input."""
        fs, var, nf = C.c_int(), C.c_int(), C.c_int64()
        self._ck(self.lib.jt_input_frame_layout(self.h, C.byref(fs), C.byref(var), C.byref(nf), None, C.c_int64(0)))
        lens = None
        if var.value:
            lens = np.zeros(nf.value, np.int32)
            self._ck(self.lib.jt_input_frame_layout(self.h, None, None, None, lens.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int64(nf.value)))
        return fs.value, bool(var.value), nf.value, lens

    def op_decode_audio(self, data, want_i32=True):
        """Decoded samples back on the host: (int32 [frames, ch] or None, f32 [frames, ch], meta)."""
        buf = np.frombuffer(bytes(data), np.uint8); m = L.AudioMeta()
        self._ck(self.lib.jt_op_decode_audio(self.h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(buf.size), None, None,
                                             C.c_int64(0), C.byref(m)))
        n = m.frames * m.channels
        i32 = np.empty(n, np.int32) if want_i32 and not m.is_float else None
        f32 = np.empty(n, np.float32)
        self._ck(self.lib.jt_op_decode_audio(self.h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(buf.size),
                                             i32.ctypes.data_as(C.POINTER(C.c_int32)) if i32 is not None else None,
                                             f32.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(n), C.byref(m)))
        sh = (m.frames, m.channels)
        return (i32.reshape(sh) if i32 is not None else None), f32.reshape(sh), _st2dict(m)

    # ---- passes
    def pass1(self, n_frames_input, frame_samples=4096, sample_rate=48000, want_meta=True):
        nfr = (n_frames_input + frame_samples - 1) // frame_samples
        ss = np.zeros(nfr); pk = np.zeros(nfr)
        cap = n_frames_input // (sample_rate // 10) + 2
        meta = (L.FrameMeta * cap)()
        a = L.Analysis()
        self._ck(self.lib.jt_pass1(self.h, C.c_int(frame_samples), C.byref(a),
                                   ss.ctypes.data_as(C.POINTER(C.c_double)), pk.ctypes.data_as(C.POINTER(C.c_double)),
                                   C.c_int64(nfr), meta if want_meta else None, C.c_int64(cap if want_meta else 0)))
        out = _st2dict(a)
        out["frame_sumsq"], out["frame_peak"] = ss, pk
        if want_meta:
            nm = a.n_frames_meta
            out["meta"] = [_st2dict(meta[i]) for i in range(nm)]
        return out

    def band_rms(self, start_s, dur_s, lo, hi):
        lo = np.ascontiguousarray(lo, np.float64); hi = np.ascontiguousarray(hi, np.float64)
        out = np.zeros(lo.size); ok = np.zeros(lo.size, np.int32)
        self._ck(self.lib.jt_band_rms(self.h, C.c_double(start_s), C.c_double(dur_s),
                                      lo.ctypes.data_as(C.POINTER(C.c_double)), hi.ctypes.data_as(C.POINTER(C.c_double)),
                                      C.c_int(lo.size), out.ctypes.data_as(C.POINTER(C.c_double)), ok.ctypes.data_as(C.POINTER(C.c_int))))
        return out, ok

    def pass2(self, params):
        a = L.Analysis()
        self._ck(self.lib.jt_pass2(self.h, C.byref(params), C.byref(a)))
        return _st2dict(a)

    def pass2_prefetch(self, params):
        """Start the head of the Pass-2 chain (biquads + anlmdn) early (jt_pass2_prefetch)."""
        self._ck(self.lib.jt_pass2_prefetch(self.h, C.byref(params)))

    def pass2_prefetch_after_pass1(self, params):
        """Announce the Pass-2 head; the next pass1 queues it behind its own kernels (jt_pass2_prefetch_after_pass1)."""
        self._ck(self.lib.jt_pass2_prefetch_after_pass1(self.h, C.byref(params)))

    def region_measure(self, stage, start_s, dur_s):
        r = L.RegionSample()
        self._ck(self.lib.jt_region_measure(self.h, C.c_int(stage), C.c_double(start_s), C.c_double(dur_s), C.byref(r)))
        return _st2dict(r)

    def region_prefetch(self, stage, starts, durs):
        """Announce the two MeasureOutputRegions ranges before the stage's pass runs (jt_region_prefetch)."""
        self._ck(self.lib.jt_region_prefetch(self.h, C.c_int(stage), (C.c_double * 2)(*starts), (C.c_double * 2)(*durs)))

    def region_measure_pair(self, stage, starts, durs):
        pair = (L.RegionSample * 2)()
        self._ck(self.lib.jt_region_measure_pair(self.h, C.c_int(stage), (C.c_double * 2)(*starts), (C.c_double * 2)(*durs), pair))
        return [_st2dict(pair[0]), _st2dict(pair[1])]

    def pass3(self, limiter=None, target_i=-16.0, target_tp=-1.0, target_lra=20.0):
        s = L.LoudnormStats()
        lim = limiter if limiter is not None else L.LimiterPlan(0, 0.0, 1.0)
        self._ck(self.lib.jt_pass3(self.h, C.byref(lim), C.c_double(target_i), C.c_double(target_tp), C.c_double(target_lra), C.byref(s)))
        return _st2dict(s)

    def pass4(self, limiter, apply):
        a = L.Analysis(); s = L.LoudnormStats()
        lim = limiter if limiter is not None else L.LimiterPlan(0, 0.0, 1.0)
        self._ck(self.lib.jt_pass4(self.h, C.byref(lim), C.byref(apply), C.byref(a), C.byref(s)))
        return _st2dict(a), _st2dict(s)

    def download_s16(self, stage):
        n = C.c_int64()
        self._ck(self.lib.jt_output_len(self.h, C.c_int(stage), C.byref(n)))
        out = np.empty(n.value, np.int16)
        self._ck(self.lib.jt_download_s16(self.h, C.c_int(stage), out.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int64(out.size), C.byref(n)))
        return out

    def download_s16_into(self, stage, out):
        """jt_download_s16 into a caller buffer (e.g. pinned memory); returns the sample count."""
        n = C.c_int64()
        self._ck(self.lib.jt_download_s16(self.h, C.c_int(stage), out.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int64(out.size), C.byref(n)))
        return n.value

    def flac_encode(self, stage, md5=True, return_info=False):
        """The stage output as a finished .flac file image (bytes), encoded on the GPU (include/jtgpu.h: jt_flac_encode)."""
        data = C.POINTER(C.c_uint8)(); n = C.c_int64(); info = L.FlacInfo()
        self._ck(self.lib.jt_flac_encode(self.h, C.c_int(stage), C.c_int(L.JT_FLAC_MD5 if md5 else 0), C.byref(data), C.byref(n),
                                         C.byref(info)))
        out = C.string_at(data, n.value)
        return (out, _flac_info(info)) if return_info else out

    def op_flac_encode(self, pcm, sample_rate, md5=True, return_info=False):
        x = np.ascontiguousarray(pcm, np.int16)
        data = C.POINTER(C.c_uint8)(); n = C.c_int64(); info = L.FlacInfo()
        self._ck(self.lib.jt_op_flac_encode_s16(self.h, x.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int64(x.size), C.c_int(sample_rate),
                                                C.c_int(L.JT_FLAC_MD5 if md5 else 0), C.byref(data), C.byref(n), C.byref(info)))
        out = C.string_at(data, n.value)
        return (out, _flac_info(info)) if return_info else out

    def timers(self):
        t = L.Timers()
        self._ck(self.lib.jt_get_timers(self.h, C.byref(t)))
        return _st2dict(t)

    # ---- operator-level (parity tests)
    def op_biquad(self, x, sr, hp=(1, 80.0, 0.707), lp=(1, 20500.0, 0.707)):
        x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x)
        self._ck(self.lib.jt_op_biquad_f32(self.h, x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                                           C.c_int64(x.size), C.c_int(sr), C.c_int(hp[0]), C.c_double(hp[1]), C.c_double(hp[2]),
                                           C.c_int(lp[0]), C.c_double(lp[1]), C.c_double(lp[2])))
        return y

    def op_anlmdn(self, x, sr, s=0.00001, p=0.006, r=0.002, m=3.0):
        x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x)
        self._ck(self.lib.jt_op_anlmdn_f32(self.h, x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                                           C.c_int64(x.size), C.c_int(sr), C.c_double(s), C.c_double(p), C.c_double(r), C.c_double(m)))
        return y

    def op_afftdn(self, x, sr, nr=12.0, nf=-50.0, band_noise=None, track=False, return_floor=False):
        x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x)
        bn = None
        if band_noise is not None:
            arr = np.ascontiguousarray(band_noise, np.float64)
            bn = arr.ctypes.data_as(C.POINTER(C.c_double))
        fl = C.c_double(0.0)
        self._ck(self.lib.jt_op_afftdn_tn_f32(self.h, x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                                              C.c_int64(x.size), C.c_int(sr), C.c_double(nr), C.c_double(nf), bn,
                                              C.c_int(1 if track else 0), C.byref(fl)))
        return (y, fl.value) if return_floor else y

    def op_dynamics(self, x, sr, params):
        x = np.ascontiguousarray(x, np.float32); y = np.empty_like(x)
        self._ck(self.lib.jt_op_dynamics(self.h, x.ctypes.data_as(C.POINTER(C.c_float)), y.ctypes.data_as(C.POINTER(C.c_float)),
                                         C.c_int64(x.size), C.c_int(sr), C.byref(params)))
        return y

    def op_alimiter(self, x, sr, limit, attack=5.0, release=100.0):
        x = np.ascontiguousarray(x, np.float64); y = np.empty_like(x)
        self._ck(self.lib.jt_op_alimiter_f64(self.h, x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)),
                                             C.c_int64(x.size), C.c_int(sr), C.c_double(limit), C.c_double(attack), C.c_double(release)))
        return y

    def op_adeclick(self, x, sr, t=1.7, w=55.0, o=50.0, method="s", return_count=False):
        x = np.ascontiguousarray(x, np.float64); y = np.empty_like(x); cnt = C.c_int64()
        self._ck(self.lib.jt_op_adeclick_f64(self.h, x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)),
                                             C.c_int64(x.size), C.c_int(sr), C.c_double(t), C.c_double(w), C.c_double(o),
                                             C.c_int(1 if method == "s" else 0), C.byref(cnt)))
        return (y, cnt.value) if return_count else y

    def op_resample_s16(self, x, in_rate, out_rate):
        x = np.ascontiguousarray(x, np.float32)
        cap = int(np.ceil(x.size * out_rate / in_rate)) + 8
        y = np.empty(cap, np.int16); n = C.c_int64()
        self._ck(self.lib.jt_op_resample_f32_to_s16(self.h, x.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(x.size), C.c_int(in_rate),
                                                    C.c_int(out_rate), y.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int64(cap), C.byref(n)))
        return y[:n.value].copy()

    def op_ebur128(self, x, sr, dualmono=True):
        x = np.ascontiguousarray(x, np.float32)
        cap = x.size // (sr // 10) + 2
        m = np.zeros(cap); s = np.zeros(cap); tp = np.zeros(cap); sp = np.zeros(cap)
        r = L.R128(); nb = C.c_int64()
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        self._ck(self.lib.jt_op_ebur128(self.h, x.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(x.size), C.c_int(sr), C.c_int(int(dualmono)),
                                        C.byref(r), dp(m), dp(s), dp(tp), dp(sp), C.c_int64(cap), C.byref(nb)))
        d = _st2dict(r); k = nb.value
        d.update(M=m[:k].copy(), S=s[:k].copy(), TP=tp[:k].copy(), SP=sp[:k].copy())
        return d

    def op_astats(self, x, sr):
        x = np.ascontiguousarray(x, np.float32)
        a = L.Astats()
        self._ck(self.lib.jt_op_astats(self.h, x.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(x.size), C.c_int(sr), C.byref(a)))
        return _st2dict(a)

    def op_aspectralstats(self, x, sr):
        x = np.ascontiguousarray(x, np.float32)
        cap = (x.size + 1023) // 1024 + 1
        hops = (L.Spectral * cap)(); n = C.c_int64()
        self._ck(self.lib.jt_op_aspectralstats(self.h, x.ctypes.data_as(C.POINTER(C.c_float)), C.c_int64(x.size), C.c_int(sr), hops, C.c_int64(cap), C.byref(n)))
        out = np.zeros((n.value, 13))
        for i in range(n.value):
            for j, k in enumerate(L.SPECTRAL_KEYS):
                out[i, j] = getattr(hops[i], k)
        return out

    def op_loudnorm_dynamic(self, x192, target_i=-16.0, target_lra=20.0, target_tp=-1.0, measured=None, offset=0.0):
        """af_loudnorm's dynamic mode on a mono stream already at 192 kHz; measured = (I, LRA, TP, thresh) or None (first pass)."""
        x = np.ascontiguousarray(x192, np.float64); y = np.empty_like(x)
        mi, ml, mt, mth = measured if measured is not None else (0.0, 0.0, 99.0, -70.0)
        ap = L.LoudnormApply(target_i, target_tp, target_lra, mi, mt, ml, mth, offset, 0, 0.0, 0.0, 0.0, 0, 1.0)
        s = L.LoudnormStats()
        self._ck(self.lib.jt_op_loudnorm_dynamic_f64(self.h, x.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(x.size), C.byref(ap),
                                                     y.ctypes.data_as(C.POINTER(C.c_double)), C.byref(s)))
        return y, _st2dict(s)

    def op_loudnorm_measure_s16(self, x, sr, limiter=None):
        x = np.ascontiguousarray(x, np.int16)
        s = L.LoudnormStats()
        lim = limiter if limiter is not None else L.LimiterPlan(0, 0.0, 1.0)
        self._ck(self.lib.jt_op_loudnorm_measure_s16(self.h, x.ctypes.data_as(C.POINTER(C.c_int16)), C.c_int64(x.size), C.c_int(sr), C.byref(lim), C.byref(s)))
        return _st2dict(s)
