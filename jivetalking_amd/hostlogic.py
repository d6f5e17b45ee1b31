"""ctypes binding of include/jt_host.h — the C++ host mirror of the reference's per-file control logic
(internal/processor: VAD, AdaptConfig, filter-spec strings, normalisation planning, ProcessAudio)."""
import ctypes as C
import numpy as np
from . import _lib as L

JT_MAX_REGIONS = 512


class Interval(C.Structure):
    _fields_ = [("timestamp_ns", C.c_int64), ("rms_level", C.c_double), ("peak_level", C.c_double),
                ("spectral", L.Spectral), ("spectral_found", C.c_int),
                ("momentary_lufs", C.c_double), ("shortterm_lufs", C.c_double),
                ("true_peak", C.c_double), ("sample_peak", C.c_double)]


class Region(C.Structure):
    _fields_ = [("start_ns", C.c_int64), ("end_ns", C.c_int64), ("duration_ns", C.c_int64)]


class RegionMetrics(C.Structure):
    _fields_ = [("rms_level", C.c_double), ("peak_level", C.c_double), ("crest_factor", C.c_double),
                ("spectral", L.Spectral), ("momentary_lufs", C.c_double), ("shortterm_lufs", C.c_double),
                ("true_peak", C.c_double), ("sample_peak", C.c_double)]


class NoiseProfile(C.Structure):
    _fields_ = [("start_ns", C.c_int64), ("duration_ns", C.c_int64), ("measured_noise_floor", C.c_double),
                ("peak_level", C.c_double), ("crest_factor", C.c_double), ("entropy", C.c_double),
                ("spectral", L.Spectral), ("band_noise", C.c_double * 15), ("band_noise_n", C.c_int),
                ("bands_measured", C.c_int), ("warning", C.c_int)]


class SpeechCandidate(C.Structure):
    _fields_ = [("region", Region), ("sample", RegionMetrics), ("voicing_density", C.c_double),
                ("body_band_rms", C.c_double), ("sib_band_rms", C.c_double), ("bands_measured", C.c_int),
                ("score", C.c_double), ("original_start_ns", C.c_int64), ("original_duration_ns", C.c_int64),
                ("was_refined", C.c_int)]


class Measurements(C.Structure):
    _fields_ = [
        ("input_i", C.c_double), ("input_tp", C.c_double), ("input_lra", C.c_double), ("input_thresh", C.c_double),
        ("target_offset", C.c_double), ("momentary", C.c_double), ("shortterm", C.c_double), ("sample_peak", C.c_double),
        ("dynamics", L.Astats), ("floor_astats", C.c_double), ("spectral", L.Spectral),
        ("floor", C.c_double), ("floor_source", C.c_int), ("floor_prescan", C.c_double),
        ("room_tone_detect_level", C.c_double), ("voice_activated", C.c_int), ("floored_fraction", C.c_double),
        ("reduction_headroom", C.c_double),
        ("n_speech_regions", C.c_int), ("speech_regions", Region * JT_MAX_REGIONS),
        ("n_candidates", C.c_int), ("candidates", SpeechCandidate * JT_MAX_REGIONS),
        ("has_speech_profile", C.c_int), ("speech_profile", SpeechCandidate),
        ("has_noise_profile", C.c_int), ("noise_profile", NoiseProfile),
        ("has_room_tone_sample", C.c_int), ("room_tone_sample", RegionMetrics),
        ("voiced_low_percentile", C.c_double), ("noise_high_percentile", C.c_double), ("gate_separation_db", C.c_double),
        ("duration_s", C.c_double), ("vad_split", C.c_double), ("vad_margin", C.c_double), ("vad_gap_tol", C.c_int)]


class BiquadCfg(C.Structure):
    _fields_ = [("enabled", C.c_int), ("frequency", C.c_double), ("poles", C.c_int), ("width", C.c_double),
                ("mix", C.c_double), ("transform_tdii", C.c_int)]


class HostConfig(C.Structure):
    _fields_ = [
        ("downmix_enabled", C.c_int), ("analysis_enabled", C.c_int),
        ("resample_enabled", C.c_int), ("resample_rate", C.c_int), ("resample_frame", C.c_int),
        ("rumble_hp", BiquadCfg), ("bandlimit_lp", BiquadCfg),
        ("nr_enabled", C.c_int), ("nr_strength", C.c_double), ("nr_patch_s", C.c_double), ("nr_research_s", C.c_double), ("nr_smooth", C.c_double),
        ("afftdn_enabled", C.c_int), ("afftdn_nr", C.c_double), ("afftdn_custom", C.c_int), ("afftdn_track_noise", C.c_int),
        ("afftdn_noise_floor", C.c_double), ("afftdn_band_noise", C.c_char * 256),
        ("gate_enabled", C.c_int), ("gate_threshold", C.c_double), ("gate_ratio", C.c_double), ("gate_attack", C.c_double),
        ("gate_release", C.c_double), ("gate_range", C.c_double), ("gate_knee", C.c_double), ("gate_makeup", C.c_double), ("gate_detection_set", C.c_int),
        ("comp_enabled", C.c_int), ("comp_threshold_db", C.c_double), ("comp_ratio", C.c_double), ("comp_attack", C.c_double),
        ("comp_release", C.c_double), ("comp_makeup_db", C.c_double), ("comp_knee", C.c_double), ("comp_mix", C.c_double),
        ("deess_enabled", C.c_int), ("deess_intensity", C.c_double), ("deess_amount", C.c_double), ("deess_frequency", C.c_double),
        ("adeclick_enabled", C.c_int), ("adeclick_threshold", C.c_double), ("adeclick_window", C.c_double), ("adeclick_overlap", C.c_double), ("adeclick_method_s", C.c_int),
        ("loudnorm_enabled", C.c_int), ("target_i", C.c_double), ("target_tp", C.c_double), ("target_lra", C.c_double),
        ("dual_mono", C.c_int), ("linear", C.c_int)]


class AdaptiveDiag(C.Structure):
    _fields_ = [("gate_quiet_speech_estimate", C.c_double), ("gate_separation", C.c_double), ("gate_speech_headroom", C.c_double),
                ("gate_threshold_unclamped", C.c_double), ("gate_depth_db", C.c_double), ("gate_narrow_gap", C.c_int),
                ("afftdn_enabled", C.c_int), ("afftdn_noise_floor_db", C.c_double), ("afftdn_disabled_voice_activated", C.c_int),
                ("afftdn_custom", C.c_int)]


class LimiterDecision(C.Structure):
    _fields_ = [("pre_gain_db", C.c_double), ("ceiling_db", C.c_double), ("gain_db", C.c_double), ("filtered_tp", C.c_double),
                ("needed", C.c_int), ("clamped", C.c_int), ("pass3_prefix", C.c_char * 256)]


class ProcessResult(C.Structure):
    _fields_ = [("input", Measurements), ("effective", HostConfig), ("diag", AdaptiveDiag),
                ("filtered", L.Analysis), ("limiter", LimiterDecision), ("measure", L.LoudnormStats),
                ("effective_target_i", C.c_double), ("offset", C.c_double), ("linear_possible", C.c_int),
                ("final_", L.Analysis), ("loudnorm", L.LoudnormStats),
                ("filtered_room_tone", L.RegionSample), ("filtered_speech", L.RegionSample),
                ("final_room_tone", L.RegionSample), ("final_speech", L.RegionSample), ("has_region_samples", C.c_int),
                ("output_lufs", C.c_double), ("output_tp_db", C.c_double), ("input_lufs", C.c_double), ("input_tp_db", C.c_double),
                ("within_target", C.c_int), ("pass2_spec", C.c_char * 2048), ("pass4_spec", C.c_char * 2048),
                ("pass_ms", C.c_double * 4), ("stage_ms", C.c_double * 10)]


SIZEOF_IDS = {0: Interval, 1: Measurements, 2: HostConfig, 3: ProcessResult, 4: SpeechCandidate, 5: NoiseProfile,
              6: LimiterDecision, 7: AdaptiveDiag, 8: L.FilterParams, 9: L.LoudnormApply, 10: L.Analysis, 11: L.RegionSample,
              12: L.FlacInfo, 13: L.AudioMeta, 15: L.Timers}

HOST_SYMBOLS = ["jt_host_build_intervals", "jt_host_build_intervals_v", "jt_host_detect", "jt_host_finish_measurements", "jt_host_afftdn_band_edges",
                "jt_host_default_config", "jt_host_adapt", "jt_host_filter_spec", "jt_host_filter_params",
                "jt_host_calculate_limiter_ceiling", "jt_host_calculate_pre_gain", "jt_host_plan_limiter",
                "jt_host_calculate_linear_mode_target", "jt_host_loudnorm_internal_target_tp", "jt_host_pass4_spec",
                "jt_process_audio", "jt_analyse_only", "jt_host_vad_detect", "jt_host_vad_split", "jt_host_vad_speech_runs",
                "jt_host_vad_gap_tolerance", "jt_host_vad_gate_stats", "jt_host_vad_noise_seed",
                "jt_host_vad_pick_low_cluster", "jt_host_vad_floored_fraction", "jt_host_sizeof",
                "jt_host_lufs_filename_value", "jt_host_output_path", "jt_process_audio_cb", "jt_process_file", "jt_process_files",
                "jt_host_test_inject_fault", "jt_host_device_numa_node", "jt_host_hist_index_check", "jt_process_files_multi", "jt_handle_pool_open", "jt_handle_pool_workers", "jt_handle_pool_process_files", "jt_handle_pool_close", "jt_handle_pool_stats",
                "jt_host_score_speech_candidate", "jt_host_level_variance", "jt_host_find_best_speech_region", "jt_host_frame_level_s16", "jt_process_audio_ticks",
                "jt_host_run_record_json", "jt_host_intervals_jsonl", "jt_host_last_intervals", "jt_host_candidates_jsonl", "jt_host_intervals_in_range",
                "jt_host_score_interval_window", "jt_host_score_speech_interval_window", "jt_host_measure_speech_candidate", "jt_host_refine_golden_speech", "jt_host_loudnorm_json"]


def lib(engine=None):
    """The library an engine was opened on (default or A/B build), with the host entry points' return types set; the default build otherwise."""
    l = engine.lib if engine is not None else L.load()
    if getattr(l, "_jt_host_ready", False):
        return l
    l.jt_host_sizeof.restype = C.c_int64
    l.jt_host_build_intervals.restype = C.c_int64
    l.jt_host_build_intervals_v.restype = C.c_int64
    l.jt_host_loudnorm_internal_target_tp.restype = C.c_double
    l.jt_host_vad_floored_fraction.restype = C.c_double
    l.jt_host_finish_measurements.restype = None
    l.jt_host_afftdn_band_edges.restype = None
    l.jt_host_default_config.restype = None
    l.jt_host_adapt.restype = None
    l.jt_host_filter_params.restype = None
    l.jt_host_calculate_limiter_ceiling.restype = None
    l.jt_host_calculate_pre_gain.restype = None
    l.jt_host_plan_limiter.restype = None
    l.jt_host_calculate_linear_mode_target.restype = None
    l.jt_host_vad_split.restype = None
    l.jt_host_vad_gate_stats.restype = None
    l.jt_host_test_inject_fault.restype = None
    l.jt_host_score_speech_candidate.restype = C.c_double
    l.jt_host_level_variance.restype = C.c_double
    l.jt_host_frame_level_s16.restype = C.c_double
    l.jt_host_run_record_json.restype = C.c_int64
    l.jt_host_intervals_jsonl.restype = C.c_int64
    l.jt_host_candidates_jsonl.restype = C.c_int64
    l.jt_host_last_intervals.restype = C.c_int64
    l.jt_host_intervals_in_range.restype = C.c_int64
    l.jt_host_score_interval_window.restype = C.c_double
    l.jt_host_score_speech_interval_window.restype = C.c_double
    l.jt_handle_pool_close.restype = None
    l._jt_host_ready = True
    return l


def default_config():
    c = HostConfig()
    lib().jt_host_default_config(C.byref(c))
    return c


def filter_spec(cfg, pass_no=2):
    buf = C.create_string_buffer(4096)
    lib().jt_host_filter_spec(C.byref(cfg), C.c_int(pass_no), buf, C.c_int(4096))
    return buf.value.decode()


def adapt(base, meas):
    eff = HostConfig(); diag = AdaptiveDiag()
    lib().jt_host_adapt(C.byref(base), C.byref(meas), C.byref(eff), C.byref(diag))
    return eff, diag


def make_intervals(rows):
    """rows: list of dicts (timestamp_ns, rms_level, momentary_lufs, centroid, entropy, ...)"""
    arr = (Interval * len(rows))()
    for i, r in enumerate(rows):
        iv = arr[i]
        iv.timestamp_ns = int(r.get("timestamp_ns", i * 250_000_000))
        iv.rms_level = r.get("rms_level", 0.0); iv.peak_level = r.get("peak_level", 0.0)
        iv.momentary_lufs = r.get("momentary_lufs", 0.0); iv.shortterm_lufs = r.get("shortterm_lufs", 0.0)
        iv.true_peak = r.get("true_peak", 0.0); iv.sample_peak = r.get("sample_peak", 0.0)
        iv.spectral_found = int(r.get("spectral_found", 1))
        for k in L.SPECTRAL_KEYS:
            setattr(iv.spectral, k, float(r.get(k, 0.0)))
    return arr


def process_audio(engine, base=None, frame_samples=4096, analyse_only=False):
    """jt_process_audio / jt_analyse_only on the PCM already uploaded to `engine`."""
    l = lib(engine)
    base = base or default_config()
    res = ProcessResult()
    fn = l.jt_analyse_only if analyse_only else l.jt_process_audio
    rc = fn(engine.h, C.byref(base), C.c_int(frame_samples), C.byref(res))
    if rc != 0:
        raise L.JtError(rc, l.jt_last_error(engine.h).decode())
    return res


class ProgressUpdate(C.Structure):
    _fields_ = [("pass_", C.c_int), ("pass_name", C.c_char_p), ("progress", C.c_double), ("level", C.c_double), ("duration", C.c_double),
                ("measurements", C.POINTER(Measurements)), ("config", C.POINTER(HostConfig)), ("diag", C.POINTER(AdaptiveDiag)),
                ("has_limiter", C.c_int), ("limiter_enabled", C.c_int), ("limiter_ceiling", C.c_double)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(ProgressUpdate))


def process_audio_with_progress(engine, on_update, base=None, frame_samples=4096, ticks=False):
    """jt_process_audio_cb: `on_update(ProgressUpdate)` is called synchronously at every pass start / end; ticks=True
    (jt_process_audio_ticks) adds the reference's every-100-frames and band ticks."""
    l = lib(engine)
    base = base or default_config()
    res = ProcessResult()
    fn = PROGRESS_FN(lambda user, u: on_update(u.contents))
    rc = (l.jt_process_audio_ticks if ticks else l.jt_process_audio_cb)(engine.h, C.byref(base), C.c_int(frame_samples), fn, None, C.byref(res))
    if rc != 0:
        raise L.JtError(rc, l.jt_last_error(engine.h).decode())
    return res


def process_file(engine, input_path, base=None, frame_samples=0, md5=True):
    """jt_process_file: file in (FLAC / WAV), "<name>-LUFS-<n>-processed.flac" out.  Returns (result, output path, io_ms).
    frame_samples = 0 (here and in the batch entry points): the file's own decoder-frame cadence, as the reference sees it."""
    l = lib(engine)
    base = base or default_config()
    res = ProcessResult(); out = C.create_string_buffer(4096); io = (C.c_double * 4)()
    rc = l.jt_process_file(engine.h, str(input_path).encode(), C.byref(base), C.c_int(frame_samples), C.c_int(1 if md5 else 0),
                           None, None, C.byref(res), out, C.c_int(4096), io)
    if rc != 0:
        raise L.JtError(rc, l.jt_last_error(engine.h).decode())
    return res, out.value.decode(), list(io)


class FileResult(C.Structure):
    _fields_ = [("rc", C.c_int), ("error", C.c_char * 256), ("output_path", C.c_char * 1024), ("wall_ms", C.c_double),
                ("result", ProcessResult)]


def process_files(paths, device=0, in_flight=2, base=None, frame_samples=0, md5=True):
    """jt_process_files: at most `in_flight` files at a time on one GPU, one result per path (failures do not stop the others)."""
    l = lib()
    base = base or default_config()
    n = len(paths)
    arr = (C.c_char_p * n)(*[str(p).encode() for p in paths])
    res = (FileResult * n)()
    failed = l.jt_process_files(C.c_int(device), arr, C.c_int(n), C.c_int(in_flight), C.byref(base), C.c_int(frame_samples),
                                C.c_int(1 if md5 else 0), res)
    if failed < 0:
        raise L.JtError(failed, "jt_process_files: bad arguments")
    return failed, res


def process_files_multi(paths, devices=(0,), in_flight_per_device=2, base=None, frame_samples=0, md5=True):
    """jt_process_files_multi: one shared queue over several GPUs.  Returns (failed, results, device_of_file)."""
    l = lib()
    base = base or default_config()
    n = len(paths)
    arr = (C.c_char_p * n)(*[str(p).encode() for p in paths])
    res = (FileResult * n)(); dev = (C.c_int * n)()
    dv = (C.c_int * len(devices))(*devices)
    failed = l.jt_process_files_multi(dv, C.c_int(len(devices)), arr, C.c_int(n), C.c_int(in_flight_per_device), C.byref(base),
                                      C.c_int(frame_samples), C.c_int(1 if md5 else 0), res, dev)
    if failed < 0:
        raise L.JtError(failed, "jt_process_files_multi: bad arguments")
    return failed, res, list(dev)


class Pool:
    """jt_handle_pool_*: handles opened once, reused by every batch (jt_process_files_multi opens and closes a pool per call)."""

    def __init__(self, devices=(0,), in_flight_per_device=2, max_workers=0, ab=None):
        if ab is None:
            import os
            ab = bool(os.environ.get("JT_USE_AB_LIB"))               # (as Engine: how tools/ select the A/B build)
        self.lib = lib() if not ab else lib(type("E", (), {"lib": L.load(ab=True)})())
        self.p = C.c_void_p()
        dv = (C.c_int * len(devices))(*devices)
        rc = self.lib.jt_handle_pool_open(dv, C.c_int(len(devices)), C.c_int(in_flight_per_device), C.c_int(max_workers), C.byref(self.p))
        if rc != 0:
            raise L.JtError(rc, "jt_handle_pool_open: bad arguments")

    def workers(self):
        """The device of every open handle (sorted by device); empty when no device could be opened."""
        cap = 4096
        dv = (C.c_int * cap)()
        n = self.lib.jt_handle_pool_workers(self.p, dv, C.c_int(cap))
        return [int(dv[i]) for i in range(min(n, cap))]

    def process_files(self, paths, base=None, frame_samples=0, md5=True):
        """Returns (failed, results, device_of_file) like process_files_multi."""
        base = base or default_config()
        n = len(paths)
        arr = (C.c_char_p * n)(*[str(p).encode() for p in paths])
        res = (FileResult * n)(); dev = (C.c_int * n)()
        failed = self.lib.jt_handle_pool_process_files(self.p, arr, C.c_int(n), C.byref(base), C.c_int(frame_samples), C.c_int(1 if md5 else 0), res, dev)
        if failed < 0:
            raise L.JtError(failed, "jt_handle_pool_process_files: bad arguments")
        return failed, res, list(dev)

    def stats(self):
        """jt_handle_pool_stats of the last batch as a dict of per-file means (ms)."""
        v = (C.c_double * 9)()
        self.lib.jt_handle_pool_stats(self.p, v, C.c_int(9))
        n = max(1.0, v[8])
        keys = ("wait_io_set", "read", "decode", "passes", "encode", "wait_finisher", "md5", "write")
        return {k: round(v[i] / n, 2) for i, k in enumerate(keys)}

    def close(self):
        if self.p:
            self.lib.jt_handle_pool_close(self.p)
            self.p = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


SIZEOF_IDS[14] = FileResult


def output_path(input_path, lufs_value):
    buf = C.create_string_buffer(4096)
    n = lib().jt_host_output_path(input_path.encode(), C.c_int(lufs_value), buf, C.c_int(4096))
    if n < 0:
        raise ValueError("output path too long")
    return buf.value.decode()


def lufs_filename_value(lufs):
    return lib().jt_host_lufs_filename_value(C.c_double(lufs))


def inject_fault(create_temp=0, write=0, rename=0):
    """Test seam of jt_process_file's publish discipline (include/jt_host.h: jt_host_test_inject_fault)."""
    lib().jt_host_test_inject_fault(C.c_int(create_temp), C.c_int(write), C.c_int(rename))


class RunProvenance(C.Structure):
    _fields_ = [("input_file", C.c_char_p), ("version", C.c_char_p), ("executable", C.c_char_p), ("processed_at", C.c_char_p),
                ("duration_s", C.c_double), ("sample_rate_hz", C.c_int), ("channels", C.c_int)]


def _sized(call):
    n = call(None, 0)
    if n < 0:
        raise L.JtError(int(n), "run record: bad arguments")
    buf = C.create_string_buffer(int(n) + 1)
    call(buf, int(n) + 1)
    return buf.value.decode()


def run_record_json(engine, res, input_file="", version="dev", executable="", processed_at="", duration_s=0.0, sample_rate_hz=0,
                    channels=0, analysis_only=False):
    """jt_host_run_record_json: the reference's RunRecord document (runrecord.go) as text."""
    enc = lambda v: v if isinstance(v, bytes) else v.encode()
    pv = RunProvenance(enc(input_file), version.encode(), executable.encode(), processed_at.encode(), duration_s, sample_rate_hz, channels)
    return _sized(lambda b, c: lib(engine).jt_host_run_record_json(engine.h if engine else None, C.byref(res), C.byref(pv), C.c_int(int(analysis_only)), b, C.c_int64(c)))


def intervals_jsonl(engine):
    return _sized(lambda b, c: lib(engine).jt_host_intervals_jsonl(engine.h, b, C.c_int64(c)))


def candidates_jsonl(res):
    return _sized(lambda b, c: lib().jt_host_candidates_jsonl(C.byref(res), b, C.c_int64(c)))


def loudnorm_json(stats):
    return _sized(lambda b, c: lib().jt_host_loudnorm_json(C.byref(stats), b, C.c_int(c)))
