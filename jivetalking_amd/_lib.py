"""ctypes binding of include/jtgpu.h.  Fails loudly when the HIP library is missing: there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libjtgpu.so")
LIB_PATH_AB = os.path.join(_HERE, "lib", "libjtgpu_ab.so")
_LIBS = {}

JT_OK, JT_E_INVAL, JT_E_NOGPU, JT_E_HIP, JT_E_STATE, JT_E_UNSUPPORTED, JT_E_CANCELLED, JT_E_SILENT = 0, -1, -2, -3, -4, -5, -6, -7


class JtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"jtgpu error {code}: {msg}")
        self.code = code


SPECTRAL_KEYS = ["mean", "variance", "centroid", "spread", "skewness", "kurtosis", "entropy",
                 "flatness", "crest", "flux", "slope", "decrease", "rolloff"]
ASTATS_KEYS = ["dc_offset", "min_level", "max_level", "min_difference", "max_difference", "mean_difference",
               "rms_difference", "peak_level", "rms_level", "rms_peak", "rms_trough", "crest_factor", "flat_factor",
               "peak_count", "noise_floor", "noise_floor_count", "entropy", "dynamic_range", "zero_crossings",
               "zero_crossings_rate", "number_of_samples", "bit_depth"]
R128_KEYS = ["integrated", "lra", "lra_low", "lra_high", "momentary", "shortterm", "true_peak", "sample_peak",
             "target_threshold"]


class Spectral(C.Structure):
    _fields_ = [(k, C.c_double) for k in SPECTRAL_KEYS]


class Astats(C.Structure):
    _fields_ = [(k, C.c_double) for k in ASTATS_KEYS]


class R128(C.Structure):
    _fields_ = [(k, C.c_double) for k in R128_KEYS]


class FrameMeta(C.Structure):
    _fields_ = [("momentary", C.c_double), ("shortterm", C.c_double), ("true_peak", C.c_double),
                ("sample_peak", C.c_double), ("spectral", Spectral)]


class Analysis(C.Structure):
    _fields_ = [("astats", Astats), ("r128", R128), ("spectral_mean", Spectral),
                ("n_frames_meta", C.c_int64), ("n_input_frames", C.c_int64)]


class FilterParams(C.Structure):
    _fields_ = [
        ("hp_enabled", C.c_int), ("hp_freq", C.c_double), ("hp_q", C.c_double),
        ("lp_enabled", C.c_int), ("lp_freq", C.c_double), ("lp_q", C.c_double),
        ("nlm_enabled", C.c_int), ("nlm_strength", C.c_double), ("nlm_patch_s", C.c_double),
        ("nlm_research_s", C.c_double), ("nlm_smooth", C.c_double),
        ("fft_enabled", C.c_int), ("fft_nr", C.c_double), ("fft_nf", C.c_double),
        ("fft_custom", C.c_int), ("fft_band_noise", C.c_double * 15),
        ("fft_track_noise", C.c_int),
        ("gate_enabled", C.c_int), ("gate_threshold", C.c_double), ("gate_ratio", C.c_double),
        ("gate_attack_ms", C.c_double), ("gate_release_ms", C.c_double), ("gate_range", C.c_double),
        ("gate_knee", C.c_double), ("gate_makeup", C.c_double),
        ("comp_enabled", C.c_int), ("comp_threshold", C.c_double), ("comp_ratio", C.c_double),
        ("comp_attack_ms", C.c_double), ("comp_release_ms", C.c_double), ("comp_makeup", C.c_double),
        ("comp_knee", C.c_double), ("comp_mix", C.c_double),
        ("deess_enabled", C.c_int), ("deess_i", C.c_double), ("deess_m", C.c_double), ("deess_f", C.c_double),
        ("out_rate", C.c_int), ("out_frame_samples", C.c_int),
    ]


class RegionSample(C.Structure):
    _fields_ = [("rms_level", C.c_double), ("peak_level", C.c_double), ("crest_factor", C.c_double),
                ("spectral", Spectral), ("momentary", C.c_double), ("shortterm", C.c_double),
                ("true_peak", C.c_double), ("sample_peak", C.c_double), ("frames", C.c_int64)]


class LimiterPlan(C.Structure):
    _fields_ = [("needed", C.c_int), ("pre_gain_db", C.c_double), ("limit", C.c_double)]


class LoudnormStats(C.Structure):
    _fields_ = [("input_i", C.c_double), ("input_tp", C.c_double), ("input_lra", C.c_double),
                ("input_thresh", C.c_double), ("output_i", C.c_double), ("output_tp", C.c_double),
                ("output_lra", C.c_double), ("output_thresh", C.c_double), ("target_offset", C.c_double),
                ("normalization_type_dynamic", C.c_int)]


class LoudnormApply(C.Structure):
    _fields_ = [("target_i", C.c_double), ("target_tp", C.c_double), ("target_lra", C.c_double),
                ("measured_i", C.c_double), ("measured_tp", C.c_double), ("measured_lra", C.c_double),
                ("measured_thresh", C.c_double), ("offset", C.c_double),
                ("adeclick_enabled", C.c_int), ("adeclick_threshold", C.c_double),
                ("adeclick_window_ms", C.c_double), ("adeclick_overlap_pct", C.c_double), ("adeclick_method", C.c_int),
                ("brickwall_limit", C.c_double)]


class Timers(C.Structure):
    _fields_ = [("pass1_ms", C.c_double), ("pass2_ms", C.c_double), ("pass3_ms", C.c_double),
                ("pass4_ms", C.c_double), ("nlm_ms", C.c_double), ("nlm_launches", C.c_int64),
                ("declick_repaired", C.c_int64), ("declick_ms", C.c_double), ("declick_heavy_windows", C.c_int64),
                ("tp_units_total", C.c_int64), ("tp_units_evaluated", C.c_int64),
                ("ln_stream_frames", C.c_int64), ("ln_stream_why", C.c_int64)]


class FlacInfo(C.Structure):
    _fields_ = [("bytes", C.c_int64), ("frames", C.c_int64), ("total_samples", C.c_int64),
                ("sample_rate", C.c_int), ("channels", C.c_int), ("bits_per_sample", C.c_int), ("block_size", C.c_int),
                ("min_frame_bytes", C.c_int), ("max_frame_bytes", C.c_int), ("header_bytes", C.c_int),
                ("gpu_ms", C.c_double), ("md5_ms", C.c_double), ("total_ms", C.c_double), ("md5", C.c_uint8 * 16)]


JT_FLAC_MD5 = 1


class AudioMeta(C.Structure):
    _fields_ = [("format", C.c_int), ("sample_rate", C.c_int), ("channels", C.c_int), ("bits_per_sample", C.c_int),
                ("is_float", C.c_int), ("frames", C.c_int64), ("duration_s", C.c_double), ("flac_frames", C.c_int64),
                ("flac_candidates", C.c_int), ("gpu_ms", C.c_double), ("total_ms", C.c_double),
                ("decoder_frame_samples", C.c_int), ("decoder_frames_variable", C.c_int), ("decoder_frames", C.c_int64),
                ("channel_mask", C.c_uint64)]

# every symbol include/jtgpu.h declares
SYMBOLS = [
    "jt_device_count", "jt_open", "jt_open_ex", "jt_close", "jt_last_error", "jt_version", "jt_set_option", "jt_build_flags", "jt_cancel", "jt_reset_cancel", "jt_begin_job", "jt_end_job", "jt_pass3_plan_hook", "jt_set_source_format",
    "jt_upload_pcm", "jt_upload_pcm_layout", "jt_attach_device_pcm", "jt_upload_s16", "jt_load_audio", "jt_op_decode_audio", "jt_input_frame_layout",
    "jt_pass1", "jt_band_rms", "jt_pass2", "jt_pass2_prefetch", "jt_pass2_prefetch_after_pass1", "jt_region_measure", "jt_region_measure_pair", "jt_region_prefetch", "jt_pass3", "jt_pass4",
    "jt_output_len", "jt_download_s16", "jt_output_frame_levels", "jt_flac_encode", "jt_op_flac_encode_s16", "jt_get_timers",
    "jt_op_biquad_f32", "jt_op_anlmdn_f32", "jt_op_afftdn_f32", "jt_op_afftdn_tn_f32", "jt_op_dynamics", "jt_op_alimiter_f64", "jt_op_adeclick_f64",
    "jt_op_resample_f32_to_s16", "jt_op_ebur128", "jt_op_astats", "jt_op_aspectralstats",
    "jt_op_loudnorm_measure_s16", "jt_op_loudnorm_dynamic_f64",
]


def load(ab=False):
    """Load libjtgpu.so (ab=True: libjtgpu_ab.so, the A/B build with the superseded kernel generations and tuning knobs, `make ab`).
    Raises (never falls back) when the HIP extension has not been built."""
    key = "ab" if ab else "default"
    if key in _LIBS:
        return _LIBS[key]
    path = LIB_PATH_AB if ab else LIB_PATH
    if not ab and os.environ.get("JT_LIB_PATH"):          # (tools/: A/B of two builds of the library inside one gpurun call)
        path = os.environ["JT_LIB_PATH"]
    if not os.path.exists(path):
        raise ImportError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    # ROCclr's hardware-queue count is read when the process first touches HIP and belongs to the host application (jtgpu.h, jt_set_option):
    # the library does not touch the environment, this binding -- the host here -- asks for one queue per stream of a handle
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    lib = C.CDLL(path)
    lib.jt_last_error.restype = C.c_char_p
    lib.jt_version.restype = C.c_char_p
    lib.jt_close.restype = None
    lib.jt_cancel.restype = None
    lib.jt_reset_cancel.restype = None
    lib.jt_begin_job.restype = None
    lib.jt_end_job.restype = None
    assert bool(lib.jt_build_flags() & 1) == bool(ab), "library flavour does not match its file name"
    _LIBS[key] = lib
    return lib


def device_count():
    """jt_device_count: HIP devices visible to this process."""
    return int(load().jt_device_count())


def set_global_option(key, value, ab=False):
    """jt_set_option(NULL, ...): process-wide keys (graveyard_gb, poison_alloc) of one library flavour."""
    rc = load(ab).jt_set_option(None, str(key).encode(), str(value).encode())
    if rc != 0:
        raise JtError(rc, f"jt_set_option(NULL, {key!r}, {value!r})")
