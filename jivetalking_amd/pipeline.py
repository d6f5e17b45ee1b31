"""Four-pass glue over the engine (mirrors processor.ProcessAudio's orchestration, processor.go:78-216).

Scalar planning between the passes follows normalise.go exactly (calculateLimiterCeiling :373-396,
calculatePreGain :411-431, loudnormInternalTargetTP :583-585, calculateLinearModeTarget :614-632,
loudnormTPTargets :1198-1203) with FFmpeg's string-formatted precisions applied where the reference passes values
through filter-spec strings.  The adaptive Pass-2 parameters come from the host mirror of AdaptConfig when
available (jivetalking_amd.hostlogic); `minimal_adapt` below is the reduced stand-in used until then.
"""
import math
import numpy as np
from . import _lib as L
from .engine import default_filter_params

TARGET_I, TARGET_TP, TARGET_LRA = -16.0, -1.0, 20.0
MIN_LIMITER_CEILING_DB = -24.0
BRICKWALL_HEADROOM_DB = 0.9
MEASUREMENT_CUSHION_DB = 0.2
LINEAR_SAFETY_MARGIN = 0.1


def _fmt(x, spec):
    return float(format(x, spec))


def q_lin3(x):
    """ebur128 metadata peaks are printed '%.3f' (linear) before the Go side converts to dB."""
    return _fmt(x, ".3f")


def lin_to_db(x):
    return -120.0 if x <= 0 else 20.0 * math.log10(x)


def plan_limiter(output_i, output_tp_db, target_i=TARGET_I, target_tp=TARGET_TP):
    """planLimiterForLoudnorm (normalise.go:539-561)."""
    gain = target_i - output_i
    projected = output_tp_db + gain
    needed, clamped, ceiling = False, False, 0.0
    if projected > target_tp:
        needed = True
        ceiling = target_tp - gain
        if ceiling < MIN_LIMITER_CEILING_DB:
            ceiling, clamped = MIN_LIMITER_CEILING_DB, True
    ideal = target_tp - gain
    pre_gain, rederived = 0.0, 0.0
    if ideal < MIN_LIMITER_CEILING_DB:
        pre_gain = MIN_LIMITER_CEILING_DB - ideal
        rederived = target_tp - (target_i - (output_i + pre_gain))
    if clamped:
        ceiling = rederived
    plan = L.LimiterPlan(0, 0.0, 1.0)
    if needed:
        plan.needed = 1
        plan.pre_gain_db = _fmt(pre_gain, ".1f") if pre_gain > 0 else 0.0      # volume=%.1fdB
        plan.limit = _fmt(10 ** (ceiling / 20.0), ".6f")                        # alimiter=limit=%.6f
    return plan, dict(needed=needed, clamped=clamped, ceiling_db=ceiling, pre_gain_db=pre_gain, gain_db=gain)


def plan_loudnorm_apply(meas, target_i=TARGET_I, target_tp=TARGET_TP, target_lra=TARGET_LRA):
    """Pass-4 loudnorm options from the Pass-3 measurement (normalise.go:861-873,1198-1291)."""
    mi, mtp = _fmt(meas["input_i"], ".2f"), _fmt(meas["input_tp"], ".2f")       # loudnorm JSON is '%.2f'
    mlra, mth = _fmt(meas["input_lra"], ".2f"), _fmt(meas["input_thresh"], ".2f")
    internal_tp = mtp + (target_i - mi) + LINEAR_SAFETY_MARGIN + MEASUREMENT_CUSHION_DB
    max_linear_i = internal_tp - mtp + mi - LINEAR_SAFETY_MARGIN
    if target_i <= max_linear_i:
        eff_i, linear_possible = target_i, True
    else:
        eff_i, linear_possible = max_linear_i, False
    offset = eff_i - mi
    emitted_tp = max(-9.0, min(internal_tp, 0.0))
    brick_db = target_tp - BRICKWALL_HEADROOM_DB
    ap = L.LoudnormApply(_fmt(eff_i, ".2f"), _fmt(emitted_tp, ".2f"), _fmt(target_lra, ".1f"),
                         mi, mtp, mlra, mth, _fmt(offset, ".2f"),
                         1, 1.7, 55.0, 50.0, _fmt(10 ** (brick_db / 20.0), ".6f"))
    return ap, dict(effective_i=eff_i, offset=offset, linear_possible=linear_possible, emitted_tp=emitted_tp)


def minimal_adapt(a1):
    """Reduced AdaptConfig stand-in: static afftdn floor from the momentary-loudness p10 (the VAD's floor axis,
    analyser_vad.go:311), everything else at DefaultFilterConfig.  Replaced by hostlogic.adapt_config()."""
    ms = np.array([m["momentary"] for m in a1.get("meta", []) if np.isfinite(m["momentary"])])
    ms = ms[ms > -100]
    nf = -50.0
    if ms.size:
        s = np.sort(ms)
        nf = float(s[int(0.10 * (s.size - 1))])
    nf = max(-80.0, min(-20.0, nf))
    return default_filter_params(fft_nf=float("%g" % nf), fft_track_noise=0)


def process_resident(engine, n_frames, sample_rate, frame_samples=4096, adapt=None, want_meta=True):
    """All four passes over the PCM already resident on the engine's device.  Returns a result dict; the final
    s16 stays on the device (engine.download_s16(4))."""
    a1 = engine.pass1(n_frames, frame_samples, sample_rate, want_meta=want_meta)
    params = (adapt or minimal_adapt)(a1)
    a2 = engine.pass2(params)
    tp2_db = lin_to_db(q_lin3(a2["r128"]["true_peak"]))
    i2 = _fmt(a2["r128"]["integrated"], ".3f")
    lim, lim_info = plan_limiter(i2, tp2_db)
    m3 = engine.pass3(lim)
    if not math.isfinite(m3["input_i"]) or m3["input_i"] < -70.0:
        raise RuntimeError("cannot normalise silent audio (measured %.1f LUFS)" % m3["input_i"])
    ap, ap_info = plan_loudnorm_apply(m3)
    a4, s4 = engine.pass4(lim, ap)
    return dict(input=a1, filtered=a2, limiter=lim_info, measure=m3, apply=ap_info, final=a4, loudnorm=s4,
                output_lufs=a4["r128"]["integrated"], output_tp_db=lin_to_db(a4["r128"]["true_peak"]))
