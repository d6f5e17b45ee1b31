"""oracle/chain.py — the reference's Pass-3 / Pass-4 filter chains composed from the oracle's pieces, driven by the SPEC STRING the
host logic prints (what FFmpeg would parse: normalise.go:1231-1334 buildLoudnormFilterSpec, :199-346 measureWithLoudnorm).

TEST INFRASTRUCTURE (like everything under oracle/): used by tests/ and by bench.py's checker legs only; nothing in the product path
imports it.  Sequential, one core: adeclick's dense solve makes it ~8 xRT.
"""
import numpy as np

from . import orc


def parse_spec(spec):
    """'a=x=1:y=2,b=3dB' -> [('a', {'x': '1', 'y': '2'}, 'x=1:y=2'), ('b', {}, '3dB')]"""
    out = []
    for f in spec.split(","):
        name, _, args = f.partition("=")
        kv = {}
        for a in args.split(":") if args else []:
            k, eq, v = a.partition("=")
            if eq:
                kv[k] = v
        out.append((name, kv, args))
    return out


def pass4(p2_s16, rate, spec, stop_before=None):
    """The Pass-4 graph on the Pass-2 output (s16 at `rate`): [volume] -> [alimiter prefix] -> loudnorm (linear when af_loudnorm's
    init() accepts the measured values, its dynamic mode at 192 kHz otherwise) -> adeclick -> alimiter -> flt -> s16.
    Returns {'s16', 'dynamic', 'loudnorm', 'pre_adeclick'} ; stop_before='adeclick' returns after the loudnorm stage."""
    x = np.asarray(p2_s16, np.int16).astype(np.float64) / 32768.0
    info = {"dynamic": 0, "loudnorm": None}
    seen_loudnorm = False
    for name, kv, raw in parse_spec(spec.decode() if isinstance(spec, bytes) else spec):
        if name == "volume":
            db = float(raw.replace("dB", ""))
            x = (x.astype(np.float32) * np.float32(10 ** (db / 20.0))).astype(np.float64)        # af_volume, precision=float
        elif name == "alimiter":
            x = orc.alimiter(x, rate, float(kv["limit"]), float(kv["attack"]), float(kv["release"]))
        elif name == "loudnorm":
            seen_loudnorm = True
            ti, ttp, tlra = float(kv["I"]), float(kv["TP"]), float(kv["LRA"])
            mi, mtp, mlra, mth = float(kv["measured_I"]), float(kv["measured_TP"]), float(kv["measured_LRA"]), float(kv["measured_thresh"])
            off = float(kv.get("offset", "0"))
            # af_loudnorm.c init(): linear only when every measured_* is supplied and the projected peak / LRA fit
            offset_db = ti - mi
            linear = (mtp != 99 and mth != -70 and mlra != 0 and mi != 0) and (mtp + offset_db <= ttp) and (mlra <= tlra)
            if linear:
                x = x * 10 ** (offset_db / 20.0)
            else:
                up = orc.swr_f64(x, rate, 192000, True)
                y192, st = orc.loudnorm_dynamic(up, ti, tlra, ttp, measured=(mi, mlra, mtp, mth), offset=off)
                x = orc.swr_f64(y192, 192000, rate, True)[: x.size]
                info["dynamic"] = int(st["dynamic"]); info["loudnorm"] = st
        elif name == "aresample":
            assert int(raw) == rate, "the chain keeps the Pass-2 rate"
        elif name == "adeclick":
            if stop_before == "adeclick":
                break
            info["pre_adeclick"] = x
            x = orc.adeclick(x, rate, float(kv["t"]), float(kv["w"]), float(kv["o"]), method=kv.get("m", "a")[0])
        elif name in ("astats", "aspectralstats", "ebur128", "aformat", "asetnsamples"):
            continue
        else:
            raise ValueError("oracle chain: filter not restated: " + name)
    assert seen_loudnorm
    info["f64"] = x
    info["s16"] = orc.f64_to_s16(x.astype(np.float32).astype(np.float64))                        # dbl -> flt (aspectralstats link) -> s16
    return info


def landing(s16, rate):
    """Integrated loudness and true peak (dBTP) of a delivered file, as the reference's final ebur128 measures it."""
    e = orc.ebur128(np.asarray(s16, np.int16).astype(np.float64) / 32768.0, rate, True, True)
    return {"output_lufs": float(e["integrated"]), "output_dbtp": float(20 * np.log10(e["true_peak"])) if e["true_peak"] > 0 else float("-inf"),
            "lra": float(e["lra"])}
