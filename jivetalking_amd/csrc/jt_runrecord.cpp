// jt_runrecord.cpp — the reference's run record (internal/processor/runrecord.go:15-51) and its two .jsonl sidecars
// (runrecord_write.go:37-45) assembled from a jt_process_result, so that the unchanged report / UI code above the seam reads the
// same JSON it reads today.  The reference builds a map[string]any tree (sanitiseValue: NaN / Inf -> null, omitempty
// honoured, embedded structs promoted, wrapper types turning time.Duration nanoseconds into *_s seconds) and prints it with
// json.MarshalIndent — i.e. keys SORTED at every level, two-space indent, encoding/json's number and string formats.  The same tree
// is built here; field names are the reference's json tags (analyser.go:28-308, analyser_metrics.go:697-711, filters.go:111-316,
// normalise.go:64-75,640-674, runrecord.go:24-215, runrecord_units.go:143-340).
#include "jt_internal.h"
#include "../../include/jt_host.h"
#include <algorithm>
#include <cstdlib>
#include <map>
#include <memory>

namespace {
struct J;
typedef std::shared_ptr<J> JP;
struct J {
    enum Kind { Null, Bool, Int, Num, Str, Arr, Obj } kind = Null;
    bool b = false; long long i = 0; double d = 0; std::string s; std::vector<JP> a; std::map<std::string, JP> o;
};
JP jnull() { return std::make_shared<J>(); }
JP jbool(bool v) { auto p = std::make_shared<J>(); p->kind = J::Bool; p->b = v; return p; }
JP jint(long long v) { auto p = std::make_shared<J>(); p->kind = J::Int; p->i = v; return p; }
// sanitiseValue: non-finite floats become null
JP jnum(double v) { auto p = std::make_shared<J>(); if (std::isfinite(v)) { p->kind = J::Num; p->d = v; } return p; }
JP jstr(const std::string &v) { auto p = std::make_shared<J>(); p->kind = J::Str; p->s = v; return p; }
JP jobj() { auto p = std::make_shared<J>(); p->kind = J::Obj; return p; }
JP jarr() { auto p = std::make_shared<J>(); p->kind = J::Arr; return p; }

// encoding/json floatEncoder: strconv.AppendFloat(f, 'f' or 'e', -1, 64) -- the shortest digits that round-trip; 'e' when
// |f| < 1e-6 or >= 1e21, with a two-digit exponent's leading zero dropped (e-07 -> e-7)
std::string go_float(double f)
{
    if (f == 0) return std::signbit(f) ? "-0" : "0";
    char buf[64]; int prec = 1; int exp10 = 0; std::string digits;
    for (; prec <= 17; ++prec) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, f);
        if (std::strtod(buf, nullptr) == f) break;
    }
    snprintf(buf, sizeof buf, "%.*e", prec - 1, f);
    std::string t(buf);
    const bool neg = t[0] == '-';
    if (neg) t = t.substr(1);
    const size_t epos = t.find('e');
    std::string mant = t.substr(0, epos); exp10 = std::atoi(t.c_str() + epos + 1);
    for (char c : mant) if (c != '.') digits.push_back(c);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const double af = std::fabs(f);
    std::string out = neg ? "-" : "";
    if (af < 1e-6 || af >= 1e21) {
        out += digits.substr(0, 1);
        if (digits.size() > 1) out += "." + digits.substr(1);
        char eb[16]; snprintf(eb, sizeof eb, "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
        std::string e(eb);
        if (e.size() == 4 && e[2] == '0') e = e.substr(0, 2) + e.substr(3);          // e-07 -> e-7
        return out + e;
    }
    const int nd = (int)digits.size();
    if (exp10 >= 0) {
        if (nd <= exp10 + 1) out += digits + std::string((size_t)(exp10 + 1 - nd), '0');
        else out += digits.substr(0, (size_t)exp10 + 1) + "." + digits.substr((size_t)exp10 + 1);
    } else out += "0." + std::string((size_t)(-exp10 - 1), '0') + digits;
    return out;
}
// encoding/json string encoding with the default HTML escaping (encode.go appendString, Go 1.22+: short forms for \b \f \n \r \t,
// \u00XX for the other control bytes, \u003c \u003e \u0026, U+2028 / U+2029 escaped, every invalid UTF-8 byte replaced by \ufffd)
std::string go_string(const std::string &s)
{
    std::string o = "\"";
    const size_t n = s.size();
    for (size_t i = 0; i < n;) {
        const unsigned char c = (unsigned char)s[i];
        if (c < 0x80) {
            switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '<': o += "\\u003c"; break;
            case '>': o += "\\u003e"; break;
            case '&': o += "\\u0026"; break;
            default:
                if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
                else o.push_back((char)c);
            }
            ++i;
            continue;
        }
        // decode one UTF-8 sequence as utf8.DecodeRuneInString does (shortest form, no surrogates, <= U+10FFFF)
        int len = 0; unsigned cp = 0;
        if (c >= 0xC2 && c <= 0xDF) { len = 2; cp = c & 0x1F; }
        else if (c >= 0xE0 && c <= 0xEF) { len = 3; cp = c & 0x0F; }
        else if (c >= 0xF0 && c <= 0xF4) { len = 4; cp = c & 0x07; }
        bool ok = len != 0 && i + (size_t)len <= n;
        for (int k = 1; ok && k < len; ++k) {
            const unsigned char d = (unsigned char)s[i + (size_t)k];
            if ((d & 0xC0) != 0x80) ok = false;
            cp = (cp << 6) | (d & 0x3F);
        }
        if (ok && ((len == 3 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)))) ok = false;
        if (!ok) { o += "\\ufffd"; ++i; continue; }
        if (cp == 0x2028 || cp == 0x2029) { o += cp == 0x2028 ? "\\u2028" : "\\u2029"; i += (size_t)len; continue; }
        o.append(s, i, (size_t)len);
        i += (size_t)len;
    }
    return o + "\"";
}
void emit(const JP &v, std::string &out, int depth, bool indent)
{
    auto nl = [&](int d) { if (indent) { out.push_back('\n'); out.append((size_t)d * 2, ' '); } };
    switch (v->kind) {
    case J::Null: out += "null"; break;
    case J::Bool: out += v->b ? "true" : "false"; break;
    case J::Int: out += std::to_string(v->i); break;
    case J::Num: out += go_float(v->d); break;
    case J::Str: out += go_string(v->s); break;
    case J::Arr:
        if (v->a.empty()) { out += "[]"; break; }
        out.push_back('[');
        for (size_t k = 0; k < v->a.size(); ++k) { if (k) out.push_back(','); nl(depth + 1); emit(v->a[k], out, depth + 1, indent); }
        nl(depth); out.push_back(']');
        break;
    case J::Obj: {
        if (v->o.empty()) { out += "{}"; break; }
        out.push_back('{');
        bool first = true;
        for (auto &kv : v->o) {                       // std::map: byte-wise sorted keys, as encoding/json sorts map keys
            if (!first) out.push_back(',');
            first = false;
            nl(depth + 1); out += go_string(kv.first); out += indent ? ": " : ":"; emit(kv.second, out, depth + 1, indent);
        }
        nl(depth); out.push_back('}');
        break; }
    }
}
double secs(int64_t ns) { return (double)ns / 1e9; }      // time.Duration.Seconds()

JP spectral(const jt_spectral &s)
{
    JP o = jobj();
    o->o["mean"] = jnum(s.mean); o->o["variance"] = jnum(s.variance); o->o["centroid_hz"] = jnum(s.centroid); o->o["spread_hz"] = jnum(s.spread);
    o->o["skewness"] = jnum(s.skewness); o->o["kurtosis"] = jnum(s.kurtosis); o->o["entropy"] = jnum(s.entropy); o->o["flatness"] = jnum(s.flatness);
    o->o["crest"] = jnum(s.crest); o->o["flux"] = jnum(s.flux); o->o["slope"] = jnum(s.slope); o->o["decrease"] = jnum(s.decrease);
    o->o["rolloff_hz"] = jnum(s.rolloff);
    return o;
}
void region_sample_fields(JP &o, double rms, double peak, double crest, const jt_spectral &sp, double m, double st, double tp, double spk)
{
    o->o["rms_level_dbfs"] = jnum(rms); o->o["peak_level_dbfs"] = jnum(peak); o->o["crest_factor_db"] = jnum(crest);
    o->o["spectral"] = spectral(sp);
    o->o["momentary_lufs"] = jnum(m); o->o["short_term_lufs"] = jnum(st); o->o["true_peak_dbtp"] = jnum(tp); o->o["sample_peak_dbfs"] = jnum(spk);
}
double lin2db(double v) { return v > 0 ? 20.0 * std::log10(v) : -INFINITY; }
// RegionSample of an output stage (analyser_output.go:95-227: crest dB = peak - rms, peaks 20 log10 of the last linear value, RMS -60 fallback)
JP region_sample_out(const jt_region_sample &r)
{
    JP o = jobj();
    region_sample_fields(o, r.rms_level, r.peak_level, r.peak_level - r.rms_level, r.spectral, r.momentary, r.shortterm, lin2db(r.true_peak), lin2db(r.sample_peak));
    return o;
}
JP region_sample_in(const jt_region_metrics &r)
{
    JP o = jobj();
    region_sample_fields(o, r.rms_level, r.peak_level, r.crest_factor, r.spectral, r.momentary_lufs, r.shortterm_lufs, r.true_peak, r.sample_peak);
    return o;
}
JP dynamics(const jt_astats &a)
{
    JP o = jobj();
    o->o["dynamic_range_db"] = jnum(a.dynamic_range); o->o["rms_level_dbfs"] = jnum(a.rms_level); o->o["peak_level_dbfs"] = jnum(a.peak_level);
    o->o["rms_trough_dbfs"] = jnum(a.rms_trough); o->o["rms_peak_dbfs"] = jnum(a.rms_peak); o->o["dc_offset"] = jnum(a.dc_offset);
    o->o["flat_factor"] = jnum(a.flat_factor); o->o["crest_factor_astats_db"] = jnum(a.crest_factor); o->o["zero_crossings_rate"] = jnum(a.zero_crossings_rate);
    o->o["zero_crossings_count"] = jnum(a.zero_crossings); o->o["max_difference"] = jnum(a.max_difference); o->o["min_difference"] = jnum(a.min_difference);
    o->o["mean_difference"] = jnum(a.mean_difference); o->o["rms_difference"] = jnum(a.rms_difference); o->o["entropy"] = jnum(a.entropy);
    o->o["min_level_dbfs"] = jnum(a.min_level); o->o["max_level_dbfs"] = jnum(a.max_level); o->o["noise_floor_count"] = jnum(a.noise_floor_count);
    o->o["bit_depth"] = jnum(a.bit_depth); o->o["number_of_samples"] = jnum(a.number_of_samples);
    return o;
}
// astats of an output stage arrives raw (crest linear, min / max level linear): the Go-side conversions (analyser_metrics.go:663-692)
double qf(double v) { if (!std::isfinite(v)) return v; char b[400]; snprintf(b, sizeof b, "%f", v); return std::strtod(b, nullptr); }      // astats metadata is "%f"
jt_astats convert_astats(const jt_astats &raw0)
{
    jt_astats raw = raw0;
    { double *p = &raw.dc_offset; for (int i = 0; i < 22; ++i) p[i] = qf(p[i]); }
    jt_astats a = raw;
    a.crest_factor = raw.crest_factor <= 0 ? -120.0 : 20.0 * std::log10(raw.crest_factor);                    // linearRatioToDB
    auto lvl = [](double v) -> double { double av = std::fabs(v); if (av <= 0) return -120.0; if (av > 1.0) av /= 32768.0; if (av > 1.0) av = 1.0; return 20.0 * std::log10(av); };   // linearSampleToDBFS
    a.min_level = lvl(raw.min_level); a.max_level = lvl(raw.max_level);
    return a;
}
double q3(double v) { char b[64]; snprintf(b, sizeof b, "%.3f", v); return std::strtod(b, nullptr); }     // ebur128 metadata is "%.3f"
// OutputLoudnessMetrics as finalizeOutputMeasurements fills it (analyser_metrics.go:987-1040): the last ebur128 metadata values
// ("%.3f"), peaks 20 log10 of the linear value; f_ebur128.c exports no lavfi.r128.target_threshold key, so the thresh falls back to
// I - 10; TargetOffset is initialised to 0.0 and never assigned for output stages.
JP out_loudness(const jt_r128 &r)
{
    JP o = jobj();
    const double I = q3(r.integrated);
    o->o["momentary_lufs"] = jnum(q3(r.momentary)); o->o["short_term_lufs"] = jnum(q3(r.shortterm)); o->o["sample_peak_dbfs"] = jnum(lin2db(q3(r.sample_peak)));
    o->o["integrated_lufs"] = jnum(I); o->o["true_peak_dbtp"] = jnum(lin2db(q3(r.true_peak))); o->o["lra_lu"] = jnum(q3(r.lra));
    o->o["thresh_lufs"] = jnum(I != 0.0 ? I - 10.0 : 0.0); o->o["target_offset_db"] = jnum(0.0);
    return o;
}
const char *floor_source_name(int s) { static const char *n[4] = {"astats", "rms_estimate", "ebur128_estimate", "vad_percentile"}; return (s >= 0 && s < 4) ? n[s] : ""; }
std::string fmt2(double v) { char b[64]; snprintf(b, sizeof b, "%.2f", v); return b; }
} // namespace

static std::string build_record(const jt_ctx *h, const jt_process_result *res, const jt_run_provenance *pv, bool analysis_only)
{
    const jt_measurements &m = res->input;
    JP root = jobj();
    root->o["schema_version"] = jint(1);
    {
        JP r = jobj();
        r->o["input_file"] = jstr(pv && pv->input_file ? pv->input_file : ""); r->o["version"] = jstr(pv && pv->version ? pv->version : "");
        r->o["executable"] = jstr(pv && pv->executable ? pv->executable : ""); r->o["processed_at"] = jstr(pv && pv->processed_at ? pv->processed_at : "");
        r->o["duration_s"] = jnum(pv && pv->duration_s > 0 ? pv->duration_s : m.duration_s);
        r->o["sample_rate_hz"] = jint(pv ? pv->sample_rate_hz : 0); r->o["channels"] = jint(pv ? pv->channels : 0);
        root->o["run"] = r;
    }
    JP lst = jobj(), dst = jobj(), sst = jobj();
    {
        JP li = jobj();
        li->o["momentary_lufs"] = jnum(m.momentary); li->o["short_term_lufs"] = jnum(m.shortterm); li->o["sample_peak_dbfs"] = jnum(m.sample_peak);
        li->o["integrated_lufs"] = jnum(m.input_i); li->o["true_peak_dbtp"] = jnum(m.input_tp); li->o["lra_lu"] = jnum(m.input_lra);
        li->o["thresh_lufs"] = jnum(m.input_thresh); li->o["target_offset_db"] = jnum(m.target_offset);
        lst->o["input"] = li; dst->o["input"] = dynamics(m.dynamics); sst->o["input"] = spectral(m.spectral);
    }
    const bool have2 = !analysis_only && res->filtered.n_frames_meta > 0;
    const bool have4 = !analysis_only && res->final_.n_frames_meta > 0;
    if (have2) { lst->o["filtered"] = out_loudness(res->filtered.r128); dst->o["filtered"] = dynamics(convert_astats(res->filtered.astats)); sst->o["filtered"] = spectral(res->filtered.spectral_mean); }
    if (have4) { lst->o["final"] = out_loudness(res->final_.r128); dst->o["final"] = dynamics(convert_astats(res->final_.astats)); sst->o["final"] = spectral(res->final_.spectral_mean); }
    { JP l = jobj(); l->o["target_i_lufs"] = jnum(-16.0); l->o["stages"] = lst; root->o["loudness"] = l; }
    { JP d = jobj(); d->o["stages"] = dst; root->o["dynamics"] = d; }
    { JP s = jobj(); s->o["stages"] = sst; root->o["spectral"] = s; }
    {
        JP n = jobj();
        n->o["floor_dbfs"] = jnum(m.floor); n->o["floor_source"] = jstr(floor_source_name(m.floor_source)); n->o["floor_prescan_dbfs"] = jnum(m.floor_prescan);
        n->o["floor_astats_dbfs"] = jnum(m.floor_astats); n->o["room_tone_detect_level_dbfs"] = jnum(m.room_tone_detect_level);
        n->o["voice_activated"] = jbool(m.voice_activated != 0); n->o["floored_fraction"] = jnum(m.floored_fraction); n->o["reduction_headroom_db"] = jnum(m.reduction_headroom);
        root->o["noise"] = n;
    }
    {
        JP rg = jobj(), rt = jobj(), sp = jobj(), rts = jobj(), sps = jobj();
        if (m.has_noise_profile) {
            const jt_noise_profile &p = m.noise_profile;
            JP e = jobj();
            e->o["start_s"] = jnum(secs(p.start_ns)); e->o["duration_s"] = jnum(secs(p.duration_ns));
            e->o["measured_floor_dbfs"] = jnum(p.measured_noise_floor); e->o["peak_level_dbfs"] = jnum(p.peak_level); e->o["crest_factor_db"] = jnum(p.crest_factor);
            e->o["entropy"] = jnum(p.entropy);
            if (p.warning == 1 || p.warning == 2) {
                char b[160];
                if (p.warning == 1) snprintf(b, sizeof b, "using short room tone region (%.1fs) - ideally need >=%ds", secs(p.duration_ns), 8);
                else snprintf(b, sizeof b, "using long room tone region (%.1fs) - ideally <=%ds", secs(p.duration_ns), 18);
                e->o["extraction_warning"] = jstr(b);
            }
            const jt_spectral &s = p.spectral;
            e->o["spectral_mean"] = jnum(s.mean); e->o["spectral_variance"] = jnum(s.variance); e->o["spectral_centroid_hz"] = jnum(s.centroid);
            e->o["spectral_spread_hz"] = jnum(s.spread); e->o["spectral_skewness"] = jnum(s.skewness); e->o["spectral_kurtosis"] = jnum(s.kurtosis);
            e->o["spectral_entropy"] = jnum(s.entropy); e->o["spectral_flatness"] = jnum(s.flatness); e->o["spectral_crest"] = jnum(s.crest);
            e->o["spectral_flux"] = jnum(s.flux); e->o["spectral_slope"] = jnum(s.slope); e->o["spectral_decrease"] = jnum(s.decrease);
            e->o["spectral_rolloff_hz"] = jnum(s.rolloff);
            if (p.band_noise_n > 0) { JP a = jarr(); for (int i = 0; i < p.band_noise_n; ++i) a->a.push_back(jnum(p.band_noise[i])); e->o["band_noise_dbfs"] = a; }
            if (p.bands_measured) e->o["band_noise_measured"] = jbool(true);
            rt->o["elected"] = e;
        }
        if (m.has_room_tone_sample) rts->o["input"] = region_sample_in(m.room_tone_sample);
        if (m.has_speech_profile) {
            const jt_speech_candidate &c = m.speech_profile;
            JP e = jobj(), r = jobj();
            r->o["start_s"] = jnum(secs(c.region.start_ns)); r->o["end_s"] = jnum(secs(c.region.end_ns)); r->o["duration_s"] = jnum(secs(c.region.duration_ns));
            e->o["region"] = r;
            region_sample_fields(e, c.sample.rms_level, c.sample.peak_level, c.sample.crest_factor, c.sample.spectral, c.sample.momentary_lufs,
                                 c.sample.shortterm_lufs, c.sample.true_peak, c.sample.sample_peak);
            if (c.voicing_density != 0) e->o["voicing_density"] = jnum(c.voicing_density);
            if (c.body_band_rms != 0) e->o["speech_band_body_rms_dbfs"] = jnum(c.body_band_rms);
            if (c.sib_band_rms != 0) e->o["speech_band_sib_rms_dbfs"] = jnum(c.sib_band_rms);
            if (c.bands_measured) e->o["speech_bands_measured"] = jbool(true);
            e->o["score"] = jnum(c.score);
            if (c.original_start_ns != 0) e->o["original_start_s"] = jnum(secs(c.original_start_ns));
            if (c.original_duration_ns != 0) e->o["original_duration_s"] = jnum(secs(c.original_duration_ns));
            if (c.was_refined) e->o["was_refined"] = jbool(true);
            sp->o["elected"] = e;
            sps->o["input"] = region_sample_in(c.sample);
        }
        if (m.n_candidates > 0) {
            JP cs = jobj(); cs->o["evaluated_count"] = jint(m.n_candidates);
            if (m.has_speech_profile) cs->o["elected_score"] = jnum(m.speech_profile.score);
            sp->o["candidates_summary"] = cs;
        }
        if (!analysis_only && res->has_region_samples) {
            if (have2 && res->filtered_room_tone.frames > 0) rts->o["filtered"] = region_sample_out(res->filtered_room_tone);
            if (have2 && res->filtered_speech.frames > 0) sps->o["filtered"] = region_sample_out(res->filtered_speech);
            if (have4 && res->final_room_tone.frames > 0) rts->o["final"] = region_sample_out(res->final_room_tone);
            if (have4 && res->final_speech.frames > 0) sps->o["final"] = region_sample_out(res->final_speech);
        }
        rt->o["samples"] = rts; sp->o["samples"] = sps;
        JP gs = jobj();
        gs->o["voiced_low_percentile_dbfs"] = jnum(m.voiced_low_percentile); gs->o["noise_high_percentile_dbfs"] = jnum(m.noise_high_percentile);
        gs->o["gate_separation_db"] = jnum(m.gate_separation_db);
        rg->o["room_tone"] = rt; rg->o["speech"] = sp; rg->o["gate_statistics"] = gs;
        root->o["regions"] = rg;
    }
    if (!analysis_only) {
        // filters: EffectiveFilterConfig (json tags filters.go:119-237), gate threshold / range as honest dB (newFiltersBlock), diagnostics
        const jt_host_config &c = res->effective; const jt_adaptive_diag &d = res->diag;
        auto biquad = [](const jt_biquad_cfg &b) {
            JP o = jobj(); o->o["enabled"] = jbool(b.enabled != 0); o->o["frequency_hz"] = jnum(b.frequency); o->o["poles_count"] = jint(b.poles);
            o->o["width"] = jnum(b.width); o->o["mix"] = jnum(b.mix); o->o["transform"] = jstr(b.transform_tdii ? "tdii" : ""); return o; };
        JP f = jobj();
        f->o["rumble_highpass"] = biquad(c.rumble_hp); f->o["bandlimit_lowpass"] = biquad(c.bandlimit_lp);
        {
            JP n = jobj();
            n->o["enabled"] = jbool(c.nr_enabled != 0); n->o["strength"] = jnum(c.nr_strength); n->o["patch_s"] = jnum(c.nr_patch_s);
            n->o["research_s"] = jnum(c.nr_research_s); n->o["smooth"] = jnum(c.nr_smooth);
            n->o["afftdn_enabled"] = jbool(c.afftdn_enabled != 0); n->o["afftdn_noise_reduction_db"] = jnum(c.afftdn_nr);
            n->o["afftdn_noise_type"] = jstr(c.afftdn_custom ? "custom" : "w"); n->o["afftdn_track_noise"] = jbool(c.afftdn_track_noise != 0);
            n->o["afftdn_noise_floor_db"] = jnum(c.afftdn_noise_floor);
            if (c.afftdn_band_noise[0]) n->o["afftdn_band_noise"] = jstr(c.afftdn_band_noise);
            f->o["noise_reduction"] = n;
        }
        {
            JP g = jobj();
            g->o["enabled"] = jbool(c.gate_enabled != 0);
            g->o["threshold_db"] = jnum(c.gate_threshold > 0 ? 20.0 * std::log10(c.gate_threshold) : c.gate_threshold);
            g->o["ratio"] = jnum(c.gate_ratio); g->o["attack_ms"] = jnum(c.gate_attack); g->o["release_ms"] = jnum(c.gate_release);
            g->o["range_db"] = jnum(c.gate_range > 0 ? 20.0 * std::log10(c.gate_range) : c.gate_range);
            g->o["knee"] = jnum(c.gate_knee); g->o["makeup"] = jnum(c.gate_makeup); g->o["detection"] = jstr(c.gate_detection_set ? "rms" : "");
            f->o["speech_gate"] = g;
        }
        {
            JP k = jobj();
            k->o["enabled"] = jbool(c.comp_enabled != 0); k->o["threshold_db"] = jnum(c.comp_threshold_db); k->o["ratio"] = jnum(c.comp_ratio);
            k->o["attack_ms"] = jnum(c.comp_attack); k->o["release_ms"] = jnum(c.comp_release); k->o["makeup_db"] = jnum(c.comp_makeup_db);
            k->o["knee"] = jnum(c.comp_knee); k->o["mix"] = jnum(c.comp_mix);
            f->o["levelling_compressor"] = k;
        }
        {
            JP e = jobj();
            e->o["enabled"] = jbool(c.deess_enabled != 0); e->o["intensity"] = jnum(c.deess_intensity); e->o["amount"] = jnum(c.deess_amount);
            e->o["frequency"] = jnum(c.deess_frequency);
            f->o["deesser"] = e;
        }
        {
            JP g = jobj();
            g->o["bandlimit_lowpass_reason"] = jstr("20.5 kHz band-limit (always on)");
            g->o["dynamic_range_db"] = jnum(0.0);
            g->o["quiet_speech_estimate_dbfs"] = jnum(d.gate_quiet_speech_estimate); g->o["separation_db"] = jnum(d.gate_separation);
            g->o["speech_headroom_db"] = jnum(d.gate_speech_headroom); g->o["threshold_unclamped_db"] = jnum(d.gate_threshold_unclamped);
            g->o["clamp_reason"] = jstr(m.has_speech_profile ? (d.gate_narrow_gap ? "narrow_gap" : "none") : "");
            g->o["speech_gate_depth_db"] = jnum(d.gate_depth_db); g->o["narrow_gap"] = jbool(d.gate_narrow_gap != 0);
            g->o["afftdn_enabled"] = jbool(d.afftdn_enabled != 0); g->o["afftdn_noise_floor_db"] = jnum(d.afftdn_noise_floor_db);
            g->o["afftdn_disable_reason"] = jstr(d.afftdn_disabled_voice_activated ? "voice_activated" : "");
            g->o["afftdn_noise_type"] = jstr((d.afftdn_disabled_voice_activated || m.floor == 0) ? "" : (d.afftdn_custom ? "custom" : "w"));
            f->o["diagnostics"] = g;
        }
        root->o["filters"] = f;
        if (res->effective.loudnorm_enabled && have4) {
            // normalisation: NormalisationResult (normalise.go:649-674) + embedded LimiterDiagnostics, region_measurement_s, numeric loudnorm_measured
            const jt_limiter_decision &l = res->limiter;
            JP n = jobj();
            n->o["input_lufs"] = jnum(res->measure.input_i); n->o["input_dbtp"] = jnum(res->measure.input_tp);
            n->o["output_lufs"] = jnum(res->output_lufs); n->o["output_dbtp"] = jnum(res->output_tp_db);
            n->o["gain_applied_db"] = jnum(res->offset); n->o["within_target"] = jbool(res->within_target != 0); n->o["skipped"] = jbool(false);
            n->o["requested_target_lufs"] = jnum(res->effective.target_i); n->o["effective_target_lufs"] = jnum(res->effective_target_i);
            n->o["linear_mode_forced"] = jbool(res->linear_possible == 0); n->o["actual_norm_dynamic"] = jbool(res->loudnorm.normalization_type_dynamic != 0);
            n->o["limiter_enabled"] = jbool(l.needed != 0); n->o["ceiling_dbtp"] = jnum(l.ceiling_db); n->o["gain_db"] = jnum(l.gain_db);
            n->o["filtered_dbtp"] = jnum(l.filtered_tp); n->o["pre_gain_db"] = jnum(l.pre_gain_db); n->o["limiter_clamped"] = jbool(l.clamped != 0);
            n->o["pass3_filter_prefix"] = jstr(l.pass3_prefix);
            n->o["region_measurement_s"] = jnum(res->stage_ms[9] / 1e3);
            // loudnorm's JSON carries "%.2f" strings; loudnormMeasuredNumeric parses them back to numbers
            JP lm = jobj();
            const jt_loudnorm_stats &s = res->loudnorm;
            auto put = [&](const char *k, double v) { if (std::isfinite(v) || std::isinf(v)) { JP x = jnum(std::strtod(fmt2(v).c_str(), nullptr)); if (x->kind == J::Num) lm->o[k] = x; } };
            put("input_integrated_lufs", s.input_i); put("input_true_peak_dbtp", s.input_tp); put("input_lra_lu", s.input_lra); put("input_thresh_lufs", s.input_thresh);
            put("output_integrated_lufs", s.output_i); put("output_true_peak_dbtp", s.output_tp); put("output_lra_lu", s.output_lra); put("output_thresh_lufs", s.output_thresh);
            put("target_offset_db", s.target_offset);
            lm->o["normalization_type"] = jstr(s.normalization_type_dynamic ? "dynamic" : "linear");
            n->o["loudnorm_measured"] = lm;
            root->o["normalisation"] = n;
        }
    }
    // interval_summary (runrecord_summary.go): count, RMS distribution by integer index, largest gap between adjacent sorted values
    if (h && !h->last_intervals.empty()) {
        const std::vector<jt_interval> &iv = h->last_intervals;
        JP s = jobj(); s->o["count"] = jint((long long)iv.size());
        std::vector<double> v;
        for (auto &x : iv) if (x.rms_level > -120) v.push_back(x.rms_level);
        if (v.size() >= 10) {
            std::sort(v.begin(), v.end());
            const size_t n = v.size();
            JP d = jobj();
            d->o["min_dbfs"] = jnum(v[0]); d->o["p10_dbfs"] = jnum(v[n / 10]); d->o["p25_dbfs"] = jnum(v[n / 4]); d->o["p50_dbfs"] = jnum(v[n / 2]);
            d->o["p75_dbfs"] = jnum(v[n * 3 / 4]); d->o["p90_dbfs"] = jnum(v[n * 9 / 10]); d->o["max_dbfs"] = jnum(v[n - 1]);
            s->o["rms_distribution"] = d;
            double gap = 0; for (size_t i = 1; i < n; ++i) gap = std::max(gap, v[i] - v[i - 1]);
            s->o["largest_gap_db"] = jnum(gap);
        }
        root->o["interval_summary"] = s;
    }
    std::string out; emit(root, out, 0, true);
    return out;
}

static int copy_out(const std::string &s, char *buf, int64_t cap)
{
    if (buf && cap > 0) { const size_t n = std::min<size_t>(s.size(), (size_t)cap - 1); std::memcpy(buf, s.data(), n); buf[n] = 0; }
    return (int)s.size();
}

extern "C" int64_t jt_host_run_record_json(const jt_ctx *h, const jt_process_result *res, const jt_run_provenance *pv, int analysis_only, char *buf, int64_t cap)
{
    if (!res) return JT_E_INVAL;
    return copy_out(build_record(h, res, pv, analysis_only != 0), buf, cap);
}

// .intervals.jsonl: json.Encoder over IntervalSample (MarshalJSON flattens the spectral block: analyser_metrics.go:34-58), struct field
// order, one object per line.  encoding/json refuses NaN / Inf in a struct encode, so the reference never holds them here (levels
// floor at -120); a non-finite value is written as null.
static std::string num_or_null(double v) { return std::isfinite(v) ? go_float(v) : std::string("null"); }
extern "C" int64_t jt_host_intervals_jsonl(const jt_ctx *h, char *buf, int64_t cap)
{
    if (!h) return JT_E_INVAL;
    std::string out;
    for (const jt_interval &x : h->last_intervals) {
        const jt_spectral &s = x.spectral;
        out += "{\"timestamp\":" + std::to_string((long long)x.timestamp_ns) + ",\"rms_level\":" + num_or_null(x.rms_level) + ",\"peak_level\":" + num_or_null(x.peak_level);
        const char *names[13] = {"mean", "variance", "centroid", "spread", "skewness", "kurtosis", "entropy", "flatness", "crest", "flux", "slope", "decrease", "rolloff"};
        const double *pv = &s.mean;
        for (int k = 0; k < 13; ++k) out += std::string(",\"spectral_") + names[k] + "\":" + num_or_null(pv[k]);
        out += ",\"momentary_lufs\":" + num_or_null(x.momentary_lufs) + ",\"short_term_lufs\":" + num_or_null(x.shortterm_lufs) +
               ",\"true_peak\":" + num_or_null(x.true_peak) + ",\"sample_peak\":" + num_or_null(x.sample_peak) + "}\n";
    }
    return copy_out(out, buf, cap);
}

// .candidates.jsonl: {"kind":"speech", <the candidate's sanitised map, keys sorted>} per line (runrecord_write.go:47-72)
extern "C" int64_t jt_host_candidates_jsonl(const jt_process_result *res, char *buf, int64_t cap)
{
    if (!res) return JT_E_INVAL;
    std::string out;
    const jt_measurements &m = res->input;
    for (int i = 0; i < m.n_candidates; ++i) {
        const jt_speech_candidate &c = m.candidates[i];
        JP e = jobj(), r = jobj();
        // (the sidecar keeps time.Duration nanoseconds: only the record's elected-profile wrapper converts to seconds)
        r->o["start"] = jint(c.region.start_ns); r->o["end"] = jint(c.region.end_ns); r->o["duration"] = jint(c.region.duration_ns);
        e->o["region"] = r;
        region_sample_fields(e, c.sample.rms_level, c.sample.peak_level, c.sample.crest_factor, c.sample.spectral, c.sample.momentary_lufs,
                             c.sample.shortterm_lufs, c.sample.true_peak, c.sample.sample_peak);
        if (c.voicing_density != 0) e->o["voicing_density"] = jnum(c.voicing_density);
        if (c.body_band_rms != 0) e->o["speech_band_body_rms_dbfs"] = jnum(c.body_band_rms);
        if (c.sib_band_rms != 0) e->o["speech_band_sib_rms_dbfs"] = jnum(c.sib_band_rms);
        if (c.bands_measured) e->o["speech_bands_measured"] = jbool(true);
        e->o["score"] = jnum(c.score);
        if (c.original_start_ns != 0) e->o["original_start"] = jint(c.original_start_ns);
        if (c.original_duration_ns != 0) e->o["original_duration"] = jint(c.original_duration_ns);
        if (c.was_refined) e->o["was_refined"] = jbool(true);
        std::string body; emit(e, body, 0, false);
        out += "{\"kind\":\"speech\"," + body.substr(1) + "\n";
    }
    return copy_out(out, buf, cap);
}

// loudnorm print_format=json as af_loudnorm.c prints it (the body parseLoudnormStatsFile reads: normalise.go:143-165, fixture
// normalise_statsfile_test.go:54-55): ten string fields, "%.2f"
extern "C" int jt_host_loudnorm_json(const jt_loudnorm_stats *s, char *buf, int cap)
{
    if (!s) return JT_E_INVAL;
    auto f = [](double v) { return std::isnan(v) ? std::string("nan") : (std::isinf(v) ? std::string(v < 0 ? "-inf" : "inf") : fmt2(v)); };
    std::string o = "{\n\t\"input_i\" : \"" + f(s->input_i) + "\",\n\t\"input_tp\" : \"" + f(s->input_tp) + "\",\n\t\"input_lra\" : \"" + f(s->input_lra) +
                    "\",\n\t\"input_thresh\" : \"" + f(s->input_thresh) + "\",\n\t\"output_i\" : \"" + f(s->output_i) + "\",\n\t\"output_tp\" : \"" + f(s->output_tp) +
                    "\",\n\t\"output_lra\" : \"" + f(s->output_lra) + "\",\n\t\"output_thresh\" : \"" + f(s->output_thresh) +
                    "\",\n\t\"normalization_type\" : \"" + (s->normalization_type_dynamic ? "dynamic" : "linear") + "\",\n\t\"target_offset\" : \"" + f(s->target_offset) + "\"\n}\n";
    return copy_out(o, buf, cap);
}
