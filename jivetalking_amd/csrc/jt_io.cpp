// jt_io.cpp — the file legs of the C ABI: FLAC / WAV file image -> PCM on the device (jt_load_audio) and stage output -> finished
// .flac image (jt_flac_encode).  Host side only: metadata / chunk walks, the frame chain walk, STREAMINFO; the kernels are in
// k_flacdec.hip and k_flac.hip.  Reference interfaces replaced: reader.go:29-169 (OpenAudioFile / ReadFrame), encoder.go:54-215.
#include "jt_internal.h"
#include <algorithm>
#include <chrono>
#include <vector>

// ---------------------------------------------------------------- FLAC output leg (k_flac.hip)
namespace {
double flac_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct FlacSummary { long long total; int min_frame, max_frame, mismatches, pad; };

void put_be(uint8_t *&p, uint64_t v, int bytes) { for (int i = bytes - 1; i >= 0; i--) *p++ = (uint8_t)(v >> (8 * i)); }
void put_le32(uint8_t *&p, uint32_t v) { for (int i = 0; i < 4; i++) *p++ = (uint8_t)(v >> (8 * i)); }

// d_pcm: n mono s16 samples on the device
void flac_encode_core(jt_ctx *h, const int16_t *d_pcm, int64_t n, int rate, int flags, const uint8_t **data, int64_t *len,
                      jt_flac_info *info)
{
    JT_REQUIRE(n > 0, JT_E_STATE, "flac: no samples to encode");
    JT_REQUIRE(n < ((int64_t)1 << 36), JT_E_INVAL, "flac: more samples than STREAMINFO can describe");
    JT_REQUIRE(rate > 0 && rate < (1 << 20), JT_E_INVAL, "flac: sample rate out of range");
    const double t0 = flac_now_ms();
    const int64_t nframes = (n + 4095) / 4096;
    JT_REQUIRE(nframes < ((int64_t)1 << 31), JT_E_INVAL, "flac: too many frames");
    // sample-rate code of the frame header (RFC 9639 §9.1.2)
    static const int rates[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
    int sr_code = 0, sr_bytes = 0, sr_val = 0;
    for (int c = 1; c < 12; c++) if (rates[c] == rate) sr_code = c;
    if (!sr_code) {
        if (rate % 1000 == 0 && rate / 1000 < 256) { sr_code = 12; sr_bytes = 1; sr_val = rate / 1000; }
        else if (rate < 65536) { sr_code = 13; sr_bytes = 2; sr_val = rate; }
        else if (rate % 10 == 0 && rate / 10 < 65536) { sr_code = 14; sr_bytes = 2; sr_val = rate / 10; }
    }
    const bool defer_md5 = (flags & JT_FLAC_MD5_DEFER) != 0, want_md5 = (flags & JT_FLAC_MD5) != 0 || defer_md5;
    h->flac_deferred = {};
    h->fl_rec.ensure(jt_flac_rec_bytes(nframes));
    h->fl_off.ensure((size_t)nframes + 8);
    unsigned char *d_sum = h->fl_rec.p + (jt_flac_rec_bytes(nframes) - 256);
    // the PCM travels to the host on an auxiliary stream while the analysis runs (only the MD5 needs it)
    int16_t *h_pcm = nullptr;
    // (a pool's Pass 4 has copied the PCM already and somebody is hashing it: jt_ctx::p4_output_hook)
    const bool pcm_there = defer_md5 && h->pcm_early.pcm && h->pcm_early.n == (size_t)n && d_pcm == h->s16_p4.p;
    if (want_md5 && !pcm_there) {
        h->pin_pcm().begin(sizeof(int16_t) * (size_t)n + 64);
        h_pcm = h->pin_pcm().take<int16_t>((size_t)n);
        JT_HIP(hipEventRecord(h->ev_fork, h->stream));
        JT_HIP(hipStreamWaitEvent(h->aux[0], h->ev_fork, 0));
        JT_HIP(hipMemcpyAsync(h_pcm, d_pcm, sizeof(int16_t) * (size_t)n, hipMemcpyDeviceToHost, h->aux[0]));
    }
    JT_HIP(hipEventRecord(h->ev0, h->stream));
    launch_flac_analyse(d_pcm, n, sr_code, sr_bytes, sr_val, h->fl_rec.p, h->fl_off.p, d_sum, h->stream);
    h->io_small.begin(4096);
    FlacSummary &sum = *h->io_small.take<FlacSummary>(1), &sum2 = *h->io_small.take<FlacSummary>(1);
    sum = FlacSummary{}; sum2 = FlacSummary{};
    JT_HIP(hipMemcpyAsync(&sum, d_sum, sizeof sum, hipMemcpyDeviceToHost, h->stream));
    JT_HIP(jt_stream_sync(h, h->stream));
    JT_REQUIRE(sum.total > 0, JT_E_HIP, "flac: analysis produced no frames");

    static const char vendor[] = "jivetalking-amd jtgpu 0.1";
    const int vlen = (int)sizeof(vendor) - 1;
    const int header_bytes = 4 + 4 + 34 + 4 + 4 + vlen + 4;
    h->fl_out.ensure((size_t)sum.total + 64);
    h->pin_flac().begin((size_t)header_bytes + (size_t)sum.total + 64);
    uint8_t *file = h->pin_flac().take<uint8_t>((size_t)header_bytes + (size_t)sum.total);
    launch_flac_emit(d_pcm, n, sr_code, sr_bytes, sr_val, h->fl_rec.p, h->fl_off.p, h->fl_out.p, d_sum, h->stream);
    JT_HIP(hipEventRecord(h->ev1, h->stream));
    JT_HIP(hipMemcpyAsync(file + header_bytes, h->fl_out.p, (size_t)sum.total, hipMemcpyDeviceToHost, h->stream));
    JT_HIP(hipMemcpyAsync(&sum2, d_sum, sizeof sum2, hipMemcpyDeviceToHost, h->stream));

    uint8_t md5[16] = {0};
    double md5_ms = 0.0;
    if (want_md5 && !pcm_there) {
        JT_HIP(jt_stream_sync(h, h->aux[0]));
        const double m0 = flac_now_ms();
        if (defer_md5) { h->flac_deferred.pcm = h_pcm; h->flac_deferred.n = (size_t)n; }
        else jt_md5(h_pcm, sizeof(int16_t) * (size_t)n, md5);
        md5_ms = flac_now_ms() - m0;
    }
    JT_HIP(jt_stream_sync(h, h->stream));
    JT_REQUIRE(sum2.mismatches == 0, JT_E_HIP, "flac: emitted size differs from the analysed size");
    float gpu_ms = 0.f;
    JT_HIP(hipEventElapsedTime(&gpu_ms, h->ev0, h->ev1));

    // fLaC marker, STREAMINFO (RFC 9639 §8.2), VORBIS_COMMENT (§8.6, vendor string only)
    uint8_t *p = file;
    memcpy(p, "fLaC", 4); p += 4;
    put_be(p, 0x00, 1); put_be(p, 34, 3);
    put_be(p, 4096, 2); put_be(p, 4096, 2);
    put_be(p, (uint64_t)sum.min_frame, 3); put_be(p, (uint64_t)sum.max_frame, 3);
    // 20 bits rate | 3 bits channels-1 | 5 bits depth-1 | 36 bits total samples
    put_be(p, ((uint64_t)rate << 44) | ((uint64_t)0 << 41) | ((uint64_t)15 << 36) | (uint64_t)n, 8);
    memcpy(p, md5, 16); p += 16;
    put_be(p, 0x84, 1); put_be(p, (uint64_t)(4 + vlen + 4), 3);
    put_le32(p, (uint32_t)vlen); memcpy(p, vendor, (size_t)vlen); p += vlen; put_le32(p, 0);

    *data = file; *len = header_bytes + sum.total;
    if (info) {
        memset(info, 0, sizeof *info);
        info->bytes = *len; info->frames = nframes; info->total_samples = n; info->sample_rate = rate; info->channels = 1;
        info->bits_per_sample = 16; info->block_size = 4096; info->min_frame_bytes = sum.min_frame; info->max_frame_bytes = sum.max_frame;
        info->header_bytes = header_bytes; info->gpu_ms = gpu_ms; info->md5_ms = md5_ms; info->total_ms = flac_now_ms() - t0;
        memcpy(info->md5, md5, 16);
    }
}
} // namespace

static void flac_encode_stage(jt_ctx *h, int stage, int flags, const uint8_t **data, int64_t *len, jt_flac_info *info)
{
    JT_REQUIRE(data && len, JT_E_INVAL, "flac: null output arguments");
    const int16_t *src = stage == 2 ? h->s16_p2.p : (stage == 4 ? h->s16_p4.p : nullptr);
    const int64_t m = stage == 2 ? h->m_p2 : (stage == 4 ? h->m_p4 : 0);
    JT_REQUIRE(src && m > 0, JT_E_STATE, "flac: stage output not on device");
    flac_encode_core(h, src, m, h->out_rate, flags, data, len, info);
}
extern "C" int jt_flac_encode(jt_ctx *h, int stage, int flags, const uint8_t **data, int64_t *len, jt_flac_info *info)
{
    JT_API_BEGIN(h)
    flac_encode_stage(h, stage, flags & JT_FLAC_MD5, data, len, info);
    JT_API_END(h)
}
// jt_process_file's encoder call: with JT_FLAC_MD5_DEFER the signature is left zero and h->flac_deferred names the pinned PCM to hash
int jt_flac_encode_file(jt_ctx *h, int stage, int flags, const uint8_t **data, int64_t *len, jt_flac_info *info)
{
    JT_API_BEGIN(h)
    flac_encode_stage(h, stage, flags & (JT_FLAC_MD5 | JT_FLAC_MD5_DEFER), data, len, info);
    JT_API_END(h)
}

extern "C" int jt_op_flac_encode_s16(jt_ctx *h, const int16_t *pcm, int64_t n, int sample_rate, int flags,
                                     const uint8_t **data, int64_t *len, jt_flac_info *info)
{
    JT_API_BEGIN(h)
    JT_REQUIRE(pcm && n > 0 && data && len, JT_E_INVAL, "flac: bad arguments");
    h->fl_pcm.ensure((size_t)n);
    JT_HIP(hipMemcpyAsync(h->fl_pcm.p, pcm, sizeof(int16_t) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    flac_encode_core(h, h->fl_pcm.p, n, sample_rate, flags, data, len, info);
    JT_API_END(h)
}

// ---------------------------------------------------------------- input leg: FLAC / WAV file image -> PCM on the device
namespace {
struct AudioDecoded { int64_t frames = 0; int channels = 0, rate = 0, bits = 0, is_float = 0, format = 0; int64_t flac_frames = 0; int cands = 0;
                      // the decoder's frame cadence (jtgpu.h: jt_audio_meta): constant length, or per-frame lengths when they differ
                      int dec_frame_samples = 4096; int64_t dec_frames = 0; std::vector<int32_t> frame_lens;
                      unsigned long long ch_mask = 0; };      // 0: no layout in the file (swr_init then takes the default of the channel count)

uint32_t rd_le32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
uint32_t rd_le16(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }

// FLAC: metadata walk on the host (a few hundred bytes), everything else on the device
void decode_flac(jt_ctx *h, const uint8_t *file, int64_t len, int64_t start, bool want_i32, bool want_f32, AudioDecoded *out)
{
    JtFlacStream st; st.len = len;
    const double tm0 = flac_now_ms(); double tm1 = tm0, tm2 = tm0, tm3 = tm0;
    int64_t pos = start + 4; bool last = false, have = false;
    while (!last) {
        JT_REQUIRE(pos + 4 <= len, JT_E_INVAL, "flac: truncated metadata");
        last = (file[pos] & 0x80) != 0;
        const int type = file[pos] & 0x7f;
        const int64_t blen = (int64_t)file[pos + 1] << 16 | (int64_t)file[pos + 2] << 8 | file[pos + 3];
        pos += 4;
        JT_REQUIRE(pos + blen <= len, JT_E_INVAL, "flac: truncated metadata block");
        JT_REQUIRE(type != 127, JT_E_INVAL, "flac: invalid metadata block type");
        if (type == 0) {
            JT_REQUIRE(blen == 34, JT_E_INVAL, "flac: bad STREAMINFO length");
            const uint8_t *p = file + pos;
            st.min_blocksize = p[0] << 8 | p[1]; st.max_blocksize = p[2] << 8 | p[3];
            st.sample_rate = p[10] << 12 | p[11] << 4 | p[12] >> 4;
            st.channels = ((p[12] >> 1) & 7) + 1;
            st.bps = (((p[12] & 1) << 4) | (p[13] >> 4)) + 1;
            st.total_samples = (int64_t)(p[13] & 15) << 32 | (int64_t)p[14] << 24 | (int64_t)p[15] << 16 | (int64_t)p[16] << 8 | p[17];
            have = true;
        }
        pos += blen;
    }
    JT_REQUIRE(have, JT_E_INVAL, "flac: no STREAMINFO block");
    JT_REQUIRE(st.bps >= 4 && st.bps <= 24, JT_E_UNSUPPORTED, "flac: only 4..24 bits per sample are supported");
    JT_REQUIRE(st.max_blocksize >= 16 && st.min_blocksize <= st.max_blocksize, JT_E_INVAL, "flac: bad block sizes in STREAMINFO");
    JT_REQUIRE(st.sample_rate > 0, JT_E_INVAL, "flac: sample rate 0");
    st.audio_offset = pos;
    JT_REQUIRE(pos < len, JT_E_INVAL, "flac: no audio frames");

    // file image on the device, zero padded so header probes and bit-reader refills never leave the allocation
    const size_t padded = ((size_t)len + 1024 + 15) & ~(size_t)15;      // the bit readers fetch up to 64 dwords ahead
    h->in_file.ensure(padded);
    JT_HIP(hipMemsetAsync(h->in_file.p + ((size_t)len & ~(size_t)3), 0, padded - ((size_t)len & ~(size_t)3), h->stream));
    JT_HIP(hipMemcpyAsync(h->in_file.p, file, (size_t)len, hipMemcpyHostToDevice, h->stream));

    // candidates: expected frame count plus room for look-alikes
    const int64_t expect = st.total_samples > 0 ? st.total_samples / std::max(16, st.min_blocksize) + 2 : (len - pos) / 16 + 2;
    int cap = (int)std::min<int64_t>((int64_t)1 << 28, expect * 2 + (len - pos) / 2048 + 4096);
    // (device -> host results land in the handle's small pinned arena: a copy into pageable memory waits inside the runtime, spinning)
    JtFlacCand *cands = nullptr; JtFlacParsed *parsed = nullptr;
    int *counts = nullptr;                                    // [0] candidates found, [1] subframe decode errors
    int ncand = 0;
    bool ahead = false;                                       // mono: the candidates' samples are decoded already, one row each
    // The DEVICE tables are sized for the capacity; the pinned host copies only for the candidates actually found (a stream
    // without a sample count or with a tiny declared block size makes the capacity ~ len / 8 entries of 112 bytes -- gigabytes of
    // pinned memory per handle for a table that then holds a few thousand rows).  A table that cannot be pinned is a verdict about
    // the FILE (JT_E_UNSUPPORTED), not about the handle.
    h->io_small.begin(4096);                                                                            // (nothing of the arena is in flight here)
    counts = h->io_small.take<int>(16); counts[0] = counts[1] = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        const size_t tab_bytes = 256 + sizeof(JtFlacCand) * (size_t)cap + sizeof(JtFlacParsed) * (size_t)cap;
        h->in_tab.ensure(tab_bytes);
        int *d_count = reinterpret_cast<int *>(h->in_tab.p);
        JtFlacCand *d_cand = reinterpret_cast<JtFlacCand *>(h->in_tab.p + 256);
        JtFlacParsed *d_parsed = reinterpret_cast<JtFlacParsed *>(h->in_tab.p + 256 + sizeof(JtFlacCand) * (size_t)cap);
        JT_HIP(hipMemsetAsync(d_count, 0, 256, h->stream));
        launch_flacdec_find(h->in_file.p, st, d_cand, d_count, cap, h->stream);
        JT_HIP(hipMemcpyAsync(&counts[0], d_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        JT_HIP(jt_stream_sync(h, h->stream));
        ncand = counts[0]; tm1 = flac_now_ms();
        if (ncand > cap) { cap = ncand + 1024; continue; }
        JT_REQUIRE(ncand > 0, JT_E_INVAL, "flac: no frame headers found");
        const size_t host_tab = (sizeof(JtFlacCand) + sizeof(JtFlacParsed)) * (size_t)ncand + 4096 + 256;
        JT_REQUIRE(host_tab <= ((size_t)1 << 31), JT_E_UNSUPPORTED, "flac: too many frame-header candidates for this file (not a stream this decoder takes)");
        try { h->io_small.begin(host_tab); }
        catch (const JtError &) { throw JtError{JT_E_UNSUPPORTED, "flac: the candidate table of this file cannot be pinned"}; }
        counts = h->io_small.take<int>(16); counts[0] = ncand; counts[1] = 0;
        cands = h->io_small.take<JtFlacCand>((size_t)ncand); parsed = h->io_small.take<JtFlacParsed>((size_t)ncand);
        // mono: one walk per candidate parses AND decodes it, into a row of its own (k_flacdec.hip, k_flac_decode_cand)
        ahead = st.channels == 1 && !h->opts.flac_no_ahead && (int64_t)ncand * st.max_blocksize < ((int64_t)1 << 31);
        if (ahead) {
            h->in_planar.ensure((size_t)ncand * (size_t)st.max_blocksize + 64);
            launch_flacdec_decode_cand(h->in_file.p, st, d_cand, ncand, h->in_planar.p, d_parsed, h->stream);
        } else launch_flacdec_parse(h->in_file.p, st, d_cand, ncand, d_parsed, h->stream);
        JT_HIP(hipMemcpyAsync(cands, d_cand, sizeof(JtFlacCand) * (size_t)ncand, hipMemcpyDeviceToHost, h->stream));
        JT_HIP(hipMemcpyAsync(parsed, d_parsed, sizeof(JtFlacParsed) * (size_t)ncand, hipMemcpyDeviceToHost, h->stream));
        JT_HIP(jt_stream_sync(h, h->stream));
        break;
    }
    JT_REQUIRE(ncand <= cap, JT_E_HIP, "flac: candidate table overflow");
    tm2 = flac_now_ms();

    // follow end -> start links from the first frame; look-alike headers inside audio data are never reached.  The candidates arrive in
    // the order the find kernel's workgroups appended them: sorted by position once (keys in ordinary memory: the pinned arena is for the
    // device's writes, not for a sort), the chain is then ONE forward walk -- a frame's successor starts behind it, so the cursor into
    // the sorted keys only ever advances.
    // (a counting sort over 2 KB buckets of the file -- frames are kilobytes apart, a bucket holds one or two -- then an insertion
    //  pass that moves almost nothing: 42 000 candidates of an hour's file in 0.3 ms where std::sort took 2)
    std::vector<std::pair<int64_t, int>> order((size_t)ncand);
    {
        const size_t nbk = (size_t)(len >> 11) + 2;
        std::vector<int> head(nbk + 1, 0);
        for (int i = 0; i < ncand; i++) {
            const int64_t ps = cands[i].pos;
            JT_REQUIRE(ps >= 0 && ps < len, JT_E_HIP, "flac: candidate position outside the file");
            head[(size_t)(ps >> 11) + 1]++;
        }
        for (size_t b = 0; b < nbk; b++) head[b + 1] += head[b];
        for (int i = 0; i < ncand; i++) { const int64_t ps = cands[i].pos; order[(size_t)head[(size_t)(ps >> 11)]++] = {ps, i}; }
        for (size_t i = 1; i < order.size(); i++) {
            const auto v = order[i]; size_t j = i;
            while (j > 0 && order[j - 1] > v) { order[j] = order[j - 1]; --j; }
            order[j] = v;
        }
    }
    std::vector<JtFlacFrame> frames; frames.reserve((size_t)ncand);
    int64_t cur = st.audio_offset, total = 0; int variable = -1;
    size_t at = 0;
    while (cur < len) {
        while (at < order.size() && order[at].first < cur) ++at;
        if (at == order.size() || order[at].first != cur) break;
        const int ci = order[at].second;
        const JtFlacCand &c = cands[ci]; const JtFlacParsed &pr = parsed[ci];
        if (!pr.ok) break;
        if (variable < 0) variable = c.variable;
        JT_REQUIRE(c.variable == variable, JT_E_INVAL, "flac: blocking strategy changes inside the stream");
        if (variable) JT_REQUIRE(c.number == total, JT_E_INVAL, "flac: frame sample number out of sequence");
        else JT_REQUIRE(c.number == (int64_t)frames.size(), JT_E_INVAL, "flac: frame number out of sequence");
        JtFlacFrame f; f.pos = c.pos; f.out_offset = total; f.blocksize = c.blocksize; f.ch_assign = c.ch_assign;
        for (int k = 0; k < 8; k++) f.sub_bit[k] = pr.sub_bit[k];
        if (ahead) f.sub_bit[7] = ci;                          // (mono: slots 1..7 are free) the candidate whose row holds this frame's samples
        frames.push_back(f);
        total += c.blocksize; cur = pr.end;
    }
    JT_REQUIRE(!frames.empty(), JT_E_INVAL, "flac: the first audio frame is damaged");
    if (st.total_samples > 0) JT_REQUIRE(total == st.total_samples, JT_E_INVAL, "flac: damaged or truncated stream (decoded sample count differs from STREAMINFO)");
    else JT_REQUIRE(cur >= len - 128, JT_E_INVAL, "flac: damaged frame inside the stream");

    tm3 = flac_now_ms();
    const size_t nvals = (size_t)total * (size_t)st.channels;
    if (!ahead) h->in_planar.ensure(nvals + 64);
    if (want_i32) h->in_i32.ensure(nvals);
    if (want_f32) h->in_owned.ensure(nvals);
    const size_t ftab = sizeof(JtFlacFrame) * frames.size();
    h->in_tab.ensure(256 + ftab);                  // the candidate tables are dead now (copied to the host above)
    int *d_err = reinterpret_cast<int *>(h->in_tab.p);
    JtFlacFrame *d_frames = reinterpret_cast<JtFlacFrame *>(h->in_tab.p + 256);
    JT_HIP(hipMemsetAsync(d_err, 0, 256, h->stream));
    JT_HIP(hipMemcpyAsync(d_frames, frames.data(), ftab, hipMemcpyHostToDevice, h->stream));
    if (ahead) {
        launch_flacdec_finish_cand(st, d_frames, (long long)frames.size(), h->in_planar.p,
                                   want_i32 ? h->in_i32.p : nullptr, want_f32 ? h->in_owned.p : nullptr, h->stream);
    } else
    launch_flacdec_decode(h->in_file.p, st, d_frames, (long long)frames.size(), total, h->in_planar.p, d_err,
                          want_i32 ? h->in_i32.p : nullptr, want_f32 ? h->in_owned.p : nullptr, h->stream);
    JT_HIP(hipMemcpyAsync(&counts[1], d_err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    JT_HIP(jt_stream_sync(h, h->stream));
    const int nerr = counts[1];
    if (jt_host_timing().load(std::memory_order_relaxed))
        fprintf(stderr, "flac decode: upload + find %.2f ms, parse + tables back %.2f, chain walk %.2f, decode %.2f (%d candidates, %zu frames)\n",
                tm1 - tm0, tm2 - tm1, tm3 - tm2, flac_now_ms() - tm3, ncand, frames.size());
    JT_REQUIRE(nerr == 0, JT_E_INVAL, "flac: subframe decode error");
    out->frames = total; out->channels = st.channels; out->rate = st.sample_rate; out->bits = st.bps; out->is_float = 0; out->format = 1;
    out->flac_frames = (int64_t)frames.size(); out->cands = ncand;
    {   // libavcodec/flac.c flac_channel_layouts[]: mono, stereo, 3.0, quad, 5.0 (side), 5.1 (side), 6.1, 7.1 -- the stream's channel
        // order is those layouts' native order (a WAVEFORMATEXTENSIBLE_CHANNEL_MASK comment that overrides it is not read)
        static const unsigned long long lay[9] = {0, 0x4, 0x3, 0x7, 0x33, 0x607, 0x60F, 0x70F, 0x63F};
        out->ch_mask = lay[st.channels];
    }
    // one AVFrame per FLAC frame (libavcodec/flacdec.c): constant when every frame but the last has the first one's length
    out->dec_frame_samples = frames[0].blocksize; out->dec_frames = (int64_t)frames.size();
    bool constant = true; int longest = 0;
    for (size_t i = 0; i < frames.size(); i++) {
        longest = std::max(longest, frames[i].blocksize);
        if (i + 1 < frames.size() ? frames[i].blocksize != frames[0].blocksize : frames[i].blocksize > frames[0].blocksize) constant = false;
    }
    if (!constant) {
        out->dec_frame_samples = longest;
        out->frame_lens.resize(frames.size());
        for (size_t i = 0; i < frames.size(); i++) out->frame_lens[i] = frames[i].blocksize;
    }
}

// RIFF/WAVE: chunk walk on the host, sample unpacking on the device
void decode_wav(jt_ctx *h, const uint8_t *file, int64_t len, bool want_i32, bool want_f32, AudioDecoded *out)
{
    JT_REQUIRE(len >= 12 && !memcmp(file + 8, "WAVE", 4), JT_E_UNSUPPORTED, "wav: not a RIFF/WAVE file");
    int64_t pos = 12; int tag = 0, ch = 0, rate = 0, bits = 0, align = 0; bool have_fmt = false;
    int64_t data_off = -1, data_len = 0;
    // RF64 / BW64 (EBU Tech 3306; libavformat's wav demuxer reads them too, wavdec.c: the "ds64" chunk): the 32-bit sizes are 0xFFFFFFFF
    // and the 64-bit ones -- RIFF size, data size, sample count -- sit in a ds64 chunk that must come first
    const bool rf64 = !memcmp(file, "RF64", 4) || !memcmp(file, "BW64", 4);
    int64_t ds64_data = -1;
    while (pos + 8 <= len) {
        const uint32_t sz = rd_le32(file + pos + 4);
        const uint8_t *body = file + pos + 8;
        if (rf64 && !memcmp(file + pos, "ds64", 4)) {
            JT_REQUIRE(sz >= 24 && pos + 8 + 24 <= len, JT_E_INVAL, "wav: short ds64 chunk");
            ds64_data = (int64_t)rd_le32(body + 8) | (int64_t)rd_le32(body + 12) << 32;
            JT_REQUIRE(ds64_data >= 0, JT_E_INVAL, "wav: bad ds64 data size");
        } else if (!memcmp(file + pos, "fmt ", 4)) {
            JT_REQUIRE(sz >= 16 && pos + 8 + 16 <= len, JT_E_INVAL, "wav: short fmt chunk");
            tag = (int)rd_le16(body); ch = (int)rd_le16(body + 2); rate = (int)rd_le32(body + 4); align = (int)rd_le16(body + 12); bits = (int)rd_le16(body + 14);
            if (tag == 0xfffe) { JT_REQUIRE(sz >= 40 && pos + 8 + 40 <= len, JT_E_INVAL, "wav: short extensible fmt chunk"); tag = (int)rd_le16(body + 24); out->ch_mask = rd_le32(body + 20); }
            have_fmt = true;
        } else if (!memcmp(file + pos, "data", 4)) {
            data_off = pos + 8;
            data_len = std::min<int64_t>((int64_t)sz, len - data_off);      // a streamed writer may leave 0xFFFFFFFF / short sizes
            if (sz == 0xffffffffu && rf64 && ds64_data >= 0) data_len = std::min<int64_t>(ds64_data, len - data_off);
            else if (sz == 0xffffffffu || sz == 0) data_len = len - data_off;
            break;
        }
        pos += 8 + (int64_t)sz + (sz & 1);
    }
    JT_REQUIRE(have_fmt && data_off >= 0, JT_E_INVAL, "wav: missing fmt or data chunk");
    JT_REQUIRE(ch >= 1 && ch <= 8 && rate > 0, JT_E_INVAL, "wav: bad channel count or rate");
    int fmt = -1;
    if (tag == 1) fmt = bits == 8 ? 0 : bits == 16 ? 1 : bits == 24 ? 2 : bits == 32 ? 3 : -1;
    else if (tag == 3) fmt = bits == 32 ? 4 : bits == 64 ? 5 : -1;
    JT_REQUIRE(fmt >= 0, JT_E_UNSUPPORTED, "wav: only PCM 8/16/24/32-bit and IEEE float 32/64-bit are supported");
    const int bytes = bits / 8;
    JT_REQUIRE(align == 0 || align == bytes * ch, JT_E_INVAL, "wav: block alignment does not match the sample format");
    const int64_t frames = data_len / ((int64_t)bytes * ch);
    JT_REQUIRE(frames > 0, JT_E_INVAL, "wav: empty data chunk");
    const size_t nvals = (size_t)frames * (size_t)ch, raw = nvals * (size_t)bytes;
    h->in_file.ensure(raw + 16);
    JT_HIP(hipMemcpyAsync(h->in_file.p, file + data_off, raw, hipMemcpyHostToDevice, h->stream));
    if (want_i32) h->in_i32.ensure(nvals);
    if (want_f32) h->in_owned.ensure(nvals);
    launch_pcm_convert(h->in_file.p, (long long)nvals, fmt, want_f32 ? h->in_owned.p : nullptr, want_i32 ? h->in_i32.p : nullptr, h->stream);
    JT_HIP(jt_stream_sync(h, h->stream));
    out->frames = frames; out->channels = ch; out->rate = rate; out->bits = bits; out->is_float = tag == 3; out->format = 2;
    // libavformat/wavdec.c wav_read_packet: size = max_size (AVOption, default 4096 bytes); with block_align > 1 at least one block and
    // rounded down to whole blocks; the PCM decoder makes one AVFrame of every packet
    {
        const int balign = bytes * ch;
        int size = 4096;
        if (balign > 1) { if (size < balign) size = balign; size = size / balign * balign; }
        out->dec_frame_samples = size / balign;
        out->dec_frames = (frames + out->dec_frame_samples - 1) / out->dec_frame_samples;
    }
}

void decode_audio(jt_ctx *h, const uint8_t *file, int64_t len, bool want_i32, bool want_f32, AudioDecoded *out, jt_audio_meta *meta)
{
    JT_REQUIRE(file && len > 12, JT_E_INVAL, "audio: empty file image");
    const double t0 = flac_now_ms();
    JT_HIP(hipEventRecord(h->ev0, h->stream));
    int64_t start = 0;
    if (!memcmp(file, "ID3", 3) && len > 10) {      // ID3v2 tag in front of the stream: 28-bit syncsafe size (+ footer)
        start = 10 + ((int64_t)(file[6] & 0x7f) << 21 | (int64_t)(file[7] & 0x7f) << 14 | (int64_t)(file[8] & 0x7f) << 7 | (file[9] & 0x7f));
        if (file[5] & 0x10) start += 10;
        JT_REQUIRE(start + 4 < len, JT_E_INVAL, "audio: ID3 tag longer than the file");
    }
    if (!memcmp(file + start, "fLaC", 4)) decode_flac(h, file, len, start, want_i32, want_f32, out);
    else if (!memcmp(file, "RIFF", 4) || !memcmp(file, "RF64", 4) || !memcmp(file, "BW64", 4)) decode_wav(h, file, len, want_i32, want_f32, out);
    else throw JtError{JT_E_UNSUPPORTED, "audio: only FLAC and RIFF/WAVE inputs are decoded on the device"};
    JT_HIP(hipEventRecord(h->ev1, h->stream));
    JT_HIP(jt_stream_sync(h, h->stream));
    float gpu_ms = 0.f;
    JT_HIP(hipEventElapsedTime(&gpu_ms, h->ev0, h->ev1));
    if (meta) {
        memset(meta, 0, sizeof *meta);
        meta->format = out->format; meta->sample_rate = out->rate; meta->channels = out->channels; meta->bits_per_sample = out->bits;
        meta->is_float = out->is_float; meta->frames = out->frames; meta->duration_s = (double)out->frames / (double)out->rate;
        meta->flac_frames = out->flac_frames; meta->flac_candidates = out->cands; meta->gpu_ms = gpu_ms; meta->total_ms = flac_now_ms() - t0;
        meta->channel_mask = out->ch_mask ? out->ch_mask : jt_default_layout(out->channels);
        meta->decoder_frame_samples = out->dec_frame_samples; meta->decoder_frames_variable = out->frame_lens.empty() ? 0 : 1; meta->decoder_frames = out->dec_frames;
    }
}
} // namespace

extern "C" int jt_load_audio(jt_ctx *h, const uint8_t *file, int64_t len, jt_audio_meta *meta)
{
    JT_API_BEGIN(h)
    AudioDecoded d;
    // the decode overwrites the owned input buffer: until it has succeeded the handle holds no input at all
    h->n = 0; h->in_raw = nullptr; h->in_mono = nullptr; h->m_p2 = h->m_p4 = 0;
    decode_audio(h, file, len, false, true, &d, meta);
    h->in_raw = h->in_owned.p;
    h->src_fmt = d.is_float ? 0 : (d.bits <= 16 ? 1 : 2);       // what libavcodec would hand to abuffer: flt/dbl, (u8/)s16, s32
    jt_set_input_common(h, d.frames, d.rate, d.channels, d.ch_mask);
    // the file's own frame cadence: what frame_samples = 0 means from here on (jt_pass1, jt_process_audio, ...)
    h->dec_frame_samples = d.dec_frame_samples; h->dec_frames = d.dec_frames; h->dec_frame_lens = std::move(d.frame_lens);
    if (!h->dec_frame_lens.empty()) {
        const size_t nf = h->dec_frame_lens.size();
        std::vector<int64_t> off(nf + 1); off[0] = 0;
        for (size_t i = 0; i < nf; i++) off[i + 1] = off[i] + h->dec_frame_lens[i];
        h->d_frame_off.ensure(nf + 1);
        JT_HIP(hipMemcpy(h->d_frame_off.p, off.data(), sizeof(int64_t) * (nf + 1), hipMemcpyHostToDevice));
    }
    JT_HIP(jt_stream_sync(h, h->stream));
    JT_API_END(h)
}

extern "C" int jt_op_decode_audio(jt_ctx *h, const uint8_t *file, int64_t len, int32_t *pcm_i32, float *pcm_f32, int64_t cap_values,
                                  jt_audio_meta *meta)
{
    JT_API_BEGIN(h)
    AudioDecoded d;
    // the decode reuses the owned input buffer: whatever input the handle held is gone afterwards (as in jt_load_audio)
    h->n = 0; h->in_raw = nullptr; h->in_mono = nullptr; h->m_p2 = h->m_p4 = 0;
    decode_audio(h, file, len, pcm_i32 != nullptr, true, &d, meta);
    const int64_t nvals = d.frames * d.channels;
    if (pcm_i32 || pcm_f32) JT_REQUIRE(cap_values >= nvals, JT_E_INVAL, "decode: output buffer too small");
    if (pcm_i32) JT_HIP(hipMemcpyAsync(pcm_i32, h->in_i32.p, sizeof(int32_t) * (size_t)nvals, hipMemcpyDeviceToHost, h->stream));
    if (pcm_f32) JT_HIP(hipMemcpyAsync(pcm_f32, h->in_owned.p, sizeof(float) * (size_t)nvals, hipMemcpyDeviceToHost, h->stream));
    JT_HIP(jt_stream_sync(h, h->stream));
    JT_API_END(h)
}

