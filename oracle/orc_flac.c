/*
 * orc_flac.c — ORACLE (test infrastructure only): FLAC stream decoder, plus a small sequential encoder used to make
 * decoder test streams.
 *
 * The reference's on-disk format on both sides of the hot path: it reads FLAC/WAV through libavformat (reader.go:61-169) and
 * writes FLAC s16, compression_level 5, 4096-sample frames through FFmpeg's flac encoder (encoder.go:54-110, asetnsamples
 * n=4096 at filters.go / normalise.go:1318-1330).  FLAC is lossless and the container is an IETF standard, so the oracle for
 * this row is the format itself: a sequential decoder restated from RFC 9639 ("Free Lossless Audio Codec"), §9 (frame
 * structure), §9.2 (subframes), §9.2.7 (coded residual), §10 (stereo decorrelation), §8.2 (STREAMINFO) and §9.3 (CRC-16
 * x^16+x^15+x^2+1, CRC-8 x^8+x^2+x+1), with every integrity field checked: header CRC-8, frame CRC-16, STREAMINFO MD5 of the
 * decoded little-endian interleaved PCM.  It is PINNED by the RFC's own worked examples (Appendix D.1 and D.3, carried in
 * tests/test_oracle_kat.py with their MD5 signatures), not by the reference executable.
 */
#include "jt_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- MD5 (RFC 1321), for the STREAMINFO signature */
typedef struct { uint32_t a, b, c, d; uint64_t len; uint8_t buf[64]; int fill; } md5_t;
static const uint32_t MD5_K[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
    0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
    0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
    0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
    0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
    0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
static const int MD5_S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                              5, 9, 14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21,
                              6, 10, 15, 21, 6, 10, 15, 21};
static void md5_block(md5_t *m, const uint8_t *p)
{
    uint32_t w[16], a = m->a, b = m->b, c = m->c, d = m->d;
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] | (uint32_t)p[4 * i + 1] << 8 | (uint32_t)p[4 * i + 2] << 16 | (uint32_t)p[4 * i + 3] << 24;
    for (int i = 0; i < 64; i++) {
        uint32_t f; int g;
        if (i < 16) { f = (b & c) | (~b & d); g = i; }
        else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
        else { f = c ^ (b | ~d); g = (7 * i) & 15; }
        const uint32_t t = a + f + MD5_K[i] + w[g];
        a = d; d = c; c = b; b = b + ((t << MD5_S[i]) | (t >> (32 - MD5_S[i])));
    }
    m->a += a; m->b += b; m->c += c; m->d += d;
}
static void md5_init(md5_t *m) { m->a = 0x67452301; m->b = 0xefcdab89; m->c = 0x98badcfe; m->d = 0x10325476; m->len = 0; m->fill = 0; }
static void md5_update(md5_t *m, const uint8_t *p, size_t n)
{
    m->len += n;
    while (n) {
        if (m->fill == 0 && n >= 64) { md5_block(m, p); p += 64; n -= 64; continue; }
        size_t take = 64 - (size_t)m->fill; if (take > n) take = n;
        memcpy(m->buf + m->fill, p, take); m->fill += (int)take; p += take; n -= take;
        if (m->fill == 64) { md5_block(m, m->buf); m->fill = 0; }
    }
}
static void md5_final(md5_t *m, uint8_t out[16])
{
    const uint64_t bits = m->len * 8; uint8_t pad = 0x80;
    md5_update(m, &pad, 1); pad = 0;
    while (m->fill != 56) md5_update(m, &pad, 1);
    uint8_t l[8]; for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (8 * i));
    md5_update(m, l, 8);
    const uint32_t v[4] = {m->a, m->b, m->c, m->d};
    for (int i = 0; i < 16; i++) out[i] = (uint8_t)(v[i >> 2] >> (8 * (i & 3)));
}
void orc_md5(const uint8_t *data, int64_t len, uint8_t out[16]) { md5_t m; md5_init(&m); md5_update(&m, data, (size_t)len); md5_final(&m, out); }

/* ---------------------------------------------------------------- CRCs (RFC 9639 §9.1.8, §9.3), MSB first, init 0 */
uint8_t orc_flac_crc8(const uint8_t *p, int64_t n)
{
    uint8_t c = 0;
    for (int64_t i = 0; i < n; i++) { c ^= p[i]; for (int k = 0; k < 8; k++) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1)); }
    return c;
}
uint16_t orc_flac_crc16(const uint8_t *p, int64_t n)
{
    uint16_t c = 0;
    for (int64_t i = 0; i < n; i++) { c ^= (uint16_t)(p[i] << 8); for (int k = 0; k < 8; k++) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1)); }
    return c;
}

/* ---------------------------------------------------------------- bit reader (big-endian, MSB first) */
typedef struct { const uint8_t *p; int64_t len, pos; /* pos in bits */ int err; } br_t;
static uint64_t br_u(br_t *b, int n)
{
    uint64_t v = 0;
    for (int i = 0; i < n; i++) {
        if ((b->pos >> 3) >= b->len) { b->err = 1; return 0; }
        v = (v << 1) | ((b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1u); b->pos++;
    }
    return v;
}
static int64_t br_s(br_t *b, int n)
{
    if (n == 0) return 0;
    const uint64_t v = br_u(b, n);
    return (v >> (n - 1)) & 1 ? (int64_t)v - ((int64_t)1 << n) : (int64_t)v;
}
static uint32_t br_unary(br_t *b)               /* count of 0 bits before the terminating 1 */
{
    uint32_t q = 0;
    for (;;) { if ((b->pos >> 3) >= b->len) { b->err = 1; return 0; } if (br_u(b, 1)) return q; q++; }
}

/* §9.2.7 coded residual */
static int decode_residual(br_t *b, int64_t *s, int blocksize, int order)
{
    const int method = (int)br_u(b, 2);
    if (method > 1) return -1;
    const int pbits = method ? 5 : 4, esc = method ? 31 : 15;
    const int porder = (int)br_u(b, 4);
    if ((blocksize >> porder) << porder != blocksize && porder > 0) return -1;
    const int psize = blocksize >> porder;
    if (psize < order && porder > 0) return -1;
    int i = order;
    for (int part = 0; part < (1 << porder); part++) {
        const int n = part == 0 ? psize - order : psize;
        if (n < 0) return -1;
        const int k = (int)br_u(b, pbits);
        if (k == esc) {
            const int w = (int)br_u(b, 5);
            for (int j = 0; j < n; j++) s[i++] = br_s(b, w);
        } else {
            for (int j = 0; j < n; j++) {
                const uint64_t q = br_unary(b);
                const uint64_t u = (q << k) | br_u(b, k);
                s[i++] = (u & 1) ? -(int64_t)(u >> 1) - 1 : (int64_t)(u >> 1);
                if (b->err) return -1;
            }
        }
    }
    return b->err ? -1 : 0;
}

/* §9.2: one subframe into s[0..blocksize) */
static int decode_subframe(br_t *b, int64_t *s, int blocksize, int bps)
{
    if (br_u(b, 1)) return -1;
    const int type = (int)br_u(b, 6);
    int wasted = 0;
    if (br_u(b, 1)) { wasted = (int)br_unary(b) + 1; bps -= wasted; if (bps <= 0) return -1; }
    if (type == 0) {
        const int64_t v = br_s(b, bps);
        for (int i = 0; i < blocksize; i++) s[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < blocksize; i++) s[i] = br_s(b, bps);
    } else if (type >= 8 && type <= 12) {
        const int order = type - 8;
        if (order > blocksize) return -1;
        for (int i = 0; i < order; i++) s[i] = br_s(b, bps);
        if (decode_residual(b, s, blocksize, order)) return -1;
        for (int i = order; i < blocksize; i++) {
            int64_t p = 0;
            switch (order) {
            case 1: p = s[i - 1]; break;
            case 2: p = 2 * s[i - 1] - s[i - 2]; break;
            case 3: p = 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; break;
            case 4: p = 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4]; break;
            default: break;
            }
            s[i] += p;
        }
    } else if (type >= 32) {
        const int order = type - 31;
        if (order > blocksize) return -1;
        for (int i = 0; i < order; i++) s[i] = br_s(b, bps);
        const int prec = (int)br_u(b, 4) + 1;
        if (prec == 16) return -1;
        const int shift = (int)br_s(b, 5);
        if (shift < 0) return -1;
        int64_t c[32];
        for (int j = 0; j < order; j++) c[j] = br_s(b, prec);
        if (decode_residual(b, s, blocksize, order)) return -1;
        for (int i = order; i < blocksize; i++) {
            int64_t p = 0;
            for (int j = 0; j < order; j++) p += c[j] * s[i - 1 - j];
            s[i] += p >> shift;
        }
    } else
        return -1;
    if (wasted) for (int i = 0; i < blocksize; i++) s[i] = s[i] * ((int64_t)1 << wasted);
    return b->err ? -1 : 0;
}

static const int BS_TAB[16] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
static const int SR_TAB[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
static const int SS_TAB[8] = {0, 8, 12, -1, 16, 20, 24, 32};

int orc_flac_decode(const uint8_t *data, int64_t len, int32_t *out, int64_t cap_frames, orc_flac_info *info)
{
    memset(info, 0, sizeof *info);
    if (len < 42 || memcmp(data, "fLaC", 4)) return -1;
    int64_t pos = 4; int last = 0, have_si = 0;
    while (!last) {
        if (pos + 4 > len) return -2;
        last = data[pos] >> 7;
        const int type = data[pos] & 0x7f;
        const int64_t blen = (int64_t)data[pos + 1] << 16 | (int64_t)data[pos + 2] << 8 | data[pos + 3];
        pos += 4;
        if (pos + blen > len) return -2;
        if (type == 0) {
            if (blen != 34) return -2;
            const uint8_t *p = data + pos;
            info->min_blocksize = p[0] << 8 | p[1]; info->max_blocksize = p[2] << 8 | p[3];
            info->min_framesize = p[4] << 16 | p[5] << 8 | p[6]; info->max_framesize = p[7] << 16 | p[8] << 8 | p[9];
            info->sample_rate = p[10] << 12 | p[11] << 4 | p[12] >> 4;
            info->channels = ((p[12] >> 1) & 7) + 1;
            info->bps = (((p[12] & 1) << 4) | (p[13] >> 4)) + 1;
            info->total_samples = (int64_t)(p[13] & 15) << 32 | (int64_t)p[14] << 24 | (int64_t)p[15] << 16 | (int64_t)p[16] << 8 | p[17];
            memcpy(info->md5_stored, p + 18, 16);
            have_si = 1;
        }
        info->metadata_blocks++;
        pos += blen;
    }
    if (!have_si) return -2;
    info->audio_offset = pos;
    const int nch = info->channels;
    md5_t md; md5_init(&md);
    int64_t *ch[8]; for (int c = 0; c < 8; c++) ch[c] = NULL;
    int64_t done = 0; int rc = 0;
    info->obs_min_framesize = 1 << 30;
    while (pos < len) {
        br_t b = {data + pos, len - pos, 0, 0};
        if (br_u(&b, 15) != 0x7ffc) { rc = -3; break; }
        const int variable = (int)br_u(&b, 1);
        const int bsc = (int)br_u(&b, 4), src = (int)br_u(&b, 4), cha = (int)br_u(&b, 4), ssc = (int)br_u(&b, 3);
        if (br_u(&b, 1)) { rc = -3; break; }
        /* UTF-8-like coded number */
        uint64_t num; {
            const int b0 = (int)br_u(&b, 8); int extra = 0;
            if (b0 < 0x80) { num = (uint64_t)b0; }
            else { int m = 0x40; while (b0 & m) { extra++; m >>= 1; } if (extra == 0 || extra > 6) { rc = -3; break; } num = (uint64_t)(b0 & (m - 1));
                   for (int i = 0; i < extra; i++) { const int bb = (int)br_u(&b, 8); if ((bb & 0xc0) != 0x80) { rc = -3; } num = num << 6 | (uint64_t)(bb & 0x3f); } }
            if (rc) break;
        }
        int bs = BS_TAB[bsc];
        if (bsc == 0) { rc = -3; break; }
        if (bsc == 6) bs = (int)br_u(&b, 8) + 1; else if (bsc == 7) bs = (int)br_u(&b, 16) + 1;
        int sr = src < 12 ? SR_TAB[src] : 0;
        if (src == 12) sr = (int)br_u(&b, 8) * 1000; else if (src == 13) sr = (int)br_u(&b, 16); else if (src == 14) sr = (int)br_u(&b, 16) * 10;
        else if (src == 15) { rc = -3; break; }
        if (src == 0) sr = info->sample_rate;
        const int hdr_bytes = (int)(b.pos >> 3);
        const int crc8 = (int)br_u(&b, 8);
        if (b.err) { rc = -3; break; }
        if (orc_flac_crc8(data + pos, hdr_bytes) != crc8) { info->crc8_errors++; rc = -4; break; }
        int bps = SS_TAB[ssc]; if (ssc == 0) bps = info->bps; if (bps < 0) { rc = -3; break; }
        const int fch = cha < 8 ? cha + 1 : 2;
        if (cha > 10 || fch != nch || bps != info->bps || sr != info->sample_rate) { rc = -5; break; }
        const int64_t first_sample = variable ? (int64_t)num : (int64_t)num * info->min_blocksize;
        if (first_sample != done) { rc = -6; break; }
        if (variable) info->variable_blocksize = 1;
        for (int c = 0; c < nch; c++) {
            ch[c] = (int64_t *)realloc(ch[c], sizeof(int64_t) * (size_t)bs);
            const int side = (cha == 8 && c == 1) || (cha == 9 && c == 0) || (cha == 10 && c == 1);
            if (decode_subframe(&b, ch[c], bs, bps + side)) { rc = -7; break; }
        }
        if (rc) break;
        while (b.pos & 7) if (br_u(&b, 1)) { rc = -8; break; }
        if (rc) break;
        const int fbytes = (int)(b.pos >> 3);
        const int crc16 = (int)br_u(&b, 16);
        if (b.err) { rc = -3; break; }
        if (orc_flac_crc16(data + pos, fbytes) != crc16) { info->crc16_errors++; rc = -4; break; }
        if (cha == 8) for (int i = 0; i < bs; i++) ch[1][i] = ch[0][i] - ch[1][i];
        else if (cha == 9) for (int i = 0; i < bs; i++) ch[0][i] = ch[0][i] + ch[1][i];
        else if (cha == 10) for (int i = 0; i < bs; i++) {
            const int64_t side = ch[1][i], mid = (ch[0][i] * 2) | (side & 1);
            ch[0][i] = (mid + side) >> 1; ch[1][i] = (mid - side) >> 1;
        }
        const int bytes = (bps + 7) / 8;
        for (int i = 0; i < bs; i++)
            for (int c = 0; c < nch; c++) {
                const int64_t v = ch[c][i];
                if (v < -((int64_t)1 << (bps - 1)) || v >= ((int64_t)1 << (bps - 1))) rc = -9;
                uint8_t le[4]; for (int k = 0; k < bytes; k++) le[k] = (uint8_t)((uint64_t)v >> (8 * k));
                md5_update(&md, le, (size_t)bytes);
                if (out && done + i < cap_frames) out[(done + i) * nch + c] = (int32_t)v;
            }
        if (rc) break;
        done += bs; info->frames++;
        const int ftot = fbytes + 2;
        if (ftot < info->obs_min_framesize) info->obs_min_framesize = ftot;
        if (ftot > info->obs_max_framesize) info->obs_max_framesize = ftot;
        if (bs > info->obs_max_blocksize) info->obs_max_blocksize = bs;
        info->last_blocksize = bs;
        pos += ftot;
    }
    for (int c = 0; c < 8; c++) free(ch[c]);
    info->decoded_samples = done;
    md5_final(&md, info->md5_decoded);
    return rc;
}

/* ---------------------------------------------------------------- sequential encoder for decoder test streams ----------
 * Deliberately exercises the parts of the format a production encoder at one setting never emits: every fixed order,
 * LPC orders up to 32, 5-bit Rice parameters, escaped partitions, wasted bits, all four stereo modes, variable or odd block
 * sizes.  `mode` bits: 0-2 predictor (0 verbatim, 1 fixed best, 2 lpc), 3 = force escape on partition 0, 4 = 5-bit rice,
 * 5 = variable block sizes, 6-7 = stereo mode (0 independent, 1 left/side, 2 side/right, 3 mid/side). */
typedef struct { uint8_t *p; int64_t cap, pos; } bw_t;
static void bw_u(bw_t *w, uint64_t v, int n)
{
    for (int i = n - 1; i >= 0; i--) {
        if ((w->pos >> 3) >= w->cap) { w->cap = w->cap * 2 + 4096; w->p = (uint8_t *)realloc(w->p, (size_t)w->cap); }
        if ((w->pos & 7) == 0) w->p[w->pos >> 3] = 0;
        w->p[w->pos >> 3] |= (uint8_t)(((v >> i) & 1u) << (7 - (w->pos & 7))); w->pos++;
    }
}
static void bw_rice(bw_t *w, int64_t r, int k)
{
    const uint64_t u = r >= 0 ? (uint64_t)r << 1 : ((uint64_t)(-r - 1) << 1) | 1;
    for (uint64_t q = u >> k; q; q--) bw_u(w, 0, 1);
    bw_u(w, 1, 1); bw_u(w, u & (((uint64_t)1 << k) - 1), k);
}
static void enc_residual(bw_t *w, const int64_t *res, int bs, int order, int five, int force_escape)
{
    int porder = 0;
    while (porder < 4 && (bs % (2 << porder)) == 0 && (bs >> (porder + 1)) > order) porder++;
    bw_u(w, five ? 1 : 0, 2); bw_u(w, (uint64_t)porder, 4);
    const int psize = bs >> porder; int i = order;
    for (int part = 0; part < (1 << porder); part++) {
        const int n = part == 0 ? psize - order : psize;
        uint64_t sum = 0; int64_t mx = 0;
        for (int j = 0; j < n; j++) { const int64_t r = res[i + j]; const int64_t a = r < 0 ? -r - 1 : r; sum += (uint64_t)a * 2 + 1; if (a > mx) mx = a; }
        int k = 0; while (n > 0 && ((uint64_t)n << (k + 1)) < sum && k < (five ? 30 : 14)) k++;
        if ((force_escape && part == 0) || mx >= ((int64_t)1 << 28)) {
            int wbits = 1; while (mx >> (wbits - 1)) wbits++;
            bw_u(w, five ? 31 : 15, five ? 5 : 4); bw_u(w, (uint64_t)wbits, 5);
            for (int j = 0; j < n; j++) bw_u(w, (uint64_t)res[i + j] & (((uint64_t)1 << wbits) - 1), wbits);
        } else {
            bw_u(w, (uint64_t)k, five ? 5 : 4);
            for (int j = 0; j < n; j++) bw_rice(w, res[i + j], k);
        }
        i += n;
    }
}
static void enc_subframe(bw_t *w, const int64_t *x, int bs, int bps, int pred, int five, int force_escape, int lpc_order)
{
    int64_t *s = (int64_t *)malloc(sizeof(int64_t) * (size_t)bs), *res = (int64_t *)malloc(sizeof(int64_t) * (size_t)bs);
    int64_t orv = 0; int allsame = 1;
    for (int i = 0; i < bs; i++) { orv |= x[i]; if (x[i] != x[0]) allsame = 0; }
    int wasted = 0; if (orv) while (!((orv >> wasted) & 1)) wasted++;
    if (wasted >= bps) wasted = 0;
    for (int i = 0; i < bs; i++) s[i] = x[i] >> wasted;
    bps -= wasted;
    int type, order = 0;
    if (allsame) type = 0;
    else if (pred == 0) type = 1;
    else if (pred == 1) {
        uint64_t best = ~(uint64_t)0; order = 0;
        for (int o = 0; o <= 4 && o < bs; o++) {
            uint64_t sum = 0;
            for (int i = o; i < bs; i++) {
                int64_t p = 0;
                if (o == 1) p = s[i - 1]; else if (o == 2) p = 2 * s[i - 1] - s[i - 2];
                else if (o == 3) p = 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; else if (o == 4) p = 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4];
                const int64_t r = s[i] - p; sum += (uint64_t)(r < 0 ? -r : r);
            }
            if (sum < best) { best = sum; order = o; }
        }
        type = 8 + order;
    } else { order = lpc_order < bs ? lpc_order : (bs > 1 ? bs - 1 : 0); type = order ? 31 + order : 1; }
    bw_u(w, 0, 1); bw_u(w, (uint64_t)type, 6); bw_u(w, wasted ? 1 : 0, 1);
    if (wasted) { for (int i = 1; i < wasted; i++) bw_u(w, 0, 1); bw_u(w, 1, 1); }
    const uint64_t mask = bps >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << bps) - 1);
    if (type == 0) bw_u(w, (uint64_t)s[0] & mask, bps);
    else if (type == 1) for (int i = 0; i < bs; i++) bw_u(w, (uint64_t)s[i] & mask, bps);
    else if (type < 32) {
        for (int i = 0; i < order; i++) bw_u(w, (uint64_t)s[i] & mask, bps);
        for (int i = order; i < bs; i++) {
            int64_t p = 0;
            if (order == 1) p = s[i - 1]; else if (order == 2) p = 2 * s[i - 1] - s[i - 2];
            else if (order == 3) p = 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; else if (order == 4) p = 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4];
            res[i] = s[i] - p;
        }
        enc_residual(w, res, bs, order, five, force_escape);
    } else {
        /* a fixed, stable, smoothing predictor: binomial-like taps quantised to 10 bits (good enough to exercise the format) */
        const int prec = 10, shift = 8; int64_t c[32];
        for (int j = 0; j < order; j++) c[j] = (j == 0 ? 460 : (j == 1 ? -250 : (j & 1 ? -(40 / (j + 1)) : 40 / (j + 1))));
        for (int i = 0; i < order; i++) bw_u(w, (uint64_t)s[i] & mask, bps);
        bw_u(w, (uint64_t)(prec - 1), 4); bw_u(w, (uint64_t)shift, 5);
        for (int j = 0; j < order; j++) bw_u(w, (uint64_t)c[j] & ((1u << prec) - 1), prec);
        for (int i = order; i < bs; i++) { int64_t p = 0; for (int j = 0; j < order; j++) p += c[j] * s[i - 1 - j]; res[i] = s[i] - (p >> shift); }
        enc_residual(w, res, bs, order, five, force_escape);
    }
    free(s); free(res);
}

int64_t orc_flac_encode(const int32_t *pcm, int64_t nframes, int channels, int bps, int sample_rate, int blocksize, int mode,
                        int lpc_order, uint8_t *out, int64_t cap)
{
    bw_t w = {NULL, 0, 0};
    const int pred = mode & 7, force_escape = (mode >> 3) & 1, five = (mode >> 4) & 1, variable = (mode >> 5) & 1, stereo = (mode >> 6) & 3;
    bw_u(&w, 0x664c6143, 32);
    bw_u(&w, 0x80, 8); bw_u(&w, 34, 24);
    const int minbs = variable ? (blocksize / 2 > 16 ? blocksize / 2 : 16) : blocksize;
    bw_u(&w, (uint64_t)minbs, 16); bw_u(&w, (uint64_t)blocksize, 16); bw_u(&w, 0, 24); bw_u(&w, 0, 24);
    bw_u(&w, (uint64_t)sample_rate, 20); bw_u(&w, (uint64_t)(channels - 1), 3); bw_u(&w, (uint64_t)(bps - 1), 5); bw_u(&w, (uint64_t)nframes, 36);
    {   /* MD5 of the little-endian interleaved PCM */
        md5_t md; md5_init(&md); const int bytes = (bps + 7) / 8;
        for (int64_t i = 0; i < nframes * channels; i++) { uint8_t le[4]; for (int k = 0; k < bytes; k++) le[k] = (uint8_t)((uint32_t)pcm[i] >> (8 * k)); md5_update(&md, le, (size_t)bytes); }
        uint8_t dg[16]; md5_final(&md, dg); for (int i = 0; i < 16; i++) bw_u(&w, dg[i], 8);
    }
    int64_t done = 0, fno = 0;
    int64_t *chb[8];
    if (channels < 1 || channels > 8) return -1;
    for (int c = 0; c < 8; c++) chb[c] = (int64_t *)malloc(sizeof(int64_t) * (size_t)blocksize);
    while (done < nframes) {
        int bs = blocksize;
        if (variable && (fno & 1)) bs = minbs;
        if (bs > nframes - done) bs = (int)(nframes - done);
        const int64_t fstart = w.pos >> 3;
        int bsc = 0; for (int c = 1; c < 16; c++) if (BS_TAB[c] == bs) bsc = c;
        if (!bsc) bsc = bs <= 256 ? 6 : 7;
        int src = 0; for (int c = 1; c < 12; c++) if (SR_TAB[c] == sample_rate) src = c;
        int ssc = 0; for (int c = 1; c < 8; c++) if (SS_TAB[c] == bps) ssc = c;
        const int cha = channels == 2 && stereo ? 7 + stereo : channels - 1;
        bw_u(&w, 0x7ffc, 15); bw_u(&w, (uint64_t)variable, 1); bw_u(&w, (uint64_t)bsc, 4); bw_u(&w, (uint64_t)src, 4); bw_u(&w, (uint64_t)cha, 4); bw_u(&w, (uint64_t)ssc, 3); bw_u(&w, 0, 1);
        {
            const uint64_t v = variable ? (uint64_t)done : (uint64_t)fno;
            if (v < 0x80) bw_u(&w, v, 8);
            else { int nb = 2; while (nb < 7 && (v >> (5 * nb + 1))) nb++;
                   bw_u(&w, ((0xffu << (8 - nb)) & 0xff) | (v >> (6 * (nb - 1))), 8);
                   for (int i = nb - 2; i >= 0; i--) bw_u(&w, 0x80 | ((v >> (6 * i)) & 0x3f), 8); }
        }
        if (bsc == 6) bw_u(&w, (uint64_t)(bs - 1), 8); else if (bsc == 7) bw_u(&w, (uint64_t)(bs - 1), 16);
        bw_u(&w, orc_flac_crc8(w.p + fstart, (w.pos >> 3) - fstart), 8);
        for (int c = 0; c < channels; c++) for (int i = 0; i < bs; i++) chb[c][i] = pcm[(done + i) * channels + c];
        int sidech = -1;
        if (cha == 8) { for (int i = 0; i < bs; i++) chb[1][i] = chb[0][i] - chb[1][i]; sidech = 1; }
        else if (cha == 9) { for (int i = 0; i < bs; i++) chb[0][i] = chb[0][i] - chb[1][i]; sidech = 0; }
        else if (cha == 10) { for (int i = 0; i < bs; i++) { const int64_t l = chb[0][i], r = chb[1][i]; chb[0][i] = (l + r) >> 1; chb[1][i] = l - r; } sidech = 1; }
        for (int c = 0; c < channels; c++) enc_subframe(&w, chb[c], bs, bps + (c == sidech), pred, five, force_escape, lpc_order);
        while (w.pos & 7) bw_u(&w, 0, 1);
        bw_u(&w, orc_flac_crc16(w.p + fstart, (w.pos >> 3) - fstart), 16);
        done += bs; fno++;
    }
    for (int c = 0; c < 8; c++) free(chb[c]);
    const int64_t total = w.pos >> 3;
    if (out && total <= cap) memcpy(out, w.p, (size_t)total);
    free(w.p);
    return total;
}
