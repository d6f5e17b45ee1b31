// k_flacdec.hip — FLAC / WAV input leg: file bytes in HBM -> PCM in HBM (gfx950).
//
// Replaces the reference's input leg, audio.OpenAudioFile + the Reader.ReadFrame loop that every pass runs again over the
// file (reader.go:29-169; Pass 1 analyser.go, Pass 2 processor.go:78-, Pass 3/4 normalise.go:924-): libavformat demux +
// libavcodec flac/pcm decode on one CPU thread, four or more times per file.  Here the file is decoded once, on the GPU,
// straight into the buffer jt_upload_pcm would have filled.  Format per RFC 9639; checked against the RFC-pinned oracle
// decoder (oracle/orc_flac.c) bit for bit.
//
// A FLAC stream has no frame index and its Rice-coded residual is strictly serial inside a frame, so the shape is:
//   find   : every byte offset in parallel: 15-bit sync code, legal header fields that agree with STREAMINFO, CRC-8 -> candidate
//   parse  : one lane per candidate walks the whole frame without storing samples: end offset, per-channel subframe bit
//            offsets, CRC-16 of the frame bytes.  The host then follows end -> start links from the first frame, which
//            discards the rare candidates that are header look-alikes inside audio data (they fail CRC-16 or are unreachable).
//   decode : one lane per (frame, channel) decodes its subframe (Rice / escape / verbatim / constant, fixed or LPC predictor
//            with 64-bit accumulation, wasted bits); every 64 samples the wave transposes through LDS and writes 256-byte
//            runs, so the stores stay coalesced although the lanes sit in different frames.
//   finish : undo the stereo decorrelation (left/side, side/right, mid/side), interleave, convert to f32 (x 2^(1-bps)).
// Byte/integer work: the serial bit parsing bounds it (one frame per lane), not HBM.
#include "jt_internal.h"
#include <hip/hip_runtime.h>

namespace {
namespace fd {
constexpr int ROW = 65;

struct StreamInfo { int channels, bps, sample_rate, min_bs, max_bs, pad; long long audio_offset, len; };
struct Cand { long long pos; long long number; int blocksize, hdr_len, ch_assign, variable; };
struct Parsed { long long end; int ok, wasted_any; long long sub_bit[8]; };
struct Frame { long long pos, out_offset; int blocksize, ch_assign; long long sub_bit[8]; };

__device__ __forceinline__ unsigned crc8_step(unsigned c, unsigned byte)
{
    c ^= byte;
#pragma unroll
    for (int k = 0; k < 8; k++) c = (c & 0x80) ? ((c << 1) ^ 0x07) & 0xff : (c << 1) & 0xff;
    return c;
}

// ---------------------------------------------------------------------------------------------------------------- find
__global__ __launch_bounds__(256) void k_flac_find(const uint8_t *__restrict__ d, StreamInfo si, Cand *__restrict__ out,
                                                   int *__restrict__ count, int cap)
{
    const long long p = si.audio_offset + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p + 6 > si.len) return;
    if (d[p] != 0xff || (d[p + 1] & 0xfe) != 0xf8) return;
    const int variable = d[p + 1] & 1;
    const int bsc = d[p + 2] >> 4, src = d[p + 2] & 15, cha = d[p + 3] >> 4, ssc = (d[p + 3] >> 1) & 7;
    if ((d[p + 3] & 1) || bsc == 0 || src == 15 || cha > 10 || ssc == 3) return;
    // the file is padded with 32 zero bytes, so the header reads below never leave the allocation
    int q = 4;
    const unsigned b0 = d[p + q++];
    long long num;
    if (b0 < 0x80) num = b0;
    else {
        int extra = 0; unsigned m = 0x40;
        while (b0 & m) { extra++; m >>= 1; }
        if (extra == 0 || extra > 6) return;
        num = b0 & (m - 1);
        for (int i = 0; i < extra; i++) { const unsigned bb = d[p + q++]; if ((bb & 0xc0) != 0x80) return; num = (num << 6) | (bb & 0x3f); }
    }
    int bs;
    if (bsc == 1) bs = 192; else if (bsc <= 5) bs = 576 << (bsc - 2); else if (bsc == 6) bs = d[p + q++] + 1;
    else if (bsc == 7) { bs = ((d[p + q] << 8) | d[p + q + 1]) + 1; q += 2; } else bs = 256 << (bsc - 8);
    int sr = 0;
    switch (src) {
    case 0: sr = si.sample_rate; break; case 1: sr = 88200; break; case 2: sr = 176400; break; case 3: sr = 192000; break;
    case 4: sr = 8000; break; case 5: sr = 16000; break; case 6: sr = 22050; break; case 7: sr = 24000; break;
    case 8: sr = 32000; break; case 9: sr = 44100; break; case 10: sr = 48000; break; case 11: sr = 96000; break;
    case 12: sr = d[p + q++] * 1000; break;
    case 13: sr = (d[p + q] << 8) | d[p + q + 1]; q += 2; break;
    default: sr = ((d[p + q] << 8) | d[p + q + 1]) * 10; q += 2; break;
    }
    const int bps = ssc == 0 ? si.bps : ssc == 1 ? 8 : ssc == 2 ? 12 : ssc == 4 ? 16 : ssc == 5 ? 20 : ssc == 6 ? 24 : 32;
    const int nch = cha < 8 ? cha + 1 : 2;
    if (bps != si.bps || nch != si.channels || sr != si.sample_rate || bs > si.max_bs) return;
    if (p + q + 1 > si.len) return;
    unsigned c8 = 0;
    for (int i = 0; i < q; i++) c8 = crc8_step(c8, d[p + i]);
    if (c8 != d[p + q]) return;
    const int idx = atomicAdd(count, 1);
    if (idx < cap) { Cand c; c.pos = p; c.number = num; c.blocksize = bs; c.hdr_len = q + 1; c.ch_assign = cha; c.variable = variable; out[idx] = c; }
}

// ---------------------------------------------------------------------------------------------------------------- bit reader
struct BitReader {
    const unsigned *w;          // the file as big-endian dwords (read through bswap)
    long long next;             // next dword index to load
    unsigned long long acc;     // valid bits left-aligned
    int cnt;
    long long used;             // bits consumed since init
    long long limit;            // bits available from the start position to the end of the file
    bool err;
    // lane-private window of the stream in LDS: ring[(dword index & 63) * 64 + lane] holds dwords [.., fill).  Lanes sit in different
    // frames, so a refill straight from memory would make the whole wave wait for one lane's miss at nearly every sample;
    // topup() brings 64 bytes per lane at wave-uniform checkpoints instead (all loads in flight together, one wait).
    unsigned *ring; int lane; long long fill;
    long long wend;             // first dword index that must not be fetched: the image is zero padded by >= 1024 bytes past its end, a
                                // reader that runs that far (forged escape widths / block sizes in a look-alike header) stops with err
    __device__ __forceinline__ void init(const uint8_t *base, long long bitpos, long long total_bits, unsigned *ring_, int lane_)
    {
        w = reinterpret_cast<const unsigned *>(base);
        ring = ring_; lane = lane_;
        next = bitpos >> 5; acc = 0; cnt = 0; used = 0; limit = total_bits - bitpos; err = false;
        wend = ((total_bits >> 5) + 224) & ~3ll;           // <= len + 896 bytes (+ one 64-byte top-up) < the padded allocation
        fill = next & ~3ll;
        topup(); topup(); topup();
        refill();
        const int skip = (int)(bitpos & 31);
        acc <<= skip; cnt -= skip;
    }
    __device__ __forceinline__ void topup()
    {
        if (fill - next < 40) {
            if (fill + 16 > wend) { err = true; return; }
            const uint4 *src = reinterpret_cast<const uint4 *>(w + fill);
            const uint4 a = src[0], b = src[1], c = src[2], e = src[3];
            const unsigned v[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, e.x, e.y, e.z, e.w};
#pragma unroll
            for (int j = 0; j < 16; j++) ring[(int)((fill + j) & 63) * 64 + lane] = v[j];
            fill += 16;
        }
    }
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32) {
            unsigned wv = 0;
            if (next < fill) wv = ring[(int)(next & 63) * 64 + lane];
            else if (next < wend) wv = w[next];
            else err = true;
            next++;
            acc |= (unsigned long long)__builtin_bswap32(wv) << (32 - cnt); cnt += 32;
        }
    }
    __device__ __forceinline__ unsigned u(int n)                  // n <= 32
    {
        if (n == 0) return 0;
        refill();
        const unsigned v = (unsigned)(acc >> (64 - n));
        acc <<= n; cnt -= n; used += n;
        return v;
    }
    __device__ __forceinline__ int s(int n)                       // n <= 32
    {
        if (n == 0) return 0;
        const unsigned v = u(n);
        return (int)(v << (32 - n)) >> (32 - n);
    }
    __device__ __forceinline__ unsigned unary()                   // zeros before the terminating one
    {
        unsigned q = 0;
        for (;;) {
            refill();
            if (acc == 0) {
                q += (unsigned)cnt; used += cnt; cnt = 0;
                if (used > limit) { err = true; return q; }
                continue;
            }
            const int z = __clzll((long long)acc);
            q += (unsigned)z; acc <<= z; acc <<= 1; cnt -= z + 1; used += z + 1;
            return q;
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------- subframe
// Per-lane decoding state of one subframe; step() yields the next sample (STORE) or just consumes its bits (!STORE).
struct SubState {
    int type;            // 0 constant, 1 verbatim, 2 predicted (fixed or lpc)
    int order, shift, bps, wasted, blocksize, prec;
    int k, esc_width, part_left, psize, pbits;
    bool escaped;
    int constv;
    bool bad;
};

template <bool STORE>
__device__ __forceinline__ void sub_begin(BitReader &br, SubState &st, int blocksize, int bps, int *hist, int *coef, int lane)
{
    st.bad = false; st.blocksize = blocksize; st.order = 0; st.shift = 0; st.prec = 0; st.wasted = 0; st.constv = 0; st.type = 1;
    st.k = 0; st.esc_width = 0; st.part_left = 0; st.psize = 0; st.pbits = 4; st.escaped = false;
    if (br.u(1)) st.bad = true;
    const int t = (int)br.u(6);
    if (br.u(1)) { st.wasted = (int)br.unary() + 1; bps -= st.wasted; }
    if (bps <= 0 || bps > 32) { st.bad = true; bps = 1; }
    st.bps = bps;
    if (t == 0) { st.type = 0; st.constv = br.s(bps); }
    else if (t == 1) st.type = 1;
    else if ((t >= 8 && t <= 12) || t >= 32) {
        st.type = 2;
        const bool lpc = t >= 32;
        st.order = lpc ? t - 31 : t - 8;
        if (st.order > blocksize) { st.bad = true; st.order = 0; }
        for (int i = 0; i < st.order; i++) { const int v = br.s(bps); if (STORE) hist[(i & 31) * 64 + lane] = v; }
        if (lpc) {
            const int prec = (int)br.u(4) + 1;
            if (prec == 16) st.bad = true;
            st.prec = prec;
            st.shift = br.s(5);
            if (st.shift < 0) { st.bad = true; st.shift = 0; }
            for (int j = 0; j < st.order; j++) { const int c = br.s(prec); if (STORE) coef[j * 64 + lane] = c; }
        } else if (STORE) {
            st.prec = 4;
            // fixed predictors as LPC taps with shift 0: 1 | 2,-1 | 3,-3,1 | 4,-6,4,-1
            const int o = st.order;
            if (o >= 1) coef[0 * 64 + lane] = o;
            if (o >= 2) coef[1 * 64 + lane] = o == 2 ? -1 : (o == 3 ? -3 : -6);
            if (o >= 3) coef[2 * 64 + lane] = o == 3 ? 1 : 4;
            if (o >= 4) coef[3 * 64 + lane] = -1;
        }
        const int method = (int)br.u(2);
        if (method > 1) st.bad = true;
        st.pbits = method ? 5 : 4;
        const int porder = (int)br.u(4);
        st.psize = blocksize >> porder;
        if (porder > 0 && ((st.psize << porder) != blocksize || st.psize < st.order)) st.bad = true;
        if (porder == 0 && blocksize < st.order) st.bad = true;
        st.part_left = -st.order;              // the first partition holds psize - order residuals
    } else st.bad = true;
}

// next residual of a predicted subframe (partition bookkeeping, Rice or escaped)
__device__ __forceinline__ int sub_residual(BitReader &br, SubState &st)
{
    for (int guard = 0; st.part_left <= 0; guard++) {
        // a new partition starts here (part_left is -order before the first one; a partition may be empty)
        const int k = (int)br.u(st.pbits);
        st.escaped = k == (st.pbits == 4 ? 15 : 31);
        if (st.escaped) st.esc_width = (int)br.u(5);
        st.k = k;
        st.part_left += st.psize;
        if (st.psize <= 0 || guard > 64) { st.bad = true; st.part_left = 1; }
    }
    st.part_left--;
    if (st.escaped) return br.s(st.esc_width);
    const unsigned q = br.unary();
    const unsigned uu = (q << st.k) | br.u(st.k);
    return (int)(uu >> 1) ^ -(int)(uu & 1);
}

// sample index i (0-based in the block); returns the decoded sample (before the wasted-bits shift) when STORE.
// General path: history and taps in LDS (any order up to 32), 64-bit accumulation.
template <bool STORE>
__device__ __forceinline__ int sub_step(BitReader &br, SubState &st, int i, int *hist, const int *coef, int lane)
{
    if (st.type == 0) return st.constv;
    if (st.type == 1) return br.s(st.bps);
    if (i < st.order) return STORE ? hist[(i & 31) * 64 + lane] : 0;
    const int r = sub_residual(br, st);
    if (!STORE) return 0;
    long long acc = 0;
    for (int j = 0; j < st.order; j++) acc += (long long)coef[j * 64 + lane] * (long long)hist[((i - 1 - j) & 31) * 64 + lane];
    const int v = r + (int)(acc >> st.shift);
    hist[(i & 31) * 64 + lane] = v;
    return v;
}

// ---------------------------------------------------------------------------------------------------------------- parse
__global__ __launch_bounds__(64) void k_flac_parse(const uint8_t *__restrict__ d, StreamInfo si, const Cand *__restrict__ cands,
                                                   int ncand, Parsed *__restrict__ out)
{
    __shared__ unsigned crc_tab[256];
    __shared__ unsigned ring[64 * 64];
    for (int t = threadIdx.x; t < 256; t += 64) {
        unsigned c = (unsigned)t << 8;
        for (int k = 0; k < 8; k++) c = (c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1);
        crc_tab[t] = c & 0xffff;
    }
    __syncthreads();
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= ncand) return;
    const Cand c = cands[idx];
    Parsed pr; pr.ok = 0; pr.end = 0; pr.wasted_any = 0;
    for (int ch = 0; ch < 8; ch++) pr.sub_bit[ch] = 0;
    BitReader br;
    const long long start_bit = (c.pos + c.hdr_len) * 8;
    br.init(d, start_bit, si.len * 8, ring, threadIdx.x);
    bool bad = false;
    for (int ch = 0; ch < si.channels && !bad; ch++) {
        const int side = (c.ch_assign == 8 && ch == 1) || (c.ch_assign == 9 && ch == 0) || (c.ch_assign == 10 && ch == 1);
        pr.sub_bit[ch] = start_bit + br.used;
        SubState st;
        sub_begin<false>(br, st, c.blocksize, si.bps + side, nullptr, nullptr, 0);
        if (st.wasted) pr.wasted_any = 1;
        if (st.type == 1) {
            // verbatim: skip blocksize * bps bits without touching them one by one
            long long skip = (long long)c.blocksize * st.bps;
            for (int it = 0; skip > 0 && !br.err; it++) {
                if ((it & 7) == 0) br.topup();
                const int n = skip > 32 ? 32 : (int)skip; br.u(n); skip -= n; if (br.used > br.limit) br.err = true;
            }
        } else if (st.type == 2) {
            for (int i = st.order; i < c.blocksize && !st.bad && !br.err; i++) {
                if ((i & 15) == 0) { br.topup(); if (br.used > br.limit) br.err = true; }      // escaped partitions never reach unary()'s check
                sub_residual(br, st);
            }
        }
        bad = st.bad || br.err || br.used > br.limit;
    }
    if (!bad) {
        const int padbits = (int)((8 - ((start_bit + br.used) & 7)) & 7);
        if (br.u(padbits) != 0) bad = true;
        const long long crc_pos = (start_bit + br.used) >> 3;          // byte offset of the CRC-16
        const unsigned stored = br.u(16);
        if (br.used > br.limit) bad = true;
        if (!bad) {
            unsigned crc = 0;
            for (long long b = c.pos; b < crc_pos; b++) crc = ((crc << 8) ^ crc_tab[((crc >> 8) ^ d[b]) & 0xff]) & 0xffff;
            if (crc == stored) { pr.ok = 1; pr.end = crc_pos + 2; }
        }
    }
    out[idx] = pr;
}

// ---------------------------------------------------------------------------------------------------------------- decode
// One lane per (frame, channel) subframe; planar int32 output: plane[ch][out_offset + i]
__global__ __launch_bounds__(64) void k_flac_decode(const uint8_t *__restrict__ d, StreamInfo si, const Frame *__restrict__ frames,
                                                    long long nsub, long long total, int *__restrict__ planar, int *__restrict__ errs)
{
    __shared__ int tile[64 * ROW];
    __shared__ int hist[32 * 64], coef[32 * 64];
    __shared__ unsigned ring[64 * 64];
    __shared__ long long rowdst[64];
    __shared__ int rowbs[64];
    constexpr int NT = 12;                                              // taps held in registers (libFLAC's subset maximum)
    const int lane = threadIdx.x;
    const long long sidx = (long long)blockIdx.x * 64 + lane;
    const bool live = sidx < nsub;
    const long long f = live ? sidx / si.channels : 0;
    const int ch = live ? (int)(sidx % si.channels) : 0;
    Frame fr; fr.blocksize = 0; fr.ch_assign = 0; fr.out_offset = 0; fr.pos = 0;
    BitReader br; SubState st; st.type = 0; st.constv = 0; st.wasted = 0; st.bad = false; st.order = 0; st.prec = 0; st.bps = 1; st.shift = 0;
    br.init(d, live ? frames[f].sub_bit[ch] : si.audio_offset * 8, si.len * 8, ring, lane);
    if (live) {
        fr = frames[f];
        const int side = (fr.ch_assign == 8 && ch == 1) || (fr.ch_assign == 9 && ch == 0) || (fr.ch_assign == 10 && ch == 1);
        sub_begin<true>(br, st, fr.blocksize, si.bps + side, hist, coef, lane);
    }
    const int bs = live ? fr.blocksize : 0;
    rowbs[lane] = bs;
    rowdst[lane] = (long long)ch * total + fr.out_offset;
    int maxbs = bs;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_xor(maxbs, m); maxbs = o > maxbs ? o : maxbs; }
    // wave-uniform choice of the predictor datapath (as libFLAC's decoder: 32-bit when bps + precision + log2(order) <= 32)
    const bool pred = live && st.type == 2 && st.order > 0;
    const bool in_regs = !__any(pred && st.order > NT);
    const bool narrow = !__any(pred && st.bps + st.prec + (31 - __clz(st.order > 0 ? st.order : 1)) > 32) && si.bps <= 22;
    int cr[NT], hr[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) { cr[j] = (pred && j < st.order) ? coef[j * 64 + lane] : 0; hr[j] = 0; }
    __syncthreads();
    for (int i0 = 0; i0 < maxbs; i0 += 64) {
        if (i0 < bs) {
            const int n = bs - i0 < 64 ? bs - i0 : 64;
            if (in_regs) {
                for (int c = 0; c < n; c++) {
                    if ((c & 15) == 0) br.topup();
                    const int i = i0 + c;
                    int v;
                    if (st.type == 0) v = st.constv;
                    else if (st.type == 1) v = br.s(st.bps);
                    else if (i < st.order) v = hist[(i & 31) * 64 + lane];
                    else {
                        const int r = sub_residual(br, st);
                        if (narrow) {
                            int a = 0;
#pragma unroll
                            for (int j = 0; j < NT; j++) a += __mul24(cr[j], hr[j]);
                            v = r + (a >> st.shift);
                        } else {
                            long long a = 0;
#pragma unroll
                            for (int j = 0; j < NT; j++) a += (long long)cr[j] * (long long)hr[j];
                            v = r + (int)(a >> st.shift);
                        }
                    }
#pragma unroll
                    for (int j = NT - 1; j > 0; j--) hr[j] = hr[j - 1];
                    hr[0] = v;
                    tile[lane * ROW + c] = (int)((unsigned)v << st.wasted);
                }
            } else {
                for (int c = 0; c < n; c++) {
                    if ((c & 15) == 0) br.topup();
                    const int v = sub_step<true>(br, st, i0 + c, hist, coef, lane);
                    tile[lane * ROW + c] = (int)((unsigned)v << st.wasted);
                }
            }
        }
        __syncthreads();
        for (int r = 0; r < 64; r++) {
            const int rb = rowbs[r];
            if (i0 + lane < rb) planar[rowdst[r] + i0 + lane] = tile[r * ROW + lane];
        }
        __syncthreads();
    }
    if (live && (st.bad || br.err)) atomicAdd(errs, 1);
}

// ---------------------------------------------------------------------------------------------------------------- decode ahead of the chain (mono)
// A mono frame's only subframe starts right behind the frame header: no parse result is needed to decode it, and the walk that decodes
// it is the walk k_flac_parse makes to find the frame's end.  So for mono streams ONE kernel does both: one lane per CANDIDATE
// (look-alikes included: they decode to garbage nobody reads, inside the reader's own bounds), the samples into the candidate's own row
// of max_blocksize entries, the frame's end / padding / CRC-16 verdict into the table the host's chain walk reads; k_flac_finish_cand
// then gathers the rows of the frames the chain kept.  (Both kernels are one lane per frame, one wave per SIMD: an hour's file and a
// ten minutes' alike took 3.3 ms each; side by side on two streams they took 6.6 -- the walk is bound by something they share --
// so the parse had to go, not move.)  The datapath is k_flac_decode's (the same functions, the same wave-uniform choices).
__global__ __launch_bounds__(64) void k_flac_decode_cand(const uint8_t *__restrict__ d, StreamInfo si, const Cand *__restrict__ cands, int ncand,
                                                         int *__restrict__ rows, Parsed *__restrict__ out)
{
    __shared__ unsigned crc_tab[256];
    for (int t = threadIdx.x; t < 256; t += 64) {
        unsigned c = (unsigned)t << 8;
        for (int k = 0; k < 8; k++) c = (c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1);
        crc_tab[t] = c & 0xffff;
    }
    __shared__ int tile[64 * ROW];
    __shared__ int hist[32 * 64], coef[32 * 64];
    __shared__ unsigned ring[64 * 64];
    __shared__ int rowbs[64];
    constexpr int NT = 12;
    const int lane = threadIdx.x;
    const int idx = blockIdx.x * 64 + lane;
    const bool live = idx < ncand;
    Cand c; c.pos = si.audio_offset; c.hdr_len = 0; c.blocksize = 0; c.number = 0; c.ch_assign = 0; c.variable = 0;
    if (live) c = cands[idx];
    BitReader br; SubState st; st.type = 0; st.constv = 0; st.wasted = 0; st.bad = false; st.order = 0; st.prec = 0; st.bps = 1; st.shift = 0;
    const long long start_bit = (c.pos + c.hdr_len) * 8;
    br.init(d, start_bit, si.len * 8, ring, lane);
    if (live) sub_begin<true>(br, st, c.blocksize, si.bps, hist, coef, lane);
    const int bs = live ? c.blocksize : 0;
    rowbs[lane] = bs;
    int maxbs = bs;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_xor(maxbs, m); maxbs = o > maxbs ? o : maxbs; }
    const bool pred = live && st.type == 2 && st.order > 0;
    const bool in_regs = !__any(pred && st.order > NT);
    const bool narrow = !__any(pred && st.bps + st.prec + (31 - __clz(st.order > 0 ? st.order : 1)) > 32) && si.bps <= 22;
    int cr[NT], hr[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) { cr[j] = (pred && j < st.order) ? coef[j * 64 + lane] : 0; hr[j] = 0; }
    __syncthreads();
    const long long row0 = (long long)blockIdx.x * 64 * si.max_bs;
    for (int i0 = 0; i0 < maxbs; i0 += 64) {
        if (i0 < bs) {
            const int n = bs - i0 < 64 ? bs - i0 : 64;
            if (in_regs) {
                for (int q = 0; q < n; q++) {
                    if ((q & 15) == 0) br.topup();
                    const int i = i0 + q;
                    int v;
                    if (st.type == 0) v = st.constv;
                    else if (st.type == 1) v = br.s(st.bps);
                    else if (i < st.order) v = hist[(i & 31) * 64 + lane];
                    else {
                        const int r = sub_residual(br, st);
                        if (narrow) {
                            int a = 0;
#pragma unroll
                            for (int j = 0; j < NT; j++) a += __mul24(cr[j], hr[j]);
                            v = r + (a >> st.shift);
                        } else {
                            long long a = 0;
#pragma unroll
                            for (int j = 0; j < NT; j++) a += (long long)cr[j] * (long long)hr[j];
                            v = r + (int)(a >> st.shift);
                        }
                    }
#pragma unroll
                    for (int j = NT - 1; j > 0; j--) hr[j] = hr[j - 1];
                    hr[0] = v;
                    tile[lane * ROW + q] = (int)((unsigned)v << st.wasted);
                }
            } else {
                for (int q = 0; q < n; q++) {
                    if ((q & 15) == 0) br.topup();
                    const int v = sub_step<true>(br, st, i0 + q, hist, coef, lane);
                    tile[lane * ROW + q] = (int)((unsigned)v << st.wasted);
                }
            }
        }
        __syncthreads();
        for (int r = 0; r < 64; r++) {
            const int rb = rowbs[r];
            if (i0 + lane < rb) rows[row0 + (long long)r * si.max_bs + i0 + lane] = tile[r * ROW + lane];
        }
        __syncthreads();
    }
    // what k_flac_parse establishes about a frame, from the same walk: zero padding to the byte, the CRC-16 over the whole frame, its end
    if (live) {
        Parsed pr; pr.ok = 0; pr.end = 0; pr.wasted_any = st.wasted ? 1 : 0;
        for (int ch = 0; ch < 8; ch++) pr.sub_bit[ch] = 0;
        pr.sub_bit[0] = start_bit;
        bool bad = st.bad || br.err || br.used > br.limit;
        if (!bad) {
            const int padbits = (int)((8 - ((start_bit + br.used) & 7)) & 7);
            if (br.u(padbits) != 0) bad = true;
            const long long crc_pos = (start_bit + br.used) >> 3;
            const unsigned stored = br.u(16);
            if (br.used > br.limit || br.err) bad = true;
            if (!bad) {
                unsigned crc = 0;
                for (long long b = c.pos; b < crc_pos; b++) crc = ((crc << 8) ^ crc_tab[((crc >> 8) ^ d[b]) & 0xff]) & 0xffff;
                if (crc == stored) { pr.ok = 1; pr.end = crc_pos + 2; }
            }
        }
        out[idx] = pr;
    }
}
// one block per frame of the chain: its candidate's row (index in sub_bit[7]) to the output
__global__ __launch_bounds__(256) void k_flac_finish_cand(const Frame *__restrict__ frames, int max_bs, const int *__restrict__ rows,
                                                          int *__restrict__ out_i32, float *__restrict__ out_f32, float scale)
{
    const Frame fr = frames[blockIdx.x];
    const long long ci = fr.sub_bit[7];
    const int *src = rows + ci * max_bs;
    for (int i = threadIdx.x; i < fr.blocksize; i += blockDim.x) {
        const long long s = fr.out_offset + i;
        const int v = src[i];
        if (out_i32) out_i32[s] = v;
        if (out_f32) out_f32[s] = (float)v * scale;
    }
}

// ---------------------------------------------------------------------------------------------------------------- finish
// one block per frame: stereo decorrelation, interleave, optional f32 conversion
__global__ __launch_bounds__(256) void k_flac_finish(const Frame *__restrict__ frames, int channels, long long total,
                                                     const int *__restrict__ planar, int *__restrict__ out_i32,
                                                     float *__restrict__ out_f32, float scale)
{
    const Frame fr = frames[blockIdx.x];
    for (int i = threadIdx.x; i < fr.blocksize; i += blockDim.x) {
        const long long s = fr.out_offset + i;
        if (channels == 2) {
            int a = planar[s], b = planar[total + s];
            if (fr.ch_assign == 8) b = a - b;
            else if (fr.ch_assign == 9) a = a + b;
            else if (fr.ch_assign == 10) { const int side = b; const int mid = (int)(((unsigned)a << 1) | ((unsigned)side & 1u)); a = (mid + side) >> 1; b = (mid - side) >> 1; }
            if (out_i32) { out_i32[2 * s] = a; out_i32[2 * s + 1] = b; }
            if (out_f32) { out_f32[2 * s] = (float)a * scale; out_f32[2 * s + 1] = (float)b * scale; }
        } else {
            for (int c = 0; c < channels; c++) {
                const int v = planar[(long long)c * total + s];
                if (out_i32) out_i32[s * channels + c] = v;
                if (out_f32) out_f32[s * channels + c] = (float)v * scale;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- WAV / raw PCM
// fmt: 0 u8, 1 s16le, 2 s24le packed, 3 s32le, 4 f32le, 5 f64le  ->  f32 (and int32 for the integer formats when asked)
__global__ __launch_bounds__(256) void k_pcm_convert(const uint8_t *__restrict__ raw, long long nvals, int fmt,
                                                     float *__restrict__ out_f32, int *__restrict__ out_i32)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvals) return;
    float f = 0.f; int v = 0;
    if (fmt == 0) { v = (int)raw[i] - 128; f = (float)v * (1.0f / 128.0f); }
    else if (fmt == 1) { v = (short)(raw[2 * i] | (raw[2 * i + 1] << 8)); f = (float)v * (1.0f / 32768.0f); }
    else if (fmt == 2) { v = (int)((unsigned)raw[3 * i] << 8 | (unsigned)raw[3 * i + 1] << 16 | (unsigned)raw[3 * i + 2] << 24) >> 8; f = (float)v * (1.0f / 8388608.0f); }
    else if (fmt == 3) { v = (int)((unsigned)raw[4 * i] | (unsigned)raw[4 * i + 1] << 8 | (unsigned)raw[4 * i + 2] << 16 | (unsigned)raw[4 * i + 3] << 24); f = (float)v * (1.0f / 2147483648.0f); }
    else if (fmt == 4) { const unsigned u = (unsigned)raw[4 * i] | (unsigned)raw[4 * i + 1] << 8 | (unsigned)raw[4 * i + 2] << 16 | (unsigned)raw[4 * i + 3] << 24; f = __uint_as_float(u); }
    else { unsigned long long u = 0; for (int b = 0; b < 8; b++) u |= (unsigned long long)raw[8 * i + b] << (8 * b); f = (float)__longlong_as_double((long long)u); }
    if (out_f32) out_f32[i] = f;
    if (out_i32) out_i32[i] = v;
}
} // namespace fd
} // namespace

// ---------------------------------------------------------------------------------------------------------------- launchers
static fd::StreamInfo mk_si(const JtFlacStream &s)
{
    fd::StreamInfo si; si.channels = s.channels; si.bps = s.bps; si.sample_rate = s.sample_rate; si.min_bs = s.min_blocksize;
    si.max_bs = s.max_blocksize; si.pad = 0; si.audio_offset = s.audio_offset; si.len = s.len;
    return si;
}
static_assert(sizeof(fd::Cand) == sizeof(JtFlacCand), "candidate layout");
static_assert(sizeof(fd::Parsed) == sizeof(JtFlacParsed), "parsed layout");
static_assert(sizeof(fd::Frame) == sizeof(JtFlacFrame), "frame layout");

void launch_flacdec_find(const uint8_t *file, const JtFlacStream &s, JtFlacCand *cands, int *count, int cap, hipStream_t st)
{
    const long long span = s.len - s.audio_offset;
    if (span <= 0) return;
    hipLaunchKernelGGL(fd::k_flac_find, dim3((unsigned)((span + 255) / 256)), dim3(256), 0, st, file, mk_si(s),
                       reinterpret_cast<fd::Cand *>(cands), count, cap);
    JT_HIP(hipGetLastError());
}
void launch_flacdec_parse(const uint8_t *file, const JtFlacStream &s, const JtFlacCand *cands, int ncand, JtFlacParsed *out,
                          hipStream_t st)
{
    if (ncand <= 0) return;
    hipLaunchKernelGGL(fd::k_flac_parse, dim3((unsigned)((ncand + 63) / 64)), dim3(64), 0, st, file, mk_si(s),
                       reinterpret_cast<const fd::Cand *>(cands), ncand, reinterpret_cast<fd::Parsed *>(out));
    JT_HIP(hipGetLastError());
}
void launch_flacdec_decode(const uint8_t *file, const JtFlacStream &s, const JtFlacFrame *frames, long long nframes, long long total,
                           int *planar, int *errs, int *out_i32, float *out_f32, hipStream_t st)
{
    if (nframes <= 0) return;
    const long long nsub = nframes * s.channels;
    hipLaunchKernelGGL(fd::k_flac_decode, dim3((unsigned)((nsub + 63) / 64)), dim3(64), 0, st, file, mk_si(s),
                       reinterpret_cast<const fd::Frame *>(frames), nsub, total, planar, errs);
    hipLaunchKernelGGL(fd::k_flac_finish, dim3((unsigned)nframes), dim3(256), 0, st, reinterpret_cast<const fd::Frame *>(frames),
                       s.channels, total, planar, out_i32, out_f32, (float)(1.0 / (double)(1ull << (s.bps - 1))));
    JT_HIP(hipGetLastError());
}
void launch_flacdec_decode_cand(const uint8_t *file, const JtFlacStream &s, const JtFlacCand *cands, int ncand, int *rows, JtFlacParsed *parsed, hipStream_t st)
{
    if (ncand <= 0) return;
    hipLaunchKernelGGL(fd::k_flac_decode_cand, dim3((unsigned)((ncand + 63) / 64)), dim3(64), 0, st, file, mk_si(s),
                       reinterpret_cast<const fd::Cand *>(cands), ncand, rows, reinterpret_cast<fd::Parsed *>(parsed));
    JT_HIP(hipGetLastError());
}
void launch_flacdec_finish_cand(const JtFlacStream &s, const JtFlacFrame *frames, long long nframes, const int *rows,
                                int *out_i32, float *out_f32, hipStream_t st)
{
    if (nframes <= 0) return;
    hipLaunchKernelGGL(fd::k_flac_finish_cand, dim3((unsigned)nframes), dim3(256), 0, st, reinterpret_cast<const fd::Frame *>(frames), s.max_blocksize,
                       rows, out_i32, out_f32, (float)(1.0 / (double)(1ull << (s.bps - 1))));
    JT_HIP(hipGetLastError());
}
void launch_pcm_convert(const uint8_t *raw, long long nvals, int fmt, float *out_f32, int *out_i32, hipStream_t st)
{
    if (nvals <= 0) return;
    hipLaunchKernelGGL(fd::k_pcm_convert, dim3((unsigned)((nvals + 255) / 256)), dim3(256), 0, st, raw, nvals, fmt, out_f32, out_i32);
    JT_HIP(hipGetLastError());
}
