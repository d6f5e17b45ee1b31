"""FLAC output leg: the GPU encoder behind jt_flac_encode / jt_op_flac_encode_s16 against the RFC 9639 oracle
decoder.  FLAC is lossless, so the bar is bit-exact PCM after decode, every header CRC-8 / frame CRC-16 right, and a STREAMINFO
that describes the stream truthfully (rate, channels, depth, total samples, min/max frame size, MD5 of the PCM) with the
reference's fixed 4096-sample blocks (encoder.go:93-100)."""
import hashlib
import struct
import os

import numpy as np
import pytest

from jivetalking_amd import synth, _lib as L
from jivetalking_amd.engine import Engine, default_filter_params

pytestmark = pytest.mark.gpu


def roundtrip(engine, oracle, pcm, rate=44100, md5=True):
    pcm = np.ascontiguousarray(pcm, np.int16)
    data, info = engine.op_flac_encode(pcm, rate, md5=md5, return_info=True)
    rc, dec, oi = oracle.flac_decode(data)
    assert rc == 0, f"oracle decoder rejected the stream (rc={rc})"
    assert oi.crc8_errors == 0 and oi.crc16_errors == 0
    assert dec.shape == (pcm.size, 1) and np.array_equal(dec[:, 0], pcm.astype(np.int32))
    assert (oi.sample_rate, oi.channels, oi.bps, oi.total_samples) == (rate, 1, 16, pcm.size)
    assert oi.min_blocksize == 4096 and oi.max_blocksize == 4096 and not oi.variable_blocksize
    assert oi.frames == (pcm.size + 4095) // 4096 and oi.last_blocksize == (pcm.size - 1) % 4096 + 1
    assert (oi.min_framesize, oi.max_framesize) == (oi.obs_min_framesize, oi.obs_max_framesize)
    assert info["bytes"] == len(data) and info["frames"] == oi.frames
    if md5:
        assert bytes(oi.md5_stored) == bytes(oi.md5_decoded) == hashlib.md5(pcm.tobytes()).digest()
    else:
        assert bytes(oi.md5_stored) == bytes(16)
    return data, info


def speech_s16(seconds, sr=44100, seed=1, dbfs=-20.0):
    x = np.asarray(synth.speech_like(seconds, sr, seed=seed, speech_dbfs=dbfs), np.float64)
    return np.clip(np.rint(x * 32768), -32768, 32767).astype(np.int16)


def test_speech_roundtrip_and_compression(engine, oracle):
    pcm = speech_s16(30.0)
    data, _ = roundtrip(engine, oracle, pcm)
    # a linear predictor + Rice coder must beat raw s16 clearly on speech at -20 dBFS
    assert len(data) < 0.6 * pcm.nbytes
    # deterministic bytes
    assert engine.op_flac_encode(pcm, 44100) == data


@pytest.mark.parametrize("n", [1, 2, 9, 16, 17, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 8191, 8192, 12289 + 37])
def test_every_length_class(engine, oracle, n):
    rng = np.random.default_rng(n)
    roundtrip(engine, oracle, (1500 * rng.standard_normal(n)).astype(np.int16))


def test_degenerate_signals(engine, oracle):
    rng = np.random.default_rng(5)
    data, _ = roundtrip(engine, oracle, np.zeros(50000, np.int16))
    assert len(data) < 400                                            # CONSTANT subframes
    roundtrip(engine, oracle, np.full(10000, -1234, np.int16))
    roundtrip(engine, oracle, np.full(4096 * 3, 32767, np.int16))
    roundtrip(engine, oracle, np.full(4096 * 3, -32768, np.int16))
    white = rng.integers(-32768, 32768, 30000).astype(np.int16)
    data, _ = roundtrip(engine, oracle, white)
    assert len(data) < white.nbytes * 1.01                            # VERBATIM fallback bounds the expansion
    roundtrip(engine, oracle, np.tile(np.array([32767, -32768], np.int16), 6000))
    imp = np.zeros(20000, np.int16); imp[::997] = 32767; imp[5::1013] = -32768
    roundtrip(engine, oracle, imp)                                    # long unary runs inside large partitions
    roundtrip(engine, oracle, rng.integers(-1, 2, 40000).astype(np.int16))
    # window-edge only energy: the Welch window gives sample 0 zero weight -> no usable autocorrelation
    edge = np.zeros(8192, np.int16); edge[0] = 900; edge[4096] = -77
    roundtrip(engine, oracle, edge)
    # a frame that switches from silence to full-scale noise half way (Rice parameters differ wildly per partition)
    mix = np.zeros(4096 * 4, np.int16); mix[6000:9000] = rng.integers(-32768, 32768, 3000)
    roundtrip(engine, oracle, mix)
    t = np.arange(100000) / 44100.0
    data, _ = roundtrip(engine, oracle, (20000 * np.sin(2 * np.pi * 1000 * t)).astype(np.int16))
    assert len(data) < 0.3 * 200000


@pytest.mark.parametrize("rate", [8000, 11000, 12345, 22050, 44100, 48000, 96000, 192000, 352800, 655350])
def test_sample_rate_codes(engine, oracle, rate):
    roundtrip(engine, oracle, speech_s16(1.0, seed=3)[:30000], rate)


def test_frame_numbers_cross_utf8_lengths(engine, oracle):
    # frame numbers 127->128 (2-byte) and 2047->2048 (3-byte) in the UTF-8-like coding: 2100 frames
    pcm = np.tile(speech_s16(8.0, seed=4), 25)[: 4096 * 2100 + 11]
    roundtrip(engine, oracle, pcm)


def test_md5_flag_off_leaves_the_signature_unknown(engine, oracle):
    roundtrip(engine, oracle, speech_s16(2.0, seed=6), md5=False)


def test_stage_outputs_encode_to_the_downloaded_pcm(oracle):
    from jivetalking_amd import hostlogic
    e = Engine()
    try:
        sr = 48000
        x = synth.speech_like(20.0, sr, seed=11)
        e.upload_pcm(x, sr, 1)
        with pytest.raises(Exception):
            e.flac_encode(4)                                          # nothing on the device yet
        res = hostlogic.process_audio(e)                              # the four passes, as the reference's ProcessAudio
        assert res is not None
        for stage in (2, 4):
            pcm = e.download_s16(stage)
            data, info = e.flac_encode(stage, return_info=True)
            rc, dec, oi = oracle.flac_decode(data)
            assert rc == 0 and np.array_equal(dec[:, 0], pcm.astype(np.int32))
            assert oi.sample_rate == 44100 and oi.total_samples == pcm.size
            assert bytes(oi.md5_stored) == hashlib.md5(pcm.tobytes()).digest()
            assert info["sample_rate"] == 44100 and info["block_size"] == 4096
    finally:
        e.close()


# ================================================================ input leg: jt_load_audio / jt_op_decode_audio
def coverage_signal(rng, n, ch, bps):
    x = rng.standard_normal((n, ch)).cumsum(0) * (1 << (bps - 6)) / 30
    return x.clip(-(1 << (bps - 1)), (1 << (bps - 1)) - 1).astype(np.int32)


# oracle coverage-encoder modes (oracle/orc_flac.c): predictor | 8 escape | 16 five-bit rice | 32 variable blocks | 64.. stereo
@pytest.mark.parametrize("mode", [0, 1, 2, 2 | 8, 2 | 16, 1 | 32, 2 | 32, 2 | 64, 2 | 128, 2 | 192, 1 | 16 | 8])
def test_decoder_covers_the_format(engine, oracle, mode):
    rng = np.random.default_rng(mode)
    for ch in (1, 2, 3, 6):
        if ch > 2 and mode >= 64:
            continue
        for bps, order in ((8, 3), (16, 8), (24, 32), (16, 12), (20, 5), (12, 2)):
            x = coverage_signal(rng, 20000 + 77, ch, bps)
            if mode & 8:
                x[:512] &= ~7                                        # wasted bits
            data = oracle.flac_encode(x, 44100, bps, 1152 if mode & 32 else 4096, mode, order)
            rc, ref, _ = oracle.flac_decode(data)
            assert rc == 0 and np.array_equal(ref, x)                 # the oracle agrees with itself first
            i32, f32, meta = engine.op_decode_audio(data)
            assert np.array_equal(i32, x), (mode, ch, bps)
            # libswresample s16/s32 -> flt: exact scaling by 2^(1-bits)
            assert np.array_equal(f32, (x.astype(np.float64) / (1 << (bps - 1))).astype(np.float32))
            assert (meta["sample_rate"], meta["channels"], meta["bits_per_sample"], meta["frames"]) == (44100, ch, bps, x.shape[0])
            assert abs(meta["duration_s"] - x.shape[0] / 44100) < 1e-12


def test_decoder_block_sizes_and_rates(engine, oracle):
    rng = np.random.default_rng(3)
    for bs, rate in ((16, 8000), (192, 22050), (576, 48000), (1000, 96000), (4608, 192000), (65535, 44100), (256, 12345)):
        x = coverage_signal(rng, bs * 3 + bs // 2 + 1, 2, 16)
        data = oracle.flac_encode(x, rate, 16, bs, 2, 6)
        i32, _, meta = engine.op_decode_audio(data)
        assert np.array_equal(i32, x) and meta["sample_rate"] == rate


def test_gpu_encoder_to_gpu_decoder_roundtrip(engine):
    pcm = np.tile(speech_s16(10.0, seed=9), 30)                       # 5 minutes, 3 230 frames
    data = engine.op_flac_encode(pcm, 44100)
    i32, f32, meta = engine.op_decode_audio(data)
    assert np.array_equal(i32[:, 0], pcm.astype(np.int32))
    assert np.array_equal(f32[:, 0], pcm.astype(np.float32) / 32768.0)
    assert meta["flac_frames"] == (pcm.size + 4095) // 4096


def _crc8(b):
    c = 0
    for x in b:
        c ^= x
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xff if c & 0x80 else (c << 1) & 0xff
    return c


def _crc16(b):
    c = 0
    for x in b:
        c ^= x << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
    return c


def test_header_lookalikes_inside_audio_data_are_not_frames(engine, oracle):
    """Verbatim frames carry the PCM bytes as they are, so a valid-looking frame can sit inside one.  A header with a right
    CRC-8 (fails its CRC-16), and a complete 11-byte frame with both CRCs right (parses, but nothing links to it)."""
    rng = np.random.default_rng(8)
    x = rng.integers(-32768, 32768, 4096 * 6).astype(np.int16)       # white noise: the GPU encoder emits VERBATIM subframes
    hdr = bytes([0xff, 0xf8, 0xc9, 0x08, 0x02])                       # fixed 4096, 44.1 kHz, mono, 16 bit, frame number 2
    fake1 = hdr + bytes([_crc8(hdr)])
    body = hdr + bytes([_crc8(hdr)]) + bytes([0x00, 0x12, 0x34])      # CONSTANT subframe, value 0x1234
    fake2 = body + _crc16(body).to_bytes(2, "big") + b"\x00"
    for at, fake in ((5000, fake1), (4096 * 3 + 100, fake2)):
        x[at:at + len(fake) // 2] = np.frombuffer(fake, ">i2")
    data = engine.op_flac_encode(x, 44100)
    rc, ref, _ = oracle.flac_decode(data)
    assert rc == 0 and np.array_equal(ref[:, 0], x)
    i32, _, meta = engine.op_decode_audio(data)
    assert np.array_equal(i32[:, 0], x.astype(np.int32))
    assert meta["flac_candidates"] >= meta["flac_frames"] + 2         # both look-alikes were seen and rejected


def test_damaged_and_foreign_inputs_fail_loudly(engine):
    pcm = speech_s16(3.0, seed=2)
    data = bytearray(engine.op_flac_encode(pcm, 44100))
    for pos in (len(data) // 2, len(data) - 1, 120):
        bad = bytearray(data); bad[pos] ^= 0x04
        with pytest.raises(L.JtError) as ei:
            engine.op_decode_audio(bytes(bad))
        assert ei.value.code == L.JT_E_INVAL
    with pytest.raises(L.JtError) as ei:
        engine.op_decode_audio(bytes(data[: len(data) * 2 // 3]))     # truncated
    assert ei.value.code == L.JT_E_INVAL
    with pytest.raises(L.JtError) as ei:
        engine.op_decode_audio(b"OggS" + bytes(200))
    assert ei.value.code == L.JT_E_UNSUPPORTED
    # an ID3v2 tag in front of the stream is skipped
    tag = b"ID3\x04\x00\x00" + bytes([0, 0, 0, 20]) + bytes(20)
    i32, _, _ = engine.op_decode_audio(tag + bytes(data))
    assert np.array_equal(i32[:, 0], pcm.astype(np.int32))


def _wav(fmt_tag, bits, ch, rate, payload, extensible=False, extra_chunks=True):
    import struct
    align = ch * bits // 8
    if extensible:
        guid = struct.pack("<H", fmt_tag) + bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack("<HHIIHHHHI", 0xfffe, ch, rate, rate * align, align, bits, 22, bits, 0) + guid
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, ch, rate, rate * align, align, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if extra_chunks:
        chunks += b"LIST" + struct.pack("<I", 5) + b"INFOx" + b"\x00"      # odd-sized chunk + pad byte
    chunks += b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


def test_wav_formats(engine):
    rng = np.random.default_rng(4)
    n, ch = 30011, 2
    f = rng.standard_normal((n, ch)).astype(np.float32) * 0.3
    s32 = rng.integers(-2**31, 2**31, (n, ch)).astype(np.int64)
    cases = []
    u8 = rng.integers(0, 256, (n, ch)).astype(np.uint8)
    cases.append((1, 8, u8.tobytes(), (u8.astype(np.int32) - 128), (u8.astype(np.float32) - 128) / 128, False))
    s16 = (s32 >> 16).astype(np.int16)
    cases.append((1, 16, s16.astype("<i2").tobytes(), s16.astype(np.int32), s16.astype(np.float32) / 32768, False))
    s24 = (s32 >> 8).astype(np.int32)
    cases.append((1, 24, s24.astype("<i4").reshape(-1, 1).view(np.uint8)[:, :3].tobytes(), s24, s24.astype(np.float32) / 8388608, True))
    cases.append((1, 32, s32.astype("<i4").tobytes(), s32.astype(np.int32), (s32.astype(np.int32)).astype(np.float32) / np.float32(2147483648.0), False))
    cases.append((3, 32, f.astype("<f4").tobytes(), None, f, False))
    d = f.astype(np.float64) * 1.0000001
    cases.append((3, 64, d.astype("<f8").tobytes(), None, d.astype(np.float32), True))
    for tag, bits, payload, want_i, want_f, ext in cases:
        i32, f32, meta = engine.op_decode_audio(_wav(tag, bits, ch, 48000, payload, extensible=ext))
        assert (meta["format"], meta["sample_rate"], meta["channels"], meta["bits_per_sample"], meta["frames"]) == (2, 48000, ch, bits, n)
        assert meta["is_float"] == (tag == 3)
        if want_i is not None:
            assert np.array_equal(i32, want_i), bits
        assert np.array_equal(f32, want_f.astype(np.float32)), (tag, bits)
    with pytest.raises(L.JtError) as ei:
        engine.op_decode_audio(_wav(2, 4, 1, 8000, bytes(100)))          # ADPCM
    assert ei.value.code == L.JT_E_UNSUPPORTED


def test_rf64_files_are_read_with_their_ds64_sizes(engine):
    """RF64 / BW64 (EBU Tech 3306: what a recorder writes once a WAV passes 4 GB; libavformat's wav demuxer reads them, wavdec.c): magic
    "RF64", every 32-bit size 0xFFFFFFFF, the real data size in the ds64 chunk -- and trailing chunks BEHIND the data must not be taken
    for samples.  Same samples, same cadence (the wav demuxer's 4096-byte packets) as the RIFF form of the file."""
    rng = np.random.default_rng(77)
    x = (rng.standard_normal(20000 * 2) * 0.1).astype("<f4")
    payload = x.tobytes()
    fmt = struct.pack("<HHIIHH", 3, 2, 48000, 48000 * 8, 8, 32)
    ds64 = struct.pack("<QQQI", 0, len(payload), 20000, 0)
    tail = b"LIST" + struct.pack("<I", 8) + b"INFOxxxx"                       # a chunk behind the data
    for magic in (b"RF64", b"BW64"):
        img = (magic + struct.pack("<I", 0xFFFFFFFF) + b"WAVE" + b"ds64" + struct.pack("<I", len(ds64)) + ds64 +
               b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", 0xFFFFFFFF) + payload + tail)
        _, f32, meta = engine.op_decode_audio(img, want_i32=False)
        assert (meta["format"], meta["channels"], meta["frames"], meta["is_float"], meta["decoder_frame_samples"]) == (2, 2, 20000, 1, 512)
        assert np.array_equal(f32.reshape(-1), x)


def test_layouts_beyond_the_restated_matrix_are_refused_not_averaged(engine, oracle):
    """Round 6: surround layouts of FL FR FC LFE BL BR FLC FRC BC SL SR are down-mixed with libswresample's default matrix
    (tests/test_gpu_round6.py holds them against the oracle's); what stays refused is a layout with channels the restated matrix has no
    row for (top channels) and a mask that does not match the channel count -- JT_E_UNSUPPORTED, never an average with made-up weights."""
    x = np.zeros((4000, 3), np.float32).reshape(-1)
    for mask in (0x804, 0x3):                      # FC + top centre (bit 11); two bits for three channels
        with pytest.raises(L.JtError) as ei:
            engine.upload_pcm(x, 48000, 3, channel_mask=mask)
        assert ei.value.code == L.JT_E_UNSUPPORTED
    x6 = coverage_signal(np.random.default_rng(2), 9000, 6, 16)
    data = oracle.flac_encode(x6, 48000, 16, 4096, 2, 8)
    meta = engine.load_audio(data)                                     # 5.1 (side): libavcodec's layout for a six-channel FLAC
    assert (meta["channels"], meta["channel_mask"]) == (6, 0x60F)
    i32, _, _ = engine.op_decode_audio(data)
    assert np.array_equal(i32, x6)


def test_load_audio_feeds_the_passes_like_upload_pcm(oracle):
    """jt_load_audio of a 24-bit stereo FLAC == jt_upload_pcm of the same samples as f32: identical Pass-1 analysis."""
    from jivetalking_amd import hostlogic
    sr = 48000
    x = np.asarray(synth.speech_like(12.0, sr, seed=21), np.float64)
    st = np.stack([x, 0.8 * np.roll(x, 7)], 1)
    pcm = np.clip(np.rint(st * (1 << 23)), -(1 << 23), (1 << 23) - 1).astype(np.int32)
    data = oracle.flac_encode(pcm, sr, 24, 4096, 2 | 192, 8)
    a = Engine(); b = Engine()
    try:
        meta = a.load_audio(data)
        assert (meta["frames"], meta["channels"], meta["sample_rate"], meta["bits_per_sample"]) == (pcm.shape[0], 2, sr, 24)
        b.upload_pcm((pcm.astype(np.float64) / (1 << 23)).astype(np.float32), sr, 2)
        b.set_source_format(24, False)              # what jt_load_audio records for a 24-bit file (its band graphs run in s32p)
        ra = hostlogic.process_audio(a, analyse_only=True)
        rb = hostlogic.process_audio(b, analyse_only=True)
        import ctypes as C
        assert bytes(C.string_at(C.addressof(ra.input), C.sizeof(ra.input))) == bytes(C.string_at(C.addressof(rb.input), C.sizeof(rb.input)))
    finally:
        a.close(); b.close()


def test_process_file_flac_and_wav_in_flac_out(tmp_path, oracle):
    """jt_process_file == ProcessAudio(inputPath): the same result for a FLAC and a WAV carrying the same samples, output named
    <name>-LUFS-<n>-processed.flac (processor.go:379-388), 44.1 kHz / 16 bit / mono (filters.go:20), decodable, MD5 right, and
    equal to the in-memory path (upload_pcm + process_audio + download_s16)."""
    import struct
    from jivetalking_amd import hostlogic
    sr = 48000
    x = np.asarray(synth.speech_like(15.0, sr, seed=31), np.float64)
    pcm = np.clip(np.rint(x * 32768), -32768, 32767).astype(np.int16)
    flac_in = tmp_path / "take one.flac"
    flac_in.write_bytes(oracle.flac_encode(pcm.astype(np.int32), sr, 16, 4096, 2, 8))
    wav_in = tmp_path / "take two.wav"
    wav_in.write_bytes(_wav(1, 16, 1, sr, pcm.astype("<i2").tobytes(), extra_chunks=False))
    e = Engine()
    try:
        e.upload_pcm(pcm.astype(np.float32) / 32768.0, sr, 1)
        e.set_source_format(16, False)              # as jt_load_audio records for a 16-bit file (its band graphs run in s16p)
        ref = hostlogic.process_audio(e)
        want = e.download_s16(4)
        outs = []
        for src in (flac_in, wav_in):
            res, out_path, io_ms = hostlogic.process_file(e, src)
            n = int(round(abs(res.output_lufs)))
            assert out_path == str(tmp_path / f"{src.stem}-LUFS-{n}-processed.flac")
            data = open(out_path, "rb").read()
            rc, dec, oi = oracle.flac_decode(data)
            assert rc == 0 and (oi.sample_rate, oi.channels, oi.bps) == (44100, 1, 16)
            assert bytes(oi.md5_stored) == bytes(oi.md5_decoded)
            assert np.array_equal(dec[:, 0], want.astype(np.int32))
            assert res.output_lufs == ref.output_lufs and res.output_tp_db == ref.output_tp_db
            outs.append(data)
            assert all(v >= 0 for v in io_ms)
        assert outs[0] == outs[1]
        with pytest.raises(L.JtError) as ei:
            hostlogic.process_file(e, tmp_path / "missing.flac")
        assert ei.value.code == L.JT_E_INVAL
    finally:
        e.close()


def test_metadata_blocks_of_every_kind_are_skipped(engine, oracle):
    """Real files carry SEEKTABLE / VORBIS_COMMENT / PADDING / APPLICATION / PICTURE blocks between STREAMINFO and the audio; a
    big PICTURE block full of sync-code look-alikes must not confuse the frame search (it starts after the metadata)."""
    rng = np.random.default_rng(12)
    x = coverage_signal(rng, 30000, 2, 16)
    base = oracle.flac_encode(x, 44100, 16, 4096, 2 | 192, 8)
    assert base[4] == 0x80 and base[5:8] == bytes([0, 0, 34])        # STREAMINFO is the only (last) block
    streaminfo = bytes([0x00]) + base[5:8] + base[8:42]               # same block, "last" flag cleared
    def block(kind, payload, last=False):
        return bytes([kind | (0x80 if last else 0)]) + len(payload).to_bytes(3, "big") + payload
    lookalike = bytes([0xff, 0xf8, 0xc9, 0x18, 0x00]) * 5000          # 25 kB of frame-header prefixes
    blocks = (block(3, bytes(18 * 10)) +                              # SEEKTABLE with placeholder points
              block(4, (4).to_bytes(4, "little") + b"test" + (0).to_bytes(4, "little")) +
              block(2, b"ABCD" + bytes(60)) +                         # APPLICATION
              block(6, lookalike) +                                   # PICTURE (opaque here)
              block(1, bytes(8192), last=True))                       # PADDING, last
    data = base[:4] + streaminfo + blocks + base[42:]
    rc, ref, _ = oracle.flac_decode(data)
    assert rc == 0 and np.array_equal(ref, x)
    i32, _, meta = engine.op_decode_audio(data)
    assert np.array_equal(i32, x) and meta["flac_candidates"] == meta["flac_frames"]


def test_process_files_pool_semantics(tmp_path, oracle):
    """jt_process_files == the reference's bounded pool (pool.go:122-228): results per path, the same bytes as one file at a
    time, and a file that cannot be opened fails alone."""
    from jivetalking_amd import hostlogic
    sr = 48000
    paths = []
    for k, seconds in enumerate((12.0, 7.5, 15.0)):
        x = np.asarray(synth.speech_like(seconds, sr, seed=41 + k), np.float64)
        pcm = np.clip(np.rint(x * 32768), -32768, 32767).astype(np.int16)
        p = tmp_path / f"ep{k}.wav"
        p.write_bytes(_wav(1, 16, 1, sr, pcm.astype("<i2").tobytes(), extra_chunks=False))
        paths.append(p)
    e = Engine()
    try:
        want = []
        for p in paths:
            res, out_path, _ = hostlogic.process_file(e, p)
            want.append((res.output_lufs, open(out_path, "rb").read()))
    finally:
        e.close()
    mixed = [paths[0], tmp_path / "nope.flac", paths[1], paths[2]]
    failed, res = hostlogic.process_files(mixed, in_flight=3)
    assert failed == 1 and res[1].rc == L.JT_E_INVAL and b"failed to open" in res[1].error
    for i, j in ((0, 0), (2, 1), (3, 2)):
        assert res[i].rc == 0 and res[i].result.output_lufs == want[j][0]
        assert open(res[i].output_path.decode(), "rb").read() == want[j][1]
        assert res[i].wall_ms > 0


def test_process_files_multi_shared_queue(tmp_path, oracle):
    """jt_process_files_multi over devices = {0, 0} (the only topology a one-GPU box offers): one shared queue, both worker sets get
    work, every file has its own result identical to the single-file path, longest file first, failures isolated."""
    from jivetalking_amd import hostlogic
    sr = 48000
    paths = []
    for k, seconds in enumerate((12.0, 7.5, 15.0, 9.0, 13.0)):
        x = np.asarray(synth.speech_like(seconds, sr, seed=41 + k), np.float64)
        pcm = np.clip(np.rint(x * 32768), -32768, 32767).astype(np.int16)
        p = tmp_path / f"multi{k}.wav"
        p.write_bytes(_wav(1, 16, 1, sr, pcm.astype("<i2").tobytes(), extra_chunks=False))
        paths.append(p)
    e = Engine()
    try:
        want = [hostlogic.process_file(e, p)[0].output_lufs for p in paths]
    finally:
        e.close()
    for p in tmp_path.glob("*-processed.flac"):
        p.unlink()
    failed, res, dev = hostlogic.process_files_multi(paths + [tmp_path / "absent.wav"], devices=(0, 0), in_flight_per_device=1)
    assert failed == 1 and res[5].rc == L.JT_E_INVAL
    assert all(d == 0 for d in dev)
    for k in range(5):
        assert res[k].rc == 0 and res[k].result.output_lufs == want[k] and os.path.exists(res[k].output_path.decode())
    assert not list(tmp_path.glob(".processing-*"))
    with pytest.raises(L.JtError):
        hostlogic.process_files_multi(paths, devices=(), in_flight_per_device=1)
