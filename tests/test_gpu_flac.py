"""FLAC output leg: the GPU encoder behind jt_flac_encode / jt_op_flac_encode_s16 against the RFC 9639 oracle
decoder.  FLAC is lossless, so the bar is bit-exact PCM after decode, every header CRC-8 / frame CRC-16 right, and a STREAMINFO
that describes the stream truthfully (rate, channels, depth, total samples, min/max frame size, MD5 of the PCM) with the
reference's fixed 4096-sample blocks (encoder.go:93-100)."""
import hashlib

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
