"""Round-3 behaviours: BASELINE configs[3]'s per-GPU share (ten-minute files queued over one GPU, several in flight) gives the
bytes the one-at-a-time path gives; a device that cannot be opened does not take files away from the devices that can
(cmd/jivetalking/pool.go:122-153: one failure never stops the others)."""
import os

import numpy as np
import pytest

from jivetalking_amd import synth, hostlogic as H, _lib as L
from jivetalking_amd.engine import Engine

pytestmark = pytest.mark.gpu
SR = 48000


def _ten_minute_flacs(engine, d, count, minutes=10.0):
    """`count` different files: a seeded 60 s talker repeated to length (numpy: torch's HIP runtime cannot be initialised after the
    library's in one process), every second file with a plosive-like burst every 1.5 s so that its plan needs the limiter prefix."""
    paths = []
    for k in range(count):
        base = np.asarray(synth.speech_like(min(60.0, minutes * 60.0), SR, seed=300 + k), np.float64)
        x = np.tile(base, int(np.ceil(minutes * 60.0 / (base.size / SR))))[: int(minutes * 60.0 * SR)].copy()
        if k % 2:
            w = int(0.02 * SR)
            burst = 0.35 * np.hanning(w) * np.sin(2 * np.pi * 180.0 * np.arange(w) / SR)
            for pos in range(SR, x.size - SR, int(1.5 * SR)):
                x[pos:pos + w] += burst
        pcm = np.clip(np.rint(x * 32768), -32768, 32767).astype(np.int16)
        p = os.path.join(str(d), f"ten{k}.flac")
        with open(p, "wb") as f:
            f.write(engine.op_flac_encode(pcm, SR, md5=True))
        paths.append(p)
    return paths


def test_ten_minute_files_in_flight_equal_one_at_a_time(engine, tmp_path):
    """8 ten-minute files with 4 in flight (configs[3], one GPU's share in small) vs the same files one at a time on one handle: the
    output files are byte-identical and every result field that is a measurement is equal."""
    paths = _ten_minute_flacs(engine, tmp_path, 8)
    want = []
    for p in paths:
        res, out_path, _ = H.process_file(engine, p)
        want.append((res.output_lufs, res.output_tp_db, res.input_lufs, int(res.limiter.needed), open(out_path, "rb").read()))
        os.unlink(out_path)
    assert {w[3] for w in want} == {0, 1}                      # both plans occur: with and without the limiter prefix
    failed, res, dev = H.process_files_multi(paths, devices=(0,), in_flight_per_device=4)
    assert failed == 0 and all(d == 0 for d in dev)
    for k in range(8):
        r = res[k]
        assert r.rc == 0 and r.wall_ms > 0
        assert (r.result.output_lufs, r.result.output_tp_db, r.result.input_lufs, int(r.result.limiter.needed)) == want[k][:4]
        assert open(r.output_path.decode(), "rb").read() == want[k][4], f"file {k}: bytes differ with 4 in flight"
        assert abs(r.result.output_lufs + 16.0) <= 0.1 and r.result.output_tp_db <= -1.0
    assert not [q for q in os.listdir(str(tmp_path)) if q.startswith(".processing-")]


def test_a_device_that_cannot_be_opened_takes_no_files(engine, tmp_path):
    """a worker whose jt_open failed used to pop files and fail them in microseconds.  devices = {0, 99}: every file is
    served by device 0; devices = {99}: every file fails with the open error, none is left in its initial state."""
    paths = _ten_minute_flacs(engine, tmp_path, 5, minutes=0.25)
    failed, res, dev = H.process_files_multi(paths, devices=(0, 99), in_flight_per_device=1)
    assert failed == 0 and all(d == 0 for d in dev) and all(res[k].rc == 0 for k in range(5))
    failed, res, dev = H.process_files_multi(paths[:1], devices=(99, 0), in_flight_per_device=1)   # fewer files than devices: the
    assert failed == 0 and dev == [0]                                                                # worker moves on to a spare one
    failed, res, dev = H.process_files_multi(paths, devices=(99,), in_flight_per_device=2)
    assert failed == 5 and all(res[k].rc != 0 and b"jt_open(99)" in res[k].error for k in range(5)) and dev == [-1] * 5
