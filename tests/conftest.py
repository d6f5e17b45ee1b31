import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "hosttable: the reference's own Go tables against the shipped .so; CPU-only arithmetic that is "
                                       "ALSO selected by -m gpu when a GPU is present, so the GPU box's record shows it")


HOST_TABLE_FILES = ("test_host_golden.py", "test_host_election.py", "test_host_intervals.py", "test_host_runrecord.py", "test_abi.py")


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    """The host-logic tables (VAD, elections, AdaptConfig strings, limiter planning, run record) need no GPU, so `-m "not gpu"` runs
    them here.  On a box that has one they are marked `gpu` as well: the driver's `-m gpu` run then exercises them against the same
    libjtgpu.so the kernels ship in."""
    on_gpu_box = os.path.exists("/dev/kfd") and os.environ.get("JT_HOST_TABLES_CPU_ONLY") != "1"
    for it in items:
        if os.path.basename(str(it.fspath)) in HOST_TABLE_FILES:
            it.add_marker(pytest.mark.hosttable)
            if on_gpu_box and it.get_closest_marker("gpu") is None:
                it.add_marker(pytest.mark.gpu)


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): built on demand from oracle/*.c with gcc."""
    from oracle import orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def engine():
    """One engine handle on cuda:0 through the C ABI.  Fails loudly if the HIP library is missing."""
    from jivetalking_amd import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="session")
def engine_ab():
    """A handle on the A/B build of the library (libjtgpu_ab.so, `make ab`): the superseded kernel generations and tuning knobs the
    default build does not contain, for the tests that hold the current kernels against the ones they replaced."""
    from jivetalking_amd import Engine
    e = Engine(0, ab=True)
    yield e
    e.close()


import contextlib

# what jt_set_option's keys fall back to
_OPTION_DEFAULTS = {"region_rot": "-1", "tp_prune_min": str(1 << 20)}


@contextlib.contextmanager
def options(engine, **kv):
    """jt_set_option for the length of a with-block (the switches that used to be JT_* environment variables), restored afterwards."""
    try:
        for k, v in kv.items():
            engine.set_option(k, v)
        yield engine
    finally:
        for k in kv:
            engine.set_option(k, _OPTION_DEFAULTS.get(k, "0"))


_TALKER_DIR = None


def bench_talker(seconds, sr=48000, seed=1000, plosives=40.0, sib_gain=0.25, sib_band=False):
    """bench.py's talker (synth.speech_like_torch: aperiodic, generated on the device) as a numpy array.  torch's HIP runtime and the
    library's cannot both be initialised in one process, so a child process generates it; cached for the session under /dev/shm."""
    import atexit
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    global _TALKER_DIR
    if _TALKER_DIR is None:
        _TALKER_DIR = tempfile.mkdtemp(prefix="jttalk", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        atexit.register(shutil.rmtree, _TALKER_DIR, True)
    path = os.path.join(_TALKER_DIR, f"t{seconds:g}_{sr}_{seed}_{plosives:g}_{sib_gain:g}_{int(sib_band)}.npy")
    if not os.path.exists(path):
        code = ("import sys, numpy as np; sys.path.insert(0, %r); from jivetalking_amd import synth; "
                "x = synth.speech_like_torch(%r, %d, seed=%d, device='cuda:0', plosives_per_min=%r, sib_gain=%r, sib_band=%r); np.save(%r, x.cpu().numpy())"
                % (ROOT, float(seconds), int(sr), int(seed), float(plosives), float(sib_gain), bool(sib_band), path))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path, mmap_mode="r")
