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
_OPTION_DEFAULTS = {"region_rot": "-1"}


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
