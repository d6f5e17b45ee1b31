import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
