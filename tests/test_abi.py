"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/jtgpu.h declares."""
import os
import re
import ctypes as C
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from jivetalking_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib.load()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "jtgpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(jt_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported(lib):
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/jtgpu.h but not exported"


def test_binding_lists_every_header_symbol():
    from jivetalking_amd import _lib
    assert sorted(_lib.SYMBOLS) == header_symbols()


def test_open_without_gpu_fails_loudly(lib):
    import os
    if os.path.exists("/dev/kfd"):                # (no torch here: a second HIP runtime initialised in the test process hides the GPU from ours)
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.jt_open(0, C.byref(h))
    assert rc == -2  # JT_E_NOGPU: no silent CPU fallback


def test_struct_sizes_match_header():
    """ctypes mirrors must be layout-compatible with the C structs (doubles + ints, natural alignment)."""
    from jivetalking_amd import _lib as L
    assert C.sizeof(L.Spectral) == 13 * 8
    assert C.sizeof(L.Astats) == 22 * 8
    assert C.sizeof(L.R128) == 9 * 8
    assert C.sizeof(L.FrameMeta) == 4 * 8 + 13 * 8
    assert C.sizeof(L.Analysis) == (22 + 9 + 13) * 8 + 16
    assert C.sizeof(L.LimiterPlan) == 24
    assert C.sizeof(L.LoudnormStats) == 9 * 8 + 8
