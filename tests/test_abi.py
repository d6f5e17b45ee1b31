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


def test_product_library_does_not_touch_the_environment():
    """a Go host runs dozens of goroutines; getenv / setenv inside the library would race with them.  The default build
    imports neither (checked on the dynamic symbol table, not by reading the sources); the A/B build imports getenv and nothing that
    writes."""
    import subprocess
    from jivetalking_amd import _lib
    und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {ln.split()[-1].split("@")[0] for ln in und.splitlines() if ln.strip()}
    assert not names & {"getenv", "secure_getenv", "setenv", "putenv", "unsetenv", "clearenv"}, names & {"getenv", "setenv", "putenv"}
    if os.path.exists(_lib.LIB_PATH_AB):
        und = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH_AB], capture_output=True, text=True, check=True).stdout
        names = {ln.split()[-1].split("@")[0] for ln in und.splitlines() if ln.strip()}
        assert "getenv" in names and not names & {"setenv", "putenv", "unsetenv", "clearenv"}


def test_superseded_kernel_generations_are_not_in_the_product_library():
    """The round-1 hop-pair anlmdn, the one-wave dynamic loudnorm, the one-kernel adeclick, the two-sweep K-weighting and the
    frame-at-a-time 2048 / 4096-point afftdn exist in the A/B build only."""
    import subprocess
    from jivetalking_amd import _lib
    gone = ("k_anlmdn_pair<", "k_loudnorm_dynamic(", "k_kw<", "k_afftdn<11", "k_afftdn<12", "k_adeclick_fast<512, 32, 32, true, 0, 0>",
            "k_adeclick_fast<512, 32, 32, false, 0, 0>")
    sym = subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for g in gone:
        assert g not in sym, g
    assert "k_anlmdn_pair3<3>" in sym and "k_afftdn_grp<11" in sym and "k_dk_solve<32" in sym       # (the current ones are)
    if os.path.exists(_lib.LIB_PATH_AB):
        sym = subprocess.run(["nm", "-C", _lib.LIB_PATH_AB], capture_output=True, text=True, check=True).stdout
        for g in gone:
            assert g in sym, g


def test_options_without_a_handle(lib):
    """jt_set_option(NULL, ...): the process-wide keys; fractions of a gigabyte are kept ((size_t)atof("0.5") was 0)."""
    from jivetalking_amd import _lib
    assert lib.jt_build_flags() == 0
    assert lib.jt_set_option(None, b"graveyard_gb", b"0.5") == 0
    assert lib.jt_set_option(None, b"graveyard_gb", b"24") == 0
    assert lib.jt_set_option(None, b"graveyard_gb", b"-1") == _lib.JT_E_INVAL
    assert lib.jt_set_option(None, b"graveyard_gb", b"lots") == _lib.JT_E_INVAL
    assert lib.jt_set_option(None, b"poison_alloc", b"0") == 0
    assert lib.jt_set_option(None, b"adeclick_exact", b"1") == _lib.JT_E_INVAL          # a per-handle key needs a handle
    assert lib.jt_set_option(None, None, b"1") == _lib.JT_E_INVAL
    if os.path.exists(_lib.LIB_PATH_AB):
        assert _lib.load(ab=True).jt_build_flags() == 1
