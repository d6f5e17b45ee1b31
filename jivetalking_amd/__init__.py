"""jivetalking_amd — MI355X (gfx950) engine for jivetalking's four-pass speech-mastering hot path.

Only what the path needs: csrc/ (HIP kernels + C ABI -> lib/libjtgpu.so), the ctypes binding
(_lib.py), a thin per-file engine wrapper (engine.py), the host mirror of the reference's
internal/processor control logic (processor.py -> C++ in csrc/host via the same .so), and the
deterministic synthetic inputs used by tests and bench (synth.py).
"""
from ._lib import load, JtError  # noqa: F401
from .engine import Engine  # noqa: F401
