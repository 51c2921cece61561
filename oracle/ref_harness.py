"""TEST INFRASTRUCTURE — imports the real reference from /root/reference (this container only).

Used by tests/golden/make_golden.py to generate golden vectors and by the
``-m "not gpu"`` tests to re-validate the oracle when /root/reference is present.  The
reference never travels to the GPU box; nothing on the product path, in ``-m gpu`` tests,
``smoke()`` or ``bench.py`` may import this module.

Two shims are needed to import /root/reference/src/scale_calculator.py on this image:
  * a stub ``cv2`` module (imported at /root/reference/src/estimate_road_norm.py:3; only
    used by drawing code, /root/reference/src/scale_calculator.py:607-609);
  * ``np.float = float`` (used at /root/reference/src/scale_calculator.py:32; removed in
    NumPy >= 1.24).
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_SRC, "scale_calculator.py"))


_cached = None


def load_reference():
    """Return the reference's ``scale_calculator`` module (imported once)."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError("reference not present at " + REFERENCE_SRC)
    import numpy as np
    import matplotlib
    matplotlib.use("Agg")
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if not hasattr(np, "float"):
        np.float = float  # noqa: NPY001 - shim for the reference only
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import scale_calculator  # type: ignore
    _cached = scale_calculator
    return scale_calculator


@contextlib.contextmanager
def quiet():
    """The reference prints from inside the hot path; swallow it."""
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield buf
