"""First-use self-check of the Qhull replay against the INSTALLED SciPy.

The reference calls ``scipy.spatial.Delaunay`` with whatever Qhull its SciPy bundles
(/root/reference/src/scale_calculator.py:12,257,266 — no version pin anywhere), and its vote reads the rotation of every
row (:113-115).  ``qhull_rows_kernel`` (and the host replay ``mvosr_qhull_rows_host``) replay the decisions of ONE Qhull
build — qhull_r 7.3.2 (2019.1.r) as bundled with SciPy 1.15.3.  On a host whose SciPy bundles another Qhull the replay and the
installed library could disagree, and a batch call (rows from the replay) and a per-frame call or a declined frame (rows from
the installed SciPy) of the same estimator would then vote on different rotations, both claiming to be "the reference's
result".  So, once per process and device, before the first estimator uses the replay:

* a handful of fixed point sets (seeds below) are triangulated by the replay and by the installed ``scipy.spatial.Delaunay``;
* rows must be identical — set, order, rotation — for every set the replay accepts;
* on ANY difference: one warning naming both versions, and the estimator runs ``triangulation="scipy"`` — the host path,
  which is by construction what the reference computes on this box.

``MVOSR_QHULL_SELFCHECK=0`` skips the check (the replay is trusted).  The result is kept in ``LAST`` per device and travels in
``bench.py``'s line next to the numbers it guards.
"""
from __future__ import annotations

import os
import warnings

import numpy as np

REPLAYED_QHULL = "qhull_r 7.3.2 (2019.1.r 2019/06/21), as bundled with SciPy 1.15.3"

# (kind, points, seed): frame-like sets of the synthetic generator at the sizes the path sees, uniform sets, float32-rounded
# positions (what an OpenCV tracker hands over), a small set.  Fixed: the check is the same on every box.
CHECK_SETS = (("frame", 2000, 9101), ("frame", 1200, 9102), ("frame", 600, 9103), ("frame", 150, 9104),
              ("uniform", 900, 9105), ("uniform", 40, 9106), ("f32", 1500, 9107), ("f32", 300, 9108))

LAST = {}          # device ordinal -> result dict of the last check in this process


def check_point_sets():
    """The fixed point sets of the check, (n, 2) float64 each."""
    from . import synth
    sets = []
    for kind, n, seed in CHECK_SETS:
        if kind == "frame":
            _, f2 = synth.synth_frame(0, n, base_seed=seed)
            sets.append(np.ascontiguousarray(f2))
        else:
            rng = np.random.default_rng(seed)
            p = np.stack([rng.uniform(0.0, synth.IMG_W, n), rng.uniform(186.0, synth.IMG_H, n)], axis=1)
            if kind == "f32":
                p = p.astype(np.float32).astype(np.float64)
            sets.append(np.ascontiguousarray(p))
    return sets


def scipy_versions():
    import scipy
    return {"scipy": scipy.__version__, "numpy": np.__version__}


def compare_rows(replay_rows, scipy_rows):
    """The decision, on plain arrays (no device needed): ``replay_rows[k]`` is the (T,3) int32 array of the replay or ``None``
    where it declined; ``scipy_rows[k]`` the installed SciPy's ``simplices`` or the exception it raised.  Returns
    ``(ok, detail)``: ok iff every set both sides produced is identical row for row (order and rotation) AND at least half of
    the sets were compared (a replay that declines everything checks nothing)."""
    compared, declined, differ = 0, 0, []
    for k, (a, b) in enumerate(zip(replay_rows, scipy_rows)):
        if a is None or isinstance(b, Exception):
            declined += 1
            continue
        compared += 1
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape or not np.array_equal(a, b):
            same_set = a.shape == b.shape and np.array_equal(_canon(a), _canon(b))
            differ.append({"set": k, "rows_replay": int(a.shape[0]), "rows_scipy": int(b.shape[0]),
                           "same_triangle_set": bool(same_set)})
    ok = not differ and 2 * compared >= len(replay_rows) and compared > 0
    return ok, {"sets": len(replay_rows), "compared": compared, "declined": declined, "different": differ}


def _canon(t):
    t = np.sort(np.asarray(t, dtype=np.int64).reshape(-1, 3), axis=1)
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


def enabled():
    return os.environ.get("MVOSR_QHULL_SELFCHECK", "1").strip() not in ("0", "off", "false", "no")


def run(ctx, force=False, host_replay=None):
    """Run (or return the cached result of) the check on ``ctx``'s device.  ``host_replay``: optional callable
    ``points -> rows or None`` — the C replay on the host (``packing.qhull_rows_host``), checked on the same sets."""
    dev = int(ctx.device)
    if not force and dev in LAST:
        return LAST[dev]
    from . import packing
    res = dict(scipy_versions())
    res["replayed"] = REPLAYED_QHULL
    if not enabled():
        res.update(ok=True, skipped=True)
        LAST[dev] = res
        return res
    sets = check_point_sets()
    ref = []
    for p in sets:
        try:
            ref.append(packing.delaunay_simplices(p))
        except Exception as exc:          # (QhullError of the installed SciPy: nothing to compare on this set)
            ref.append(exc)
    from . import _lib
    try:
        replay = packing.delaunay_gpu(ctx, sets, rows="qhull")
    except _lib.MvosrAllocError as exc:
        # the device cannot even hold the check's 11 MB workspace right now (or mvosr_ctx_workspace_limit forbids it): nothing was
        # compared — this estimator takes the host path; the result is NOT kept, the next construction checks again
        res.update(ok=False, skipped=False, unchecked=str(exc))
        return res
    ok, detail = compare_rows(replay, ref)
    res.update(ok=ok, skipped=False, device=detail)
    if host_replay is not None:
        ok_h, detail_h = compare_rows([host_replay(p) for p in sets], ref)
        res.update(ok=ok and ok_h, host=detail_h)
    if not res["ok"]:
        warnings.warn("mvoscalerecovery_amd: the Qhull replay (%s) and the installed scipy.spatial.Delaunay (SciPy %s) give "
                      "different rows on the self-check sets (%s); estimators constructed with the default / "
                      "triangulation='gpu', check_triangle='reference' run triangulation='scipy' on this box"
                      % (REPLAYED_QHULL, res["scipy"], {k: res[k] for k in ("device", "host") if k in res}), RuntimeWarning,
                      stacklevel=3)
    LAST[dev] = res
    return res
