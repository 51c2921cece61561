#!/usr/bin/env python3
"""Soak of the per-frame call of the reference-exact estimator: the one-SciPy-call path (stand-in second triangulation, product
kernels only, exact level on read) against the two-SciPy-call path it replaced, frame by frame on two estimators fed the same
sequence: scales, stds, the window, flat_feature; height_level read every few frames.
   python profiles/soak_single_exact.py [frames] [lo] [hi]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth                                     # noqa: E402
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator            # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    mode = os.environ.get("SOAK_MODE", "reference")            # "fixed": the explicit speed mode's per-frame call (product kernels only)
    a = ScaleEstimator(1.75, window_size=5, delaunay_workers=0, triangulation="gpu", check_triangle=mode)
    # (reference mode, round 6: the comparator is the host-SciPy estimator itself — two scipy.spatial.Delaunay calls per frame, no
    # replay of Qhull's run anywhere — so that the soak also holds mvosr_qhull_rows_host to SciPy on every frame)
    b = ScaleEstimator(1.75, window_size=5, delaunay_workers=0, triangulation="scipy" if mode == "reference" else "gpu", check_triangle=mode)
    b.GPU_EXACT_SINGLE_FAST = False
    b.GPU_SINGLE_HOT = False
    rng = np.random.default_rng(77)
    bad = fast = 0
    ta = tb = 0.0
    for i in range(F):
        f3, f2 = synth.synth_frame(i, int(rng.integers(lo, hi + 1)), base_seed=31337, upper_fraction=0.1)
        if i % 97 == 50:                                  # a three-feature frame now and then
            f2 = f2.copy()
            low = np.nonzero(f2[:, 1] > a.vanish)[0]
            f2[low[3:], 1] = a.vanish - 5.0
        x, y = f3.copy(), f3.copy()
        t0 = time.perf_counter(); ra = a.scale_calculation(x, f2); ta += time.perf_counter() - t0
        t0 = time.perf_counter(); rb = b.scale_calculation(y, f2); tb += time.perf_counter() - t0
        fast += a.__dict__.get("_level_thunk") is not None
        ok = ra == rb and np.array_equal(x, y) and list(a.scale_queue) == list(b.scale_queue)
        ok = ok and ((a.flat_feature is None and b.flat_feature is None) or
                     (a.flat_feature is not None and b.flat_feature is not None and np.array_equal(a.flat_feature, b.flat_feature)))
        if i % 5 == 0:
            ok = ok and a.height_level == b.height_level
        if not ok:
            bad += 1
            print("frame %d differs: %r %r" % (i, ra, rb), flush=True)
    print("%d per-frame calls of %d-%d features: %d differ; %d took the one-SciPy-call path, %d were handed back to the host's path, "
          "%d exact levels computed on demand; %.2f ms per call against %.2f" % (
              F, lo, hi, bad, fast, getattr(a, "single_fast_redone", 0), getattr(a, "single_fast_levels", 0), 1e3 * ta / F, 1e3 * tb / F))


def fuzz(count=480):
    """The frame-level fuzz set (duplicates, tied depths, walls, tiny frames, the level at zero ...): what each call returns or
    raises, the window and height_level after it, on the two paths."""
    mode = os.environ.get("SOAK_MODE", "reference")
    a = ScaleEstimator(1.75, window_size=5, delaunay_workers=0, triangulation="gpu", check_triangle=mode)
    # (reference mode, round 6: the comparator is the host-SciPy estimator itself — two scipy.spatial.Delaunay calls per frame, no
    # replay of Qhull's run anywhere — so that the soak also holds mvosr_qhull_rows_host to SciPy on every frame)
    b = ScaleEstimator(1.75, window_size=5, delaunay_workers=0, triangulation="scipy" if mode == "reference" else "gpu", check_triangle=mode)
    b.GPU_EXACT_SINGLE_FAST = False
    b.GPU_SINGLE_HOT = False
    bad = raised = fast = 0
    for i in range(count):
        f3, f2 = synth.fuzz_frame(i)
        outs = []
        for est in (a, b):
            x = f3.copy()
            try:
                outs.append(("ok", est.scale_calculation(x, f2)))
            except Exception as exc:                     # noqa: BLE001
                outs.append(("raised", type(exc).__name__))
        fast += a.__dict__.get("_level_thunk") is not None
        raised += outs[0][0] == "raised"
        def eq(p, q):                                      # (NaN scales and levels are legitimate results here)
            if p is None or q is None or isinstance(p, str):
                return p == q
            return np.array_equal(np.asarray(p, dtype=np.float64), np.asarray(q, dtype=np.float64), equal_nan=True)
        same = outs[0][0] == outs[1][0] and eq(outs[0][1], outs[1][1]) and eq(list(a.scale_queue), list(b.scale_queue)) and \
            eq(getattr(a, "height_level", None), getattr(b, "height_level", None))
        if not same:
            bad += 1
            if bad <= 12:
                print("fuzz frame %d (kind %d) differs: %r / %r; window equal %s; levels %r / %r" % (
                    i, i % 10, outs[0], outs[1], list(a.scale_queue) == list(b.scale_queue), getattr(a, "height_level", None), getattr(b, "height_level", None)), flush=True)
    print("fuzz: %d frames, %d differ; %d raised (same exception type on both paths), %d took the one-SciPy-call path, %d handed back" % (
        count, bad, raised, fast, getattr(a, "single_fast_redone", 0)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fuzz":
        fuzz()
    else:
        main()
