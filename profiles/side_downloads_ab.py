#!/usr/bin/env python3
"""Same-process A/B of GPU_SIDE_DOWNLOADS (results copied to page-locked memory by a kernel) against downloads queued by hipMemcpyAsync on the
compute stream behind each chunk's kernels: one estimator per path, the knob flipped between calls.  LABNOTES 10.14.
    python profiles/side_downloads_ab.py [exact|fixed|rescale] [frames] [features | lo:hi] [pairs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "exact"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
feat = sys.argv[3] if len(sys.argv) > 3 else "2000"
pairs = int(sys.argv[4]) if len(sys.argv) > 4 else 6
P = min(F, 4096)
if ":" in feat:
    lo, hi = (int(x) for x in feat.split(":"))
    sizes = [int(v) for v in np.random.default_rng(4541).integers(lo, hi + 1, P)]
else:
    sizes = [int(feat)] * P
pool = [synth.synth_frame(200000 + i, sizes[i], base_seed=2024) for i in range(P)]
f3, f2 = [pool[i % P][0] for i in range(F)], [pool[i % P][1] for i in range(F)]
if which == "rescale":
    from mvoscalerecovery_amd.rescale import ScaleEstimator
    est = ScaleEstimator(1.75, window_size=5, triangulation="gpu", delaunay_workers=0, ransac_seed=2024)
else:
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", delaunay_workers=0,
                         check_triangle="reference" if which == "exact" else "fixed")
for _ in range(2):
    est.scale_calculation_batch(f3, f2)
t = {True: [], False: []}
for rep in range(2 * pairs):
    est.GPU_SIDE_DOWNLOADS = rep % 2 == 0
    t0 = time.perf_counter()
    est.scale_calculation_batch(f3, f2)
    t[rep % 2 == 0].append(time.perf_counter() - t0)
fmt = lambda ts: " ".join("%.0f" % (F / x / 1e3) for x in ts)
print("%-7s %6d frames of %s features: side downloads %s k frames/s (median %.1f k); queued behind the kernels %s (median %.1f k)" % (
    which, F, feat, fmt(t[True]), F / sorted(t[True])[pairs // 2] / 1e3, fmt(t[False]), F / sorted(t[False])[pairs // 2] / 1e3), flush=True)
