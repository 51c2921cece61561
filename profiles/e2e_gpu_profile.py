#!/usr/bin/env python3
"""Where the host time of the drop-in batch call goes with triangulation="gpu" (cProfile, cumulative):
    python profiles/e2e_gpu_profile.py [frames] [features] [scale|rescale]
(third argument "rescale": the estimator /root/reference/src/main.py:20 imports, device-resident; "exact": check_triangle="reference")"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
WHICH = sys.argv[3] if len(sys.argv) > 3 else "scale"
P = min(F, 4096)                      # distinct frames (larger than the host's last-level cache)
pool = [synth.synth_frame(i, N, base_seed=2024) for i in range(P)]
SNAP = float(os.environ.get("SNAP_FRACTION", "0"))        # that share of the frames with pixel coordinates rounded to 1/4 px: declined by the device triangulations
if SNAP > 0:
    import numpy as np
    pool = [(a, np.ascontiguousarray(np.round(b * 4) / 4)) if ((i * 2654435761) % (1 << 32)) / float(1 << 32) < SNAP else (a, b) for i, (a, b) in enumerate(pool)]
f3, f2 = [pool[i % P][0] for i in range(F)], [pool[i % P][1] for i in range(F)]
if os.environ.get("SNAP_ONE"):        # ONE frame of the call (that index) snapped to 1/4 px: what a single declined frame costs (LABNOTES 10.11)
    import numpy as np
    for i_ in os.environ["SNAP_ONE"].split(","):
        f2[int(i_)] = np.ascontiguousarray(np.round(f2[int(i_)] * 4) / 4)
if WHICH == "rescale":
    from mvoscalerecovery_amd.rescale import ScaleEstimator as RescaleEstimator
    est = RescaleEstimator(1.75, window_size=5, triangulation="gpu", delaunay_workers=0, ransac_seed=2024)
elif WHICH == "exact":                # the reference's own vote on Qhull's rows built on the device (no declared deviation)
    est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", check_triangle="reference",
                         delaunay_workers=None if SNAP > 0 else 0)
else:
    est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", delaunay_workers=None if SNAP > 0 else 0)
est.scale_calculation_batch(f3, f2)
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
s, e = est.scale_calculation_batch(f3, f2)
pr.disable()
dt = time.perf_counter() - t0
print(WHICH, "triangulation=gpu N=%d frames=%d: %.0f frames/s (%.3f ms/frame)%s" % (N, F, F / dt, 1e3 * dt / F, "; declined %d" % est.declined_total if SNAP > 0 else ""))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
