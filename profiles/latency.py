"""Per-frame latency of the drop-in ScaleEstimator.scale_calculation (the main.py loop shape):
where the host time goes besides the two SciPy Delaunay calls.  python profiles/latency.py [n_feat] [frames]"""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mvoscalerecovery_amd import synth                                   # noqa: E402
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 60
frames = [synth.synth_frame(i, n, base_seed=77) for i in range(F)]
est = ScaleEstimator(1.75, window_size=5, triangulation="scipy")
for f3, f2 in frames[:5]:
    est.scale_calculation(f3.copy(), f2)
t = []
for f3, f2 in frames:
    a = f3.copy()
    t0 = time.perf_counter()
    est.scale_calculation(a, f2)
    t.append(time.perf_counter() - t0)
t = np.array(t) * 1e3
print("per-frame ms: median %.3f  p10 %.3f  p90 %.3f" % (np.median(t), np.percentile(t, 10), np.percentile(t, 90)))
pr = cProfile.Profile()
pr.enable()
for f3, f2 in frames:
    est.scale_calculation(f3.copy(), f2)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
