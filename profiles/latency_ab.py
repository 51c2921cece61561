"""Per-frame latency of the two estimators with the device triangulation, 200 frames after 10 warm-up calls (for same-box
comparisons of libraries: MVOSR_LIB_PATH=profiles/ab/libmvosr_<tag>.so python profiles/latency_ab.py [features])."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
from mvoscalerecovery_amd.rescale import ScaleEstimator as RescaleEstimator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
fr = [synth.synth_frame(300000 + i, n, base_seed=2024) for i in range(200)]
for name, est in (("scale", ScaleEstimator(1.75, window_size=5, delaunay_workers=0, triangulation="gpu")),
                  ("rescale", RescaleEstimator(1.75, window_size=5, delaunay_workers=0, triangulation="gpu", ransac_seed=2024))):
    for f3, f2 in fr[:10]:
        est.scale_calculation(f3.copy(), f2)
    t = []
    for f3, f2 in fr:
        a = f3.copy()
        t0 = time.perf_counter()
        est.scale_calculation(a, f2)
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    print("%-8s %d features: median %.3f ms  p10 %.3f  p90 %.3f" % (name, n, np.median(t), np.percentile(t, 10), np.percentile(t, 90)))
