#!/usr/bin/env python3
"""Where the HOST spends a per-frame call: cProfile over 300 calls.   python profiles/latency_cprofile.py [rescale|scale|exact] [features]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth      # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "rescale"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
fr = [synth.synth_frame(300000 + i, n, base_seed=2024) for i in range(100)]
if which == "rescale":
    from mvoscalerecovery_amd.rescale import ScaleEstimator
    est = ScaleEstimator(1.75, window_size=5, device=0, delaunay_workers=0, triangulation="gpu", ransac_seed=2024)
elif which == "exact":                 # the default construction: the reference's result
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    est = ScaleEstimator(1.75, window_size=5, device=0, delaunay_workers=0)
else:
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    est = ScaleEstimator(1.75, window_size=5, device=0, delaunay_workers=0, triangulation="gpu")
for f3, f2 in fr[:5]:
    est.scale_calculation(f3.copy(), f2)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    for f3, f2 in fr:
        est.scale_calculation(f3.copy(), f2)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
