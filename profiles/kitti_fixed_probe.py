#!/usr/bin/env python3
"""The fixed-vote e2e leg on KITTI-sized (ragged 300-1500) frames, with knobs, to bisect a change of its rate:
    EARLY=0|1 (ScaleEstimator.GPU_REDO_EARLY)  [MVOSR_LIB_PATH=...]  python profiles/kitti_fixed_probe.py [frames] [calls]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator

F = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(4541)
sizes = [int(v) for v in rng.integers(300, 1501, 4096)]
pool = [synth.synth_frame(200000 + i, sizes[i], base_seed=2024) for i in range(4096)]
f3s, f2s = [pool[i % 4096][0] for i in range(F)], [pool[i % 4096][1] for i in range(F)]
ScaleEstimator.GPU_REDO_EARLY = os.environ.get("EARLY", "1") != "0"
est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", delaunay_workers=0, check_triangle="fixed")
for _ in range(2):
    est.scale_calculation_batch(f3s, f2s)
ts = []
for _ in range(calls):
    t0 = time.perf_counter(); est.scale_calculation_batch(f3s, f2s); ts.append(time.perf_counter() - t0)
print("EARLY=%s lib=%s: %s k frames/s (median %.0f k), declined %d" % (os.environ.get("EARLY", "1"), os.path.basename(os.environ.get("MVOSR_LIB_PATH", "product")),
      " ".join("%.0f" % (F / t / 1e3) for t in ts), F / sorted(ts)[len(ts) // 2] / 1e3, est.declined_total))
