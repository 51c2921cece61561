#!/usr/bin/env python3
"""The exact path on KITTI-sized (ragged 300-1500) frames: how many frames the exact pass redoes, by which route, and where the host's time goes."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth                                     # noqa: E402
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator            # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(4541)
sizes = [int(v) for v in rng.integers(300, 1501, 1024)]
pool = [synth.synth_frame(200000 + i, sizes[i % 1024], base_seed=2024) for i in range(4096)]
f3, f2 = [pool[i % 4096][0] for i in range(F)], [pool[i % 4096][1] for i in range(F)]
for host_redo in (True, False, True, False):
    est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=None)
    est.GPU_EXACT_HOST_REDO = host_redo
    est.scale_calculation_batch(f3, f2)
    for k in ("exact_redone_on_host", "exact_redone_on_device"):
        setattr(est, k, 0)
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        est.scale_calculation_batch(f3, f2)
        t.append(time.perf_counter() - t0)
    print("host_redo %s: %.1f ms = %.1f k frames/s; per call: redone on the host %d, on the device %d, declined %d" % (
        host_redo, 1e3 * sorted(t)[1], F / sorted(t)[1] / 1e3, est.exact_redone_on_host // 3, est.exact_redone_on_device // 3, est.declined_total), flush=True)
est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=None)
est.scale_calculation_batch(f3, f2)
pr = cProfile.Profile(); pr.enable(); est.scale_calculation_batch(f3, f2); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
