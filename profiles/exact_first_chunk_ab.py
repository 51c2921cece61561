#!/usr/bin/env python3
"""A/B of the exact path's first chunk (scale_calculator.GPU_EXACT_FIRST_CHUNK) on one box: calls of 16 384 / 32 768 / 65 536 frames."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth                                     # noqa: E402
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator            # noqa: E402

import numpy as np                                                          # noqa: E402
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000                         # 0: ragged 300-1500 (config C3's sizes)
rng = np.random.default_rng(4541)
sizes = [int(v) for v in rng.integers(300, 1501, 1024)] if N == 0 else [N]
pool = [synth.synth_frame(200000 + i, sizes[i % len(sizes)], base_seed=2024) for i in range(4096)]
est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=0)
for F in (16384, 65536):
    f3, f2 = [pool[i % 4096][0] for i in range(F)], [pool[i % 4096][1] for i in range(F)]
    for first in (0, 4096, 0, 4096):
        est.GPU_EXACT_FIRST_CHUNK = first
        est.scale_calculation_batch(f3, f2)
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            est.scale_calculation_batch(f3, f2)
            t.append(time.perf_counter() - t0)
        print("%d features, %6d frames, first chunk %5d: %.1f ms = %.1f k frames/s" % (N, F, first, 1e3 * sorted(t)[1], F / sorted(t)[1] / 1e3), flush=True)
