#!/usr/bin/env python3
"""End-to-end throughput of the drop-in batch call INCLUDING the host stages (packing, both SciPy
Delaunay calls per frame on a process pool, uploads) — the number a `main_offline`-shaped user sees.
    python profiles/e2e.py [frames] [features] [workers]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import packing, synth
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator

F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
W = int(sys.argv[3]) if len(sys.argv) > 3 else packing.available_cpus()
frames = [synth.synth_frame(i, N, base_seed=2024) for i in range(F)]
f3, f2 = [f[0] for f in frames], [f[1] for f in frames]
for workers in sorted({1, W // 2, W, 2 * W}):
    est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=workers, triangulation="scipy")
    est.scale_calculation_batch(f3[:8], f2[:8])
    n = F if workers > 1 else min(F, 64)
    t0 = time.perf_counter()
    s, e = est.scale_calculation_batch(f3[:n], f2[:n])
    dt = time.perf_counter() - t0
    print("e2e N=%d frames=%d delaunay_workers=%d (usable CPUs %d of %d): %.1f frames/s (%.2f ms/frame)"
          % (N, n, workers, packing.available_cpus(), os.cpu_count() or 0, n / dt, 1e3 * dt / n))
