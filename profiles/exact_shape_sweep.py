#!/usr/bin/env python3
"""The reference-exact batch call's shape knobs once more, now that its time no longer depends on the process (LABNOTES 10.14): first chunk
(GPU_EXACT_FIRST_CHUNK), chunk per context (GPU_EXACT_CHUNK), chunks in flight (GPU_PIPELINE) — one process, alternating, median of 3.
    python profiles/exact_shape_sweep.py [features (0: ragged 300-1500)]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(4541)
sizes = [int(v) for v in rng.integers(300, 1501, 1024)] if N == 0 else [N]
pool = [synth.synth_frame(200000 + i, sizes[i % len(sizes)], base_seed=2024) for i in range(4096)]
est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=0)
base = dict(GPU_EXACT_FIRST_CHUNK=est.GPU_EXACT_FIRST_CHUNK, GPU_EXACT_CHUNK=est.GPU_EXACT_CHUNK, GPU_PIPELINE=est.GPU_PIPELINE)
variants = [("as shipped", {}), ("first chunk 0", dict(GPU_EXACT_FIRST_CHUNK=0)), ("first chunk 2048", dict(GPU_EXACT_FIRST_CHUNK=2048)),
            ("chunk 8192 (4096 per context)", dict(GPU_EXACT_CHUNK=8192)), ("chunk 24576 (12288 per context)", dict(GPU_EXACT_CHUNK=24576)),
            ("three chunks in flight", dict(GPU_PIPELINE=3)), ("one chunk in flight", dict(GPU_PIPELINE=1))]
for F in (16384, 65536):
    f3, f2 = [pool[i % 4096][0] for i in range(F)], [pool[i % 4096][1] for i in range(F)]
    for rep in range(2):
        for name, kv in variants:
            for k, v in base.items():
                setattr(est, k, v)
            for k, v in kv.items():
                setattr(est, k, v)
            est.scale_calculation_batch(f3, f2)
            t = []
            for _ in range(3):
                t0 = time.perf_counter(); est.scale_calculation_batch(f3, f2); t.append(time.perf_counter() - t0)
            print("%4d features %6d frames  %-34s %.1f ms = %.1f k frames/s" % (N, F, name, 1e3 * sorted(t)[1], F / sorted(t)[1] / 1e3), flush=True)
