#!/usr/bin/env python3
"""The device-triangulation batch call against its chunk size (frames per chunk; the points cap scaled along):
    python profiles/e2e_chunk_sweep.py [frames] [features]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator

F = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
pool = [synth.synth_frame(i, N, base_seed=2024) for i in range(256)]
f3, f2 = [pool[i % 256][0] for i in range(F)], [pool[i % 256][1] for i in range(F)]
for chunk in (512, 1024, 1536, 2048, 3072, 4096):
    est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", delaunay_workers=0)
    est.GPU_CHUNK = chunk
    est.GPU_CHUNK_POINTS = max(est.GPU_CHUNK_POINTS, chunk * N + 1)
    est.scale_calculation_batch(f3, f2)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        est.scale_calculation_batch(f3, f2)
        best = min(best, time.perf_counter() - t0)
    print("chunk %5d frames: %.0f frames/s" % (chunk, F / best))
    est.close() if hasattr(est, "close") else None
