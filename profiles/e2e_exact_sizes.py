#!/usr/bin/env python3
"""The reference-exact device path end to end as a function of the call's length (a call pays its first chunk's pack + upload and its
last chunk's tail once): frames/s for calls of 16 384 ... 65 536 frames of 2000 features, and the chunk size.
   python profiles/e2e_exact_sizes.py [features] [frames,frames,...] [chunk,chunk,...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth                                     # noqa: E402
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator            # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16384,32768,65536").split(",")]
    chunks = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "16384").split(",")]
    pool = [synth.synth_frame(i, n, base_seed=2024) for i in range(2048)]
    for C in chunks:
        est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", check_triangle="reference", delaunay_workers=0)
        est.GPU_EXACT_CHUNK = C
        est.GPU_EXACT_CHUNK_POINTS = max(est.GPU_EXACT_CHUNK_POINTS, C * (n + 16))
        for F in sizes:
            f3s = [pool[i % len(pool)][0] for i in range(F)]
            f2s = [pool[i % len(pool)][1] for i in range(F)]
            est.scale_calculation_batch(f3s, f2s)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                est.scale_calculation_batch(f3s, f2s)
                ts.append(time.perf_counter() - t0)
            print("chunk cap %6d (per context: half): %7.1f ms per call of %6d frames of %d features = %6.1f k frames/s (declined in last chunk %d)" % (
                C, min(ts) * 1e3, F, n, F / min(ts) / 1e3, est.last_declined), flush=True)


if __name__ == "__main__":
    main()
