#!/usr/bin/env python3
"""End-to-end calls of the reference-exact device path (triangulation="gpu", check_triangle="reference") with and without its two round-5 levers (two contexts taking the chunks in turn; the second triangulation as a stand-in).
   python profiles/e2e_exact_probe.py [frames] [features]     (MVOSR_QH_WAVES=4|5|6|8 picks the kernel's register budget)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth                                     # noqa: E402
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator            # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    pool = [synth.synth_frame(i, n, base_seed=2024) for i in range(min(F, 2048))]
    f3s = [pool[i % len(pool)][0] for i in range(F)]
    f2s = [pool[i % len(pool)][1] for i in range(F)]
    for two, standin in ((True, True), (False, True), (False, False)):
        est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", check_triangle="reference", delaunay_workers=0)
        est.GPU_EXACT_TWO_CONTEXTS = two
        est.GPU_EXACT_STANDIN = standin
        est.scale_calculation_batch(f3s, f2s)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            est.scale_calculation_batch(f3s, f2s)
            ts.append(time.perf_counter() - t0)
        print("two contexts %-5s stand-in second triangulation %-5s: %.1f ms per call of %d frames = %.1f k frames/s (declined in last chunk %d)" % (
            two, standin, min(ts) * 1e3, F, F / min(ts) / 1e3, est.last_declined), flush=True)


if __name__ == "__main__":
    main()
