#!/usr/bin/env python3
"""A/B of the fixed-mode batch path's last chunk in two pieces (an experimental scale_calculator.GPU_RAMP_DOWN, round 6: no gain —
profiles/r06_ramp_down_ab.txt — and not in the product; the script needs that knob re-added): calls of 32 768 / 65 536 frames."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth                                     # noqa: E402
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator            # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
pool = [synth.synth_frame(200000 + i, N, base_seed=2024) for i in range(4096)]
est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=0, triangulation="gpu")
for F in (32768, 65536):
    f3, f2 = [pool[i % 4096][0] for i in range(F)], [pool[i % 4096][1] for i in range(F)]
    for down in (False, True, False, True):
        est.GPU_RAMP_DOWN = down
        est.scale_calculation_batch(f3, f2)
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            est.scale_calculation_batch(f3, f2)
            t.append(time.perf_counter() - t0)
        print("%d features, %6d frames, last chunk in two pieces %5s: %.1f ms = %.1f k frames/s" % (N, F, down, 1e3 * sorted(t)[2], F / sorted(t)[2] / 1e3), flush=True)
