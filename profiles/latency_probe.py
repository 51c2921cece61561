#!/usr/bin/env python3
"""Per-frame calls of the device-resident estimators, for a kernel trace of ONE frame's chain:
   rocprofv3 --kernel-trace -d gpurun_out/lat -o lat -- python3 profiles/latency_probe.py [rescale|scale|exact] [frames] [features]
prints the median wall time per call; profiles/sum_kernel_trace.py on the trace gives each kernel's share of it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth      # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "rescale"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    fr = [synth.synth_frame(300000 + i, n, base_seed=2024) for i in range(frames)]
    if which == "rescale":
        from mvoscalerecovery_amd.rescale import ScaleEstimator
        est = ScaleEstimator(1.75, window_size=5, device=0, delaunay_workers=0, triangulation="gpu", ransac_seed=2024)
    elif which == "exact":            # the default construction: the reference's result (one SciPy call per frame since round 5)
        from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
        est = ScaleEstimator(1.75, window_size=5, device=0, delaunay_workers=0, triangulation="gpu", check_triangle="reference")
        est.GPU_EXACT_SINGLE_FAST = os.environ.get("SINGLE_FAST", "1") == "1"
        if os.environ.get("HOST_REPLAY", "1") != "1":     # the first triangulation by SciPy, as in round 5 (A/B of mvosr_qhull_rows_host)
            est._host_replay = False
    else:
        from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
        est = ScaleEstimator(1.75, window_size=5, device=0, delaunay_workers=0, triangulation="gpu")
    for f3, f2 in fr[:5]:
        est.scale_calculation(f3.copy(), f2)
    t = []
    for f3, f2 in fr:
        a = f3.copy()
        t0 = time.perf_counter()
        est.scale_calculation(a, f2)
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    if which == "exact":
        print("   frames redone through the host's path: %d, exact levels computed on demand: %d" % (getattr(est, "single_fast_redone", 0), getattr(est, "single_fast_levels", 0)))
    print("%s: %d per-frame calls of %d features: median %.3f ms, p10 %.3f, p90 %.3f" % (which, frames, n, np.median(t), np.percentile(t, 10), np.percentile(t, 90)))


if __name__ == "__main__":
    main()
