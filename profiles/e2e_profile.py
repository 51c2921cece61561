"""Where the host time of the batched drop-in call goes (cProfile): python profiles/e2e_profile.py [frames] [workers]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth                                   # noqa: E402
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator         # noqa: E402

if __name__ == "__main__":
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    W = int(sys.argv[2]) if len(sys.argv) > 2 else min(64, os.cpu_count() or 1)
    frames = [synth.synth_frame(i, 2000, base_seed=2024) for i in range(F)]
    f3, f2 = [f[0] for f in frames], [f[1] for f in frames]
    est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=W, triangulation="scipy")
    est.scale_calculation_batch(f3[:256], f2[:256])
    t0 = time.perf_counter()
    est.scale_calculation_batch(f3, f2)
    print("workers %d: %.0f frames/s" % (W, F / (time.perf_counter() - t0)))
    pr = cProfile.Profile()
    pr.enable()
    est.scale_calculation_batch(f3, f2)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
