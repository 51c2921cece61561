"""Where the per-frame call of the device-resident rescale estimator spends its time: host (cProfile, by own time) and GPU
(sum of kernel durations from HIP events around the call would need the library's profile hooks: here the wall time with and
without a final sync is compared).   python profiles/latency_rescale_profile.py [features]"""
import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.rescale import ScaleEstimator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
frames = [synth.synth_frame(i, n, base_seed=5) for i in range(264)]
est = ScaleEstimator(1.75, window_size=5, triangulation="gpu", delaunay_workers=0, ransac_seed=3)
for f3, f2 in frames[:8]:
    est.scale_calculation(f3, f2)
t0 = time.perf_counter()
for f3, f2 in frames[8:136]:
    est.scale_calculation(f3, f2)
print("per frame %.3f ms (128 frames)" % (1e3 * (time.perf_counter() - t0) / 128))
pr = cProfile.Profile(); pr.enable()
for f3, f2 in frames[136:]:
    est.scale_calculation(f3, f2)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
