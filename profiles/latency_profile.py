import cProfile, pstats, sys, time, os
sys.path.insert(0, os.getcwd())
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
frames=[synth.synth_frame(i,2000,base_seed=5) for i in range(64)]
est=ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu")
for f3,f2 in frames[:8]: est.scale_calculation(f3,f2)
t0=time.perf_counter()
pr=cProfile.Profile(); pr.enable()
for f3,f2 in frames[8:]: est.scale_calculation(f3,f2)
pr.disable()
print("per frame %.3f ms"%(1e3*(time.perf_counter()-t0)/56))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
