import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.rescale import ScaleEstimator
F=32768; P=4096
pool=[synth.synth_frame(200000+i, 2000, base_seed=2024) for i in range(P)]
f3,f2=[pool[i%P][0] for i in range(F)],[pool[i%P][1] for i in range(F)]
est=ScaleEstimator(1.75, window_size=5, triangulation="gpu", delaunay_workers=0, ransac_seed=2024)
os.environ["MVOSR_TRACE_CHUNKS"]="1"
for _ in range(2): est.scale_calculation_batch(f3,f2)
out=[]
for rep in range(14):
    t0=time.perf_counter(); est.scale_calculation_batch(f3,f2); dt=time.perf_counter()-t0
    out.append((F/dt/1e3, est.chunk_trace))
print(" ".join("%.0f"%o[0] for o in out))
slow=min(out,key=lambda o:o[0]); fast=max(out,key=lambda o:o[0])
for tag,o in (("slowest",slow),("fastest",fast)):
    print(tag, "%.0f k"%o[0])
    for what,k,n,t in o[1]:
        if what in ("launch","launched","collect"): print("   %-9s chunk %2d (%5d) %7.2f ms"%(what,k,n,1e3*t))
