#!/usr/bin/env python3
"""Host-side timeline of the device-resident rescale batch call (MVOSR_TRACE_CHUNKS): when this process began a chunk, had it
packed, had its device blocks, had its launches queued, and when it collected a chunk's results.
    python profiles/e2e_host_trace.py [frames] [features]"""
import os, sys, time
os.environ["MVOSR_TRACE_CHUNKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.rescale import ScaleEstimator
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
P = min(F, 4096)
pool = [synth.synth_frame(i, N, base_seed=2024) for i in range(P)]
f3, f2 = [pool[i % P][0] for i in range(F)], [pool[i % P][1] for i in range(F)]
est = ScaleEstimator(1.75, window_size=5, triangulation="gpu", delaunay_workers=0, ransac_seed=2024)
for _ in range(2): est.scale_calculation_batch(f3, f2)
t0 = time.perf_counter()
est.scale_calculation_batch(f3, f2)
dt = time.perf_counter() - t0
print("%.0f frames/s, call %.2f ms" % (F / dt, 1e3 * dt))
tr = est.chunk_trace
base = None
for what, k, n, t in tr:
    if what in ("packed", "blocks"):
        print("      %-9s %27.2f ms" % (what, 1e3 * t))
    else:
        print("%-9s chunk %2d (%5d frames) %8.2f ms" % (what, k, n, 1e3 * t))
