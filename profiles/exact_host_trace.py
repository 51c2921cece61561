#!/usr/bin/env python3
"""Host-side timeline of the reference-exact batch call (LABNOTES 10.7 / 10.14): when this process entered and left each chunk's launch
(_chunk_gpu: pack + upload + every kernel queued) and each chunk's collection (_chunk_gpu_finish), relative to the call's start.
    python profiles/exact_host_trace.py [frames] [features]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth, engine
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
P = min(F, 4096)
pool = [synth.synth_frame(i, N, base_seed=2024) for i in range(P)]
f3, f2 = [pool[i % P][0] for i in range(F)], [pool[i % P][1] for i in range(F)]
est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", check_triangle="reference", delaunay_workers=0)
est.GPU_SIDE_DOWNLOADS = os.environ.get("SIDE", "1") != "0"          # SIDE=0: the downloads queued on the compute stream behind the kernels, as before
log, T0 = [], [0.0]


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        t = time.perf_counter() - T0[0]
        try:
            return fn(*a, **k)
        finally:
            log.append((t, time.perf_counter() - T0[0], label, len(a[0]) if a and hasattr(a[0], "__len__") else 0))
    setattr(obj, name, w)


wrap(est, "_chunk_gpu", "launch")
wrap(est, "_chunk_gpu_finish", "collect")
wrap(est, "_chunk_gpu_complete_all", "re-runs")
wrap(engine, "pack_upload_native", "  pack+upload")
_tri = engine.DeviceBatch.triangulate


def tri(self, *a, **k):
    t = time.perf_counter() - T0[0]
    try:
        return _tri(self, *a, **k)
    finally:
        log.append((t, time.perf_counter() - T0[0], "  triangulate (launches)", self.n_frames))


engine.DeviceBatch.triangulate = tri
for _ in range(2):
    est.scale_calculation_batch(f3, f2)
for rep in range(2):
    del log[:]
    T0[0] = time.perf_counter()
    est.scale_calculation_batch(f3, f2)
    dt = time.perf_counter() - T0[0]
    print("call %d: %.0f frames/s, %.2f ms" % (rep, F / dt, 1e3 * dt))
for a, b, label, n in sorted(log):
    print("%8.2f .. %8.2f ms  %-26s %s" % (1e3 * a, 1e3 * b, label, n if n else ""))
