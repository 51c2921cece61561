#!/usr/bin/env python3
"""Would two compute streams help?  Two estimators (two contexts: a stream each) in two threads, each with half of the frames,
against one estimator with all of them.  The second triangulation is bound by its longest star, the first by instruction
issue: chunks on different streams could fill each other's waits.     python profiles/two_streams_probe.py [rescale|scale]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth, _lib
which = sys.argv[1] if len(sys.argv) > 1 else "rescale"
if which == "rescale":
    from mvoscalerecovery_amd.rescale import ScaleEstimator
    make = lambda ctx: ScaleEstimator(1.75, window_size=5, triangulation="gpu", delaunay_workers=0, ransac_seed=1)
else:
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    make = lambda ctx: ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", delaunay_workers=0)
F, N = 32768, 2000
pool = [synth.synth_frame(i, N, base_seed=2024) for i in range(4096)]
f3 = [pool[i % 4096][0] for i in range(F)]; f2 = [pool[i % 4096][1] for i in range(F)]
one = make(None)
one.scale_calculation_batch(f3, f2)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); one.scale_calculation_batch(f3, f2); ts.append(time.perf_counter() - t0)
print("one stream: %s k frames/s" % " ".join("%.0f" % (F / t / 1e3) for t in ts), flush=True)
ests = []
orig = _lib.default_context
for k in (0, 1):
    c = _lib.Context(0)                        # (a context of its own: its own compute and upload streams, its own caches)
    _lib.default_context = lambda device=0, c=c: c
    ests.append(make(None))
_lib.default_context = orig
H = F // 2
def run(k):
    ests[k].scale_calculation_batch(f3[k * H:(k + 1) * H], f2[k * H:(k + 1) * H])
for k in (0, 1): run(k)
ts = []
for _ in range(3):
    th = [threading.Thread(target=run, args=(k,)) for k in (0, 1)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; ts.append(time.perf_counter() - t0)
print("two streams (two threads, half the frames each): %s k frames/s" % " ".join("%.0f" % (F / t / 1e3) for t in ts), flush=True)
