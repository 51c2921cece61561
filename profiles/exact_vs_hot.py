#!/usr/bin/env python3
"""HOT (product) against EXACT (height_level summed in NumPy's order for every frame) scale kernel on a resident batch:
what it costs the drop-in batch paths to ask for the exact level of EVERY frame instead of keeping track of the frames
whose level a later step reads.   python profiles/exact_vs_hot.py [features] [frames]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import packing, synth
from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=8, triangulation="scipy")
pool = [synth.synth_frame(i, N, base_seed=7) for i in range(64)]
eng = est.engine
pf = packing.pack_features([p[0] for p in pool], [p[1] for p in pool])
packing.attach_tri1(pf, None, 8)
db = DeviceBatch(eng.ctx, pf, with_tri2=False)
o = DeviceOutputs(eng.ctx, db, counts=True, stage=True)
eng.outlier_vote_batch(db, o)
eng.ctx.sync()
c = o.get("vote_counters")
masks = [c[pf.frame_slice(f)] >= 0 for f in range(64)]
packing.attach_tri2(pf, None, masks, 8)
big = packing.tile_frames(pf, F // 64)
dbb = DeviceBatch(eng.ctx, big)
out = DeviceOutputs(eng.ctx, dbb, counts=True)
for exact in (False, True, False, True):
    eng.scale_batch(dbb, out, exact=exact)
    eng.ctx.sync()
    e0, e1 = eng.ctx.event(), eng.ctx.event()
    eng.ctx.record(e0)
    for _ in range(5):
        eng.scale_batch(dbb, out, exact=exact)
    eng.ctx.record(e1)
    ms = eng.ctx.elapsed_ms(e0, e1) / 5
    print("%s: %.3f ms per %d frames of %d features = %.2f M frames/s" % ("EXACT" if exact else "HOT  ", ms, big.n_frames, N, big.n_frames / ms / 1e3))
