#!/usr/bin/env python3
"""Diagnostic: per-phase timeline of the fused kernel from in-kernel s_memtime stamps.
Needs a library built with -DMVOSR_STAMPS (never the shipped build):
    bash profiles/ab_build.sh stamps "-DMVOSR_STAMPS -DMVOSR_ABLATE"
    MVOSR_DEBUG_SKIP=16 MVOSR_LIB_PATH=profiles/ab/libmvosr_stamps.so python profiles/stamps.py [frames] [features]
(MVOSR_DEBUG_SKIP=16: no road-model launch — its histogram output shares the buffer the stamps are written to.)
Reports SHARES of a workgroup's life per phase (not absolute run time: stamps perturb it)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, packing, synth
from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = _lib.default_context(0)
eng = ScaleEngine(1.75, ctx=ctx)
pool = 64
frames = [synth.synth_frame(i, N, base_seed=2024) for i in range(pool)]
pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames])
packing.attach_tri1(pf)
db = DeviceBatch(ctx, pf, with_tri2=False)
out = DeviceOutputs(ctx, db, counts=True, stage=True)
eng.outlier_vote_batch(db, out); ctx.sync()
c = out.get("vote_counters")
masks = [c[pf.frame_slice(f)] >= 0 for f in range(pool)]
packing.attach_tri2(pf, None, masks)
pf = packing.tile_frames(pf, F // pool)
db = DeviceBatch(ctx, pf)
out = DeviceOutputs(ctx, db, counts=True, hist=True)
for _ in range(3):
    eng.scale_batch(db, out)
ctx.sync()
raw = out.get("hist").reshape(pf.n_frames, -1).view(np.uint64)
h = raw[:, :10].astype(np.float64)
road = raw[:, 16:23].astype(np.float64)
cols = [0, 1, 2, 3, 4, 5, 9]            # stamps the scale kernel sets (6-8 belonged to the old fused road phase)
names = ["load y,z,v + barrier", "vote sweep (tri1) + barrier", "compaction (2 barriers)", "select sweep 1 + reduce",
         "select sweep 2 + reduce", "pack selected + store list"]
h = h[:, cols]
d = np.diff(h, axis=1)
tot = h[:, -1] - h[:, 0]
print("workgroup life: mean %.0f ticks (s_memtime), median %.0f" % (tot.mean(), np.median(tot)))
for i, n in enumerate(names):
    print("%-36s mean %9.0f  share %5.1f%%" % (n, d[:, i].mean(), 100 * d[:, i].mean() / tot.mean()))

rn = ["start -> cnt/off/level loaded", "values loaded + histogram atomics", "hist read, ballots, max/min", "near flags, kept pass, mean reduce", "std pass + reduce", "mode logic + return"]
rd = np.diff(road, axis=1)
rt = road[:, -1] - road[:, 0]
print("road-model wave life: mean %.0f ticks, median %.0f" % (rt.mean(), np.median(rt)))
for i, n in enumerate(rn):
    print("%-36s mean %9.0f  share %5.1f%%" % (n, rd[:, i].mean(), 100 * rd[:, i].mean() / rt.mean()))
