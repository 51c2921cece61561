#!/usr/bin/env python3
"""Diagnostic: where a workgroup of the tiled dense kernel spends its life (in-kernel s_memtime deltas, wave 0's view).
Needs a library built with -DMVOSR_STAMPS:  profiles/ab_build.sh stamps -DMVOSR_STAMPS;  MVOSR_LIB_PATH=profiles/ab/libmvosr_stamps.so python profiles/stamps_tiled.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, packing, synth
from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
os.environ["MVOSR_DEBUG_SKIP"] = "16"      # no road-model launch: it would write its histograms over the stamps
packing.start_pool(None)
ctx = _lib.default_context(0)
eng = ScaleEngine(1.75, ctx=ctx)
pool = 16
frames = [synth.synth_frame(i, N, base_seed=2024) for i in range(pool)]
pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames])
packing.attach_tri1(pf, None, None)
packing.apply_tile_order(pf)
db = DeviceBatch(ctx, pf, with_tri2=False)
out = DeviceOutputs(ctx, db, counts=True, stage=True)
eng.outlier_vote_batch(db, out); ctx.sync()
c = out.get("vote_counters")
masks = [c[pf.frame_slice(f)] >= 0 for f in range(pool)]
packing.attach_tri2(pf, None, masks, None, feature_ids=True)
pf = packing.tile_frames(pf, F // pool)
db = DeviceBatch(ctx, pf)
out = DeviceOutputs(ctx, db, counts=True, hist=True)
for _ in range(3):
    eng.scale_batch(db, out)
ctx.sync()
raw = out.get("hist").reshape(pf.n_frames, -1).view(np.uint64)[:, :12].astype(np.float64)
names = ["prologue (init, tiles 0-1, index check)", "far rows", "vote rows (all steps)", "select rows (all steps)", "barrier A wait",
         "retire + tile store/load", "barrier B wait", "final checks + reductions", "ambiguity scan", "ambiguity resolution (cold sweep)", "count + ordered store"]
tot = raw[:, :11].sum(axis=1)
print("frames %d, workgroup life: mean %.0f ticks (100 MHz s_memtime), median %.0f, max %.0f" % (len(tot), tot.mean(), np.median(tot), tot.max()))
for i, nm in enumerate(names):
    print("%-44s mean %9.0f  share %5.1f%%   max %9.0f" % (nm, raw[:, i].mean(), 100 * raw[:, i].mean() / tot.mean(), raw[:, i].max()))
print("frames that ran the cold sweep: %.1f%%" % (100 * np.mean(raw[:, 9] > 200)))
