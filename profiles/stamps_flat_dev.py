#!/usr/bin/env python3
"""Diagnostic: phase durations of flat_selection_kernel<device> (a build with -DMVOSR_FS_STAMPS overwrites the frame's
results with them):   ONLY=mvosr_rescale bash profiles/ab_build.sh fsst -DMVOSR_FS_STAMPS
    MVOSR_LIB_PATH=profiles/ab/libmvosr_fsst.so python profiles/stamps_flat_dev.py [features]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.rescale import ScaleEstimator
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
F = 2048
frames = [synth.synth_frame(i, N, base_seed=4242) for i in range(F)]
est = ScaleEstimator(1.75, window_size=5, triangulation="gpu", delaunay_workers=0, ransac_seed=7)
try:
    est.scale_calculation_batch([f[0] for f in frames], [f[1] for f in frames])
except Exception as e:                      # (the "results" are stamps: the cross-frame tail may object)
    print("(tail:", type(e).__name__, ")")
L = est.last
cols = [L["model"][:, 0], L["model"][:, 1], L["model"][:, 2], L["model"][:, 3], L["raw_scale"], L["height_level"], L["best_ic"].astype(float), L["used"].astype(float)]
names = ["load planes (keep-compaction)", "normals, flags, bounds", "median (histogram select)", "kept rows -> point list", "hypotheses' planes",
         "multiplicities + distinct vertices", "inlier counts", "best hypothesis, outputs (one thread)"]
tot = sum(np.nanmean(c) for c in cols)
print("flat_selection_kernel<device>, %d features: %.0f ticks per frame" % (N, tot))
for nm, c in zip(names, cols):
    print("  %-40s %8.0f  %5.1f %%" % (nm, np.nanmean(c), 100 * np.nanmean(c) / tot))
