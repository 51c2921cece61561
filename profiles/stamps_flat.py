#!/usr/bin/env python3
"""Diagnostic: phase shares of flat_selection_kernel from in-kernel s_memtime stamps (a build with -DMVOSR_FS_STAMPS; the
stamps overwrite each frame's first five heights):
    ONLY=mvosr_rescale bash profiles/ab_build.sh fsst -DMVOSR_FS_STAMPS
    MVOSR_LIB_PATH=profiles/ab/libmvosr_fsst.so python profiles/stamps_flat.py"""
import ctypes as C
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, packing, synth, rescale
from mvoscalerecovery_amd.engine import DeviceBatch

F, POOL, N = 4096, 64, 2000
est = rescale.ScaleEstimator(1.75, window_size=5, ransac_seed=7, delaunay_workers=0)
ctx, lib = est.ctx, est.ctx.lib
frames = [synth.synth_frame(i, N, base_seed=4242) for i in range(POOL)]
est.feature_selection_batch([f[0] for f in frames], [f[1] for f in frames])
pf2 = packing.tile_frames(est.last["pf2"], F // POOL)
db2 = DeviceBatch(ctx, pf2, with_tri2=True)
nt = int(pf2.tri2_off[-1])
tri_h, tri_f = ctx.zeros(nt, np.float64), ctx.zeros(nt, np.uint8)
level, nkept, st2 = ctx.zeros(F, np.float64), ctx.zeros(F, np.int32), ctx.zeros(F, np.int32)
b2 = db2.struct()
max_tri = int(np.max(np.diff(pf2.tri2_off)))
for _ in range(3):
    _lib.check(lib.mvosr_flat_selection_batch(ctx.handle, C.byref(b2), -80.0, -85.0, 0.9, tri_h.ptr, tri_f.ptr, level.ptr, nkept.ptr,
                                              st2.ptr, max_tri), "flat")
ctx.sync()
h = tri_h.download()
st = np.stack([h[int(o):int(o) + 5] for o in pf2.tri2_off[:-1]])
names = ["load x,y,z + barrier", "phase 1 (normals, flags, bounds)", "median (histogram select)", "kept flags + outputs"]
tot = st[:, 4].mean()
print("workgroup life: mean %.0f ticks" % tot)
for i, nm in enumerate(names):
    d = (st[:, i + 1] - st[:, i]).mean()
    print("%-36s mean %9.0f  share %5.1f%%" % (nm, d, 100 * d / tot))
