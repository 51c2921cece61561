"""Time of the vote kernel alone (mvosr_outlier_vote_batch) on tiled 2000-feature frames: python profiles/vote_only.py [frames]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, packing, synth                     # noqa: E402
from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine   # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = _lib.default_context(0)
eng = ScaleEngine(1.75, ctx=ctx)
pool = 64
frames = [synth.synth_frame(i, 2000, base_seed=2024) for i in range(pool)]
pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames])
packing.attach_tri1(pf)
pf = packing.tile_frames(pf, F // pool)
db = DeviceBatch(ctx, pf, with_tri2=False)
out = DeviceOutputs(ctx, db, counts=True, stage=True)
for _ in range(3):
    eng.outlier_vote_batch(db, out)
ctx.sync()
a, b = ctx.event(), ctx.event()
ctx.record(a)
for _ in range(10):
    eng.outlier_vote_batch(db, out)
ctx.record(b)
ctx.sync()
ms = ctx.elapsed_ms(a, b) / 10
c = out.get("vote_counters")
print("vote kernel: %.4f ms per %d frames (%.2f M frames/s); survivors of frame 0: %d" % (ms, F, F / ms / 1e3, int((c[pf.frame_slice(0)] >= 0).sum())))
