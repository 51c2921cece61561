import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np
from mvoscalerecovery_amd import _lib, packing, synth
from scipy.spatial import Delaunay
ctx = _lib.default_context(0)
rng = np.random.default_rng(17)
for _ in range(7): rng.uniform()
frames = [synth.synth_frame(i, int(m), base_seed=4242, upper_fraction=0.1)[1] for i, m in enumerate(np.random.default_rng(5).integers(100, 2300, 400))]
frames = [f[f[:, 1] > 185] for f in frames]
refs = [packing.canonical_rows(Delaunay(p).simplices) for p in frames]
out = []
for rep in range(6):
    g = packing.delaunay_gpu(ctx, frames)
    st = packing.delaunay_gpu.last_status
    for k, (t, r) in enumerate(zip(g, refs)):
        if t is None or not np.array_equal(t, r):
            out.append({"rep": rep, "frame": k, "n": len(frames[k]), "why": int(st[k]) >> 8, "none": t is None})
print(json.dumps({"bad": len(out), "first": out[:5]}))
