"""GPU check + timing of mvosr_delaunay_batch against scipy.spatial.Delaunay (canonical rows).
    python profiles/dt_check.py [n_points] [n_sets]
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvoscalerecovery_amd import _lib, packing, synth      # noqa: E402
from scipy.spatial import Delaunay                          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = _lib.default_context(0)
out = {"max_points": packing.delaunay_gpu_max_points()}

# ---- correctness on a mix of sets
rng = np.random.default_rng(17)
sets = [synth.synth_frame(i, m, base_seed=606)[1] for i, m in enumerate((2000, 1500, 300, 64, 7, 3, 4000, 4400, 1000, 500, 120, 33))]
sets.append(rng.normal(0.0, 1.0, (900, 2)) * [1.0, 1e-3])
sets.append(np.concatenate([rng.uniform(0, 100, (500, 2)), rng.uniform(40, 41, (500, 2))]))
sets.append(rng.uniform(0, 1, (2500, 2)))
th = rng.uniform(0, 2 * np.pi, 300)
sets.append(np.stack([np.cos(th), np.sin(th)], axis=1) * rng.uniform(0.999, 1.001, (300, 1)) * 50 + 100)   # a noisy ring (big hull)
got = packing.delaunay_gpu(ctx, sets)
st = packing.delaunay_gpu.last_status
bad = []
for k, (pts, tri) in enumerate(zip(sets, got)):
    ref = packing.canonical_rows(Delaunay(pts).simplices)
    ok = tri is not None and tri.shape == ref.shape and np.array_equal(tri, ref)
    if not ok:
        bad.append({"set": k, "n": len(pts), "status": int(st[k]), "why": int(st[k]) >> 8, "rows": None if tri is None else int(tri.shape[0]), "ref_rows": int(ref.shape[0])})
out["mismatch"] = bad[:6]
out["mismatch_count"] = len(bad)
# keep mask
pts = sets[0]
keep = np.where(rng.uniform(size=len(pts)) < 0.9, 1, -1).astype(np.int32)
g2 = packing.delaunay_gpu(ctx, [pts], [keep])[0]
ref2 = packing.canonical_rows(Delaunay(pts[keep >= 0]).simplices)
out["keep_ok"] = bool(g2 is not None and np.array_equal(g2, ref2)) and int(packing.delaunay_gpu.last_used[0]) == int((keep >= 0).sum())
# degenerate inputs are declined
grid = np.stack(np.meshgrid(np.arange(20.0), np.arange(15.0)), axis=-1).reshape(-1, 2)
dup = sets[2].copy(); dup[10] = dup[200]
line = np.stack([np.arange(50.0), 2.0 * np.arange(50.0)], axis=1)
dec = packing.delaunay_gpu(ctx, [grid, dup, line, sets[2][:2]])
out["declined"] = [t is None for t in dec]
out["declined_why"] = [int(s) >> 8 for s in packing.delaunay_gpu.last_status]
again = packing.delaunay_gpu(ctx, sets[:3])
out["deterministic"] = all(np.array_equal(a, b) for a, b in zip(got[:3], again))

# ---- many random frames: all must equal SciPy
frames = [synth.synth_frame(i, int(m), base_seed=4242, upper_fraction=0.1)[1] for i, m in enumerate(rng.integers(100, 2300, 200))]
frames = [f[f[:, 1] > 185] for f in frames]
g = packing.delaunay_gpu(ctx, frames)
stl = packing.delaunay_gpu.last_status
nbad = 0
for p, t, s in zip(frames, g, stl):
    ref = packing.canonical_rows(Delaunay(p).simplices)
    if t is None or not np.array_equal(t, ref):
        nbad += 1
out["random_frames_bad"] = nbad
out["random_frames_declined"] = int(sum(t is None for t in g))

# ---- timing: F sets of n points resident in HBM
pool = [synth.synth_frame(i, n, base_seed=99)[1] for i in range(64)]
cnt = np.full(F, n, dtype=np.int32)
off = (np.arange(F, dtype=np.int64) * n)
uv = np.concatenate([pool[i % 64] for i in range(F)])
d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
d_off, d_cnt, d_toff = ctx.to_device(off), ctx.to_device(cnt), ctx.to_device(2 * off)
d_tri = ctx.empty((2 * F * n, 3), np.int32)
d_tcnt, d_st, d_used = ctx.zeros(F, np.int32), ctx.zeros(F, np.int32), ctx.zeros(F, np.int32)
def launch():
    _lib.check(ctx.lib.mvosr_delaunay_batch(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, n, d_toff.ptr,
                                            d_tri.ptr, d_tcnt.ptr, d_used.ptr, d_st.ptr), "dt")
launch(); ctx.sync()
e0, e1 = ctx.event(), ctx.event()
reps = 5
ctx.record(e0)
for _ in range(reps):
    launch()
ctx.record(e1)
ms = ctx.elapsed_ms(e0, e1) / reps
out["timing"] = {"n": n, "sets": F, "ms_per_launch": ms, "sets_per_s": F / ms * 1e3, "declined": int((d_st.download() != 0).sum())}
print(json.dumps(out))
