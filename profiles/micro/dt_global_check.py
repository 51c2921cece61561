"""The global-memory variant of delaunay_kernel (frames beyond the LDS capacity) against SciPy, and its rate."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from mvoscalerecovery_amd import _lib, packing, synth
from scipy.spatial import Delaunay
ctx = _lib.default_context(0)
out = {"lds_points": int(ctx.lib.mvosr_delaunay_lds_points()), "max_points": packing.delaunay_gpu_max_points()}
sets = [synth.synth_frame(i, n, base_seed=77)[1] for i, n in enumerate((5000, 8000, 20000, 20000, 31000))]
t0 = time.perf_counter(); got = packing.delaunay_gpu(ctx, sets); dt = time.perf_counter() - t0
st = packing.delaunay_gpu.last_status
res = []
for k, (p, t) in enumerate(zip(sets, got)):
    ref = packing.canonical_rows(Delaunay(p).simplices)
    res.append({"n": len(p), "ok": bool(t is not None and np.array_equal(t, ref)), "why": int(st[k]) >> 8})
out["sets"] = res
keep = np.where(np.random.default_rng(1).uniform(size=20000) < 0.85, 1, -1).astype(np.int32)
g2 = packing.delaunay_gpu(ctx, [sets[2]], [keep])[0]
out["keep_ok"] = bool(g2 is not None and np.array_equal(g2, packing.canonical_rows(Delaunay(sets[2][keep >= 0]).simplices)))
# rate: 256 resident sets of 20000 points
n, F = 20000, 256
pool = [synth.synth_frame(i, n, base_seed=5)[1] for i in range(8)]
cnt = np.full(F, n, dtype=np.int32); off = np.arange(F, dtype=np.int64) * n
uv = np.concatenate([pool[i % 8] for i in range(F)])
d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
d_off, d_cnt, d_toff = ctx.to_device(off), ctx.to_device(cnt), ctx.to_device(2 * off)
d_tri = ctx.empty((2 * F * n, 3), np.int32); d_tcnt, d_st = ctx.zeros(F, np.int32), ctx.zeros(F, np.int32)
launch = lambda: _lib.check(ctx.lib.mvosr_delaunay_batch(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, n, d_toff.ptr, d_tri.ptr, d_tcnt.ptr, None, d_st.ptr), "dt")
launch(); ctx.sync()
e0, e1 = ctx.event(), ctx.event(); ctx.record(e0); launch(); launch(); ctx.record(e1)
ms = ctx.elapsed_ms(e0, e1) / 2
out["rate"] = {"points": n, "sets": F, "ms": ms, "sets_per_s": F / ms * 1e3, "points_per_s": F * n / ms * 1e3, "declined": int((d_st.download() != 0).sum())}
print(json.dumps(out))
