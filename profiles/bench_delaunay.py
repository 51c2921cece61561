"""mvosr_delaunay_batch alone on resident point sets (the workload profiles/collect_delaunay.sh runs under rocprofv3):
    python profiles/bench_delaunay.py [--points 2000] [--sets 4096] [--steps 5]
Prints one JSON line: sets/s from HIP events around the launches, declines, and the size of the problem."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, synth      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=2000)
ap.add_argument("--sets", type=int, default=4096)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--ragged", default="", help="lo:hi — sets of lo..hi points (uniform), the launch sized by the largest: the shape of a chunk of KITTI-sized frames")
ap.add_argument("--seeded", action="store_true", help="time the SECOND triangulation: 85 %% of the points kept, seeded with the first one's rows")
ap.add_argument("--keep", type=float, default=0.85, help="share of the points the second triangulation keeps (the vote keeps ~0.95)")
ap.add_argument("--no-carry", action="store_true", help="seeded without the untouched stars carried over (mvosr_delaunay_batch_seeded)")
ap.add_argument("--grid", type=float, default=0.0, help="round the coordinates to this grid (pixels), e.g. 0.0625: collinear triples in most sets (the hard-point pass takes their stars)")
args = ap.parse_args()
n, F = args.points, args.sets
ctx = _lib.default_context(0)
_frame = synth.synth_frame
if args.grid > 0:
    class synth:                                   # noqa: N801  (the pool's frames on the grid)
        @staticmethod
        def synth_frame(i, n_, base_seed=99):
            f3, f2 = _frame(i, n_, base_seed=base_seed)
            return f3, np.ascontiguousarray(np.round(f2 / args.grid) * args.grid)
pool = [synth.synth_frame(i, n, base_seed=99)[1] for i in range(64)]
if args.ragged:
    lo_, hi_ = (int(x) for x in args.ragged.split(":"))
    cnt = np.random.default_rng(3).integers(lo_, hi_ + 1, F).astype(np.int32)
    n = int(cnt.max())
    pool = [synth.synth_frame(i, n, base_seed=99)[1] for i in range(64)]
    off = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64)
    uv = np.concatenate([pool[i % 64][:cnt[i]] for i in range(F)])
else:
    cnt = np.full(F, n, dtype=np.int32)
    off = np.arange(F, dtype=np.int64) * n
    uv = np.concatenate([pool[i % 64] for i in range(F)])
d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
d_off, d_cnt, d_toff = ctx.to_device(off), ctx.to_device(cnt), ctx.to_device(2 * off)
d_tri = ctx.empty((2 * int(cnt.sum()), 3), np.int32)
d_tcnt, d_st = ctx.zeros(F, np.int32), ctx.zeros(F, np.int32)


d_info = ctx.zeros(int(cnt.sum()), np.uint32)


def launch():
    _lib.check(ctx.lib.mvosr_delaunay_batch_ex(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, n, d_toff.ptr,
                                               d_tri.ptr, d_tcnt.ptr, None, d_st.ptr, None, None, None, None, d_info.ptr), "mvosr_delaunay_batch_ex")


launch()
ctx.sync()
if args.seeded:
    keep = np.where(np.random.default_rng(5).uniform(size=int(cnt.sum())) < args.keep, 1, -1).astype(np.int32)
    d_keep = ctx.to_device(keep)
    d_tri2 = ctx.empty((2 * int(cnt.sum()), 3), np.int32)
    d_tcnt2 = ctx.zeros(F, np.int32)
    d_tcnt1, d_tcnt = d_tcnt, d_tcnt2

    def launch():      # noqa: F811
        _lib.check(ctx.lib.mvosr_delaunay_batch_ex(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, d_keep.ptr, n, d_toff.ptr,
                                                   d_tri2.ptr, d_tcnt2.ptr, None, d_st.ptr, d_toff.ptr, d_tri.ptr, d_tcnt1.ptr,
                                                   None if args.no_carry else d_info.ptr, None), "mvosr_delaunay_batch_ex (seeded)")

    launch()
    ctx.sync()
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0)
for _ in range(args.steps):
    launch()
ctx.record(e1)
ms = ctx.elapsed_ms(e0, e1) / args.steps
rows = d_tcnt.download()
print(json.dumps({"what": ("seeded second triangulation over %.0f %% of the points%s" % (100 * args.keep, "" if args.no_carry else ", untouched stars carried over")) if args.seeded else "first triangulation", "points_per_set": n, "sets": F, "steps": args.steps, "kernel_ms": ms, "sets_per_s": F / ms * 1e3,
                  "points_per_s": float(cnt.sum()) / ms * 1e3, "rows_per_set": float(rows.mean()), "declined": int((d_st.download() != 0).sum())}))
