#!/usr/bin/env python3
"""GPU probe for mvosr_delaunay_qhull_batch: rows against scipy.spatial.Delaunay (order and rotation), decline reasons, and a
first throughput figure.   python profiles/qhull_gpu_check.py [frames] [points]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.spatial import Delaunay                     # noqa: E402
from mvoscalerecovery_amd import _lib, packing, synth   # noqa: E402

WHY = ["ok", "few points", "zero width", "simplex search", "flat simplex", "narrow", "inside simplex", "band", "coplanar horizon",
       "too many visible", "cone too large", "open cone", "not convex", "gauss", "not sharp", "above none", "facets full", "arena full"]


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    npts = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    ctx = _lib.Context(0)
    for n, count in ((30, 8), (150, 16), (400, 16), (1000, 8), (npts if npts > 0 else 777, 16)):
        sets = [synth.synth_frame(s, n, base_seed=31415)[1] for s in range(count)]
        got = packing.delaunay_gpu(ctx, sets, rows="qhull", order_out=True)
        ok = bad = dec = 0
        for k, (p, t) in enumerate(zip(sets, got)):
            ref = Delaunay(p).simplices
            if t is None:
                dec += 1
                print("   n %d frame %d declined: %s" % (n, k, WHY[int(packing.delaunay_gpu.last_status[k]) >> 8]))
            elif t.shape == ref.shape and np.array_equal(t, ref):
                ok += 1
            else:
                bad += 1
                same = t.shape == ref.shape and set(map(tuple, np.sort(t, 1))) == set(map(tuple, np.sort(ref, 1)))
                j = 0
                while j < min(len(t), len(ref)) and np.array_equal(t[j], ref[j]):
                    j += 1
                print("   n %d frame %d MISMATCH rows %s vs %s, same set %s, first differing row %d" % (n, k, t.shape, ref.shape, same, j))
        print("n %5d: %d frames: identical to SciPy %d, different %d, declined %d" % (n, count, ok, bad, dec), flush=True)
    # throughput: resident sets, one launch per frame count
    for fr in [int(x) for x in os.environ.get("QH_FRAMES", str(frames)).split(",")]:
        throughput(ctx, fr, npts)


def throughput(ctx, frames, npts):
    lo, hi = (npts, npts) if npts > 0 else (300, 1500)
    rng = np.random.default_rng(3)
    pool = [synth.synth_frame(s, int(rng.integers(lo, hi + 1)), base_seed=999)[1] for s in range(min(frames, 512))]
    sets = [pool[s % len(pool)] for s in range(frames)]
    cnt = np.array([len(p) for p in sets], dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(cnt.astype(np.int64))])
    uv = np.concatenate(sets)
    d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
    d_off, d_cnt, d_toff = ctx.to_device(off[:-1].astype(np.int64)), ctx.to_device(cnt), ctx.to_device((2 * off[:-1]).astype(np.int64))
    d_tri = ctx.empty((2 * int(off[-1]), 3), np.int32)
    d_tc, d_st = ctx.zeros(frames, np.int32), ctx.zeros(frames, np.int32)
    best = 1e9
    for rep in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        _lib.check(ctx.lib.mvosr_delaunay_qhull_batch(ctx.handle, frames, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, int(cnt.max()),
                                                      d_toff.ptr, d_tri.ptr, d_tc.ptr, None, d_st.ptr, None))
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    st = d_st.download()
    why = {}
    for x in st[st != 0]:
        why[WHY[int(x) >> 8]] = why.get(WHY[int(x) >> 8], 0) + 1
    print("launch of %6d sets of %s points: %8.2f ms = %7.1f k sets/s (declined %d %s)" % (
        frames, npts if npts > 0 else "300-1500", best * 1e3, frames / best / 1e3, int((st != 0).sum()), why if why else ""), flush=True)
    for b in (d_u, d_v, d_off, d_cnt, d_toff, d_tri, d_tc, d_st):
        b.free()


if __name__ == "__main__":
    main()
