#!/usr/bin/env python3
"""How often delaunay_kernel declines a frame of ordinary data (guard bands on fp64 predicates; a declined frame costs two host
Qhull calls): distinct synthetic frames (synth.synth_frame: projected road + structure points, sub-pixel coordinates) and
uniform random sets, first triangulation; the reasons (status bits 8..: 1 dup, 2 tie, 4 collinear, 8 degree, 16 rows, 32 Euler,
64 hard, 128 size).     python profiles/dt_decline_rate.py [sets] [points]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = _lib.default_context(0)
for kind in ("synth_frame", "uniform", "pixel-quantised (1/16 px)", "synth_frame on a 1/16 px grid"):
    rng = np.random.default_rng(11)
    tot_decl, why = 0, {}
    host_ok = host_same = 0
    B = 4096
    for b0 in range(0, F, B):
        if kind.startswith("synth_frame"):
            uv = np.concatenate([synth.synth_frame(1000000 + b0 + i, n, base_seed=4242)[1][:n] for i in range(B)])
            cnts = np.full(B, n, np.int32)
            if len(uv) != B * n:       # (frames may come back shorter)
                fr = [synth.synth_frame(1000000 + b0 + i, n, base_seed=4242)[1] for i in range(B)]
                cnts = np.array([len(a) for a in fr], np.int32); uv = np.concatenate(fr)
            if kind.endswith("grid"):
                uv = np.round(uv * 16) / 16
        elif kind == "uniform":
            uv = rng.uniform(0, 1241, (B * n, 2)) * np.array([1.0, 376.0 / 1241.0]); cnts = np.full(B, n, np.int32)
        else:
            uv = np.round(rng.uniform(0, 1241, (B * n, 2)) * np.array([1.0, 376.0 / 1241.0]) * 16) / 16; cnts = np.full(B, n, np.int32)
        off = np.concatenate([[0], np.cumsum(cnts)[:-1]]).astype(np.int64)
        mx = int(cnts.max())
        d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
        d_off, d_cnt, d_toff = ctx.to_device(off), ctx.to_device(cnts), ctx.to_device(2 * off)
        d_tri = ctx.empty((2 * len(uv), 3), np.int32)
        d_tcnt, d_st = ctx.zeros(B, np.int32), ctx.zeros(B, np.int32)
        _lib.check(ctx.lib.mvosr_delaunay_batch(ctx.handle, B, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, mx, d_toff.ptr, d_tri.ptr,
                                                d_tcnt.ptr, None, d_st.ptr), "dt")
        st = d_st.download()
        bad = st[st != 0]
        tot_decl += len(bad)
        # (round 6) would the HOST replay of Qhull's run take the sets this kernel declines?  (its guard bands are Qhull's own decisions',
        # this kernel's are on its in-circle predicates: where it accepts, the triangle SET is the unique Delaunay triangulation)
        from mvoscalerecovery_amd import packing
        for f in np.nonzero(st != 0)[0]:
            pts = uv[off[f]:off[f] + cnts[f]]
            rows = packing.qhull_rows_host(pts)
            host_ok = host_ok + 1 if rows is not None else host_ok
            if rows is not None:
                same = set(map(tuple, np.sort(rows, 1))) == set(map(tuple, np.sort(packing.delaunay_simplices(pts), 1)))
                host_same = host_same + 1 if same else host_same
        for s in bad:
            why[int(s) >> 8] = why.get(int(s) >> 8, 0) + 1
    print("%-28s %6d sets of %d points: declined %d (%.4f %%)  reasons %s; the host replay accepts %d of them (same triangle set as SciPy: %d)" % (
        kind, F, n, tot_decl, 100.0 * tot_decl / F, why, host_ok, host_same), flush=True)
