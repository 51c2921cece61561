#!/usr/bin/env python3
"""Soak of mvosr_delaunay_batch / _seeded against scipy.spatial.Delaunay (canonical rows): random point sets of several
distributions and sizes, plus a second triangulation over a random 40-97 % of each set seeded with the first.
    [SOAK_SETS=520 SOAK_NMAX=2150] python profiles/soak_delaunay.py [rounds]      (160 sets per round by default)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, packing
from scipy.spatial import Delaunay

from fractions import Fraction as Fr

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
# SOAK_SETS (160) sets per launch of up to SOAK_NMAX (4700) points: 512 sets and more let the launcher pick its small-frame and
# arena-out instantiations by the largest set (SOAK_SETS=520 SOAK_NMAX=2150 / 3300 / 1100 / 500)
SETS, NMAX = int(os.environ.get("SOAK_SETS", "160")), int(os.environ.get("SOAK_NMAX", "4700"))
QUANT = float(os.environ.get("SOAK_QUANT", "0"))


def empty_circle_violations(p, tris, verts):
    """Exact rational arithmetic: how many of `tris` have one of `verts` strictly inside their circumcircle."""
    bad_ = 0
    for t in tris:
        a_, b_, c_ = ([Fr(float(p[i][0])), Fr(float(p[i][1]))] for i in t)
        if (b_[0] - a_[0]) * (c_[1] - a_[1]) - (b_[1] - a_[1]) * (c_[0] - a_[0]) < 0:
            a_, b_ = b_, a_
        for v in verts:
            if v in t:
                continue
            d = [Fr(float(p[v][0])), Fr(float(p[v][1]))]
            ax, ay, bx, by, cx, cy = a_[0] - d[0], a_[1] - d[1], b_[0] - d[0], b_[1] - d[1], c_[0] - d[0], c_[1] - d[1]
            det = (ax * ax + ay * ay) * (bx * cy - cx * by) - (bx * bx + by * by) * (ax * cy - cx * ay) + (cx * cx + cy * cy) * (ax * by - bx * ay)
            if det > 0:
                bad_ += 1
                break
    return bad_


ctx = _lib.default_context(0)
lib = ctx.lib
rng = np.random.default_rng(31337)
total = bad = declined = bad2 = declined2 = qhull_inexact = device_wrong = qhull_inexact2 = device_wrong2 = 0
for rnd in range(rounds):
    sets = []
    for k in range(SETS):
        n = int(rng.integers(4, NMAX))
        kind = k % 6
        if kind == 0: p = rng.uniform(0, 1, (n, 2)) * [1241.0, 376.0]
        elif kind == 1: p = rng.normal(0, 1, (n, 2)) * [300.0, 40.0] + [600, 200]
        elif kind == 2:
            c = rng.uniform(0, 1000, (8, 2)); p = c[rng.integers(0, 8, n)] + rng.normal(0, 15, (n, 2))
        elif kind == 3:
            t = rng.uniform(0, 2 * np.pi, n); r = rng.uniform(0.2, 1.0, n) ** 0.3; p = np.stack([r * np.cos(t) * 500, r * np.sin(t) * 100], 1)
        elif kind == 4:
            p = np.concatenate([rng.uniform(0, 100, (n // 2, 2)), rng.uniform(900, 1000, (n - n // 2, 2)) * [1, 0.1]])
        else:
            p = rng.uniform(0, 1, (n, 2)) ** 3 * [2000.0, 500.0]           # strongly non-uniform density
        if QUANT:                                   # SOAK_QUANT=16: coordinates on a 1/16 px grid — collinear triples, cocircular quadruples, duplicates
            p = np.round(p * QUANT) / QUANT
        sets.append(np.ascontiguousarray(p))
    F = len(sets)
    cnt = np.array([len(p) for p in sets], dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64)
    uv = np.concatenate(sets)
    keep = np.concatenate([np.where(rng.uniform(size=n) < rng.uniform(0.4, 0.97), 1, -1) for n in cnt]).astype(np.int32)
    d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
    d_off, d_cnt, d_toff, d_keep = ctx.to_device(off), ctx.to_device(cnt), ctx.to_device(2 * off), ctx.to_device(keep)
    rows = int(2 * cnt.sum())
    tri1, tri2 = ctx.empty((rows, 3), np.int32), ctx.empty((rows, 3), np.int32)
    c1, c2, s1, s2 = (ctx.zeros(F, np.int32) for _ in range(4))
    nmax = int(cnt.max())
    # (round 4: the second triangulation carries over the stars the mask did not touch — mvosr_delaunay_batch_ex with the
    # first launch's per-point facts; every third round keeps 90-99 % of the points, the vote's regime)
    if rnd % 3 == 2:
        keep = np.concatenate([np.where(rng.uniform(size=n) < rng.uniform(0.9, 0.99), 1, -1) for n in cnt]).astype(np.int32)
        d_keep = ctx.to_device(keep)
    d_info = ctx.zeros(int(cnt.sum()), np.uint32)
    _lib.check(lib.mvosr_delaunay_batch_ex(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, nmax, d_toff.ptr, tri1.ptr, c1.ptr, None, s1.ptr,
                                           None, None, None, None, d_info.ptr), "dt1")
    _lib.check(lib.mvosr_delaunay_batch_ex(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, d_keep.ptr, nmax, d_toff.ptr, tri2.ptr, c2.ptr, None,
                                           s2.ptr, d_toff.ptr, tri1.ptr, c1.ptr, d_info.ptr, None), "dt2")
    t1, t2, n1, n2, h1, h2 = tri1.download(), tri2.download(), c1.download(), c2.download(), s1.download(), s2.download()
    for f, p in enumerate(sets):
        a = int(2 * off[f])
        total += 1
        if h1[f] != 0:
            declined += 1
        elif not np.array_equal(t1[a:a + n1[f]], packing.canonical_rows(Delaunay(p).simplices)):
            bad += 1
            ref = packing.canonical_rows(Delaunay(p).simplices)
            got = t1[a:a + n1[f]]
            gs, rs = set(map(tuple, got.tolist())), set(map(tuple, ref.tolist()))
            # who is right?  the rows only one side has, against the vertices they involve, in exact arithmetic
            og, orf = sorted(gs - rs), sorted(rs - gs)
            verts = sorted(set(v for t in og + orf for v in t))
            vg, vr = empty_circle_violations(p, og, verts), empty_circle_violations(p, orf, verts)
            if vg == 0 and vr > 0:
                qhull_inexact += 1
            else:
                device_wrong += 1
            print("MISMATCH first n=%d kind=%d rows %d vs %d; only GPU %d (violating the empty circle, exactly: %d), only SciPy %d (%d)"
                  % (len(p), f % 6, len(got), len(ref), len(og), vg, len(orf), vr))
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez("gpurun_out/dt_mismatch_%d.npz" % bad, pts=p, gpu=got, ref=ref)
        q = p[keep[off[f]:off[f] + cnt[f]] >= 0]
        if h2[f] != 0:
            declined2 += 1
        elif len(q) >= 3 and not np.array_equal(t2[a:a + n2[f]], packing.canonical_rows(Delaunay(q).simplices)):
            bad2 += 1
            ref = packing.canonical_rows(Delaunay(q).simplices)
            idx = np.flatnonzero(keep[off[f]:off[f] + cnt[f]] >= 0)              # the kernel's rows name the points of the full set
            rank = np.full(len(p), -1); rank[idx] = np.arange(len(idx))
            got = rank[t2[a:a + n2[f]]] if t2[a:a + n2[f]].max(initial=0) >= len(q) else t2[a:a + n2[f]]
            gs, rs = set(map(tuple, np.sort(got, 1).tolist())), set(map(tuple, np.sort(ref, 1).tolist()))
            og, orf = sorted(gs - rs), sorted(rs - gs)
            verts = sorted(set(v for t in og + orf for v in t))
            vg, vr = empty_circle_violations(q, og, verts), empty_circle_violations(q, orf, verts)
            if vg == 0 and vr > 0:
                qhull_inexact2 += 1
            else:
                device_wrong2 += 1
            print("MISMATCH second n=%d kept=%d kind=%d; only GPU %d (violating, exactly: %d), only SciPy %d (%d)" % (len(p), len(q), f % 6, len(og), vg, len(orf), vr))
    for b in (d_u, d_v, d_off, d_cnt, d_toff, d_keep, tri1, tri2, c1, c2, s1, s2):
        b.free()
print(("coordinates on a 1/%g grid: " % QUANT if QUANT else "") + "sets %d: first triangulation declined %d, mismatches %d (SciPy's rows violate the empty circle in exact arithmetic, the device's do not: %d; otherwise: %d); "
      "seeded second declined %d, mismatches %d (%d; %d)" % (total, declined, bad, qhull_inexact, device_wrong, declined2, bad2, qhull_inexact2, device_wrong2))
