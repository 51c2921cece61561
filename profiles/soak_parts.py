#!/usr/bin/env python3
"""Soak of delaunay_kernel's PARTS launches (a few frames, several workgroups each) against the one-workgroup launch: random
batches of 1-16 frames of 520-4700 points, first triangulation and the seeded second one with carried stars over a random mask —
rows, counts, statuses and seed words must be identical.   python profiles/soak_parts.py [batches]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, synth          # noqa: E402


def run(ctx, sets, keep):
    F = len(sets)
    cnt = np.array([len(q) for q in sets], dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64)
    uv = np.concatenate(sets)
    d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
    d_off, d_cnt, d_toff, d_keep = ctx.to_device(off), ctx.to_device(cnt), ctx.to_device(2 * off), ctx.to_device(keep)
    rows = int(2 * cnt.sum())
    t1, t2 = ctx.empty((rows, 3), np.int32), ctx.empty((rows, 3), np.int32)
    c1, c2, s1, s2 = (ctx.zeros(F, np.int32) for _ in range(4))
    info = ctx.zeros(int(cnt.sum()), np.uint32)
    n_max = int(cnt.max())
    _lib.check(ctx.lib.mvosr_delaunay_batch_ex(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, n_max, d_toff.ptr, t1.ptr, c1.ptr,
                                               None, s1.ptr, None, None, None, None, info.ptr), "first")
    _lib.check(ctx.lib.mvosr_delaunay_batch_ex(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, d_keep.ptr, n_max, d_toff.ptr, t2.ptr, c2.ptr,
                                               None, s2.ptr, d_toff.ptr, t1.ptr, c1.ptr, info.ptr, None), "second")
    # ... and seeded without the per-point words (no carried stars: the exact path's stand-in, seeded with Qhull's rows)
    t3, c3, s3 = ctx.empty((rows, 3), np.int32), ctx.zeros(F, np.int32), ctx.zeros(F, np.int32)
    _lib.check(ctx.lib.mvosr_delaunay_batch_seeded(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, d_keep.ptr, n_max, d_toff.ptr, t3.ptr, c3.ptr,
                                                   None, s3.ptr, d_toff.ptr, t1.ptr, c1.ptr), "second, seeds only")
    out = [x.download() for x in (t1, t2, c1, c2, s1, s2, info, t3, c3, s3)]
    for b in (d_u, d_v, d_off, d_cnt, d_toff, d_keep, t1, t2, c1, c2, s1, s2, info, t3, c3, s3):
        b.free()
    return out, off


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ctx = _lib.default_context(0)
    rng = np.random.default_rng(2025)
    bad = frames = declined = 0
    for b in range(batches):
        F = int(rng.integers(1, 17))
        sets = [synth.synth_frame(1000 * b + i, int(rng.integers(520, 4701)), base_seed=4242)[1] for i in range(F)]
        if b % 7 == 0:                                  # (a duplicate point now and then: declined by both)
            sets[0] = sets[0].copy(); sets[0][5] = sets[0][99]
        keep = np.where(rng.uniform(size=sum(len(q) for q in sets)) < rng.uniform(0.5, 0.98), 1, -1).astype(np.int32)
        os.environ.pop("MVOSR_DT_PARTS", None)
        a, off = run(ctx, sets, keep)
        os.environ["MVOSR_DT_PARTS"] = "0"
        c, _ = run(ctx, sets, keep)
        os.environ.pop("MVOSR_DT_PARTS", None)
        frames += F
        declined += int((a[4] != 0).sum())
        # (of a declined frame's status only the code is compared: the reason bits above it say which of several tests fired first —
        # a duplicate point is also a tie and a collinear triple — and that depends on the order the stars were walked in)
        same = all(np.array_equal(a[k], c[k]) for k in (2, 3)) and all(np.array_equal(a[k] & 0xFF, c[k] & 0xFF) for k in (4, 5))
        for f in range(F):
            lo = int(2 * off[f])
            same = same and np.array_equal(a[0][lo:lo + a[2][f]], c[0][lo:lo + c[2][f]]) and np.array_equal(a[1][lo:lo + a[3][f]], c[1][lo:lo + c[3][f]])
            # (seeds only: the same rows as with carried stars, in both launch shapes)
            same = same and a[8][f] == c[8][f] == a[3][f] and np.array_equal(a[7][lo:lo + a[8][f]], a[1][lo:lo + a[3][f]]) and np.array_equal(c[7][lo:lo + c[8][f]], a[1][lo:lo + a[3][f]])
            if a[4][f] == 0:
                same = same and np.array_equal(a[6][off[f]:off[f] + len(sets[f])], c[6][off[f]:off[f] + len(sets[f])])
        if not same:
            bad += 1
            print("batch %d (%d frames, sizes %s) differs" % (b, F, [len(q) for q in sets]), flush=True)
            for f in range(F):
                lo = int(2 * off[f])
                r1 = np.array_equal(a[0][lo:lo + a[2][f]], c[0][lo:lo + c[2][f]])
                r2 = np.array_equal(a[1][lo:lo + a[3][f]], c[1][lo:lo + c[3][f]])
                if not (r1 and r2 and (a[4][f] & 0xFF) == (c[4][f] & 0xFF) and (a[5][f] & 0xFF) == (c[5][f] & 0xFF) and a[2][f] == c[2][f] and a[3][f] == c[3][f]):
                    print("   frame %d (%d points): first rows equal %s (%d / %d rows, status %x / %x); second rows equal %s (%d / %d rows, status %x / %x)" % (
                        f, len(sets[f]), r1, a[2][f], c[2][f], a[4][f], c[4][f], r2, a[3][f], c[3][f], a[5][f], c[5][f]), flush=True)
                elif a[4][f] == 0 and not np.array_equal(a[6][off[f]:off[f] + len(sets[f])], c[6][off[f]:off[f] + len(sets[f])]):
                    d = np.nonzero(a[6][off[f]:off[f] + len(sets[f])] != c[6][off[f]:off[f] + len(sets[f])])[0]
                    print("   frame %d (%d points): seed words differ at %d points, e.g. %d: %08x / %08x" % (f, len(sets[f]), len(d), d[0], a[6][off[f] + d[0]], c[6][off[f] + d[0]]), flush=True)
    print("%d batches, %d frames of 520-4700 points: %d batches differ; first triangulations declined (by both): %d" % (batches, frames, bad, declined))


if __name__ == "__main__":
    main()
