#!/usr/bin/env python3
"""Soak of mvosr_delaunay_qhull_batch against scipy.spatial.Delaunay: many seeded point sets of many sizes and shapes, rows
compared row for row (order and rotation); the declined sets by reason; mismatches (there must be none).
    python profiles/soak_qhull.py [sets] > profiles/r05_soak_qhull.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.spatial import Delaunay                     # noqa: E402
from mvoscalerecovery_amd import _lib, packing, synth   # noqa: E402

WHY = ["ok", "few points", "zero width", "simplex search", "flat simplex", "narrow", "inside simplex", "band", "coplanar horizon",
       "too many visible", "cone too large", "open cone", "not convex", "gauss", "not sharp", "above none", "facets full", "arena full"]


def make(kind, seed, rng):
    n = int(rng.integers(40, 2300))
    if kind == "synthetic frame":
        return synth.synth_frame(seed, n, base_seed=20260000)[1]
    if kind == "float32 positions":
        return synth.synth_frame(seed, n, base_seed=20261111)[1].astype(np.float32).astype(np.float64)
    if kind == "uniform image":
        return rng.uniform([0, 186], [1241, 376], (n, 2))
    if kind == "clusters":
        cen = rng.uniform([0, 186], [1241, 376], (10, 2))
        return np.concatenate([c + rng.normal(0, 6, (max(n // 14, 3), 2)) for c in cen] + [rng.uniform([0, 186], [1241, 376], (max(n // 4, 3), 2))])
    if kind == "survivors (keep mask)":
        p = synth.synth_frame(seed, n, base_seed=20262222)[1]
        return p[rng.random(len(p)) < 0.93]
    if kind == "quarter-pixel grid":
        return np.unique(np.round(synth.synth_frame(seed, n, base_seed=20263333)[1] * 4) / 4, axis=0)
    if kind.endswith("-pixel grid") and kind.startswith("1/"):          # "1/16-pixel grid", "1/64-pixel grid" (SOAK_KINDS)
        q = float(kind[2:kind.index("-")])
        return np.unique(np.round(synth.synth_frame(seed, n, base_seed=20263333)[1] * q) / q, axis=0)
    raise ValueError(kind)


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
    ctx = _lib.Context(0)
    kinds = ["synthetic frame", "float32 positions", "uniform image", "clusters", "survivors (keep mask)", "quarter-pixel grid"]
    if os.environ.get("SOAK_KINDS"):
        kinds = [k.strip() for k in os.environ["SOAK_KINDS"].split(",")]
    rng = np.random.default_rng(7)
    t0 = time.time()
    print("mvosr_delaunay_qhull_batch against scipy.spatial.Delaunay (SciPy %s), %d point sets of 40-2300 points" % (__import__("scipy").__version__, total))
    grand = [0, 0, 0]
    for kind in kinds:
        per = total // len(kinds)
        ok = bad = 0
        why = {}
        for base in range(0, per, 512):
            sets = [make(kind, 1000 * kinds.index(kind) + base + i, rng) for i in range(min(512, per - base))]
            got = packing.delaunay_gpu(ctx, sets, rows="qhull")
            st = packing.delaunay_gpu.last_status
            for k, (p, t) in enumerate(zip(sets, got)):
                if t is None:
                    r = WHY[int(st[k]) >> 8]
                    why[r] = why.get(r, 0) + 1
                    continue
                ref = Delaunay(p).simplices
                if t.shape == ref.shape and np.array_equal(t, ref):
                    ok += 1
                else:
                    bad += 1
        dec = sum(why.values())
        grand[0] += ok; grand[1] += bad; grand[2] += dec
        print("%-24s %6d sets: identical to SciPy (rows, order, rotation) %6d, DIFFERENT %d, declined %d %s" % (kind, per, ok, bad, dec, why if why else ""), flush=True)
    print("total: identical %d, different %d, declined %d (%.2f %%)   [%.0f s]" % (grand[0], grand[1], grand[2], 100.0 * grand[2] / max(sum(grand), 1), time.time() - t0))


if __name__ == "__main__":
    main()
