#!/usr/bin/env python3
"""What determines the rotation of SciPy/Qhull's `simplices` rows — the one thing the reference's vote needs from the
triangulation beyond the triangle set (/root/reference/src/scale_calculator.py:105-119).  SciPy only (no GPU, no oracle):

    python profiles/qhull_order_probe.py > profiles/r04_qhull_order_probe.txt

Facts it prints, per 2000-feature frame of the bench's generator:
  1. rows are CCW and the relation "row[2] precedes row[0], row[1]" is acyclic: a row = the CCW triangle rotated until the
     vertex that comes first in one global order pi is last (tests/test_qhull_row_structure.py asserts it);
  2. pi looks like a random permutation to every local key: the share of rows whose last vertex is the minimum of index, u, v,
     centred radius or lifted height is 1/3 (chance); the share of points that precede all their neighbours is 1/7 (what a
     random order gives at mean degree 6);
  3. pi is (almost always) a function of the point SET: relabelling the input points leaves every row's last vertex where
     it was in most frames — it is Qhull's insertion order, the furthest point of the next facet with a non-empty outside
     set on the Qbb-scaled paraboloid, not a by-product of input order — and it is NOT invariant under translation/scaling
     of the pixels (the lifted distances change), so it cannot be derived from an affine-invariant predicate either.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth          # noqa: E402
from scipy.spatial import Delaunay              # noqa: E402


def earliest(rows):
    return {tuple(sorted(r)): r[2] for r in rows.tolist()}


def main():
    print("frame | acyclic | last=argmin(index/u/v/radius/lift) | points before all neighbours | same last vertex after: relabel / reverse / 2x+100")
    for seed in range(20):
        f3, p = synth.synth_frame(seed, 2000, base_seed=31415)
        rows = Delaunay(p).simplices
        n = len(p)
        succ = [[] for _ in range(n)]
        indeg = np.zeros(n, dtype=np.int64)
        for a, b, c in rows.tolist():
            succ[c] += [a, b]
            indeg[a] += 1
            indeg[b] += 1
        sources = int((indeg == 0).sum())
        stack, seen = [i for i in range(n) if indeg[i] == 0], 0
        while stack:
            i = stack.pop()
            seen += 1
            for j in succ[i]:
                indeg[j] -= 1
                if indeg[j] == 0:
                    stack.append(j)
        c = p.mean(0)
        keys = [np.arange(n), p[:, 0], p[:, 1], -np.hypot(p[:, 0] - c[0], p[:, 1] - c[1]), p[:, 0] ** 2 + p[:, 1] ** 2]
        hits = ["%.3f" % float(np.mean(np.argmin(k[rows], axis=1) == 2)) for k in keys]
        base = earliest(rows)
        rng = np.random.default_rng(seed)
        perm = rng.permutation(n)
        same = []
        for other in (perm[Delaunay(p[perm]).simplices], n - 1 - Delaunay(p[::-1]).simplices, Delaunay(p * 2.0 + 100.0).simplices):
            e = earliest(other)
            same.append("%.3f" % float(np.mean([base[k] == e.get(k, -1) for k in base])))
        print("%5d | %s | %s | %d of %d (%.3f) | %s" % (seed, seen == n, " ".join(hits), sources, n, sources / n, " / ".join(same)))


if __name__ == "__main__":
    main()
