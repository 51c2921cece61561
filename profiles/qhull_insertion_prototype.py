#!/usr/bin/env python3
"""Time-boxed prototype: can Qhull's INSERTION ORDER pi — the one thing the reference's vote needs from the
triangulation beyond the triangle set: SciPy's row is the CCW triangle rotated until its earliest-inserted vertex is last
(tests/test_qhull_row_structure.py) — be reproduced from first principles?  SciPy only (no GPU, no oracle):

    python profiles/qhull_insertion_prototype.py [frames] > profiles/r04_qhull_insertion_prototype.txt

What is rebuilt here, from Qhull's documented algorithm (the library's source is not in this image and its trace output is
compiled out of SciPy's copy): the sites lifted to the paraboloid, the last coordinate scaled as 'Qbb' does, optionally the
point 'at infinity' of 'Qz'; an initial simplex from the points with extreme coordinates, grown by the largest determinant;
every other point handed to an outside set; then Quickhull's loop — the furthest point of the first facet (in facet-list
order) that has an outside set, the visible facets, a cone of new facets at the end of the list, the visible facets' points
re-partitioned.  Variants of the places where the documentation leaves a choice are all tried.

The score: pi_proto against SciPy's rows — the share of rows whose LAST vertex is the one the prototype inserted first (a
random order gives 1/3) — and the insertion step at which the prototype first contradicts the rows' partial order.
"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth          # noqa: E402
from scipy.spatial import Delaunay              # noqa: E402


def lift(p, qz):
    x = p - 0.0
    z = (x ** 2).sum(1)
    # Qbb: the last coordinate scaled to [0, m], m = the largest |coordinate| of the others
    m = np.abs(x).max()
    z = (z - z.min()) / (z.max() - z.min()) * m
    pts = np.column_stack([x, z])
    if qz:                                        # Qz: a point above the paraboloid, over the centre of the sites
        pts = np.vstack([pts, [x[:, 0].mean(), x[:, 1].mean(), z.max() * 2.0 if qz == 1 else z.max()]])
    return pts


def initial_simplex(P):
    n = len(P)
    ext = []
    for k in range(3):
        ext += [int(np.argmin(P[:, k])), int(np.argmax(P[:, k]))]
    simplex = [int(np.argmin(P[:, 0])), int(np.argmax(P[:, 0]))]
    for _ in range(2):
        best, who = -1.0, -1
        for cand in list(dict.fromkeys(ext)) + ([] if best > 0 else []):
            if cand in simplex:
                continue
            d = abs(np.linalg.det(np.array([P[i] - P[simplex[0]] for i in simplex[1:] + [cand]])[:, :len(simplex)]
                                  if len(simplex) < 3 else np.array([P[i] - P[simplex[0]] for i in simplex[1:] + [cand]])))
            if d > best:
                best, who = d, cand
        if who < 0 or best < 1e-9:
            for cand in range(n):
                if cand in simplex:
                    continue
                M = np.array([P[i] - P[simplex[0]] for i in simplex[1:] + [cand]])
                d = abs(np.linalg.det(M[:, :len(simplex)] if len(simplex) < 3 else M))
                if d > best:
                    best, who = d, cand
        simplex.append(who)
    return simplex


class Hull:
    def __init__(self, P, simplex, first_wins):
        self.P, self.order = P, list(simplex)
        self.facets = {}                 # id -> [verts(3), normal, offset, outside list of (dist, point)]; dict keeps list order
        self.next_id = 0
        inner = P[simplex].mean(0)
        self.inner = inner
        for omit in range(4):
            v = [simplex[i] for i in range(4) if i != omit]
            self.add_facet(v)
        rest = [i for i in range(len(P)) if i not in simplex]
        self.partition(rest, list(self.facets), first_wins)

    def add_facet(self, v):
        a, b, c = self.P[v[0]], self.P[v[1]], self.P[v[2]]
        nrm = np.cross(b - a, c - a)
        nrm /= np.linalg.norm(nrm)
        off = -nrm.dot(a)
        if nrm.dot(self.inner) + off > 0:          # outward normals: the interior point is below
            v = [v[1], v[0], v[2]]
            nrm, off = -nrm, -off
        self.facets[self.next_id] = [v, nrm, off, []]
        self.next_id += 1
        return self.next_id - 1

    def partition(self, points, facet_ids, first_wins):
        if not points or not facet_ids:
            return
        N = np.array([self.facets[f][1] for f in facet_ids])
        O = np.array([self.facets[f][2] for f in facet_ids])
        D = self.P[points] @ N.T + O                # [points, facets]
        for r, pt in enumerate(points):
            above = np.nonzero(D[r] > 1e-12)[0]
            if not len(above):
                continue
            k = above[0] if first_wins else above[np.argmax(D[r][above])]
            self.facets[facet_ids[k]][3].append((D[r][k], pt))

    def run(self, first_wins):
        while True:
            fid = next((f for f, rec in self.facets.items() if rec[3]), None)
            if fid is None:
                break
            d, pt = max(self.facets[fid][3])         # the furthest point of the first facet with an outside set
            ids = list(self.facets)
            N = np.array([self.facets[f][1] for f in ids])
            O = np.array([self.facets[f][2] for f in ids])
            vis = [f for f, dd in zip(ids, N @ self.P[pt] + O) if dd > 1e-12]
            visset = set(vis)
            edges = {}
            for f in ids:
                v = self.facets[f][0]
                for e in ((v[0], v[1]), (v[1], v[2]), (v[2], v[0])):
                    edges[e] = f
            orphans = []
            new = []
            for f in vis:
                v = self.facets[f][0]
                for e in ((v[0], v[1]), (v[1], v[2]), (v[2], v[0])):
                    if edges.get((e[1], e[0])) not in visset:
                        new.append((e[0], e[1], pt))
                orphans += [q for _, q in self.facets[f][3] if q != pt]
            for f in vis:
                del self.facets[f]
            new_ids = [self.add_facet(list(t)) for t in new]
            self.order.append(pt)
            self.partition(orphans, new_ids, first_wins)
        return self.order


def score(rows, order, n):
    rank = np.full(n + 1, 10 ** 9)
    rank[np.array(order)] = np.arange(len(order))
    ok = np.argmin(rank[rows], axis=1) == 2
    # first insertion step that contradicts a row: the prototype inserted a vertex of the row before the row's last vertex
    bad_rows = rows[~ok]
    first_bad = int(rank[bad_rows].min()) if len(bad_rows) else -1
    return float(ok.mean()), first_bad


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    print("frame | points | variant (Qz point: 0 none / 1 far above / 2 at the top; partition: first facet / best facet) -> rows whose last vertex"
          " is the prototype's earliest | first insertion step contradicting SciPy's rows")
    tot = {}
    for seed in range(frames):
        f3, p = synth.synth_frame(seed, 400, base_seed=31415)
        rows = Delaunay(p).simplices
        n = len(p)
        out = []
        for qz, first_wins in itertools.product((0, 1, 2), (True, False)):
            P = lift(p, qz)
            H = Hull(P, initial_simplex(P), first_wins)
            order = H.run(first_wins)
            sc, fb = score(rows, [o for o in order if o < n], n)
            tot.setdefault((qz, first_wins), []).append(sc)
            out.append("qz%d/%s %.3f @%d" % (qz, "first" if first_wins else "best", sc, fb))
        print("%5d | %d | %s" % (seed, n, " | ".join(out)))
    print("mean share of rows reproduced (chance: 0.333): " + ", ".join("qz%d/%s %.3f" % (k[0], "first" if k[1] else "best", np.mean(v)) for k, v in tot.items()))


if __name__ == "__main__":
    main()
