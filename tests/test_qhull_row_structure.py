"""What SciPy/Qhull's `simplices` rows are, stated as a test (the obstacle to a reference-exact device
triangulation): every row of scipy.spatial.Delaunay(points2d).simplices is counter-clockwise and ends with the vertex
Qhull inserted FIRST among the three — Qhull keeps a simplicial facet's vertices by decreasing vertex id (= insertion
order) and SciPy swaps the first two where needed for orientation.  So the relation "row[2] was inserted before row[0] and
row[1]" over all rows is ACYCLIC, and a row is a function of the triangle set and of one global order pi of the points:
"rotate the CCW triangle until its earliest vertex is last".  The reference's vote
(/root/reference/src/scale_calculator.py:105-119) reads exactly that rotation.  pi is Qhull's insertion order (the
furthest point of the next facet with a non-empty outside set), which no simple key of the points reproduces."""
import numpy as np


def _frame(seed, n):
    from mvoscalerecovery_amd import synth
    f3, f2 = synth.synth_frame(seed, n, base_seed=31415)
    return f2


def _topological_order(n, rows):
    """A linear extension of {row[2] < row[0], row[2] < row[1]} (Kahn), or None when the relation has a cycle."""
    succ = [[] for _ in range(n)]
    indeg = np.zeros(n, dtype=np.int64)
    for a, b, c in rows:
        succ[c] += [a, b]
        indeg[a] += 1
        indeg[b] += 1
    order, stack = [], [i for i in range(n) if indeg[i] == 0]
    while stack:
        i = stack.pop()
        order.append(i)
        for j in succ[i]:
            indeg[j] -= 1
            if indeg[j] == 0:
                stack.append(j)
    return order if len(order) == n else None


def test_scipy_rows_are_ccw_with_the_earliest_vertex_last():
    from scipy.spatial import Delaunay
    for seed, n in [(s, 2000) for s in range(6)] + [(100 + s, 300 + 97 * s) for s in range(8)]:
        p = _frame(seed, n)
        rows = Delaunay(p).simplices
        a, b, c = p[rows[:, 0]], p[rows[:, 1]], p[rows[:, 2]]
        cross = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
        assert np.all(cross > 0), (seed, n)                               # every row counter-clockwise
        order = _topological_order(len(p), rows.tolist())
        assert order is not None, (seed, n)                               # the "last vertex came first" relation is acyclic
        rank = np.empty(len(p), dtype=np.int64)
        rank[order] = np.arange(len(p))
        # the rows are a function of (triangle set, rank): rotate each CCW triangle until its lowest-rank vertex is last
        k = np.argmin(rank[rows], axis=1)
        rebuilt = np.stack([rows[np.arange(len(rows)), (k + 1) % 3], rows[np.arange(len(rows)), (k + 2) % 3],
                            rows[np.arange(len(rows)), k]], axis=1)
        assert np.array_equal(rebuilt, rows), (seed, n)
        # ... and no simple key of the points is that order: none of these predicts every row's last vertex
        keys = {"index": np.arange(len(p)), "u": p[:, 0], "v": p[:, 1],
                "radius": -np.hypot(p[:, 0] - p[:, 0].mean(), p[:, 1] - p[:, 1].mean())}
        for name, key in keys.items():
            hit = float(np.mean(np.argmin(key[rows], axis=1) == 2))
            assert hit < 0.75, (name, hit)
