#!/usr/bin/env python3
"""VERDICT r4 item 1(a)+(b)+(c): Qhull's insertion order, observed, against the restatement `oracle/qhull_rows.py`.

    python profiles/qhull_order_truth.py [frames_per_size] > profiles/r05_qhull_order_truth.txt

Three independent observations of the same order pi per frame:
  TV    rank[n] = distinct vertices in Delaunay(P, qhull_options="Qbb Qc Qz Q12 TV-n") (SciPy API only; a few frames, O(n) runs)
  T1    Qhull's own `qh_addpoint` trace lines (profiles/qhull_trace.py: SciPy's bundled qhull_r 7.3.2 called directly)
  rows  every row of the full triangulation ends with its earliest-inserted vertex
and the gate: rows of `oracle.qhull_rows.delaunay_rows(P)` == `scipy.spatial.Delaunay(P).simplices`, row for row.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scipy.spatial import Delaunay                     # noqa: E402
from mvoscalerecovery_amd import synth                  # noqa: E402
from oracle.qhull_rows import QhullDelaunay2D, Declined  # noqa: E402
import qhull_trace                                      # noqa: E402


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    print("SciPy", __import__("scipy").__version__, "| Qhull", "2019.1.r (qhull_r 7.3.2) | options d Qbb Qc Qz Q12 Qt")
    # (a) the TV route on a few frames: last vertex of every row has the minimal rank; equals the T1 trace and the restatement
    for seed, n in [(0, 120), (1, 200), (2, 400)]:
        _, P = synth.synth_frame(seed, n, base_seed=31415)
        rank = qhull_trace.ranks_by_TV(P)
        rows = Delaunay(P).simplices
        r = rank[rows]
        last_min = int(np.sum(r[:, 2] <= r.min(axis=1)))
        _, adds = qhull_trace.insertion_order(P)
        q = QhullDelaunay2D(P)
        pi_trace = [a[0] for a in adds]
        pi_mine = [p for p in q.order[4:]]
        # ranks from TV: the initial simplex's three sites and the first point added share the minimal rank
        along = [int(rank[p]) for p in pi_trace if p < n]      # TV ranks can tie (a vertex in no lower facet yet): non-decreasing
        print("TV   frame seed %d n %d: rows whose last vertex has the minimal TV rank %d / %d; TV ranks non-decreasing along the T1 order: %s; "
              "restatement order == T1 order: %s" % (seed, n, last_min, len(rows), along == sorted(along),
                                                    pi_mine == pi_trace))
    # (c) the gate
    for n in (150, 400, 1000, 2000):
        ok = bad = dec = 0
        t0 = time.time()
        why = {}
        for seed in range(per):
            _, P = synth.synth_frame(seed, n, base_seed=31415)
            ref = Delaunay(P).simplices
            try:
                rows = QhullDelaunay2D(P).simplices()
            except Declined as e:
                dec += 1
                why[str(e)] = why.get(str(e), 0) + 1
                continue
            if rows.shape == ref.shape and np.array_equal(rows, ref):
                ok += 1
            else:
                bad += 1
        print("gate n %4d: %d frames: rows identical to SciPy's (order and rotation) %d, different %d, declined %d %s  (%.1f s)"
              % (n, per, ok, bad, dec, why if why else "", time.time() - t0))


if __name__ == "__main__":
    main()
