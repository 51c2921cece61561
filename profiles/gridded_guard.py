#!/usr/bin/env python3
"""VERDICT r5 #3b: do the declines of quarter-pixel-gridded point sets depend on the width of the roundoff guard?  Builds
csrc/mvosr_qhull_host.c with the guard at 64 / 16 / 4 / 3 x DISTround (gcc, no GPU) and compares every accepted set with SciPy.
    python profiles/gridded_guard.py > profiles/r06_gridded_guard.txt"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scipy.spatial import Delaunay                     # noqa: E402
from mvoscalerecovery_amd import synth                  # noqa: E402

WHY = {1: "few", 2: "zero width", 3: "flat simplex", 4: "narrow", 5: "one extreme", 6: "simplex search", 7: "degenerate facet", 8: "pivot",
       9: "initial roundoff", 10: "inside simplex", 11: "partition roundoff", 12: "above none", 13: "visibility roundoff", 14: "coplanar horizon",
       15: "open cone", 16: "not convex", 17: "overflow"}


def main():
    src = open(os.path.join(ROOT, "mvoscalerecovery_amd", "csrc", "mvosr_qhull_host.c")).read()
    sets = [np.unique(np.round(synth.synth_frame(s, 600 + (s * 37) % 1400, base_seed=20263333)[1] * 4) / 4, axis=0) for s in range(400)]
    refs = [Delaunay(p).simplices for p in sets]
    tmp = tempfile.mkdtemp()
    for g in (64, 16, 4, 3):
        c, so = os.path.join(tmp, "g%d.c" % g), os.path.join(tmp, "g%d.so" % g)
        open(c, "w").write(src.replace("S->guard = 64 * S->distround;", "S->guard = %d * S->distround;" % g))
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-std=gnu11", "-ffp-contract=off", c, "-lm", "-o", so], check=True)
        lib = C.CDLL(so)
        lib.mvosr_qhull_rows_host.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        ok = bad = 0
        why = {}
        for p, r in zip(sets, refs):
            p = np.ascontiguousarray(p)
            rows, n = np.empty((2 * len(p) + 8, 3), np.int32), np.zeros(1, np.int32)
            rc = lib.mvosr_qhull_rows_host(p.ctypes.data, len(p), 2, rows.ctypes.data, len(rows), n.ctypes.data, None)
            if rc == 0:
                ok, bad = ok + bool(np.array_equal(rows[:n[0]], r)), bad + (not np.array_equal(rows[:n[0]], r))
            else:
                why[WHY[rc]] = why.get(WHY[rc], 0) + 1
        print("guard %d x DISTround: identical %d, DIFFERENT %d, declined %s" % (g, ok, bad, why))


if __name__ == "__main__":
    main()
