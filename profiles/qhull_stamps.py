#!/usr/bin/env python3
"""Where a wavefront of qhull_rows_kernel spends its time: s_memtime sections of frame 0 (build with EXTRA=-DMVOSR_QH_STAMPS).
   python profiles/qhull_stamps.py [frames] [points]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, synth   # noqa: E402

NAMES = ["pick", "visibility search", "cone", "match + sharp", "partition targets", "placement", "records + fence", "-"]


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    ctx = _lib.Context(0)
    lib = ctx.lib
    lib.mvosr_debug_qh_stamps.argtypes = [C.c_void_p]
    d_stamps = ctx.zeros(32, np.uint64)
    lib.mvosr_debug_qh_stamps(d_stamps.ptr)
    pool = [synth.synth_frame(s, n, base_seed=999)[1] for s in range(min(frames, 256))]
    sets = [pool[s % len(pool)] for s in range(frames)]
    cnt = np.array([len(p) for p in sets], dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(cnt.astype(np.int64))])
    uv = np.concatenate(sets)
    d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
    d_off, d_cnt, d_toff = ctx.to_device(off[:-1].astype(np.int64)), ctx.to_device(cnt), ctx.to_device((2 * off[:-1]).astype(np.int64))
    d_tri = ctx.empty((2 * int(off[-1]), 3), np.int32)
    d_tc, d_st = ctx.zeros(frames, np.int32), ctx.zeros(frames, np.int32)
    for rep in range(2):
        _lib.check(lib.mvosr_delaunay_qhull_batch(ctx.handle, frames, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, int(cnt.max()),
                                                  d_toff.ptr, d_tri.ptr, d_tc.ptr, None, d_st.ptr, None))
        ctx.sync()
    st = d_stamps.download()
    tot = float(st[:7].sum())
    steps = int(st[8])
    print("%d resident frames of %d points; frame 0: %d insertions, %.0f clocks per insertion" % (frames, n, steps, tot / max(steps, 1)))
    for k in range(7):
        print("  %-20s %5.1f %%  %8.0f clocks per insertion" % (NAMES[k], 100.0 * st[k] / tot, st[k] / max(steps, 1)))
    by_size(st)


def by_size(st):
    names = ["no points to place", "1..64 points", "more than 64"]
    for c in range(3):
        n = float(st[19 + c])
        if n:
            print("  partition + placement, %-20s %6.0f insertions, %8.0f clocks each (%4.1f %% of the run)" % (names[c], n, st[16 + c] / n, 100.0 * st[16 + c] / max(float(st[:7].sum()), 1.0)))


if __name__ == "__main__":
    main()
