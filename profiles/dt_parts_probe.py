#!/usr/bin/env python3
"""One-frame launches of mvosr_delaunay_batch_ex (first triangulation, then the seeded second over 95 % of the points): wall time per
launch with hipEvents-free timing (sync, 200 launches back to back).   MVOSR_DT_PARTS=0|n   python profiles/dt_parts_probe.py [points]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, synth          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ctx = _lib.default_context(0)
p = synth.synth_frame(7, n, base_seed=99)[1]
d_u, d_v = ctx.to_device(np.ascontiguousarray(p[:, 0])), ctx.to_device(np.ascontiguousarray(p[:, 1]))
d_off, d_cnt, d_toff = ctx.to_device(np.zeros(1, np.int64)), ctx.to_device(np.array([n], np.int32)), ctx.to_device(np.zeros(1, np.int64))
tri, tri2 = ctx.empty((2 * n, 3), np.int32), ctx.empty((2 * n, 3), np.int32)
tc, st, tc2, st2 = (ctx.zeros(1, np.int32) for _ in range(4))
info = ctx.zeros(n, np.uint32)
rng = np.random.default_rng(5)
keep = ctx.to_device(np.where(rng.uniform(size=n) < 0.95, 1, -1).astype(np.int32))
lib = ctx.lib


def first():
    _lib.check(lib.mvosr_delaunay_batch_ex(ctx.handle, 1, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, n, d_toff.ptr, tri.ptr, tc.ptr, None, st.ptr,
                                           None, None, None, None, info.ptr))


def second():
    _lib.check(lib.mvosr_delaunay_batch_ex(ctx.handle, 1, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, keep.ptr, n, d_toff.ptr, tri2.ptr, tc2.ptr, None, st2.ptr,
                                           d_toff.ptr, tri.ptr, tc.ptr, info.ptr, None))


for name, fn in (("first", first), ("seeded second (95 % kept, carried stars)", second)):
    for _ in range(5):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    ctx.sync()
    dt = (time.perf_counter() - t0) / 200
    print("MVOSR_DT_PARTS=%s  %d points, %s: %.1f us per launch (status %d, rows %d)" % (
        os.environ.get("MVOSR_DT_PARTS", "auto"), n, name, dt * 1e6,
        int((st if fn is first else st2).download()[0]), int((tc if fn is first else tc2).download()[0])))
