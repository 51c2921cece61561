#!/usr/bin/env python3
"""How long do the two triangulation kernels take to DECLINE a frame?  (round 6, LABNOTES 10.11)
A launch of F ordinary 2000-point sets against the same launch with ONE set snapped to a quarter-pixel grid (declined by construction),
and a launch of that one set alone: if declining takes longer than an ordinary set's triangulation, every call with a declined frame
waits for that one workgroup.
    python profiles/decline_kernel_probe.py [sets] [points]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, synth                              # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = _lib.default_context(0)
pool = [synth.synth_frame(i, n, base_seed=2024)[1][:n] for i in range(64)]


def run(sets, kind):
    Fs = len(sets)
    cnt = np.array([len(p) for p in sets], dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(cnt.astype(np.int64))])
    uv = np.concatenate(sets)
    d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
    d_off, d_cnt, d_toff = ctx.to_device(off[:-1].copy()), ctx.to_device(cnt), ctx.to_device((2 * off[:-1]).copy())
    d_tri = ctx.empty((2 * int(off[-1]), 3), np.int32)
    d_tcnt, d_st = ctx.zeros(Fs, np.int32), ctx.zeros(Fs, np.int32)
    if kind == "qhull":
        launch = lambda: _lib.check(ctx.lib.mvosr_delaunay_qhull_batch(ctx.handle, Fs, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, int(cnt.max()), d_toff.ptr,
                                                                       d_tri.ptr, d_tcnt.ptr, None, d_st.ptr, None), "qhull")
    else:
        launch = lambda: _lib.check(ctx.lib.mvosr_delaunay_batch(ctx.handle, Fs, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, int(cnt.max()), d_toff.ptr,
                                                                 d_tri.ptr, d_tcnt.ptr, None, d_st.ptr), "delaunay")
    launch(); ctx.sync()
    ms = []
    for _ in range(5):
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); launch(); ctx.record(e1); ctx.sync()
        ms.append(ctx.elapsed_ms(e0, e1))
    st = d_st.download()
    for b in (d_u, d_v, d_off, d_cnt, d_toff, d_tri, d_tcnt, d_st):
        b.free()
    return sorted(ms)[2], int((st != 0).sum()), [int(s) for s in st[st != 0][:4]]


snapped = [np.ascontiguousarray(np.round(pool[k] * 4.0) / 4.0) for k in range(8)]
for kind in ("qhull", "delaunay"):
    base = [pool[i % 64] for i in range(F)]
    t0, d0, _ = run(base, kind)
    print("%-8s %5d ordinary sets of %d points: %8.3f ms (declined %d)" % (kind, F, n, t0, d0), flush=True)
    for k in range(4):
        one = list(base); one[F // 2] = snapped[k]
        t1, d1, why = run(one, kind)
        ta, da, _ = run([snapped[k]], kind)
        tb, db, _ = run([pool[k]], kind)
        print("%-8s ... with snapped set %d in the middle: %8.3f ms (%+.3f; declined %d, status %s); that set ALONE: %7.3f ms (declined %d); an ordinary set alone: %7.3f ms" % (
            kind, k, t1, t1 - t0, d1, ["0x%x" % w for w in why], ta, da, tb), flush=True)
