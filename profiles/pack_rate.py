#!/usr/bin/env python3
"""Host-only: the C packer's rate (mvosr_pack_fill into page-locked memory) by thread count and by pieces per call.
    python profiles/pack_rate.py [frames] [features]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, synth, packing
from mvoscalerecovery_amd.engine import frame_tables
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = _lib.default_context(0)
pool = [synth.synth_frame(i, N, base_seed=2024) for i in range(4096)]
def run(threads, pieces, reps=6, start=0):
    f3 = [pool[(start + i) % 4096][0] for i in range(F)]; f2 = [pool[(start + i) % 4096][1] for i in range(F)]
    p3, p2, npts = frame_tables(f3, f2)
    off, total = packing.pack_layout(npts)
    stage = _lib.PinnedBuffer(ctx, 5 * 8 * total + 64)
    cnt = np.zeros(F, np.int32)
    base = stage.ptr
    ts = []
    for r in range(reps):
        t0 = time.perf_counter()
        edges = [F * k // pieces for k in range(pieces + 1)]
        for a, b in zip(edges[:-1], edges[1:]):
            _lib.check(ctx.lib.mvosr_pack_fill(b - a, _lib.addr(p3) + 8 * a, _lib.addr(p2) + 8 * a, _lib.addr(npts) + 4 * a, 0.0, _lib.addr(off) + 8 * a,
                                               base, base + 8 * total, base + 16 * total, base + 24 * total, base + 32 * total, 0, 1.0, 0.0, threads,
                                               _lib.addr(cnt) + 4 * a), "pack")
        ts.append(time.perf_counter() - t0)
    stage.free()
    return min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2]
for threads in (1, 2, 4, 8, 16):
    for pieces in (1, 4):
        best, med = run(threads, pieces)
        print("threads %2d pieces %d: %.2f ms best, %.2f median for %d frames of %d (%.2f us/frame, %.1f GB/s read+written)" % (
            threads, pieces, 1e3 * best, 1e3 * med, F, N, 1e6 * best / F, 2 * 40 * N * F / best / 1e9), flush=True)
