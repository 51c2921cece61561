"""Diagnostic: where a workgroup of delaunay_kernel spends its life (in-kernel s_memtime stamps).
Needs a library built with -DMVOSR_STAMPS:   profiles/ab_build.sh stamps -DMVOSR_STAMPS   (ONLY=mvosr_delaunay to rebuild that file alone)
    MVOSR_LIB_PATH=profiles/ab/libmvosr_stamps.so python profiles/dt_stamps.py [n] [sets]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, synth          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ctx = _lib.default_context(0)
pool = [synth.synth_frame(i, n, base_seed=99)[1] for i in range(64)]
cnt = np.full(F, n, dtype=np.int32)
off = np.arange(F, dtype=np.int64) * n
uv = np.concatenate([pool[i % 64] for i in range(F)])
d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
d_off, d_cnt, d_toff = ctx.to_device(off), ctx.to_device(cnt), ctx.to_device(2 * off)
d_tri = ctx.empty((2 * F * n, 3), np.int32)
d_tcnt, d_st, d_used = ctx.zeros(F, np.int32), ctx.zeros(F, np.int32), ctx.zeros(F, np.int32)
ROWS = int(os.environ.get("DT_PART_ROWS", "0"))       # PARTS launches (a few frames): a row of stamps per part
d_stamps = ctx.zeros((max(F, F * ROWS), 64), np.uint64)
ctx.lib.mvosr_debug_dt_stamps.argtypes = [C.c_void_p]
ctx.lib.mvosr_debug_dt_stamps(d_stamps.ptr)
d_info = ctx.zeros(F * n, np.uint32)
for _ in range(2):
    _lib.check(ctx.lib.mvosr_delaunay_batch_ex(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, n, d_toff.ptr,
                                               d_tri.ptr, d_tcnt.ptr, d_used.ptr, d_st.ptr, None, None, None, None, d_info.ptr), "dt")
ctx.sync()
if os.environ.get("DT_SEEDED"):
    # the second triangulation's shape: 85 % of the points kept, seeded with the rows just computed
    rng = np.random.default_rng(5)
    KEEP = float(os.environ.get("DT_KEEP", "0.85"))
    CARRY = os.environ.get("DT_CARRY", "1") == "1"
    keep = np.where(rng.uniform(size=F * n) < KEEP, 1, -1).astype(np.int32)
    d_keep = ctx.to_device(keep)
    d_tri2 = ctx.empty((2 * F * n, 3), np.int32)
    d_tcnt2 = ctx.zeros(F, np.int32)
    seeded = os.environ["DT_SEEDED"] == "1"
    for _ in range(2):
        _lib.check(ctx.lib.mvosr_delaunay_batch_ex(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, d_keep.ptr, n, d_toff.ptr,
                                                   d_tri2.ptr, d_tcnt2.ptr, d_used.ptr, d_st.ptr,
                                                   d_toff.ptr if seeded else None, d_tri.ptr if seeded else None, d_tcnt.ptr if seeded else None,
                                                   d_info.ptr if (seeded and CARRY) else None, None), "dt2")
    ctx.sync()
    print("second triangulation over %.0f %% of the points, %s%s:" % (100 * KEEP, "seeded" if seeded else "not seeded", ", untouched stars carried over" if (seeded and CARRY) else ""))
s = d_stamps.download().astype(np.float64)
if ROWS:
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    for r in range(len(s)):
        print("  part %2d: start %7.0f  setup %7.0f  phase 1 %8.0f  tail %7.0f  end %8.0f | loop iterations %4.0f  lanes busy %5.1f %%  wide scans %4.0f (row passes %4.0f)  hard %2.0f" % (
            r, s[r, 0] - t0, s[r, 2] - s[r, 0], s[r, 3] - s[r, 2], max(s[r, 6], s[r, 5]) - s[r, 3], max(s[r, 6], s[r, 5]) - t0, s[r, 12] / 256.0, 100.0 * s[r, 13] / max(s[r, 12], 1), s[r, 31], s[r, 7], s[r, 9]))
d = np.diff(s[:, :7], axis=1)
tot = s[:, 6] - s[:, 0]
names = ["load + bbox + grid dims", "zero + count / scan / scatter", "phase 1 (one lane per point)", "(barrier)", "phase 2 (hard points, groups)", "prefix + Euler + rows out"]
print("n=%d sets=%d  declined=%d  workgroup life: median %.0f cycles" % (n, F, int((d_st.download() != 0).sum()), np.median(tot)))
for k, name in enumerate(names[:6]):
    print("  %-26s %5.1f %%   (median %.0f cycles)" % (name, 100.0 * np.median(d[:, k] / tot), np.median(d[:, k])))
print("  hard points: median %.0f   triangles taken after a search: %.0f per set, from a hint: %.0f per set" % (np.median(s[:, 9]), np.mean(s[:, 10]), np.mean(s[:, 11])))
print("  steps per point: max %.0f (mean over sets); points with <=6 / <=10 / <=16 / <=28 / more steps: %s" % (np.mean(s[:, 16]), " / ".join("%.0f" % np.mean(s[:, 17 + k]) for k in range(5))))
print("  open stars (hull vertices): %.0f per set, %.1f steps each; closed stars: %.1f steps each" % (np.mean(s[:, 24]), np.mean(s[:, 22]) / max(np.mean(s[:, 24]), 1), np.mean(s[:, 23]) / max(n - np.mean(s[:, 24]), 1)))
print("  scan steps per lane: %.1f   lanes with a point in a step: %.1f %%" % (np.mean(s[:, 12]) / 512.0, 100.0 * np.mean(s[:, 13]) / max(np.mean(s[:, 12]), 1)))
print("  searches whose hint had arrived by the time they finished: %.0f per set; whose slot held another neighbour's hint: %.0f; nearest-neighbour searches completed: %.0f" % (np.mean(s[:, 26]), np.mean(s[:, 27]), np.mean(s[:, 28])))
print("  wave-level candidate trips per set: lane pass %.0f (in %.0f wave-steps, %.0f lane-candidates), shared wide scans %.0f (in %.0f scans, %.0f row passes)" % (
    np.mean(s[:, 29]), np.mean(s[:, 8]), np.mean(s[:, 14]), np.mean(s[:, 30]), np.mean(s[:, 31]), np.mean(s[:, 7])))
SEC = ["take a point", "shared wide scans (before the step)", "edge + row ranges", "scan loop", "search finished: nearest neighbour / circle box / widen", "hints out + chain of hinted triangles",
       "star finished: rows to the arena", "shared wide scans (after) + loop end"]
tot1 = 8 * np.mean(d[:, 2])
print("  phase 1 by section, summed over the 8 wavefronts (share of 8 x phase-1 cycles):")
for k, name in enumerate(SEC):
    print("    %-58s %5.1f %%" % (name, 100 * 16 * np.mean(s[:, 32 + k]) / tot1))
print("  lane-pass scan steps by the wavefront's trips (0-11 / 12-23 / 24-35 / 36+): %s;  lanes by their own trips (0 / 1-12 / 13-24 / 25+): %s" % (
    " / ".join("%.0f" % np.mean(s[:, 44 + k]) for k in range(4)), " / ".join("%.0f" % np.mean(s[:, 40 + k]) for k in range(4))))
if os.environ.get("DT_SEEDED"):
    sub = ["zero the cell index", "count", "scan", "scatter (+ position table)", "mark the stars that lost a neighbour (seed rows, pass 1)",
           "unchanged stars' bookkeeping", "seed rows pass 2 (arena copies, hints)", "order of the points to walk (until phase 1)"]
    st = np.concatenate([s[:, 48:55], s[:, 2:3]], axis=1)
    sub = sub[:st.shape[1] - 1]
    print("  between stamps 1 and 2 (cycles, median):")
    for k, nm in enumerate(sub):
        print("    %-62s %8.0f" % (nm, np.median(st[:, k + 1] - st[:, k])))
