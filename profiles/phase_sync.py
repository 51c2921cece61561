#!/usr/bin/env python3
"""Diagnostic (round 6, LABNOTES 10.10): do the workgroups of one scale_frames_kernel launch march through their phases in step?
Equal-sized frames, one frame per workgroup, three workgroups per CU: if the 768 resident workgroups all stream their features at the
same time and all sweep at the same time, the phases' times ADD (which is what the ablation of LABNOTES 7 shows) although HBM, LDS and
VALU could overlap.  s_memtime is one chip-wide 100 MHz counter, so the stamps of a -DMVOSR_STAMPS build place every workgroup's phases
on one time axis.
    bash profiles/ab_build.sh stamps "-DMVOSR_STAMPS -DMVOSR_ABLATE" [-DMVOSR_STAGGER=2000]
    MVOSR_DEBUG_SKIP=16 MVOSR_LIB_PATH=profiles/ab/libmvosr_stamps.so python profiles/phase_sync.py [frames] [features]
Prints, per phase, the mean number of resident workgroups in it and the standard deviation over 1 us bins against the binomial value
(independent workgroups), the same per CU for the load phase, and the kernel's span."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, packing, synth
from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = _lib.default_context(0)
eng = ScaleEngine(1.75, ctx=ctx)
pool = 64
frames = [synth.synth_frame(i, N, base_seed=2024) for i in range(pool)]
pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames])
packing.attach_tri1(pf)
db = DeviceBatch(ctx, pf, with_tri2=False)
out = DeviceOutputs(ctx, db, counts=True, stage=True)
eng.outlier_vote_batch(db, out); ctx.sync()
c = out.get("vote_counters")
masks = [c[pf.frame_slice(f)] >= 0 for f in range(pool)]
packing.attach_tri2(pf, None, masks)
pf = packing.tile_frames(pf, F // pool)
print("tiled", flush=True)
db = DeviceBatch(ctx, pf)
out = DeviceOutputs(ctx, db, counts=True, hist=True)
for _ in range(3):
    eng.scale_batch(db, out)
ctx.sync()
print("ran", flush=True)
raw = out.get("hist").reshape(pf.n_frames, -1).view(np.uint64)
st = raw[:, :12].copy()
hw = st[:, 10]
cu = ((hw >> 8) & 0xF).astype(np.int64)
se = ((hw >> 13) & 0x7).astype(np.int64)
sh = ((hw >> 12) & 0x1).astype(np.int64)
xcc = ((hw >> 32) & 0xF).astype(np.int64)
cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t = st[:, [11, 0, 1, 2, 3, 4, 5, 9]].astype(np.int64)      # 11: first instruction of the workgroup; 0: its frame's counts and offsets known
ok = (t[:, 0] > 0) & (t[:, -1] > t[:, 0])
t = t[ok]; cukey = cukey[ok]
# s_memtime counts core-rate ticks (~0.5 ns here: a workgroup's life is ~50 k ticks) and the stamps of different CUs do not share an
# origin (normalising per XCD still leaves a "span" of hundreds of lives): every CU's stamps are moved so that its first workgroup starts
# at 0 — the launch's first 768 workgroups are handed out within about a microsecond, 3 % of a life
xcc_of = (cukey // (8 * 2 * 16))
for x in np.unique(cukey):
    m = cukey == x
    t[m] -= t[m, 0].min()
span = t[:, -1].max()
life_ticks = float((t[:, -1] - t[:, 0]).mean())
names = ["counts", "load", "vote", "compaction", "sweep 1", "sweep 2", "tail"]
print("%d frames of %d features: span %.0f ticks = %.1f workgroup lives, %d distinct CUs seen on %d XCDs, workgroup life mean %.0f ticks" %
      (len(t), N, span, span / life_ticks, len(np.unique(cukey)), len(np.unique(xcc_of)), life_ticks))
BIN = max(int(life_ticks / 25), 1)          # 1/25 of a workgroup's life
nb = int(span // BIN) + 1
if nb > 5_000_000:
    sys.exit('the stamps do not share a time axis (span %d bins): XCC ids %s' % (nb, np.unique(xcc_of)))
lo, hi = int(nb * 0.1), int(nb * 0.9)


def occupancy(a, b, sel=None):
    """number of workgroups with a <= bin*BIN < b, per bin (difference array)"""
    if sel is not None:
        a, b = a[sel], b[sel]
    d = np.zeros(nb + 2, np.int64)
    np.add.at(d, np.minimum((a + BIN - 1) // BIN, nb + 1), 1)
    np.add.at(d, np.minimum((b + BIN - 1) // BIN, nb + 1), -1)
    return np.cumsum(d)[:nb]


live = occupancy(t[:, 0], t[:, -1])
print("resident workgroups (middle 80 %% of the span): mean %.1f" % live[lo:hi].mean())
for i, nme in enumerate(names):
    occ = occupancy(t[:, i], t[:, i + 1])[lo:hi]
    m = occ.mean(); p = m / max(live[lo:hi].mean(), 1)
    binom = np.sqrt(max(live[lo:hi].mean() * p * (1 - p), 1e-9))
    print("%-11s in phase: mean %7.1f workgroups (%.3f of the resident), std over bins of 1/25 life %6.1f; independent workgroups would give %5.1f  -> %.1f x" %
          (nme, m, p, occ.std(), binom, occ.std() / binom))
# the chip's load demand over time: autocorrelation of the load-phase count (a peak at one workgroup life = marching in step)
occ = occupancy(t[:, 1], t[:, 2])[lo:hi].astype(np.float64)
occ -= occ.mean()
ac = np.correlate(occ, occ, "full")[len(occ) - 1:]
ac /= ac[0]
life = int(round((t[:, -1] - t[:, 0]).mean() / BIN))
print("autocorrelation of the load-phase count at lags 1/4, 1/2, 3/4, 1, 2 workgroup lives (%d bins): %s" %
      (life, " ".join("%.2f" % ac[min(int(round(life * k)), len(ac) - 1)] for k in (0.25, 0.5, 0.75, 1, 2))))
# per CU: how often are two or three of a CU's workgroups in the load phase together?
keys = np.unique(cukey)
both = np.zeros(4)
for k in keys[:64]:
    sel = cukey == k
    o = occupancy(t[:, 1], t[:, 2], sel)[lo:hi]
    for j in range(4):
        both[j] += (o == j).sum() if j < 3 else (o >= 3).sum()
both /= both.sum()
p1 = (both * np.arange(4)).sum() / 3.0
exp = [(1 - p1) ** 3, 3 * p1 * (1 - p1) ** 2, 3 * p1 * p1 * (1 - p1), p1 ** 3]
print("per CU (64 CUs sampled), share of time with 0 / 1 / 2 / 3 workgroups loading: %s; independent: %s" %
      (" ".join("%.3f" % v for v in both), " ".join("%.3f" % v for v in exp)))
# the first generation
first = np.sort(t[:, 0])[:768]
print("start of the first 768 workgroups: within %.2f lives; their ends within %.2f lives" %
      ((first[-1] - first[0]) / life_ticks, (np.sort(t[:, -1])[767] - np.sort(t[:, -1])[0]) / life_ticks))
# per generation: spread of the starts of workgroups k*768 .. (k+1)*768 in start order
ss = np.sort(t[:, 0])
for g in (0, 1, 2, 5, 10, 20):
    seg = ss[g * 768:(g + 1) * 768]
    if len(seg) == 768:
        print("generation %2d: starts spread over %.2f lives (10th-90th percentile %.2f)" %
              (g, (seg[-1] - seg[0]) / life_ticks, (np.percentile(seg, 90) - np.percentile(seg, 10)) / life_ticks))

# the slots' idle time: per CU, the k-th workgroup to start takes the slot of the (k-3)-th to end
gaps = []
res = []
for k in keys:
    sel = cukey == k
    a = np.sort(t[sel, 0]); b = np.sort(t[sel, -1])
    if len(a) > 6:
        gaps.append(a[3:] - b[:len(a) - 3])
    res.append(occupancy(t[:, 0], t[:, -1], sel)[lo:hi].mean())
gaps = np.concatenate(gaps)
print("resident workgroups per CU (middle 80 %%): mean %.2f of 3;  gap between a workgroup's last stamp and the first instruction of the next one in "
      "its slot: median %.0f ticks, mean %.0f, 90th percentile %.0f = %.3f / %.3f / %.3f of a life (%d ticks)" %
      (np.mean(res), np.median(gaps), gaps.mean(), np.percentile(gaps, 90), np.median(gaps) / life_ticks, gaps.mean() / life_ticks,
       np.percentile(gaps, 90) / life_ticks, life_ticks))
# balance: frames per CU and per XCD, and when each CU / XCD ends
per_cu = np.array([(cukey == k).sum() for k in keys])
end_cu = np.array([t[cukey == k, -1].max() for k in keys]) / life_ticks
print("frames per CU: min %d mean %.1f max %d; a CU's last end: min %.2f mean %.2f max %.2f lives" %
      (per_cu.min(), per_cu.mean(), per_cu.max(), end_cu.min(), end_cu.mean(), end_cu.max()))
for x in np.unique(xcc_of):
    m = xcc_of == x
    ks = np.unique(cukey[m])
    print("  XCD %d: %d CUs, %d frames, ends %.2f .. %.2f lives, mean life %.0f ticks" %
          (x, len(ks), m.sum(), min(t[cukey == k, -1].max() for k in ks) / life_ticks, t[m, -1].max() / life_ticks, (t[m, -1] - t[m, 0]).mean()))
