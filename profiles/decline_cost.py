#!/usr/bin/env python3
"""What a FEW declined frames cost a batch call of the device-triangulation paths (round 6, LABNOTES 10.11).
On general-position data Qhull's replay declines 0.009-0.016 % of the frames, i.e. one to three of a 16 384-frame call — most calls have
some — and the bench's pool of synthetic frames happens to have none.  This probe puts k quarter-pixel-snapped frames (declined by
construction) into an otherwise ordinary call, at the call's start, its end or spread over it, and times the call; the deferred re-run's
own share is taken from timers around its steps.
    python profiles/decline_cost.py [frames] [features] [calls]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import packing, synth                              # noqa: E402

W = os.environ.get("WORKERS")
W = None if W in (None, "") else int(W)
if W != 0:
    packing.start_pool(W)                                                   # (forked before the GPU runtime starts)
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator            # noqa: E402


def timed(obj, name, acc):
    fn = getattr(obj, name)

    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    setattr(obj, name, wrap)


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    pool = [synth.synth_frame(i, n, base_seed=2024) for i in range(min(F, 2048))]
    f3s = [pool[i % len(pool)][0] for i in range(F)]
    base2 = [pool[i % len(pool)][1] for i in range(F)]
    places = [("none", []), ("1 first", [17]), ("1 last", [F - 9]), ("2 ends", [17, F - 9]),
              ("4 spread", [int(F * (j + 0.5) / 4) for j in range(4)]), ("16 spread", [int(F * (j + 0.5) / 16) for j in range(16)])]
    places = places[:int(os.environ.get("PLACES", "99"))]
    legs = [x for x in os.environ.get("LEGS", "exact,fixed,rescale").split(",") if x]
    if os.environ.get("EXACT_ONLY"):
        legs = ["exact"]
    early = os.environ.get("EARLY", "1") != "0"                               # EARLY=0: the merged re-run at the call's end only (before §10.11)
    for leg in legs:
        exact = leg == "exact"
        if leg == "rescale":
            from mvoscalerecovery_amd.rescale import ScaleEstimator as RescaleEstimator
            est = RescaleEstimator(1.75, window_size=5, triangulation="gpu", delaunay_workers=W, ransac_seed=2024)
        else:
            est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", delaunay_workers=W,
                                 check_triangle="reference" if exact else "fixed")
        est.GPU_REDO_EARLY = early
        acc = {}
        for nm in ("_chunk_gpu_complete_all", "_chunk_begin", "_chunk_vote", "_chunk_scale", "_chunk_gpu", "_chunk_gpu_finish", "_finish_deferred"):
            if hasattr(est, nm):
                timed(est, nm, acc)
        from mvoscalerecovery_amd import engine as _eng
        for cls_, nm in ((_eng.DeviceBatch, "triangulation_status"), (packing, "delaunay_submit")):
            fn = getattr(cls_, nm)

            def wrap(*a_, _fn=fn, _nm=nm, **k_):
                t0_ = time.perf_counter()
                try:
                    return _fn(*a_, **k_)
                finally:
                    acc[_nm] = acc.get(_nm, 0.0) + time.perf_counter() - t0_
                    acc[_nm + "_calls"] = acc.get(_nm + "_calls", 0) + 1
            setattr(cls_, nm, wrap)
        for _ in range(2):
            est.scale_calculation_batch(f3s, base2)
        ref = None
        for label, where in places:
            f2s = list(base2)
            f3c = list(f3s)
            for w in where:
                # declined by construction by BOTH device triangulations: a quarter-pixel grid and a repeated site (below the vanishing
                # row, i.e. among the triangulated ones) — since round 6's collinearity change the grid alone no longer is, for the fast kernel
                a3, a2 = f3s[w].copy(), np.ascontiguousarray(np.round(base2[w] * 4.0) / 4.0)
                low = np.nonzero(a2[:, 1] > 200.0)[0]
                a2[low[1]], a3[low[1]] = a2[low[0]], a3[low[0]]
                f2s[w], f3c[w] = a2, a3
            est.scale_calculation_batch(f3c, f2s)                            # (the re-run's context and buffers of this size exist)
            if os.environ.get("AB"):
                # AB=1: the same estimator with GPU_REDO_EARLY on and off in alternating calls — one process, one box, one pool of frames
                t_on, t_off = [], []
                for rep in range(2 * calls):
                    est.GPU_REDO_EARLY = rep % 2 == 0
                    t0 = time.perf_counter()
                    est.scale_calculation_batch(f3c, f2s)
                    (t_on if rep % 2 == 0 else t_off).append(time.perf_counter() - t0)
                est.GPU_REDO_EARLY = early
                m_on, m_off = sorted(t_on)[len(t_on) // 2], sorted(t_off)[len(t_off) // 2]
                if ref is None:
                    ref = (m_on, m_off)
                print("%-8s %-9s: started early %7.2f ms (%+6.2f), merged at the end %7.2f ms (%+6.2f); declined %3d" % (
                    leg, label, m_on * 1e3, (m_on - ref[0]) * 1e3, m_off * 1e3, (m_off - ref[1]) * 1e3,
                    getattr(est, "declined_total", getattr(est, "last_declined", 0))), flush=True)
                continue
            ts, parts = [], []
            for _ in range(calls):
                acc.clear()
                ctxs = [est.engine.ctx if hasattr(est, "engine") else est.ctx] + ([est._engine2.ctx] if getattr(est, "_engine2", None) is not None else [])
                a0 = [c.alloc_stats() for c in ctxs]
                t0 = time.perf_counter()
                est.scale_calculation_batch(f3c, f2s)
                ts.append(time.perf_counter() - t0)
                a1 = [c.alloc_stats() for c in ctxs]
                acc["mallocs"] = sum(b["hip_malloc"] - a["hip_malloc"] for a, b in zip(a0, a1))
                acc["frees"] = sum(b["hip_free"] - a["hip_free"] for a, b in zip(a0, a1))
                parts.append(dict(acc))
            if os.environ.get("SHOW_CALLS"):
                print("   calls (ms): " + " ".join("%.1f" % (t * 1e3) for t in ts))
            k = int(np.argsort(ts)[len(ts) // 2])
            med = ts[k]
            if ref is None:
                ref = med
            p = parts[k]
            print("[early started %d launched %d status hits %d] " % (getattr(est, "redo_early_started", 0), getattr(est, "redo_early_launched", 0),
                                                                         getattr(est, "redo_early_status_hits", 0)), end="")
            est.redo_early_started = est.redo_early_launched = est.redo_early_status_hits = 0
            print("%-9s %-9s: %7.2f ms per call (%+6.2f), %6.1f k frames/s, declined %3d; re-run %5.2f ms = begin %5.2f + vote %5.2f + scale %5.2f + rest %5.2f; launch side %6.2f, collect side %6.2f ms; hipMalloc %d hipFree %d; status waits %6.2f ms in %d, submits %5.2f ms in %d" % (
                leg, label, med * 1e3, (med - ref) * 1e3, F / med / 1e3, getattr(est, "declined_total", getattr(est, "last_declined", 0)),
                (p.get("_chunk_gpu_complete_all", 0.0) + p.get("_finish_deferred", 0.0)) * 1e3, p.get("_chunk_begin", 0.0) * 1e3, p.get("_chunk_vote", 0.0) * 1e3, p.get("_chunk_scale", 0.0) * 1e3,
                (p.get("_chunk_gpu_complete_all", 0.0) - p.get("_chunk_begin", 0.0) - p.get("_chunk_vote", 0.0) - p.get("_chunk_scale", 0.0)) * 1e3,
                p.get("_chunk_gpu", 0.0) * 1e3, p.get("_chunk_gpu_finish", 0.0) * 1e3, p.get("mallocs", 0), p.get("frees", 0),
                p.get("triangulation_status", 0.0) * 1e3, p.get("triangulation_status_calls", 0), p.get("delaunay_submit", 0.0) * 1e3, p.get("delaunay_submit_calls", 0)), flush=True)


if __name__ == "__main__":
    main()
