"""Throughput of the `rescale` variant's GPU stages (graph vote, flat
selection, seeded RANSAC plane) on synthetic 2000-feature frames resident in HBM.

    python profiles/bench_rescale.py [--frames 4096] [--pool 64] [--features 2000] [--steps 5]

A pool of frames goes once through the drop-in ``rescale.ScaleEstimator`` (host Delaunay, vote mask,
kept-vertex lists, sample triples), its device inputs are tiled to --frames, and every stage kernel
is timed with HIP events on the launch stream.  Prints one JSON line (not the driver's bench line).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mvoscalerecovery_amd import _lib, packing, synth                     # noqa: E402
from mvoscalerecovery_amd.engine import DeviceBatch                       # noqa: E402
from mvoscalerecovery_amd import rescale                                   # noqa: E402


def timed(ctx, fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    ctx.sync()
    a, b = ctx.event(), ctx.event()
    ctx.record(a)
    for _ in range(steps):
        fn()
    ctx.record(b)
    ctx.sync()
    return ctx.elapsed_ms(a, b) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--pool", type=int, default=64)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--hyp-sweep", action="store_true", help="diagnostic: RANSAC kernel time against the number of hypotheses")
    ap.add_argument("--cpu-sample", type=int, default=8, help="frames of the pool timed through the NumPy oracle")
    args = ap.parse_args()

    est = rescale.ScaleEstimator(1.75, window_size=5, ransac_seed=7, delaunay_workers=0, triangulation="scipy")
    ctx, lib = est.ctx, est.ctx.lib
    frames = [synth.synth_frame(i, args.features, base_seed=4242) for i in range(args.pool)]
    t0 = time.perf_counter()
    sel = est.feature_selection_batch([f[0] for f in frames], [f[1] for f in frames])
    host_s = time.perf_counter() - t0
    pts = [np.ascontiguousarray(s[0]) for s in sel]
    triples = [est._triples(p.shape[0]) for p in pts]
    H = triples[0].shape[0]
    repeats = max(1, args.frames // args.pool)
    F = args.pool * repeats

    # stage 1 inputs: lower-half features + first triangulation
    pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames], est.vanish)
    packing.attach_tri1(pf, None, 0)
    pf_t = packing.tile_frames(pf, repeats)
    db1 = DeviceBatch(ctx, pf_t, with_tri2=False)
    total = ctx.zeros(pf_t.total_padded, np.int32)
    good = ctx.zeros(pf_t.total_padded, np.int32)
    st1 = ctx.zeros(F, np.int32)
    b1 = db1.struct()

    # stage 2 inputs: survivors + second triangulation (what feature_selection_batch built)
    pf2_t = packing.tile_frames(est.last["pf2"], repeats)
    db2 = DeviceBatch(ctx, pf2_t, with_tri2=True)
    nt = int(pf2_t.tri2_off[-1])
    tri_h = ctx.zeros(nt, np.float64)
    tri_f = ctx.zeros(nt, np.uint8)
    level = ctx.zeros(F, np.float64)
    nkept = ctx.zeros(F, np.int32)
    st2 = ctx.zeros(F, np.int32)
    max_tri = int(np.max(np.diff(pf2_t.tri2_off)))
    b2 = db2.struct()

    # stage 3 inputs: the kept-vertex lists and the sample triples
    cnt = np.tile(np.array([p.shape[0] for p in pts], dtype=np.int32), repeats)
    off = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64)
    allp = np.tile(np.concatenate(pts, axis=0), (repeats, 1))
    d = [ctx.to_device(np.ascontiguousarray(allp[:, i]), np.float64) for i in range(3)]
    d_off, d_cnt = ctx.to_device(off, np.int64), ctx.to_device(cnt, np.int32)
    d_tri = ctx.to_device(np.tile(np.stack(triples), (repeats, 1, 1)), np.int32)
    model = ctx.zeros((F, 4), np.float64)
    best = ctx.zeros(F, np.int32)
    used = ctx.zeros(F, np.int32)

    def graph():
        _lib.check(lib.mvosr_graph_inliers_batch(ctx.handle, C.byref(b1), C.c_uint32(est._good_bits), total.ptr, good.ptr, st1.ptr),
                   "graph")

    def flat():
        _lib.check(lib.mvosr_flat_selection_batch(ctx.handle, C.byref(b2), -80.0, -85.0, 0.9, tri_h.ptr, tri_f.ptr, level.ptr, nkept.ptr,
                                                  st2.ptr, max_tri), "flat")

    def ransac():
        _lib.check(lib.mvosr_ransac_plane_batch(ctx.handle, F, d_off.ptr, d_cnt.ptr, d[0].ptr, d[1].ptr, d[2].ptr, d_tri.ptr, H,
                                                rescale.RANSAC_THRESHOLD, rescale.RANSAC_GOAL, None, model.ptr, best.ptr, used.ptr),
                   "ransac")

    if args.hyp_sweep:
        for h2 in (1, 8, 25, 50, 100):
            if h2 > H:
                break
            d_tri2 = ctx.to_device(np.tile(np.stack([t[:h2] for t in triples]), (repeats, 1, 1)), np.int32)

            def ransac_h():
                _lib.check(lib.mvosr_ransac_plane_batch(ctx.handle, F, d_off.ptr, d_cnt.ptr, d[0].ptr, d[1].ptr, d[2].ptr, d_tri2.ptr, h2,
                                                        rescale.RANSAC_THRESHOLD, 2.0, None, model.ptr, best.ptr, used.ptr), "ransac")
            print("ransac_plane_kernel, %3d hypotheses: %.4f ms per %d frames" % (h2, timed(ctx, ransac_h, args.steps), F), file=sys.stderr)

    # row a12 (triangle_batch.py): same frames read as [u, v, depth] + the first triangulation
    tb_h = ctx.zeros(F, np.float64)
    tb_c = ctx.zeros((F, 2), np.int32)
    tb_s = ctx.zeros(F, np.int32)

    def tribatch():
        _lib.check(lib.mvosr_triangle_batch(ctx.handle, C.byref(b1), 718.856, 607.1928, 185.2157, 0.98, 3.0, tb_h.ptr, tb_c.ptr, tb_s.ptr),
                   "triangle_batch")

    ms = {"graph_inliers": timed(ctx, graph, args.steps), "flat_selection": timed(ctx, flat, args.steps),
          "ransac_plane": timed(ctx, ransac, args.steps)}
    total_ms = sum(ms.values())
    ms["triangle_batch"] = timed(ctx, tribatch, args.steps)
    n_mean = float(pf.feat_cnt.mean())
    t1 = float(pf.tri1_off[-1]) / args.pool
    t2 = float(est.last["pf2"].tri2_off[-1]) / args.pool
    m_mean = float(np.mean([p.shape[0] for p in pts]))
    # algorithmic bytes per frame: stage inputs once + stage outputs
    by = {"graph_inliers": n_mean * 16 + t1 * 12 + n_mean * 8,
          "flat_selection": float(est.last["pf2"].feat_cnt.mean()) * 24 + t2 * 12 + t2 * 9,
          "ransac_plane": m_mean * 24 + H * 12 + 48,
          "triangle_batch": n_mean * 24 + t1 * 12 + 20}

    # the same pool through the NumPy oracle (one core), triangulations supplied by the oracle itself
    cpu = None
    try:
        from oracle import rescale_oracle as ro
        k = min(args.cpu_sample, args.pool)
        oe = ro.OracleRescaleEstimator(1.75, window_size=5, sampler=lambda n: triples[0][:, :] % n)
        t0 = time.perf_counter()
        for i in range(k):
            oe.scale_calculation(frames[i][0], frames[i][1])
        cpu = {"value": k / (time.perf_counter() - t0), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": "%d frames of the pool through oracle/rescale_oracle.py incl. its two Delaunay calls" % k}
    except Exception as exc:                                  # the oracle is optional here
        cpu = {"error": repr(exc)}

    line = {
        "metric": "frames/sec, rescale-variant GPU stages (graph vote + flat selection + 100-hypothesis RANSAC plane); triangle_batch timed beside",
        "value": F / (total_ms * 1e-3), "unit": "frames/s", "frames": F, "pool": args.pool,
        "features_per_frame": n_mean, "tri1_per_frame": t1, "tri2_per_frame": t2, "ransac_points_per_frame": m_mean,
        "hypotheses": H,
        "kernel_ms": ms,
        "frames_per_s_per_kernel": {k: F / (v * 1e-3) for k, v in ms.items()},
        "algorithmic_GBps_per_kernel": {k: by[k] * F / (ms[k] * 1e-3) / 1e9 for k in ms},
        "frac_of_hbm_peak_per_kernel": {k: by[k] * F / (ms[k] * 1e-3) / 1e9 / 8000.0 for k in ms},
        "host_side_s_per_frame_pool_pass": host_s / args.pool,
        "cpu_baseline": cpu,
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
