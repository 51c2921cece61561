"""Mirror of the road-normal helpers /root/reference/src/estimate_road_norm.py.

The point-cloud helpers run on the GPU: ``get_pitch_ransac`` (:66-70) and ``get_pitch_line_ransac`` (:60-64) through
the RANSAC kernel (samples drawn on the host, seedable — the reference re-seeds from OS entropy on every call),
``get_inliers`` (:71-78) through the inlier-mask kernel.  The motion helpers (``get_norm_svd`` :20-26,
``get_pitch_svd`` :28-37, ``get_pitch`` :52-58) take a handful of 3-vectors (the camera translations
of a few frames) and stay NumPy one-liners, as in the reference: there is nothing to accelerate.  Return types are the
reference's (``get_norm_svd`` hands back a 1x3 ``np.matrix``)."""
from __future__ import annotations

import ctypes as C
import math
import random

import numpy as np

from . import _lib


def get_pitch_ransac(road_points, max_iterations, threshold, seed=None, triples=None, device=0):
    """estimate_road_norm.py:66-70: returns ``(model, best_inlier_count)``; model = unit (a,b,c,d)
    with b >= 0 (the reference's SVD null vector has an arbitrary sign)."""
    ctx = _lib.default_context(device)
    pts = np.ascontiguousarray(np.asarray(road_points, dtype=np.float64))
    m = pts.shape[0]
    if triples is None:
        rng = random.Random(seed)
        triples = np.array([rng.sample(range(m), 3) for _ in range(int(max_iterations))], dtype=np.int32)
    triples = np.ascontiguousarray(np.asarray(triples, dtype=np.int32).reshape(-1, 3))
    d = [ctx.to_device(np.ascontiguousarray(pts[:, i])) for i in range(3)]
    off, cnt = ctx.to_device(np.zeros(1, np.int64)), ctx.to_device(np.array([m], np.int32))
    tri = ctx.to_device(triples)
    model, best, used = ctx.zeros((1, 4), np.float64), ctx.zeros(1, np.int32), ctx.zeros(1, np.int32)
    _lib.check(ctx.lib.mvosr_ransac_plane_batch(ctx.handle, 1, off.ptr, cnt.ptr, d[0].ptr, d[1].ptr, d[2].ptr, tri.ptr,
                                                triples.shape[0], float(threshold), 0.8, None, model.ptr, best.ptr, used.ptr),
               "mvosr_ransac_plane_batch")
    ctx.sync()
    out = model.download()[0], int(best.download()[0])
    for buf in d + [off, cnt, tri, model, best, used]:
        buf.free()
    return out


def get_pitch_line_ransac(road_points, max_iterations, threshold, seed=None, pairs=None, device=0):
    """estimate_road_norm.py:60-64: 2-D line fit, ``(model, best_inlier_count)``; model = unit (a,b,c) of
    a x + b y + c = 0 with b >= 0 (the reference's SVD null vector has an arbitrary sign)."""
    ctx = _lib.default_context(device)
    pts = np.ascontiguousarray(np.asarray(road_points, dtype=np.float64))
    m = pts.shape[0]
    if pairs is None:
        rng = random.Random(seed)
        pairs = np.array([rng.sample(range(m), 2) for _ in range(int(max_iterations))], dtype=np.int32)
    pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    samples = np.ascontiguousarray(np.concatenate([pairs, np.zeros((pairs.shape[0], 1), np.int32)], axis=1))
    d = [ctx.to_device(np.ascontiguousarray(pts[:, i])) for i in range(2)]
    off, cnt = ctx.to_device(np.zeros(1, np.int64)), ctx.to_device(np.array([m], np.int32))
    smp = ctx.to_device(samples)
    model, best, used = ctx.zeros((1, 4), np.float64), ctx.zeros(1, np.int32), ctx.zeros(1, np.int32)
    _lib.check(ctx.lib.mvosr_ransac_line_batch(ctx.handle, 1, off.ptr, cnt.ptr, d[0].ptr, d[1].ptr, smp.ptr,
                                               samples.shape[0], float(threshold), 0.8, None, model.ptr, best.ptr, used.ptr),
               "mvosr_ransac_line_batch")
    ctx.sync()
    mm = model.download()[0]
    out = np.array([mm[0], mm[1], mm[3]]), int(best.download()[0])
    for buf in d + [off, cnt, smp, model, best, used]:
        buf.free()
    return out


def get_inliers(parameter, data, threshold, device=0):
    """estimate_road_norm.py:71-78: boolean mask |n.p + d| < threshold."""
    ctx = _lib.default_context(device)
    pts = np.ascontiguousarray(np.asarray(data, dtype=np.float64))
    par = np.ascontiguousarray(np.asarray(parameter, dtype=np.float64).reshape(-1)[:4])
    d = [ctx.to_device(np.ascontiguousarray(pts[:, i])) for i in range(3)]
    mask = ctx.zeros(pts.shape[0], np.uint8)
    _lib.check(ctx.lib.mvosr_plane_inliers(ctx.handle, pts.shape[0], d[0].ptr, d[1].ptr, d[2].ptr, _lib.addr(par),
                                           float(threshold), mask.ptr), "mvosr_plane_inliers")
    ctx.sync()
    out = mask.download().astype(bool)
    for buf in d + [mask]:
        buf.free()
    return out


def get_norm_svd(camera_motion_ts):
    """:20-26: third left singular vector of the stacked translations, sign n_y >= 0, as a 1x3 ``np.matrix``."""
    u, s, v = np.linalg.svd(np.asarray(camera_motion_ts).T, full_matrices=True)
    n = u[:, 2]
    if n[1] < 0:
        n = -n
    return np.matrix(n)


def get_pitch_svd(camera_motion_ts):
    """:28-37: asin(n_y / |n|^2) of that vector."""
    n = np.asarray(get_norm_svd(camera_motion_ts)).reshape(-1)
    return math.asin(n[1] / float(n @ n))


def get_pitch(camera_motion_ts):
    """:52-58: asin(-sum t_y / |sum t|^2)."""
    m = np.sum(np.asarray(camera_motion_ts), 0)
    return math.asin(-m[1] / float(m @ m))
