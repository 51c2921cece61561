"""GPU mirror of the legacy per-triangle batch /root/reference/src/triangle_batch.py:14-68
(SURVEY.md §8 row a12): per frame of ``[u, v, depth]`` features, the camera height as the
3-sigma-clipped mean height of the flat, below-camera Delaunay triangles.

The reference is a Python-2 script that loops over text dumps and prints one number per frame;
``camera_heights`` is the same computation as a function over a list of frames (one launch)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, packing
from .engine import DeviceBatch

FOCUS, CX, CY = 718.856, 607.1928, 182.2157        # triangle_batch.py:20-22 (cy exactly as the script has it)
S_MIN = 0.98                                        # :54
N_SIGMA = 3.0                                       # :60-61


def camera_heights(points3d_list, tris=None, focus=FOCUS, cx=CX, cy=CY, device=0, delaunay_workers=None):
    """``points3d_list``: list of (N,3) arrays ``[u, v, depth]`` (:19).  ``tris``: optional
    precomputed ``Delaunay(points[:, :2]).simplices`` per frame (:23-25).  Returns
    ``(heights[F], counts[F,2], status[F])``; a frame whose status is MVOSR_ST_ERR_SINGULAR is one
    where the reference raises ``LinAlgError`` (:36)."""
    ctx = _lib.default_context(device)
    f3 = [np.stack([p[:, 0], np.zeros(len(p)), p[:, 2]], axis=1) for p in map(np.asarray, points3d_list)]   # x=u, z=depth
    f2 = [np.asarray(p)[:, 0:2] for p in points3d_list]
    pf = packing.pack_features(f3, f2, -np.inf)
    packing.attach_tri1(pf, tris, delaunay_workers)
    for f, err in sorted(pf.extra["tri1_errors"].items()):
        raise err
    db = DeviceBatch(ctx, pf, with_tri2=False)
    F = pf.n_frames
    height, counts, status = ctx.zeros(F, np.float64), ctx.zeros((F, 2), np.int32), ctx.zeros(F, np.int32)
    b = db.struct()
    _lib.check(ctx.lib.mvosr_triangle_batch(ctx.handle, C.byref(b), focus, cx, cy, S_MIN, N_SIGMA, height.ptr, counts.ptr,
                                            status.ptr), "mvosr_triangle_batch")
    ctx.sync()
    out = height.download(), counts.download(), status.download()
    for buf in (height, counts, status):
        buf.free()
    db.free()
    return out
