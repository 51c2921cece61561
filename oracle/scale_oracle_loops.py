"""TEST INFRASTRUCTURE — the loop-faithful flavour of the CPU oracle.  NOT part of the product.

``oracle/scale_oracle.py`` restates the reference's stages with vectorised NumPy (fast enough to check
thousands of frames).  This module restates the two stages that dominate the reference's run time the way
the reference itself is written — one Python iteration per triangle:

  * ``find_outliers`` + ``check_triangle``  /root/reference/src/scale_calculator.py:151-167, :105-119
  * ``feature_selection_by_tri``            /root/reference/src/scale_calculator.py:225-248 (two list
    comprehensions over the triangles, one ``np.matrix(...).I`` per triangle, :228-229)

so that ``bench.py`` can time a "reference-shaped" CPU baseline on the GPU box, where /root/reference does not
exist.  Everything else (remap, road model, window median) is shared with
``scale_oracle``: those stages are already NumPy calls in the reference.  tests/test_oracle_golden.py checks
this flavour against the same golden vectors of the reference as the vectorised one.
"""
from __future__ import annotations

import warnings

import numpy as np

from . import scale_oracle as so

warnings.filterwarnings("ignore", category=PendingDeprecationWarning)      # np.matrix, which the reference uses (:228-229)


def check_triangle(v, d, mode="reference"):
    """scale_calculator.py:105-119 (the (0,2) pair marks vertices 0 and 1, as the reference does).
    ``mode="fixed"``: the (0,2) pair marks vertices 0 and 2 — the declared deviation of SURVEY.md §8 f1."""
    flag = [False, False, False]
    a = (v[0] - v[1]) * (d[0] - d[1])
    b = (v[0] - v[2]) * (d[0] - d[2])
    c = (v[1] - v[2]) * (d[1] - d[2])
    if a > 0:
        flag[0] = True
        flag[1] = True
    if b > 0:
        flag[0] = True
        if mode == "fixed":
            flag[2] = True
        else:
            flag[1] = True
    if c > 0:
        flag[1] = True
        flag[2] = True
    return np.array(flag)


def find_outliers(feature3d, feature2d, triangle_ids, mode="reference"):
    """scale_calculator.py:151-167: per-feature counters (start 1), one Python iteration per triangle."""
    outliers = np.ones((feature3d.shape[0]))
    for triangle_id in triangle_ids:
        depths = feature3d[triangle_id, 2]
        pixel_vs = feature2d[triangle_id, 1]
        flag = check_triangle(pixel_vs, depths, mode)
        outlier = triangle_id[flag]
        inlier = triangle_id[~flag]
        outliers[outlier] -= np.ones(outliers[outlier].shape[0])
        outliers[inlier] += np.ones(outliers[inlier].shape[0])
    return outliers


def feature_selection_by_tri(feature3d, triangle_ids):
    """scale_calculator.py:225-248: returns (selected ids, height_level); raises LinAlgError like :229."""
    b_matrix = np.ones((3, 1), float)
    triangles = np.array([np.matrix(feature3d[triangle_id]) for triangle_id in triangle_ids])          # :228
    triangles_i = np.array([np.matrix(feature3d[triangle_id]).I for triangle_id in triangle_ids])      # :229
    normals = (triangles_i @ b_matrix).reshape(-1, 3)
    with np.errstate(all="ignore"):
        normals_len = np.sqrt(np.sum(normals * normals, 1)).reshape(-1, 1)
        normals = normals / normals_len
        pitch_deg = np.arcsin(-normals[:, 1]) * 180 / np.pi
        valid_pitch_id = pitch_deg < -80
        heights = np.mean(triangles[:, :, 1], 1)
        unvalid_pitch_id = pitch_deg >= -80
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            height_level = np.mean(heights[unvalid_pitch_id])
        valid_height_id = heights > height_level
    valid_id = valid_pitch_id & valid_height_id
    return np.unique(triangle_ids[valid_id].reshape(-1)), float(height_level)


def frame_raw_scale(feature3d, feature2d, absolute_reference, tri1=None, tri2=None,
                    camera_pitch=so.CAMERA_PITCH, vanish=so.VANISH, check_triangle_mode="reference"):
    """One frame up to the window filter (scale_calculator.py:411-422), reference-shaped loops for the two hot
    stages.  Returns ``(raw_scale, status, height_level, counters, selected_ids)``.  ``check_triangle_mode="fixed"``:
    the order-invariant vote on canonical rows (``scale_oracle.canonical_rows``)."""
    fixed = check_triangle_mode == "fixed"
    f3 = so.remap(np.asarray(feature3d, dtype=np.float64), camera_pitch)
    f2 = np.asarray(feature2d, dtype=np.float64)
    low = so.lower_mask(f2, vanish)
    f3l, f2l = f3[low], f2[low]
    if tri1 is None:
        tri1 = so.delaunay(f2l)
    if fixed:
        tri1 = so.canonical_rows(tri1)
    counters = find_outliers(f3l, f2l, np.asarray(tri1), check_triangle_mode)
    valid = counters >= 0
    if not valid.shape[0] > 3:
        return np.nan, so.ST_TOO_FEW, np.nan, counters, None
    f3v, f2v = f3l[valid], f2l[valid]
    if tri2 is None:
        tri2 = so.delaunay(f2v)
    if fixed:
        tri2 = so.canonical_rows(tri2)
    try:
        selected, height_level = feature_selection_by_tri(f3v, np.asarray(tri2))
    except np.linalg.LinAlgError:
        return np.nan, so.ST_ERR_SINGULAR, np.nan, counters, None
    if selected.shape[0] == 0:
        with np.errstate(all="ignore"):
            return float(np.float64(absolute_reference) / np.float64(height_level)), so.ST_NO_FLAT, height_level, counters, selected
    road = so.road_model(f3v[selected][:, 1], height_level)
    if road.status in (so.ST_ERR_LEFT, so.ST_ERR_RIGHT):
        return np.nan, road.status, height_level, counters, selected
    with np.errstate(all="ignore"):
        return float(np.float64(absolute_reference) / np.float64(road.height)), road.status, height_level, counters, selected
