"""TEST INFRASTRUCTURE — CPU oracle for the legacy per-triangle batch of
/root/reference/src/triangle_batch.py:14-68.  NOT part of the product.

Pinned against what the reference script itself prints when executed on synthetic dumps
(tests/golden/make_golden.py: make_triangle_batch -> tests/golden/triangle_batch.json)."""
import numpy as np

FOCUS, CX, CY = 718.856, 607.1928, 182.2157        # triangle_batch.py:20-22 (cy as the script has it)
S_MIN = 0.98                                        # :54
N_SIGMA = 3                                         # :60-61


def camera_height(points3d, tri=None, focus=FOCUS, cx=CX, cy=CY):
    """points3d (N,3) = [u, v, depth] (:19).  Returns (final_mean, n_kept, n_clipped)."""
    from scipy.spatial import Delaunay
    points3d = np.asarray(points3d, dtype=np.float64)
    if tri is None:
        tri = Delaunay(points3d[:, 0:2]).simplices                                   # :23-25
    a = points3d[tri].copy()                                                         # (T,3,3) rows = vertices (:30-31)
    a[:, :, 0] = a[:, :, 2] * (a[:, :, 0] - cx) / focus                              # :32
    a[:, :, 1] = a[:, :, 2] * (a[:, :, 1] - cy) / focus                              # :33
    norm = (np.linalg.inv(a) @ np.ones((3, 1), float)).reshape(-1, 3)                # :36-37
    s = norm[:, 1] / np.sqrt(np.sum(norm * norm, 1))                                 # :38-39,:43
    height = np.mean(a[:, :, 1], 1)                                                  # :40
    keep = (s > S_MIN)
    h = height[keep]
    h = h[h > 0]                                                                     # :54-55
    mean, std = np.mean(h), np.std(h)                                                # :57-58
    h2 = h[h > mean - N_SIGMA * std]
    h2 = h2[h2 < mean + N_SIGMA * std]                                               # :60-61
    return float(np.mean(h2)), int(h.shape[0]), int(h2.shape[0])                     # :62
