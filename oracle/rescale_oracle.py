"""TEST INFRASTRUCTURE — CPU oracle for the `rescale` variant of the scale estimator
(SURVEY.md §8 rows f2, f4, a11, a13).  NOT part of the product.

NumPy restatement of /root/reference/src/rescale.py:22-193 (the estimator that
/root/reference/src/main.py:20 actually imports), /root/reference/src/graph.py:5-36,124-145
(GraphChecker), the plane helpers of /root/reference/src/estimate_road_norm.py:8-18,66-70 and the
RANSAC loop /root/reference/src/thirdparty/Ransac/ransac.py:3-23.

The reference's RANSAC re-seeds Python's `random` from OS entropy on every call (ransac.py:6), so
its output is not reproducible as shipped.  It IS a deterministic function of the sequence of
sample triples: this oracle takes the triples as an argument, and the golden vectors
(tests/golden/make_golden.py: make_rescale) are produced by running the reference's own
`rescale.ScaleEstimator` with `random.sample` patched to replay a recorded sequence of index
triples.  Pinning is therefore "exact given the sample sequence"; without a shared sequence only
statistical agreement can be claimed.
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass

import numpy as np

VANISH = 185                                     # rescale.py:30
EDGE_POTENTIAL = [[3, 1], [2, 2], [2, 2], [0, 4]]   # rescale.py:32
PROB_THRESHOLD = 0.6                             # graph.py:132
VOTE_THRESHOLD = 0.5                             # graph.py:35
MIN_VALID_FOR_RETRI = 10                         # rescale.py:133
PITCH_LOOSE_DEG = -80                            # rescale.py:85
PITCH_TIGHT_DEG = -85                            # rescale.py:86
HEIGHT_FACTOR = 0.9                              # rescale.py:91
RANSAC_MIN_POINTS = 12                           # rescale.py:152
RANSAC_ITERATIONS = 100                          # rescale.py:155
RANSAC_THRESHOLD = 0.005                         # rescale.py:155
RANSAC_GOAL = 0.8                                # estimate_road_norm.py:68
SLEW = 0.3                                       # rescale.py:169-172


def triangle_potential(edge_potential=EDGE_POTENTIAL):
    """graph.py:6-17: 8x8 table, rows = (a,b,c) inlier states of the 3 vertices, columns = observed
    edge-order code."""
    ep = np.array(edge_potential)
    tp = np.ones((8, 8))
    for row in range(8):
        r = [int(row & 4 != 0), int(row & 2 != 0), int(row & 1 != 0)]
        for col in range(8):
            c = [int(col & 4 != 0), int(col & 2 != 0), int(col & 1 != 0)]
            tp[row, col] = ep[r[0] * 2 + r[1], c[0]] * ep[r[1] * 2 + r[2], c[1]] * ep[r[0] * 2 + r[2], c[2]]
    return tp


def vertex_probabilities(tp):
    """graph.py:134-145 for each of the 8 edge codes: (pa, pb, pc)."""
    rng = np.arange(8)
    out = np.zeros((8, 3))
    for idx in range(8):
        pot = tp[:, idx]
        z = np.sum(pot)
        with np.errstate(all="ignore"):
            out[idx] = [np.sum(pot[(rng & 4) > 0]) / z, np.sum(pot[(rng & 2) > 0]) / z, np.sum(pot[(rng & 1) > 0]) / z]
    return out


def edge_code(v, d, tri):
    """graph.py:124-129, vectorised over triangles."""
    v0, v1, v2 = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
    d0, d1, d2 = d[tri[:, 0]], d[tri[:, 1]], d[tri[:, 2]]
    a = ((v0 - v1) * (d0 - d1) < 0).astype(np.int64)
    b = ((v1 - v2) * (d1 - d2) < 0).astype(np.int64)
    c = ((v0 - v2) * (d0 - d2) < 0).astype(np.int64)
    return a * 4 + b * 2 + c


def graph_inliers(v, d, tri, edge_potential=EDGE_POTENTIAL):
    """graph.py:18-36: a vertex is kept when more than half of its incident triangles give it a
    marginal > 0.6.  Returns (valid, good, total) — `good`/`total` are the integer tallies behind
    the ratio (a vertex in no triangle has 0/0 = nan, which is not > 0.5)."""
    n = v.shape[0]
    probs = vertex_probabilities(triangle_potential(edge_potential))
    good_flag = probs > PROB_THRESHOLD                      # (8,3)
    code = edge_code(v, d, np.asarray(tri))
    total = np.zeros(n, dtype=np.int64)
    good = np.zeros(n, dtype=np.int64)
    np.add.at(total, np.asarray(tri).reshape(-1), 1)
    np.add.at(good, np.asarray(tri).reshape(-1), good_flag[code].reshape(-1).astype(np.int64))
    with np.errstate(all="ignore"):
        valid = (good / total) > VOTE_THRESHOLD
    return valid, good, total


@dataclass
class FlatSelection:
    ids: np.ndarray              # vertex ids of the kept triangles, in triangle order, with repeats (rescale.py:101)
    heights_loose: np.ndarray    # heights[pitch < -80]  (second return value, rescale.py:102)
    height_level: float
    tri_valid: np.ndarray
    heights: np.ndarray
    pitch_deg: np.ndarray


def flat_selection(xyz, tri):
    """rescale.py:75-102 (same NumPy routines as the reference: inv, @, median)."""
    tri = np.asarray(tri)
    A = xyz[tri]
    normals = (np.linalg.inv(A) @ np.ones((3, 1), float)).reshape(-1, 3)
    nlen = np.sqrt(np.sum(normals * normals, 1)).reshape(-1, 1)
    unit = normals / nlen
    pitch_deg = np.arcsin(-unit[:, 1]) * 180 / np.pi
    loose = pitch_deg < PITCH_LOOSE_DEG
    tight = pitch_deg < PITCH_TIGHT_DEG
    heights = (1 / nlen).reshape(-1)
    with np.errstate(all="ignore"):
        height_level = HEIGHT_FACTOR * np.median(heights[loose])
        tri_valid = tight & (heights > height_level)
    return FlatSelection(tri[tri_valid].reshape(-1), heights[loose], float(height_level), tri_valid, heights, pitch_deg)


def estimate_plane(p3):
    """estimate_road_norm.py:8-15: unit 4-vector spanning the null space of [x y z 1] of 3 points."""
    a = np.ones((3, 4))
    a[:, :3] = np.asarray(p3)[:3]
    return np.linalg.svd(a)[-1][-1, :]


def count_inliers(m, pts, threshold):
    """estimate_road_norm.py:17-18 over all points."""
    return int(np.sum(np.abs(pts @ m[:3] + m[3]) < threshold))


def run_ransac(pts, triples, threshold=RANSAC_THRESHOLD, goal_fraction=RANSAC_GOAL, repeated_counts_zero=False):
    """ransac.py:3-23 with the sample sequence given: `triples` (H,3) row indices, consumed in order;
    stops at the first improvement that exceeds the goal.  Returns (model, best_count, n_used).
    `repeated_counts_zero` (the device-resident sampler's rule): a sample holding one point twice spends its iteration with
    zero inliers — the reference's SVD of such a rank-2 sample returns a plane that rounding noise picks from the pencil through
    two points (the reference's own run on such samples: tests/golden/rescale.npz frame 26 — never the best plane there)."""
    goal = pts.shape[0] * goal_fraction
    best_ic, best_m, used = 0, None, 0
    for t in triples:
        used += 1
        p3 = pts[list(t)]
        if repeated_counts_zero and (np.array_equal(p3[0], p3[1]) or np.array_equal(p3[0], p3[2]) or np.array_equal(p3[1], p3[2])):
            continue
        m = estimate_plane(p3)
        ic = count_inliers(m, pts, threshold)
        if ic > best_ic:
            best_ic, best_m = ic, m
            if ic > goal:
                break
    return best_m, best_ic, used


def estimate_line(p2):
    """estimate_road_norm.py:39-46: unit 3-vector spanning the null space of [x y 1] of 2 points."""
    a = np.ones((2, 3))
    a[:, :2] = np.asarray(p2)[:2]
    return np.linalg.svd(a)[-1][-1, :]


def run_ransac_line(pts, pairs, threshold, goal_fraction=RANSAC_GOAL):
    """get_pitch_line_ransac (estimate_road_norm.py:60-64) with the sample sequence given."""
    goal = pts.shape[0] * goal_fraction
    best_ic, best_m, used = 0, None, 0
    for t in pairs:
        used += 1
        m = estimate_line(pts[list(t)])
        ic = int(np.sum(np.abs(pts @ m[:2] + m[2]) < threshold))              # is_inlier_line, :48-49
        if ic > best_ic:
            best_ic, best_m = ic, m
            if ic > goal:
                break
    return best_m, best_ic, used


def road_model_ransac(pts, triples):
    """ScaleEstimator.road_model_calculation_ransac, /root/reference/src/scale_calculator.py:366-384, with the sample
    triples given: (camera height, pitch, inlier mask)."""
    m, _, _ = run_ransac(pts, triples, threshold=0.005)
    inl = np.abs(pts @ m[:3] + m[3]) < 0.01                                   # get_inliers, estimate_road_norm.py:71-78
    normal, h_bar = np.array(m[:3]), -m[3]
    if normal[1] < 0:
        normal, h_bar = -normal, -h_bar
    ln = np.sqrt(np.sum(normal * normal))
    return h_bar / ln, np.arcsin(-normal[1] / ln), inl


def scale_from_model(m, absolute_reference):
    """rescale.py:156-167: camera height from the plane, sign fixed so that n_y >= 0."""
    norm = np.array(m[:3], dtype=np.float64)
    h_bar = -m[3]
    if norm[1] < 0:
        norm, h_bar = -norm, -h_bar
    norm_norm = np.sqrt(norm @ norm) / h_bar
    return absolute_reference / (1 / norm_norm)


_M64 = (1 << 64) - 1


def mix64(x):
    """splitmix64's finaliser (the product's counter-based sample sequence, include/mvosr.h: mvosr_flat_ransac_batch)."""
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def device_triples(seed, frame_counter, ids, n_hyp=RANSAC_ITERATIONS):
    """The list positions hypothesis h = 0..n_hyp-1 of frame `frame_counter` draws under key `seed` from the point list whose
    vertex ids are `ids` (an int m: a list of m distinct points): three distinct positions, uniform — the reference draws
    them with random.sample from OS entropy, /root/reference/src/thirdparty/Ransac/ransac.py:6,10.  ONE draw per hypothesis:
    a sample that names one vertex twice (the list repeats vertices, /root/reference/src/rescale.py:101) spends its
    iteration, as in the reference (ransac.py:8-21)."""
    ids = np.arange(ids) if np.isscalar(ids) else np.asarray(ids)
    m = len(ids)
    key = mix64((seed ^ ((frame_counter * 0xD1B54A32D192ED03) & _M64)) & _M64)
    out = np.zeros((n_hyp, 3), dtype=np.int64)
    for h in range(n_hyp):
        hk = mix64((key + h) & _M64)
        r = [mix64((hk + k) & _M64) for k in range(3)]
        i0 = (r[0] * m) >> 64
        i1 = (r[1] * (m - 1)) >> 64
        if i1 >= i0:
            i1 += 1
        i2 = (r[2] * (m - 2)) >> 64
        lo, hi = min(i0, i1), max(i0, i1)
        if i2 >= lo:
            i2 += 1
        if i2 >= hi:
            i2 += 1
        out[h] = (i0, i1, i2)
    return out


def canonical_rows(tri):
    """Vertex ids ascending inside a row, rows in lexicographic order: the row form of the device triangulation, a
    function of the triangle set alone.  GraphChecker's vote (graph.py:18-36,124-145, a symmetric edge potential) and
    flat_selection's kept SET (rescale.py:75-96) do not depend on the row form; the ORDER of flat_selection's point list
    (:101) does, and with it which points a given sequence of sample positions picks — immaterial for a sampler that
    draws positions uniformly."""
    t = np.sort(np.asarray(tri, dtype=np.int32).reshape(-1, 3), axis=1)
    return t if t.shape[0] == 0 else np.ascontiguousarray(t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))])


class OracleRescaleEstimator:
    """Same call surface as rescale.ScaleEstimator; `sampler(n) -> (H,3)` supplies the RANSAC
    index triples for a list of n points.  `device_seed` (instead of a sampler): the product's counter-based sample
    sequence (device_triples) over rows in canonical form — the restatement of the device-resident path."""

    def __init__(self, absolute_reference, window_size=6, sampler=None, device_seed=None, canonical=None):
        self.absolute_reference = absolute_reference
        self.window_size = window_size
        self.vanish = VANISH
        self.scale = 1
        self.scale_queue = deque()
        self.sampler = sampler
        self.device_seed = device_seed
        self.canonical = (device_seed is not None) if canonical is None else canonical
        self.frame_counter = 0
        self.last = {}

    def initial_estimation(self, motion_matrix):
        return 0                                                   # rescale.py:36-38

    def feature_selection(self, feature3d, feature2d):
        from scipy.spatial import Delaunay
        low = feature2d[:, 1] > self.vanish                        # :115
        f2, f3 = feature2d[low], feature3d[low]
        tri = Delaunay(f2).simplices                               # :124
        if self.canonical:
            tri = canonical_rows(tri)
        valid, good, total = graph_inliers(f2[:, 1], f3[:, 2], tri)
        self.last.update(tri1=tri, valid=valid, good=good, total=total)
        if np.sum(valid) > MIN_VALID_FOR_RETRI:                    # :133
            f2, f3 = f2[valid], f3[valid]
            tri = Delaunay(f2).simplices
            if self.canonical:
                tri = canonical_rows(tri)
        fs = flat_selection(f3, tri)
        self.last.update(tri2=tri, flat=fs)
        return f3[fs.ids], fs.heights_loose

    def scale_calculation_ransac(self, pts):
        self.last.pop("model", None)
        if pts.shape[0] >= RANSAC_MIN_POINTS:                      # :152
            if self.device_seed is not None:
                triples = device_triples(self.device_seed, self.frame_counter, self.last["flat"].ids)
            else:
                triples = self.sampler(pts.shape[0])
            m, ic, used = run_ransac(np.array(pts), triples, repeated_counts_zero=self.device_seed is not None)
            self.last.update(model=m, best_ic=ic, used=used)
            scale = scale_from_model(m, self.absolute_reference)
            if scale - self.scale > SLEW:                          # :169-174
                self.scale += SLEW
            elif scale - self.scale < -SLEW:
                self.scale -= SLEW
            else:
                self.scale = scale
        self.scale_queue.append(self.scale)                        # :175-178
        if len(self.scale_queue) > self.window_size:
            self.scale_queue.popleft()
        self.frame_counter += 1
        return np.median(self.scale_queue), 1

    def scale_calculation(self, feature3d, feature2d, img=None):
        pts, _ = self.feature_selection(np.asarray(feature3d, dtype=np.float64), np.asarray(feature2d, dtype=np.float64))
        return self.scale_calculation_ransac(pts)
