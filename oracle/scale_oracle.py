"""TEST INFRASTRUCTURE — CPU oracle for the scale-recovery hot path.  NOT part of the product.

A NumPy restatement (vectorised, no per-triangle Python loops) of the deterministic
``ScaleEstimator`` of the reference, /root/reference/src/scale_calculator.py:21-497, written
stage by stage so that every intermediate the HIP kernels produce has a CPU counterpart.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  The product path
(``mvoscalerecovery_amd``) never imports it and has no CPU fallback.

Pinning: the reference has no tests or golden vectors of its own, so this
oracle is pinned against outputs of the reference itself, generated in the build container by
importing /root/reference (tests/golden/make_golden.py, fixtures in tests/golden/*.npz):
per-stage intermediates for ~20 frames, the known-answer cases of SURVEY.md §4, edge cases,
and 200- / 4541-frame output-only sequences.  tests/test_oracle_golden.py checks them (exact
on every discrete output and on the returned heights/scales, <=1e-12 on intermediate floats).

Third-party arithmetic the reference calls that is not under /root/reference (un-pinned
there: no requirements file): SciPy ``spatial.Delaunay`` (Qhull) at
scale_calculator.py:257,266 — consumed verbatim (``simplices`` rows, intra-row vertex order
matters); NumPy/LAPACK ``inv`` (:229), ``histogram`` (:326), ``median``
(:333,:400).  Where this file calls the same NumPy routine the citation says so.
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field

import numpy as np

# ---- constants of the reference's path ------------------------------------------------------
CAMERA_PITCH = -0.5 * np.pi / 180          # scale_calculator.py:24
VANISH = 185                               # scale_calculator.py:22 (ctor default)
PITCH_THRESHOLD_DEG = -80                  # scale_calculator.py:235,239
N_EDGES = 170                              # scale_calculator.py:326  range(0,170)
BIN_EDGES = np.array(range(0, N_EDGES)) * 0.1   # same expression as the reference -> same doubles
MODE_REL = 0.33                            # scale_calculator.py:461
MODE_MIN = 2                               # scale_calculator.py:451,462
MODE_GAP = 0.11                            # scale_calculator.py:474
SKEW_THRESHOLD = 0.3                       # scale_calculator.py:348

# ---- per-frame status codes (must equal mvoscalerecovery_amd.constants; a test checks) ----
ST_MODE = 0            # returned mode/10                               (:354)
ST_RIGHT = 1           # skew > 0.3 -> returned right local-min edge    (:348-352)
ST_MEDIAN = 2          # no modes -> median of remaining y              (:331-333)
ST_LEVEL = 3           # no modes and no points left -> height_level    (:334-335)
ST_NO_FLAT = 4         # selection empty -> scale = ref/height_level, std=100 (:277-279,:420-422)
ST_ERR_LEFT = 5        # IndexError at :343 (no local minimum left of the mode)
ST_ERR_RIGHT = 6       # IndexError at :344 (no local minimum right of the mode)
ST_ERR_SINGULAR = 7    # LinAlgError at :229 (exactly singular triangle)
ST_ERR_MASK = 8        # tri2 was not built on the mask the vote produces (build-side check)
ST_ERR_EMPTY = 9       # no triangles / no features handed in (build-side check)
ST_TOO_FEW = 10        # <= 3 features below the vanishing row: no second triangulation, the scale comes from
                       # the PREVIOUS frame's height_level, std = 100 (:263-270,:420-422)


# ---- a4: feature_remap ------------------------------------------------------------------------
def remap(feature3d, camera_pitch=CAMERA_PITCH):
    """scale_calculator.py:390-394: rotate (y,z) by camera_pitch.  Returns a new (N,3) array
    (the reference mutates its argument in place; values are identical)."""
    f = np.array(feature3d, dtype=np.float64, copy=True)
    c, s = np.cos(camera_pitch), np.sin(camera_pitch)
    y = feature3d[:, 1] * c - feature3d[:, 2] * s
    z = feature3d[:, 1] * s + feature3d[:, 2] * c
    f[:, 1] = y
    f[:, 2] = z
    return f


def lower_mask(feature2d, vanish=VANISH):
    """scale_calculator.py:252: keep features strictly below the vanishing row."""
    return feature2d[:, 1] > vanish


# ---- a7: find_outliers / check_triangle ---------------------------------------------------
def canonical_rows(tri):
    """The row form of the ``check_triangle="fixed"`` mode: vertex ids ascending inside every row, rows in
    lexicographic order — a function of the triangle SET alone, so SciPy's rows and the device
    triangulation's rows (mvosr_delaunay_batch emits this form) give the same array."""
    t = np.sort(np.asarray(tri).reshape(-1, 3), axis=1)
    if t.shape[0] == 0:
        return t
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


CHECK_TRIANGLE_MODES = ("reference", "fixed")


def outlier_votes(v, z, tri, check_triangle="reference"):
    """scale_calculator.py:105-119,151-167.  ``v`` pixel rows, ``z`` depths (remapped), ``tri``
    (T,3) int vertex ids exactly as SciPy emitted them.  Returns the per-feature counters
    (int64, start value 1, +1 per incident triangle that does not flag the vertex, -1 per one
    that does).  ``check_triangle="reference"`` reproduces the reference's quirk: the (0,2) pair test
    marks vertices 0 and 1 (:113-115), never vertex 2.  ``"fixed"`` is the evident intent of those lines —
    ``b > 0`` marks vertices 0 and 2 — a DECLARED DEVIATION: every vertex is then flagged
    iff one of its two pair tests fails, which does not depend on the order of the vertices inside a row.
    """
    if check_triangle not in CHECK_TRIANGLE_MODES:
        raise ValueError("check_triangle must be 'reference' or 'fixed'")
    tri = np.asarray(tri)
    n = v.shape[0]
    counters = np.ones(n, dtype=np.int64)
    if tri.shape[0] == 0:
        return counters
    v0, v1, v2 = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
    d0, d1, d2 = z[tri[:, 0]], z[tri[:, 1]], z[tri[:, 2]]
    a = (v0 - v1) * (d0 - d1) > 0
    b = (v0 - v2) * (d0 - d2) > 0
    c = (v1 - v2) * (d1 - d2) > 0
    if check_triangle == "fixed":
        flag = np.stack([a | b, a | c, b | c], axis=1)
    else:
        flag = np.stack([a | b, a | b | c, c], axis=1)      # (T,3) True = vote against
    votes = np.where(flag, -1, 1).astype(np.int64)
    np.add.at(counters, tri.reshape(-1), votes.reshape(-1))
    return counters


def votes_valid(counters):
    """scale_calculator.py:166."""
    return counters >= 0


# ---- a8: feature_selection_by_tri ---------------------------------------------------------
@dataclass
class TriSelect:
    normals: np.ndarray          # (T,3) plane normals n = A^-1 . 1 (not normalised)
    normals_len: np.ndarray      # (T,)
    pitch_deg: np.ndarray        # (T,)
    heights: np.ndarray          # (T,)
    valid_pitch: np.ndarray      # (T,) bool  pitch_deg < -80
    height_level: float
    tri_valid: np.ndarray        # (T,) bool  valid_pitch & heights > height_level
    selected_ids: np.ndarray     # sorted unique vertex ids of the kept triangles
    singular: bool = False


def tri_select(xyz, tri):
    """scale_calculator.py:225-248.  ``xyz`` (N',3) remapped features that survived the vote,
    ``tri`` (T,3) the second triangulation.  Uses the same NumPy routines the reference calls:
    ``np.linalg.inv`` (what ``np.matrix.I`` dispatches to for square input, :229) and ``@`` with
    a (3,1) ones vector (:230), so intermediates are bit-identical on the same NumPy/LAPACK.
    """
    tri = np.asarray(tri)
    tcount = tri.shape[0]
    A = xyz[tri]                                                    # (T,3,3) rows = vertices
    singular = False
    try:
        Ai = np.linalg.inv(A)
    except np.linalg.LinAlgError:
        singular = True
        Ai = np.full_like(A, np.nan)
    ones = np.ones((3, 1), float)
    normals = (Ai @ ones).reshape(-1, 3)
    with np.errstate(all="ignore"):
        nlen = np.sqrt(np.sum(normals * normals, 1)).reshape(-1, 1)
        unit = normals / nlen
        pitch_deg = np.arcsin(-unit[:, 1]) * 180 / np.pi
        valid_pitch = pitch_deg < PITCH_THRESHOLD_DEG
        heights = np.mean(A[:, :, 1], 1) if tcount else np.zeros(0)
        invalid_pitch = pitch_deg >= PITCH_THRESHOLD_DEG
        height_level = np.mean(heights[invalid_pitch]) if invalid_pitch.any() else np.nan
        tri_valid = valid_pitch & (heights > height_level)
    selected = np.unique(tri[tri_valid].reshape(-1))
    return TriSelect(normals, nlen.reshape(-1), pitch_deg, heights, valid_pitch,
                     float(height_level), tri_valid, selected, singular)


# ---- a9: road_model_calculation_static ----------------------------------------------------
def histogram_170(y):
    """scale_calculator.py:326: ``np.histogram(y, bins=arange(170)*0.1)`` restated: bin k counts
    ``edge[k] <= y < edge[k+1]``, the last bin is closed on the right, values outside
    [edge[0], edge[169]] are dropped."""
    e = BIN_EDGES
    k = np.searchsorted(e, y, side="right") - 1          # edge[k] <= y < edge[k+1]
    k = np.where(y == e[-1], N_EDGES - 2, k)
    ok = (y >= e[0]) & (y <= e[-1])
    return np.bincount(k[ok], minlength=N_EDGES - 1)[: N_EDGES - 1].astype(np.int64)


def single_bin_drop_mask(y, hist):
    """scale_calculator.py:284-293 (remove_single): True where a point lies in the interval of a
    bin whose count is exactly 1.  The interval is built from the bin's RIGHT edge r as
    [r-0.1, r] for the first such bin and (r-0.1, r] for the others — note r-0.1 is not always
    the bin's own left edge in floating point, so this is done literally."""
    right = BIN_EDGES[1:][hist == 1]
    drop = np.zeros(y.shape[0], dtype=bool)
    if np.sum(right) > 0:                                 # the reference's guard (:287)
        lo = right - 0.1
        drop |= (y >= lo[0]) & (y <= right[0])
        if right.shape[0] > 1:
            drop |= ((y[:, None] > lo[None, 1:]) & (y[:, None] <= right[None, 1:])).any(axis=1)
    return drop


def mode_flags(hist):
    """scale_calculator.py:446-465: local maxima that reach 0.33 max and >= 2; end bins only if
    they equal the maximum; no modes at all when max <= 2."""
    h = np.asarray(hist)
    flag = np.zeros(h.shape[0], dtype=bool)
    mx = h.max()
    if mx <= MODE_MIN:
        return flag
    flag[0] = h[0] == mx
    flag[-1] = h[-1] == mx
    mid = h[1:-1]
    flag[1:-1] = (mid >= h[:-2]) & (mid >= h[2:]) & (mid >= MODE_REL * mx) & (mid >= MODE_MIN)
    return flag


def mode_clusters(flag):
    """scale_calculator.py:466-483: group the modes' right edges while consecutive edges are
    closer than 0.11.  Returns a list of (first_edge, last_edge) per cluster."""
    edges = BIN_EDGES[1:][flag]
    clusters = []
    for e in edges:
        if clusters and e - clusters[-1][1] < MODE_GAP:
            clusters[-1][1] = e
        else:
            clusters.append([e, e])
    return [(a, b) for a, b in clusters]


def local_min_flags(hist):
    """scale_calculator.py:428-443 (check_reverse_mode)."""
    h = np.asarray(hist)
    flag = np.zeros(h.shape[0], dtype=bool)
    mn = h.min()
    flag[0] = h[0] == mn
    flag[-1] = h[-1] == mn
    mid, left, right = h[1:-1], h[:-2], h[2:]
    flag[1:-1] = (mid <= left) & (mid <= right) & ~((mid == right) & (mid == left))
    return flag


@dataclass
class RoadModel:
    height: float
    status: int
    hist_raw: np.ndarray
    hist: np.ndarray             # singles zeroed
    n_kept: int                  # points left after remove_single
    n_modes: int = 0
    mode_left: int = -1
    mode_right: int = -1
    left: float = np.nan
    right: float = np.nan
    skew: float = np.nan
    mean: float = np.nan
    std: float = np.nan


def road_model(y, height_level):
    """scale_calculator.py:324-354.  ``y`` = remapped y of the selected road points."""
    hist_raw = histogram_170(y)
    drop = single_bin_drop_mask(y, hist_raw)
    y_kept = y[~drop]
    hist = hist_raw.copy()
    hist[hist == 1] = 0
    flag = mode_flags(hist)
    clusters = mode_clusters(flag)
    rm = RoadModel(np.nan, ST_MODE, hist_raw, hist, int(y_kept.shape[0]), n_modes=len(clusters))
    if not clusters:
        if y_kept.shape[0] > 0:
            rm.height, rm.status = float(np.median(y_kept)), ST_MEDIAN     # :333
        else:
            rm.height, rm.status = float(height_level), ST_LEVEL          # :335
        return rm
    first, last = clusters[-1]                                            # :338-339 modes[-1]
    ml, mr = int(first * 10), int(last * 10)
    rm.mode_left, rm.mode_right = ml, mr
    mode = (ml + mr) / 2
    mins = local_min_flags(hist)
    left_c = BIN_EDGES[1:ml + 1][mins[:ml]]                               # :341,:343
    if left_c.shape[0] == 0:
        rm.status = ST_ERR_LEFT
        return rm
    rm.left = float(left_c[-1])
    right_c = BIN_EDGES[mr + 1:][mins[mr:]]                               # :342,:344
    if right_c.shape[0] == 0:
        rm.status = ST_ERR_RIGHT
        return rm
    rm.right = float(right_c[0])
    with np.errstate(all="ignore"):
        rm.mean, rm.std = float(np.mean(y_kept)), float(np.std(y_kept))
        rm.skew = (np.mean(y_kept) - mode / 10) / np.std(y_kept)          # :346,:496
    if rm.skew > SKEW_THRESHOLD:
        rm.height, rm.status = rm.right, ST_RIGHT
    else:
        rm.height, rm.status = mode / 10, ST_MODE
    return rm


# ---- a10: scale_filtering -----------------------------------------------------------------
def window_median(raw_scales, window, queue=()):
    """scale_calculator.py:396-400 applied to a whole sequence: push, trim to ``window`` from the
    left, median (mean of the two middle values while the length is even).  ``queue`` is the
    deque content carried in from earlier frames.  Returns (filtered, final_queue)."""
    q = deque(queue)
    out = np.empty(len(raw_scales), dtype=np.float64)
    for i, s in enumerate(raw_scales):
        q.append(s)
        if len(q) > window:
            q.popleft()
        out[i] = np.median(q)
    return out, list(q)


# ---- a3/a5: one frame, all stages -----------------------------------------------------------
@dataclass
class FrameResult:
    raw_scale: float
    height: float
    height_level: float
    status: int
    std: float
    lower: np.ndarray = None
    counters: np.ndarray = None
    valid: np.ndarray = None
    tri1: np.ndarray = None
    tri2: np.ndarray = None
    sel: TriSelect = None
    road: RoadModel = None
    flat_feature: np.ndarray = None
    flat_feature_2d: np.ndarray = None
    extra: dict = field(default_factory=dict)


def delaunay(points2d):
    """scale_calculator.py:257-258,266-267: SciPy/Qhull, simplices verbatim."""
    from scipy.spatial import Delaunay
    return Delaunay(points2d).simplices


def frame_raw_scale(feature3d, feature2d, absolute_reference, tri1=None, tri2=None,
                    camera_pitch=CAMERA_PITCH, vanish=VANISH, keep=True, check_triangle="reference"):
    """scale_calculator.py:411-422 up to (not including) the window filter.  ``tri1``/``tri2``
    may be supplied (the batch path takes both triangulations as inputs); otherwise they are
    computed with SciPy exactly where the reference computes them.  ``check_triangle="fixed"``: the
    order-invariant vote on rows in canonical form (``canonical_rows``) — both triangulations then enter
    the stages as functions of their triangle sets only (the declared deviation of row f1)."""
    fixed = check_triangle == "fixed"
    f3 = remap(np.asarray(feature3d, dtype=np.float64), camera_pitch)
    f2 = np.asarray(feature2d, dtype=np.float64)
    low = lower_mask(f2, vanish)
    f3l, f2l = f3[low], f2[low]
    if tri1 is None:
        tri1 = delaunay(f2l)
    if fixed:
        tri1 = canonical_rows(tri1)
    counters = outlier_votes(f2l[:, 1], f3l[:, 2], tri1, check_triangle)
    valid = votes_valid(counters)
    if not valid.shape[0] > 3:                                            # :263 (the LENGTH of the mask, as the reference tests it)
        res = FrameResult(np.nan, np.nan, np.nan, ST_TOO_FEW, 100)        # :268-270 -> None -> :420-422 with the previous level
        if keep:
            res.lower, res.counters, res.valid, res.tri1 = low, counters, valid, tri1
        return res
    f3v, f2v = f3l[valid], f2l[valid]
    if tri2 is None:
        tri2 = delaunay(f2v)
    if fixed:
        tri2 = canonical_rows(tri2)
    sel = tri_select(f3v, tri2)
    res = FrameResult(np.nan, np.nan, sel.height_level, ST_MODE, 1)
    if keep:
        res.lower, res.counters, res.valid, res.tri1, res.tri2, res.sel = low, counters, valid, tri1, tri2, sel
    if sel.singular:
        res.status = ST_ERR_SINGULAR
        return res
    if sel.selected_ids.shape[0] == 0:                                    # :274-279
        res.status, res.std = ST_NO_FLAT, 100
        with np.errstate(all="ignore"):
            res.raw_scale = float(np.float64(absolute_reference) / np.float64(sel.height_level))  # :421
        return res
    pts = f3v[sel.selected_ids]
    road = road_model(pts[:, 1], sel.height_level)
    res.height, res.status = road.height, road.status
    if keep:
        res.road, res.flat_feature, res.flat_feature_2d = road, pts, f2v[sel.selected_ids]
    if road.status in (ST_ERR_LEFT, ST_ERR_RIGHT):
        return res
    with np.errstate(all="ignore"):
        res.raw_scale = float(np.float64(absolute_reference) / np.float64(road.height))           # :419
    return res


class StatusError(Exception):
    pass


def raise_for_status(status):
    """Map an error status to the exception the reference raises at that point."""
    if status in (ST_ERR_LEFT, ST_ERR_RIGHT):
        raise IndexError("index -1 is out of bounds for axis 0 with size 0"
                         if status == ST_ERR_LEFT else
                         "index 0 is out of bounds for axis 0 with size 0")
    if status == ST_ERR_SINGULAR:
        raise np.linalg.LinAlgError("Singular matrix")
    if status in (ST_ERR_MASK, ST_ERR_EMPTY):
        raise StatusError("status %d" % status)


class OracleScaleEstimator:
    """Same call surface as the reference class (scale_calculator.py:21-46,396-423), built from
    the stage functions above.  Used by tests as the comparator for the product's
    ``ScaleEstimator`` and by the driver-loop tests as the injected CPU backend."""

    def __init__(self, absolute_reference, window_size=6, vanish=VANISH, focus=718, check_triangle="reference"):
        if check_triangle not in CHECK_TRIANGLE_MODES:
            raise ValueError("check_triangle must be 'reference' or 'fixed'")
        self.check_triangle = check_triangle
        self.absolute_reference = absolute_reference
        self.camera_pitch = CAMERA_PITCH
        self.window_size = window_size
        self.vanish = vanish
        self.focus = focus
        self.scale_queue = deque()
        self.motion_queue = deque()
        self.flat_feature = []
        self.flat_feature_2d = []
        self.last = None

    def initial_estimation(self, motion_t):
        pitch = np.arcsin(motion_t[1]) * 180 / np.pi                      # :42
        self.motion_queue.append(np.asarray(motion_t).reshape(-1))       # :44
        return pitch

    def scale_filtering(self, scale):
        out, q = window_median([scale], self.window_size, self.scale_queue)
        self.scale_queue = deque(q)
        return out[0]

    def scale_calculation(self, feature3d, feature2d, img=None, tri1=None, tri2=None):
        res = frame_raw_scale(feature3d, feature2d, self.absolute_reference, tri1, tri2,
                              self.camera_pitch, self.vanish, check_triangle=self.check_triangle)
        self.last = res
        if res.status == ST_TOO_FEW:
            # :420-422 reads self.height_level of an earlier frame (AttributeError if there is none)
            self.flat_feature = None
            with np.errstate(all="ignore"):
                res.raw_scale = float(np.float64(self.absolute_reference) / np.float64(self.height_level))
            return self.scale_filtering(res.raw_scale), res.std
        self.height_level = res.height_level
        raise_for_status(res.status)
        self.flat_feature = res.flat_feature
        if res.flat_feature is not None:
            self.flat_feature_2d = res.flat_feature_2d
        return self.scale_filtering(res.raw_scale), res.std
