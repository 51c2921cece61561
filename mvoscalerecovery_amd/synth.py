"""Synthetic KITTI-shaped optical-flow features (SURVEY.md §8d generator spec).

The reference ships no feature dumps (only image path lists with absolute paths,
/root/reference/dataset/kitti_image_00.txt), so every measured configuration runs on
synthetic frames that have the shape the reference's drivers hand to
``ScaleEstimator.scale_calculation``: ``feature3d`` (N,3) float64 camera-frame points up
to scale and ``feature2d`` (N,2) float64 pixel coordinates, produced the way
/root/reference/src/main.py:102-106 produces them (pinhole re-projection with one focal
length for both axes).  Camera constants are KITTI-00's, /root/reference/src/param.py:30-36.

Everything is seeded per frame (``default_rng(base_seed + frame_idx)``) so that a fixture
can store outputs only and regenerate the inputs (a checksum of the inputs travels with
the fixture to detect RNG drift).
"""
from __future__ import annotations

import zlib

import numpy as np

# KITTI-00 intrinsics, /root/reference/src/param.py:30-35
IMG_W = 1241.0
IMG_H = 376.0
FX = 718.856
CX = 607.1928
CY = 185.2157


def synth_frame(frame_idx: int, n_features: int = 2000, base_seed: int = 1234,
                sigma: float | None = None, h_cam: float | None = None,
                v_min: float = 186.0, upper_fraction: float = 0.0):
    """One frame of KITTI-shaped features.

    Returns ``(feature3d (N,3) f64, feature2d (N,2) f64)``.

    * ``u ~ U(0, IMG_W)``, ``v ~ U(v_min, IMG_H)``; with ``v_min=186`` every feature passes
      the reference's ``v > vanish(185)`` filter (/root/reference/src/scale_calculator.py:252)
      so N features give ~2N triangles.  ``upper_fraction`` > 0 places that share of the
      features above the vanishing row (they must be dropped by the filter).
    * features inside the road trapezoid ``|u-cx| < 250 + 2 (v-cy)`` lie on the ground plane
      at camera height ``h_cam`` (VO units): ``z = h fx / max(v-cy, 1)``; the others are
      obstacles above the ground: ``z = min(U(5,60), z_ground)``.
    * multiplicative depth noise ``z *= 1 + sigma N(0,1)``.
    """
    rng = np.random.default_rng(base_seed + frame_idx)
    if h_cam is None:
        h_cam = 0.6 + 0.4 * ((frame_idx * 0.6180339887498949) % 1.0)
    if sigma is None:
        sigma = 0.01 + 0.01 * ((frame_idx * 0.7548776662466927) % 1.0)
    u = rng.uniform(0.0, IMG_W, n_features)
    v = rng.uniform(v_min, IMG_H, n_features)
    if upper_fraction > 0.0:
        up = rng.uniform(0.0, 1.0, n_features) < upper_fraction
        v = np.where(up, rng.uniform(0.0, 185.0, n_features), v)
    dv = np.maximum(v - CY, 1.0)
    z_ground = h_cam * FX / dv
    on_road = np.abs(u - CX) < 250.0 + 2.0 * (v - CY)
    z_obst = np.minimum(rng.uniform(5.0, 60.0, n_features), z_ground)
    z = np.where(on_road & (v > CY), z_ground, z_obst)
    z = z * (1.0 + sigma * rng.standard_normal(n_features))
    z = np.maximum(z, 0.2)
    x = (u - CX) * z / FX
    y = (v - CY) * z / FX
    feature3d = np.stack([x, y, z], axis=1).astype(np.float64)
    feature2d = np.stack([u, v], axis=1).astype(np.float64)
    return feature3d, feature2d


def synth_sequence_dict(n_frames: int, base_seed: int = 77, n_lo: int = 300, n_hi: int = 1500,
                        p_not_moving: float = 0.01, p_too_few: float = 0.01,
                        fixed_n: int | None = None):
    """A dict with the schema /root/reference/src/main.py:149-154 saves and
    /root/reference/src/main_offline.py:52-55 replays: ``motions`` (12-vectors),
    ``move_flags``, ``feature2ds``, ``feature3ds`` (``[]`` for non-moving frames,
    /root/reference/src/main.py:94-95).  A few frames are non-moving and a few carry
    <= 100 features so both caller gates (/root/reference/src/main_offline.py:64,73)
    are exercised.
    """
    rng = np.random.default_rng(base_seed)
    motions, move_flags, f2, f3 = [], [], [], []
    for i in range(n_frames):
        r = rng.uniform()
        if i > 1 and r < p_not_moving:
            motions.append(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], dtype=np.float64))
            move_flags.append(False)
            f2.append([])
            f3.append([])
            continue
        if r > 1.0 - p_too_few:
            n = int(rng.integers(20, 101))
        elif fixed_n is not None:
            n = fixed_n
        else:
            n = int(rng.integers(n_lo, n_hi + 1))
        a3, a2 = synth_frame(i, n, base_seed=base_seed * 1000003, upper_fraction=0.15)
        t = rng.normal(0.0, 0.02, 3)
        t[2] = 1.0
        t /= np.linalg.norm(t)
        m = np.eye(3, 4)
        m[:, 3] = t
        motions.append(m.reshape(-1))
        move_flags.append(True)
        f2.append(a2)
        f3.append(a3)
    return {"motions": motions, "move_flags": move_flags, "feature2ds": f2, "feature3ds": f3}


def checksum(*arrays) -> int:
    """CRC32 over the raw bytes of the arrays (detects RNG drift in regenerated inputs)."""
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return c


def road_fuzz_list(i: int, seed: int = 777) -> np.ndarray:
    """The i-th list of y values of the road-model fuzz set (tests/golden/road_fuzz.npz holds what
    the reference's ``road_model_calculation_static`` returns for each): ten families aimed at the
    corners of `/root/reference/src/scale_calculator.py:284-354,428-497` — values exactly on bin
    edges, bins of count one next to modes, mode candidates around the 0.33*max and >=2 thresholds,
    several clusters, out-of-range values, very short lists and lists longer than the kernel's
    register cache (1024 values)."""
    rng = np.random.default_rng([seed, i])
    kind = i % 10
    if kind == 0:
        y = rng.normal(rng.uniform(0.4, 3.0), rng.uniform(0.03, 0.4), int(rng.integers(5, 900)))
    elif kind == 1:
        m = int(rng.integers(20, 700))
        y = np.concatenate([rng.normal(1.7, 0.06, m), 1.7 + rng.exponential(rng.uniform(0.2, 1.5), m // 2)])
    elif kind == 2:
        y = rng.uniform(-1.0, 18.0, int(rng.integers(5, 1200)))
    elif kind == 3:
        y = np.round(rng.normal(rng.uniform(0.5, 2.5), rng.uniform(0.1, 0.6), int(rng.integers(5, 600))), 1)
    elif kind == 4:
        y = rng.uniform(0.0, 3.0, int(rng.integers(1, 7)))
    elif kind == 5:
        m = int(rng.integers(1030, 2600))
        y = np.concatenate([rng.normal(1.72, rng.uniform(0.02, 0.15), m), rng.uniform(0.0, 16.9, m // 20)])
    elif kind == 6:
        centres = np.sort(rng.choice(np.arange(3, 60), size=int(rng.integers(2, 5)), replace=False)) * 0.1 + 0.05
        top = int(rng.integers(6, 40))
        parts = []
        for c in centres:
            k = int(max(1, round(top * rng.choice([1.0, 0.34, 0.33, 0.32, 0.5, 0.1]))))
            parts.append(np.full(k, c) + rng.uniform(-0.04, 0.04, k))
        y = np.concatenate(parts)
    elif kind == 7:
        k = rng.integers(0, 170, int(rng.integers(5, 300))).astype(np.float64)
        y = k * 0.1
        nudge = rng.integers(0, 3, y.size)
        y = np.where(nudge == 1, np.nextafter(y, np.inf), np.where(nudge == 2, np.nextafter(y, -np.inf), y))
    elif kind == 8:
        base = np.repeat(rng.choice(np.arange(2, 40), size=int(rng.integers(2, 6)), replace=False) * 0.1 + 0.05,
                         int(rng.integers(3, 12)))
        singles = rng.choice(np.arange(0, 60), size=int(rng.integers(1, 12)), replace=False) * 0.1 + rng.uniform(0.0, 0.1)
        y = np.concatenate([base + rng.uniform(-0.049, 0.049, base.size), singles])
    else:
        y = np.full(int(rng.integers(1, 50)), rng.choice([0.85, 0.3, 16.9, 0.0, 1.7000000000000002]))
    y = np.asarray(y, dtype=np.float64)
    rng.shuffle(y)
    return y


def fuzz_frame(i: int, seed: int = 4321):
    """The i-th frame of the frame-level fuzz set (tests/golden/frame_fuzz.npz holds what the
    reference's ``ScaleEstimator.scale_calculation`` returns or raises for each): small frames with
    the structure the happy-path generator avoids — duplicate and grid-aligned pixels, tied depths
    (vote products exactly zero), nearly collinear pixel rows, extreme depth scales, ground-only and
    obstacle-only scenes, very few points, features above the vanishing row, negative heights."""
    rng = np.random.default_rng([seed, i])
    if i >= 400:                                    # frames 400..: kind 10, added in round 4 (frames 0..399 are as they were)
        return _fuzz_level_at_zero(rng)
    kind = i % 10
    n = int(rng.integers(12, 420))
    f3, f2 = synth_frame(i, n, base_seed=seed, upper_fraction=0.0)
    u, v = f2[:, 0].copy(), f2[:, 1].copy()
    x, y, z = f3[:, 0].copy(), f3[:, 1].copy(), f3[:, 2].copy()
    if kind == 1:                                   # coarse pixel grid: many exact duplicates
        u, v = np.round(u / 40.0) * 40.0, np.round(v / 12.0) * 12.0 + 1.0
    elif kind == 2:                                 # tied depths
        z = np.maximum(np.round(z * 2.0) / 2.0, 0.5)
    elif kind == 3:                                 # a few pixel rows only
        v = 190.0 + 30.0 * rng.integers(0, 4, n) + rng.uniform(0.0, 1e-3, n)
    elif kind == 4:                                 # extreme depth scale
        z = z * (1e6 if i % 20 == 4 else 1e-6)
    elif kind == 5:                                 # ground only
        z = 0.8 * FX / np.maximum(v - CY, 1.0) * (1.0 + 1e-3 * rng.standard_normal(n))
    elif kind == 6:                                 # a wall
        z = np.full(n, 12.0) + rng.normal(0.0, 0.05, n)
    elif kind == 7:                                 # very few points
        keep = int(rng.integers(1, 12))                # 1-2: QhullError at :257; 3: the :263-270 branch
        u, v, z = u[:keep], v[:keep], z[:keep]
    elif kind == 8:                                 # a third of the features above the vanishing row
        up = rng.uniform(0.0, 1.0, len(v)) < 0.35
        v = np.where(up, rng.uniform(0.0, 185.0, len(v)), v)
    elif kind == 9:                                 # the camera below the road: negative heights
        v = 2.0 * CY - v + 190.0
    if kind in (1, 2, 3, 4, 5, 6, 7, 8, 9):
        x = (u - CX) * z / FX
        y = (v - CY) * z / FX
    return np.stack([x, y, z], axis=1).astype(np.float64), np.stack([u, v], axis=1).astype(np.float64)


def _fuzz_level_at_zero(rng):
    """Fuzz kind 10: the guard bands' worst case.  Steep triangles above AND below y' = 0 whose mean height — the
    ``height_level`` of /root/reference/src/scale_calculator.py:239-241 — is tuned to within ~1e-13 of a ground plane at
    y' = c0 ~ 1e-7: the level is within 1e-6 of zero (a guard band relative to |level| alone would be 1e-19 wide while the
    sum's rounding error is relative to mean |h| ~ 1), and every flat triangle's height is within 1e-13 of it, so the
    selection ``heights > height_level`` (:243-244) turns on the last bits of NumPy's pairwise sum.  Depth decreases
    strictly with the pixel row, so every feature survives the vote and the second triangulation is the first."""
    from scipy.spatial import Delaunay
    n = int(rng.integers(150, 420))
    u = rng.uniform(0.0, IMG_W, n)
    v = rng.uniform(190.0, IMG_H, n)
    zp = 2000.0 / (v - 150.0)                                   # z' after the remap
    ground = rng.uniform(size=n) < 0.6
    c0 = 1e-7 * (1.0 + rng.uniform())
    side = np.where(rng.uniform(size=n) < 0.5, 1.0, -1.0)
    yp = np.where(ground, c0 + rng.uniform(-1e-13, 1e-13, n), side * rng.uniform(0.3, 2.0, n))
    xp = (u - CX) * zp / FX
    tri = np.sort(np.asarray(Delaunay(np.stack([u, v], axis=1)).simplices), axis=1)
    tri = tri[np.lexsort((tri[:, 2], tri[:, 1], tri[:, 0]))]     # (a function of the triangle set: the frame must not depend on the row form)
    for _ in range(8):                                          # the level is linear in a common shift of the obstacles
        P = np.stack([xp, yp, zp], axis=1)[tri]
        nrm = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
        sgn = np.sign(np.einsum("ij,ij->i", nrm, P[:, 0]))      # n = A^-1 . 1 points away from the origin's side of the plane
        with np.errstate(all="ignore"):
            pitch = np.degrees(np.arcsin(-sgn * nrm[:, 1] / np.linalg.norm(nrm, axis=1)))
        steep = ~(pitch < -80.0)
        if not steep.any():
            break
        level = P[:, :, 1].mean(axis=1)[steep].mean()
        k = (~ground)[tri].sum(axis=1)[steep].mean() / 3.0
        if not k > 0:
            break
        yp = np.where(ground, yp, yp + (c0 - level) / k)
    c, s_ = np.cos(-0.5 * np.pi / 180), np.sin(-0.5 * np.pi / 180)   # invert feature_remap (:390-394): the path rotates back
    y = yp * c + zp * s_
    z = -yp * s_ + zp * c
    return np.stack([xp, y, z], axis=1).astype(np.float64), np.stack([u, v], axis=1).astype(np.float64)


def too_few_sequence(seed: int = 2718, n_frames: int = 12, few_at=(0, 4, 5, 9)):
    """A short sequence in which the frames ``few_at`` have exactly THREE features below the vanishing row
    (the rest above it): the reference then takes its "no enough feature for triangulation" branch
    (/root/reference/src/scale_calculator.py:263-270) and divides by the height_level an earlier frame
    left on the estimator (:420-422) — or raises AttributeError when it is the first frame.
    tests/golden/too_few.json holds what the reference returns for it.  Returns a list of (f3, f2)."""
    frames = []
    for i in range(n_frames):
        n = 220 + 17 * i
        if i in few_at:
            f3, f2 = synth_frame(i, n, base_seed=seed, upper_fraction=0.0)
            rng = np.random.default_rng([seed, i])
            low = rng.choice(n, 3, replace=False)
            v = rng.uniform(0.0, 185.0, n)
            v[low] = f2[low, 1]
            z = f3[:, 2]
            f2 = np.stack([f2[:, 0], v], axis=1)
            f3 = np.stack([f3[:, 0], (v - CY) * z / FX, z], axis=1)
        else:
            f3, f2 = synth_frame(i, n, base_seed=seed, upper_fraction=0.2)
        frames.append((f3.astype(np.float64), f2.astype(np.float64)))
    return frames


def road_long_list(k: int, seed: int = 8192) -> np.ndarray:
    """The k-th LONG list of y values (tests/golden/road_long.json holds the reference's outputs for them): longer
    than np.add.reduce's 8192-element buffer, so np.mean / np.std behind the skewness decision
    (/root/reference/src/scale_calculator.py:346,:496) are sums of several pairwise-summed chunks."""
    rng = np.random.default_rng([seed, k])
    n = (8193, 9000, 20011, 40000)[k % 4]
    y = np.concatenate([rng.normal(1.72, 0.08, n - n // 5), 1.72 + rng.exponential(0.6, n // 5)])
    rng.shuffle(y)
    return y.astype(np.float64)
