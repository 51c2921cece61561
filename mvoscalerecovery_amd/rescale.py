"""Drop-in for the reference's *other* ``ScaleEstimator`` — /root/reference/src/rescale.py:22-193,
the one /root/reference/src/main.py:20 and main_offline.py:20 import — backed by the MI355X kernels.

    from mvoscalerecovery_amd.rescale import ScaleEstimator

Per frame (rescale.py:113-148,151-178,191-193):

    host   vanishing-row filter (:115), Delaunay #1 (:124)                         [SciPy]
    GPU    GraphChecker.find_inliers tallies (graph.py:18-36)                      [mvosr_graph_inliers_batch]
    host   valid = good/total > 0.5; if more than 10 survive: filter + Delaunay #2 (:133-137)
    GPU    flat_selection: normals, 1/|n| heights, median level, kept triangles (:75-102)  [mvosr_flat_selection_batch]
    GPU    RANSAC plane over the kept triangles' vertices (:151-167)                [mvosr_ransac_plane_batch]
    host   height -> scale, +-0.3 slew limiter, window median (:167-178)

The reference's RANSAC re-seeds `random` from OS entropy on every call
(/root/reference/src/thirdparty/Ransac/ransac.py:6), so its output is not reproducible; it is a
deterministic function of the sample triples.  Here the triples are drawn on the host —
``random.sample(range(M), 3)`` like the reference, seeded with ``ransac_seed`` (None = OS entropy, the
reference's behaviour) or supplied by ``sampler(M) -> (H,3)`` — and the GPU evaluates all hypotheses
and replays the reference's sequential best/stop rule.  With the same triples the results equal the
reference's (tests/golden/rescale.npz); without, agreement is statistical only.
"""
from __future__ import annotations

import ctypes as C
import random
from collections import deque

import numpy as np

from . import _lib
from . import packing
from .engine import DeviceBatch, ScaleEngine

EDGE_POTENTIAL = [[3, 1], [2, 2], [2, 2], [0, 4]]     # rescale.py:32
VANISH = 185                                          # rescale.py:30
RANSAC_ITERATIONS = 100                               # rescale.py:155
RANSAC_THRESHOLD = 0.005                              # rescale.py:155
RANSAC_GOAL = 0.8                                     # estimate_road_norm.py:68
SLEW = 0.3                                            # rescale.py:169-172


def good_bits(edge_potential=EDGE_POTENTIAL, prob_threshold=0.6):
    """The 24 bits the vote kernel needs: for each edge-order code (graph.py:124-129) and vertex, is
    the vertex's marginal under the triangle potential (graph.py:6-17,134-145) above 0.6
    (graph.py:131-132)?  A 8x8 table evaluated once per estimator."""
    ep = np.array(edge_potential)
    tp = np.ones((8, 8))
    for row in range(8):
        r = [int(row & 4 != 0), int(row & 2 != 0), int(row & 1 != 0)]
        for col in range(8):
            c = [int(col & 4 != 0), int(col & 2 != 0), int(col & 1 != 0)]
            tp[row, col] = ep[r[0] * 2 + r[1], c[0]] * ep[r[1] * 2 + r[2], c[1]] * ep[r[0] * 2 + r[2], c[2]]
    rng = np.arange(8)
    bits = 0
    for code in range(8):
        pot = tp[:, code]
        z = np.sum(pot)
        with np.errstate(all="ignore"):
            probs = [np.sum(pot[(rng & 4) > 0]) / z, np.sum(pot[(rng & 2) > 0]) / z, np.sum(pot[(rng & 1) > 0]) / z]
        for k, p in enumerate(probs):
            if p > prob_threshold:
                bits |= 1 << (3 * code + k)
    return bits


class ScaleEstimator:
    def __init__(self, absolute_reference, window_size=6, device=0, ransac_seed=None, sampler=None,
                 delaunay_workers=None, verbose=False):
        # reference attributes (rescale.py:24-35)
        self.absolute_reference = absolute_reference
        self.camera_pitch = 0
        self.scale = 1
        self.inliers = None
        self.scale_queue = deque()
        self.window_size = window_size
        self.vanish = VANISH
        # build-side state
        self.verbose = verbose
        self.delaunay_workers = delaunay_workers
        self.engine = ScaleEngine(absolute_reference, device=device, camera_pitch=0.0)
        self.ctx = self.engine.ctx
        self._good_bits = good_bits()
        self._rng = random.Random(ransac_seed)
        self._sampler = sampler
        self.height_level = None
        self.last = {}

    def initial_estimation(self, motion_matrix):
        return 0                                                   # rescale.py:36-38

    # ---- sampling --------------------------------------------------------------------------------
    def _triples(self, m):
        if self._sampler is not None:
            t = np.asarray(self._sampler(m), dtype=np.int32)
        else:                                                      # ransac.py:10, RANSAC_ITERATIONS draws
            t = np.array([self._rng.sample(range(m), 3) for _ in range(RANSAC_ITERATIONS)], dtype=np.int32)
        return np.ascontiguousarray(t.reshape(-1, 3))

    # ---- the three GPU stages over a list of frames ------------------------------------------------
    def _graph_vote(self, pf):
        ctx, lib = self.ctx, self.ctx.lib
        db = DeviceBatch(ctx, pf, with_tri2=False)
        total = ctx.zeros(pf.total_padded, np.int32)
        good = ctx.zeros(pf.total_padded, np.int32)
        status = ctx.zeros(pf.n_frames, np.int32)
        b = db.struct()
        _lib.check(lib.mvosr_graph_inliers_batch(ctx.handle, C.byref(b), C.c_uint32(self._good_bits), total.ptr, good.ptr,
                                                 status.ptr), "mvosr_graph_inliers_batch")
        ctx.sync()
        t, g, st = total.download(), good.download(), status.download()
        for buf in (total, good, status):
            buf.free()
        db.free()
        return t, g, st

    def _flat_selection(self, pf2):
        ctx, lib = self.ctx, self.ctx.lib
        db = DeviceBatch(ctx, pf2, with_tri2=True)
        nt = max(int(pf2.tri2_off[-1]), 1)
        tri_h = ctx.zeros(nt, np.float64)
        tri_f = ctx.zeros(nt, np.uint8)
        level = ctx.zeros(pf2.n_frames, np.float64)
        nkept = ctx.zeros(pf2.n_frames, np.int32)
        status = ctx.zeros(pf2.n_frames, np.int32)
        max_tri = int(np.max(np.diff(pf2.tri2_off))) if pf2.n_frames else 0
        b = db.struct()
        _lib.check(lib.mvosr_flat_selection_batch(ctx.handle, C.byref(b), -80.0, -85.0, 0.9, tri_h.ptr, tri_f.ptr, level.ptr,
                                                  nkept.ptr, status.ptr, max_tri), "mvosr_flat_selection_batch")
        ctx.sync()
        out = tri_h.download(), tri_f.download(), level.download(), status.download()
        for buf in (tri_h, tri_f, level, nkept, status):
            buf.free()
        db.free()
        return out

    def _ransac(self, point_lists, triples):
        """point_lists: list of (M,3) arrays (M >= 12 each); triples: list of (H,3) int32, same H."""
        ctx, lib = self.ctx, self.ctx.lib
        F = len(point_lists)
        H = triples[0].shape[0]
        cnt = np.array([p.shape[0] for p in point_lists], dtype=np.int32)
        off = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int64)
        allp = np.concatenate(point_lists, axis=0)
        d = {k: ctx.to_device(np.ascontiguousarray(allp[:, i]), np.float64) for i, k in enumerate("xyz")}
        d_off, d_cnt = ctx.to_device(off, np.int64), ctx.to_device(cnt, np.int32)
        d_tri = ctx.to_device(np.ascontiguousarray(np.stack(triples)), np.int32)
        model = ctx.zeros((F, 4), np.float64)
        best = ctx.zeros(F, np.int32)
        used = ctx.zeros(F, np.int32)
        _lib.check(lib.mvosr_ransac_plane_batch(ctx.handle, F, d_off.ptr, d_cnt.ptr, d["x"].ptr, d["y"].ptr, d["z"].ptr,
                                                d_tri.ptr, H, RANSAC_THRESHOLD, RANSAC_GOAL, None, model.ptr, best.ptr,
                                                used.ptr), "mvosr_ransac_plane_batch")
        ctx.sync()
        out = model.download(), best.download(), used.download()
        for buf in list(d.values()) + [d_off, d_cnt, d_tri, model, best, used]:
            buf.free()
        return out

    # ---- reference surface ---------------------------------------------------------------------------
    def feature_selection_batch(self, feature3ds, feature2ds):
        """rescale.py:113-148 for a list of frames: returns a list of (point_selected, heights_loose)."""
        f3s = [np.asarray(a, dtype=np.float64) for a in feature3ds]
        f2s = [np.asarray(a, dtype=np.float64) for a in feature2ds]
        pf = packing.pack_features(f3s, f2s, self.vanish)                                   # :115-117
        packing.attach_tri1(pf, None, self.delaunay_workers)                                # :124
        for f, err in sorted(pf.extra["tri1_errors"].items()):
            raise err
        total, good, st = self._graph_vote(pf)
        if np.any(st == 8):
            raise _lib.MvosrLibraryError("triangulation with an out-of-range vertex id")
        low3, low2, tris, valids = [], [], [], []
        for f in range(pf.n_frames):
            sl = pf.frame_slice(f)
            with np.errstate(all="ignore"):
                valid = (good[sl] / total[sl]) > 0.5                                        # graph.py:34-35,131-132
            valids.append(valid)
            a3, a2 = f3s[f][pf.lower_index[f]], f2s[f][pf.lower_index[f]]
            tri = pf.tri1[pf.tri1_off[f]:pf.tri1_off[f + 1]]
            if self.verbose:
                print('feature rejected ', int(np.sum(~valid)))
                print('feature left     ', int(np.sum(valid)))
            if np.sum(valid) > 10:                                                          # :133-137
                a3, a2 = a3[valid], a2[valid]
                tri = None
            low3.append(a3)
            low2.append(a2)
            tris.append(tri)
        pf2 = packing.pack_features(low3, low2, -np.inf)
        need = [f for f, t in enumerate(tris) if t is None]
        if need:
            new = packing.delaunay_many([low2[f] for f in need], self.delaunay_workers)
            for f, t in zip(need, new):
                if isinstance(t, Exception):
                    raise t
                tris[f] = t
        packing.attach_tri2(pf2, tris, None)
        tri_h, tri_f, level, st2 = self._flat_selection(pf2)
        out = []
        for f in range(pf2.n_frames):
            if st2[f] == 7:
                raise np.linalg.LinAlgError("Singular matrix")                              # :79
            t0, t1 = int(pf2.tri2_off[f]), int(pf2.tri2_off[f + 1])
            fl = tri_f[t0:t1]
            ids = tris[f][(fl & 4) != 0].reshape(-1)                                          # :101
            out.append((low3[f][ids], tri_h[t0:t1][(fl & 1) != 0]))                           # :140,:102
            self.height_level = level[f]                                                    # :92
        self.last = {"valid": valids, "tris2": tris, "height_level": level, "pf2": pf2, "tri_flags": tri_f}
        return out

    def feature_selection(self, feature3d, feature2d):
        return self.feature_selection_batch([feature3d], [feature2d])[0]

    def _apply_scale(self, model):
        """rescale.py:156-178 for one frame, given the RANSAC model (or None when < 12 points)."""
        if model is not None:
            norm = np.array(model[:3], dtype=np.float64)
            h_bar = -model[3]
            if norm[1] < 0:                                                                 # :159-161 (the kernel already did)
                norm, h_bar = -norm, -h_bar
            norm_norm = np.sqrt(norm @ norm) / h_bar                                        # :162-163
            ransac_camera_height = 1 / norm_norm                                            # :165
            scale = self.absolute_reference / ransac_camera_height                          # :167
            if scale - self.scale > SLEW:                                                   # :169-174
                self.scale += SLEW
            elif scale - self.scale < -SLEW:
                self.scale -= SLEW
            else:
                self.scale = scale
        self.scale_queue.append(self.scale)                                                 # :175-177
        if len(self.scale_queue) > self.window_size:
            self.scale_queue.popleft()

    def scale_calculation_ransac_batch(self, point_lists):
        idx = [i for i, p in enumerate(point_lists) if p.shape[0] >= 12]                    # :152
        models = {}
        if idx:
            triples = [self._triples(point_lists[i].shape[0]) for i in idx]
            h = min(t.shape[0] for t in triples)
            triples = [t[:h] for t in triples]
            m, best, used = self._ransac([np.ascontiguousarray(point_lists[i]) for i in idx], triples)
            for k, i in enumerate(idx):
                models[i] = m[k]
            self.last.update(model=m, best_ic=best, used=used, ransac_frames=idx)
        raw = []
        for i in range(len(point_lists)):
            q_before = list(self.scale_queue)
            self._apply_scale(models.get(i))
            raw.append((self.scale, q_before))
        # np.median(self.scale_queue) per frame (:178) = the window-median kernel over the pushed sequence
        pushed = np.array([r[0] for r in raw], dtype=np.float64)
        filt = self.engine.window_median_host(pushed, self.window_size, raw[0][1]) if raw else np.zeros(0)
        return filt, np.ones(len(point_lists))

    def scale_calculation_ransac(self, point_selected):
        s, e = self.scale_calculation_ransac_batch([np.asarray(point_selected, dtype=np.float64)])
        return s[0], 1

    def scale_calculation(self, feature3d, feature2d, img=None):
        """rescale.py:191-193."""
        s, e = self.scale_calculation_batch([feature3d], [feature2d])
        return s[0], 1

    def scale_calculation_batch(self, feature3ds, feature2ds):
        """Equivalent to scale_calculation per frame, in order, with one launch per GPU stage."""
        sel = self.feature_selection_batch(feature3ds, feature2ds)
        return self.scale_calculation_ransac_batch([s[0] for s in sel])
