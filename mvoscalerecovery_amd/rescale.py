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

``triangulation="gpu"`` is the device-resident path: C packer -> ONE upload -> Delaunay #1 (mvosr_delaunay_batch) ->
GraphChecker vote with the decision made on the device (mvosr_graph_keep_batch) -> Delaunay #2 over the survivors, seeded
-> flat_selection + RANSAC plane in one kernel per frame (mvosr_flat_ransac_batch) -> after the last chunk the slew
limiter and the window median: a sequential recurrence over results that are on the host anyway (the statuses decide where
the reference would raise), walked by one core in C (mvosr_slew_median_host; the device form mvosr_slew_median exists and is
tested, the estimator does not use it).  No host step between the device stages.  The deterministic stages carry no deviation:
the vote's edge potential is symmetric, so its mask is a function of the triangle SET (graph.py:18-36,124-145);
flat_selection keeps a set of triangles (rescale.py:75-96); only the order of its point list (:101) follows the rows,
and the RANSAC draws list positions uniformly (ransac.py:10) — here from a counter-based sequence keyed by
``ransac_seed`` (None: OS entropy, the reference's behaviour).  A draw that names one vertex twice (the list repeats vertices:
0.5-2 % of the reference's draws) spends its iteration, as in the reference (ransac.py:8-21), and counts zero inliers — the
reference's SVD of such a rank-2 sample returns a plane picked by rounding noise, which its own recorded run shows losing to every
real hypothesis (tests/golden/rescale.npz frame 26); rounds 4-5 drew such a sample again (a declared deviation: gone).  The scale
distribution is compared with the unseeded reference's in tests/test_gpu_rescale.py.  ``triangulation="scipy", sampling="device"`` runs the
same kernels on SciPy's triangulations brought to the same row form, with bit-identical results.
"""
from __future__ import annotations

import ctypes as C
import os
import random
import time
from collections import deque

import numpy as np

from . import _lib
from . import constants as K
from . import packing
from .engine import DeviceBatch, ScaleEngine, frame_tables, pack_upload_native

EDGE_POTENTIAL = [[3, 1], [2, 2], [2, 2], [0, 4]]     # rescale.py:32
VANISH = 185                                          # rescale.py:30
RANSAC_ITERATIONS = 100                               # rescale.py:155
RANSAC_THRESHOLD = 0.005                              # rescale.py:155
RANSAC_GOAL = 0.8                                     # estimate_road_norm.py:68
RANSAC_MIN_POINTS = 12                                # rescale.py:152
MIN_VALID_FOR_RETRI = 10                              # rescale.py:133
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
                 delaunay_workers=None, verbose=False, triangulation=None, sampling=None):
        # reference attributes (rescale.py:24-35)
        self.absolute_reference = absolute_reference
        self.camera_pitch = 0
        self.scale = 1
        self.inliers = None
        self.scale_queue = deque()
        # (the reference takes any window; mvosr_slew_median_host's ring buffer holds 64 — other windows walk the recurrence in Python,
        # _push_device)
        self.window_size = window_size
        self.vanish = VANISH
        # build-side state
        self.verbose = verbose
        self.delaunay_workers = delaunay_workers
        if delaunay_workers is None or delaunay_workers > 1:
            packing.start_pool(delaunay_workers)     # fork the host stage's workers BEFORE the GPU runtime starts its threads
        self.engine = ScaleEngine(absolute_reference, device=device, camera_pitch=0.0)
        self.ctx = self.engine.ctx
        self._good_bits = good_bits()
        self._rng = random.Random(ransac_seed)
        self._sampler = sampler
        self.height_level = None
        self.last = {}
        # triangulation "scipy": both Delaunay calls on the host, as the reference; "gpu": mvosr_delaunay_batch, everything
        # device-resident.  sampling "host": the triples are drawn here with Python's random, by LIST POSITION in SciPy's row
        # order (replays a recorded reference run); "device": the kernel's counter-based sequence over rows in canonical
        # form (the default — and the only choice — with "gpu").
        # No triangulation given (the reference's own construction, /root/reference/src/main.py:55): the device-resident path
        # (round 5; the reference's RANSAC is unseeded, so no realisation of its draws is THE result — DESIGN.md §3.4 —;
        # MVOSR_TRIANGULATION=scipy restores the host default, and a host `sampler` or sampling="host" implies it).
        if triangulation is None:
            triangulation = "scipy" if (sampler is not None or sampling == "host") else os.environ.get("MVOSR_TRIANGULATION", "gpu")
        if triangulation not in ("scipy", "gpu"):
            raise ValueError("triangulation must be 'scipy' or 'gpu'")
        if sampling is None:
            sampling = "device" if triangulation == "gpu" else "host"
        if sampling not in ("host", "device"):
            raise ValueError("sampling must be 'host' or 'device'")
        if triangulation == "gpu" and sampling == "host":
            raise ValueError("triangulation='gpu' draws its samples on the device (the point list never visits the host)")
        if sampling == "device" and sampler is not None:
            raise ValueError("a host sampler needs sampling='host'; the device path takes id_triples per call")
        self.triangulation, self.sampling = triangulation, sampling
        self._seed = int.from_bytes(os.urandom(8), "little") if ransac_seed is None else int(ransac_seed) & ((1 << 64) - 1)
        self._frame_counter = 0                     # frames this estimator has processed: the sample sequence's counter
        self.stage_outputs = False                  # device path, per-frame calls: keep masks / rows / flags in self.last
        self.last_declined = 0

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
        s, e = self.scale_calculation_batch([feature3d], [feature2d], stage=self.stage_outputs)
        return s[0], 1

    def scale_calculation_batch(self, feature3ds, feature2ds, id_triples=None, stage=False):
        """Equivalent to scale_calculation per frame, in order.  sampling="host": one launch per GPU stage with host
        steps between them.  sampling="device": the batch streams through the device-resident stages in chunks
        (``id_triples``: per frame an (H,3) array of survivor-numbered vertex ids replacing the draw — a recorded
        reference run mapped to point ids; ``stage``: keep the per-frame stage outputs in ``self.last``)."""
        if self.sampling == "device":
            return self._stream_device(feature3ds, feature2ds, id_triples, stage)
        if id_triples is not None:
            raise ValueError("id_triples belong to sampling='device'")
        sel = self.feature_selection_batch(feature3ds, feature2ds)
        return self.scale_calculation_ransac_batch([s[0] for s in sel])

    # ---- the device-resident path ------------------------------------------------------------------------------
    GPU_CHUNK = 8192            # frames per chunk, at most (a call of F frames uses chunks of F/4, 512 at least)
    GPU_RESIDENT = 512          # frames the GPU works on at once (two 8-wavefront workgroups per CU)
    GPU_CHUNK_POINTS = 10000000 # ... and features per chunk
    GPU_RAMP_FRACTIONS = (1 / 6, 1 / 3, 1 / 2, 2 / 3, 5 / 6)   # the short first chunks, as fractions of a full one (scale_calculator.py)
    GPU_PIPELINE = 2            # chunks queued on the device behind the one being collected
    GPU_SIDE_DOWNLOADS = os.environ.get("MVOSR_SIDE_DOWNLOADS", "1") != "0"   # streamed batches: results to page-locked memory by a copy kernel, no copy engine (scale_calculator.ScaleEstimator.GPU_SIDE_DOWNLOADS)
    N_HYP = RANSAC_ITERATIONS

    def _rescale_params(self, frame_base):
        return _lib.RescaleParams(self._good_bits, MIN_VALID_FOR_RETRI, -80.0, -85.0, 0.9, RANSAC_MIN_POINTS, self.N_HYP,
                                  RANSAC_THRESHOLD, RANSAC_GOAL, float(self.absolute_reference), self._seed, int(frame_base))

    def _launch_flat_ransac(self, db, keep_ptr, dt2_ptr, frame_base, frame_ids, id_triples, stage, max_tri, ctx=None):
        """mvosr_flat_ransac_batch over a DeviceBatch; returns the outputs' block (download queued)."""
        ctx, F, H = ctx or self.ctx, db.n_frames, self.N_HYP
        spec = [("raw_scale", F, np.float64), ("height_level", F, np.float64), ("model", (F, 4), np.float64),
                ("best_ic", F, np.int32), ("used", F, np.int32), ("n_kept", F, np.int32), ("status", F, np.int32)]
        out = ctx.block(spec)
        side = []
        o = _lib.RescaleOutputs(out["raw_scale"].ptr, out["height_level"].ptr, out["model"].ptr, out["best_ic"].ptr,
                                out["used"].ptr, out["n_kept"].ptr, out["status"].ptr, None, None, None)
        flags = None
        if stage:
            flags = ctx.block([("tri_flags", max(db.n_rows2, 1), np.uint8)])
            o.tri_flags = flags["tri_flags"].ptr
        d_ids = d_tr = None
        if frame_ids is not None:
            d_ids = ctx.block([("frame_ids", F, np.int64)])
            d_ids.upload({"frame_ids": np.asarray(frame_ids, dtype=np.int64)})
            side.append(d_ids)
        if id_triples is not None:
            t = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.int32).reshape(H, 3) for x in id_triples]))
            d_tr = ctx.block([("id_triples", (F, H, 3), np.int32)])
            d_tr.upload({"id_triples": t})
            side.append(d_tr)
        rp = self._rescale_params(frame_base)
        b = db.struct()
        _lib.check(ctx.lib.mvosr_flat_ransac_batch(ctx.handle, C.byref(b), keep_ptr, C.byref(rp),
                                                   d_tr["id_triples"].ptr if d_tr is not None else None,
                                                   d_ids["frame_ids"].ptr if d_ids is not None else None, dt2_ptr,
                                                   C.byref(o), int(max_tri)), "mvosr_flat_ransac_batch")
        # (the results reach their page-locked image through a copy KERNEL: a hipMemcpyAsync queued here would park a copy engine behind
        # this chunk's kernels, and an upload of the next chunk that lands on that engine waits with it — DeviceBlock.mark_done; the
        # per-frame call with stage outputs keeps the queued download)
        if stage or not self.GPU_SIDE_DOWNLOADS:
            out.prefetch()
        else:
            out.mark_done()
        out.mark(True)
        for blk in side:
            blk.mark(True)
        return out, flags, side

    def _chunk_dev_gpu(self, f3s, f2s, frame_base, id_triples, stage, tables=False, early_status=False):
        """One chunk, both triangulations on the device: pack (C packer into page-locked memory) -> ONE upload -> every
        launch and the download of the results queued; nothing is waited for here."""
        ctx, lib = self.ctx, self.ctx.lib
        if tables is False:
            tables = frame_tables(f3s, f2s) if len(f3s) else None
        if tables is not None:
            pf, blk = pack_upload_native(ctx, f3s, f2s, self.vanish, None, tables=tables)             # rescale.py:115-117
            self._trace("packed", -1, len(f3s))
            if pf.max_feat > self._max_points():           # (_stream_device cuts the stream before such a frame: a direct caller's)
                blk.free()
                self._refuse_oversized(pf, frame_base)
            db = DeviceBatch(ctx, pf, with_tri2=False, device_triangulation=True, uploaded=blk)
            self._trace("blocks", -1, len(f3s))
        else:
            pf = packing.pack_features(f3s, f2s, self.vanish)
            if pf.n_frames == 0:
                return {"gpu": False}
            if pf.max_feat > self._max_points():
                self._refuse_oversized(pf, frame_base)
            db = DeviceBatch(ctx, pf, with_tri2=False, device_triangulation=True)
        bufs = db.bufs
        db.info.invalidate()
        try:
            _lib.check(lib.mvosr_delaunay_batch_ex(ctx.handle, db.n_frames, bufs["feat_off"].ptr, bufs["feat_cnt"].ptr, bufs["u"].ptr,
                                                   bufs["v"].ptr, None, int(db.max_feat), bufs["tri_off"].ptr, bufs["tri1"].ptr,
                                                   bufs["tri1_cnt"].ptr, None, bufs["dt1_status"].ptr, None, None, None, None,
                                                   bufs["dt_info"].ptr), "mvosr_delaunay_batch_ex (first triangulation)")
            if early_status:                       # (the call's last chunk: what this kernel declined, known right behind it — _chunk_dev_finish)
                db._queue_early_status()
            bs = db.struct()
            keep = bufs["vote_counters"]                                        # (the block's per-feature int32 plane: here the keep flags)
            _lib.check(lib.mvosr_graph_keep_batch(ctx.handle, C.byref(bs), C.c_uint32(self._good_bits), MIN_VALID_FOR_RETRI,
                                                  bufs["dt1_status"].ptr, keep.ptr, None, None), "mvosr_graph_keep_batch")   # graph.py:18-36, rescale.py:133
            _lib.check(lib.mvosr_delaunay_batch_ex(ctx.handle, db.n_frames, bufs["feat_off"].ptr, bufs["feat_cnt"].ptr, bufs["u"].ptr,
                                                   bufs["v"].ptr, keep.ptr, int(db.max_feat), bufs["tri_off"].ptr, bufs["tri2"].ptr,
                                                   bufs["tri2_cnt"].ptr, bufs["n2_expected"].ptr, bufs["dt2_status"].ptr,
                                                   bufs["tri_off"].ptr, bufs["tri1"].ptr, bufs["tri1_cnt"].ptr, bufs["dt_info"].ptr, None),
                       "mvosr_delaunay_batch_ex (second triangulation)")           # rescale.py:134-137
        except _lib.MvosrAllocError:
            # (MVOSR_ERR_ALLOC: the triangulation's workspace could not be allocated; nothing was launched — the host's path)
            db.free()
            self.alloc_fallbacks = getattr(self, "alloc_fallbacks", 0) + 1
            return {"gpu": False}
        db.n_rows2 = 2 * pf.total_padded
        out, flags, side = self._launch_flat_ransac(db, keep.ptr, bufs["dt2_status"].ptr, frame_base, None, id_triples, stage,
                                                    2 * int(db.max_feat))
        if stage or not self.GPU_SIDE_DOWNLOADS:
            db.prefetch_info()
        else:
            db.mark_info_done()
        db.mark()
        return {"gpu": True, "pf": pf, "db": db, "out": out, "flags": flags, "side": side}

    def _trace(self, what, chunk, frames):
        """MVOSR_TRACE_CHUNKS=1: (what, chunk, frames, seconds since the call began) into ``self.chunk_trace`` — when this process
        started and finished each chunk's pack + launches and each collection (profiles/e2e_host_trace.py reads it)."""
        tr = getattr(self, "chunk_trace", None)
        if tr is not None:
            tr.append((what, chunk, frames, time.perf_counter() - getattr(self, "_trace_t0", 0.0)))

    def _oversized_error(self, frame, count):
        """flat_selection + RANSAC hold a frame's survivors, heights and flags in ONE workgroup's LDS: a frame beyond that has no
        kernel here (the reference takes any size — /root/reference/src/rescale.py:113-148 — at seconds per frame).  Said with the
        frame named, instead of a library error from the middle of the host path."""
        return ValueError("rescale.ScaleEstimator: frame %d has %d features below the vanishing row; the device-resident path takes at "
                          "most %d per frame (one workgroup's LDS).  Thin the frame, or use scale_calculator.ScaleEstimator, whose "
                          "dense kernels take any size." % (frame, count, self._max_points()))

    def _refuse_oversized(self, pf, frame_base):
        cnt = np.asarray(pf.feat_cnt)
        f = int(np.argmax(cnt > self._max_points()))
        raise self._oversized_error(frame_base - self._frame_counter + f, int(cnt[f]))

    def _first_oversized(self, feature2ds, lens, a):
        """Index (within the call) of the first frame of ``feature2ds[a:a + len(lens)]`` with more features below the vanishing row
        than the device path takes, and that count — or None.  ``lens``: the frames' sizes (an upper bound of that count, so the
        frames are looked at only where a size exceeds the limit: never, in an ordinary call)."""
        cap = self._max_points()
        for j in np.nonzero(np.asarray(lens) > cap)[0]:
            f2 = np.asarray(feature2ds[a + int(j)], dtype=np.float64)
            below = int(np.count_nonzero(f2[:, 1] > self.vanish)) if f2.ndim == 2 and f2.size else 0
            if below > cap:
                return a + int(j), below
        return None

    def _resident_frames(self, max_pts):
        """Frames the triangulation kernel works on at a time (mvosr_delaunay_frames_per_cu x CUs)."""
        return max(self.GPU_RESIDENT, int(self.ctx.lib.mvosr_delaunay_frames_per_cu(int(max_pts))) * int(self.ctx.n_cu))

    def _max_points(self):
        """Largest frame the device-resident kernels take (flat_selection + RANSAC hold a frame's survivors, rows'
        heights and flags in one workgroup's LDS: 42 B per feature + 12 KB)."""
        return min(packing.delaunay_gpu_max_points(), int(self.ctx.lib.mvosr_delaunay_lds_points()),
                   (self.ctx.lds_per_block - 14000 - 36 * self.N_HYP) // 43)

    def _chunk_dev_host(self, f3s, f2s, frame_base, frame_ids, id_triples, stage):
        """The same kernels on the host's triangulations (SciPy's rows in canonical form): triangulation="scipy" with
        sampling="device", and the frames the device triangulation declined.  Synchronous: the four steps below, one after the other."""
        rec = self._host_begin(f3s, f2s, frame_base)
        self._host_keep_start(rec)
        self._host_tri2_start(rec)
        self._host_flat_start(rec, frame_base, frame_ids, id_triples, stage)
        return self._host_collect(rec, stage)

    # The host path in steps none of which waits for more than its own inputs (round 6: a deferred re-run advances through them while
    # the call's later chunks run, _advance_deferred): packed + first triangulations started -> (those back) uploaded, the vote
    # launched, its keep words on their way -> (those back) second triangulations started -> (those back) uploaded, flat_selection +
    # RANSAC launched, results on their way -> collected.
    def _host_begin(self, f3s, f2s, frame_base, slot=0, background=False, ctx=None):
        pf = packing.pack_features(f3s, f2s, self.vanish)                                   # rescale.py:115-117
        if pf.n_frames and pf.max_feat > self._max_points():
            self._refuse_oversized(pf, frame_base if np.isscalar(frame_base) else self._frame_counter)
        pf.extra["canonical"] = True
        return {"pf": pf, "h1": packing.submit_tri1(pf, self.delaunay_workers, slot=slot, background=background), "step": 0,   # :124
                "ctx": ctx or self.ctx}

    def _redo_context(self):
        """A second context on the device (its own stream, workspace, caches) for the re-runs that advance while later chunks run: on the
        estimator's own stream their few launches would sit BEHIND the chunks already queued — two chunks, 8 ms each — at every step."""
        if getattr(self, "_redo_ctx", None) is None:
            self._redo_ctx = _lib.Context(self.ctx.device)
        return self._redo_ctx

    def _host_keep_start(self, rec):
        ctx, lib, pf = rec["ctx"], self.ctx.lib, rec["pf"]
        packing.attach_tri1(pf, rec.pop("h1"), self.delaunay_workers)
        db = DeviceBatch(ctx, pf, with_tri2=False)
        aux = ctx.block([("keep", max(pf.total_padded, 1), np.int32)])
        bs = db.struct()
        _lib.check(lib.mvosr_graph_keep_batch(ctx.handle, C.byref(bs), C.c_uint32(self._good_bits), MIN_VALID_FOR_RETRI, None,
                                              aux["keep"].ptr, None, None), "mvosr_graph_keep_batch")
        aux.mark_done()
        rec.update(db=db, aux=aux, step=1)

    def _host_tri2_start(self, rec, slot=1, background=False):
        pf = rec["pf"]
        keep = rec["aux"]["keep"].download()
        rec["keep"] = keep
        rec["masks"] = [keep[pf.frame_slice(f)] >= 0 for f in range(pf.n_frames)]
        rec["h2"] = packing.submit_tri2(pf, rec["masks"], self.delaunay_workers, slot=slot, background=background)   # :134-137
        rec["step"] = 2

    def _host_flat_start(self, rec, frame_base, frame_ids, id_triples, stage):
        pf, db = rec["pf"], rec["db"]
        packing.attach_tri2(pf, rec.pop("h2"), rec["masks"], self.delaunay_workers)
        db.set_tri2(pf)
        db.n_rows2 = int(pf.tri2_off[-1])
        max_tri = int(np.max(np.diff(pf.tri2_off))) if pf.n_frames else 0
        rec["out"], rec["flags"], side = self._launch_flat_ransac(db, rec["aux"]["keep"].ptr, None, frame_base, frame_ids, id_triples, stage,
                                                                  max(max_tri, 1), ctx=rec["ctx"])
        rec["side"] = side + [rec["aux"]]
        rec["step"] = 3

    def _host_collect(self, rec, stage):
        pf = rec["pf"]
        host_errors = dict(pf.extra["tri2_errors"])
        host_errors.update(pf.extra["tri1_errors"])
        res = self._collect(rec["out"], pf.n_frames)
        st = {"pf": pf, "db": rec["db"], "out": rec["out"], "flags": rec["flags"], "side": rec["side"], "host_errors": host_errors, "keep": rec["keep"]}
        if stage:
            res["stage"] = self._stage_outputs(st, host=True)
        self._free_chunk(st)
        res["host_errors"] = host_errors
        return res

    @staticmethod
    def _collect(out, F):
        return {k: out[k].download() for k in ("raw_scale", "height_level", "model", "best_ic", "used", "n_kept", "status")}

    def _stage_outputs(self, st, host=False):
        """Per frame: the vote's mask (graph.py:35), the second triangulation's rows and flat_selection's flags."""
        pf, db = st["pf"], st["db"]
        F = pf.n_frames
        fl = st["flags"]["tri_flags"].download()
        if host:
            keep = st["keep"]
            rows = [pf.tri2[pf.tri2_off[f]:pf.tri2_off[f + 1]] for f in range(F)]
            flags = [fl[pf.tri2_off[f]:pf.tri2_off[f + 1]] for f in range(F)]
        else:
            keep = db.bufs["vote_counters"].download()
            tri2, cnt2 = db.bufs["tri2"].download(), db.bufs["tri2_cnt"].download()
            toff = 2 * np.asarray(pf.feat_off, dtype=np.int64)
            rows = [tri2[toff[f]:toff[f] + cnt2[f]] for f in range(F)]
            flags = [fl[toff[f]:toff[f] + cnt2[f]] for f in range(F)]
        return {"valid": [keep[pf.frame_slice(f)] > 0 for f in range(F)], "tris2": rows, "tri_flags": flags}

    @staticmethod
    def _free_chunk(st):
        for k in ("out", "flags"):
            if st.get(k) is not None:
                st[k].free()
        for blk in st.get("side") or []:
            blk.free()
        if st.get("db") is not None:
            st["db"].free()

    def _chunk_dev_finish(self, st, f3s, f2s, frame_base, id_triples, stage, defer=None):
        """Results of a chunk started by ``_chunk_dev_gpu``; frames whose triangulation the device declined (degenerate point
        sets, fewer than 3 points) are redone on the host's triangulations with their own sample counters — at once, or, with
        ``defer`` (a list; streamed batches), in ONE batch after the call's last chunk (``_finish_deferred``): the re-run is waited
        for, and between the chunks that wait drained the pipeline (80 declined frames in 16 384: 529 -> 93 k frames/s in the
        sibling estimator's fixed mode, scale_calculator._chunk_gpu_finish)."""
        F = len(f3s)
        if not st["gpu"]:
            return self._chunk_dev_host(f3s, f2s, frame_base, None, id_triples, stage)
        db = st["db"]
        early = None
        if defer is not None and not stage and self.GPU_REDO_EARLY:
            # the call's LAST chunk: the frames its first triangulation declined are known behind that kernel.  This thread has nothing
            # else to do: it takes them through their first triangulation (waited for — the chunk's other kernels are running), the
            # vote (on the re-runs' context) and the start of their second triangulation now; the call's end launches the rest
            s1e = db.early_status()
            if s1e is not None and s1e.any() and int(np.count_nonzero(s1e)) <= self.GPU_REDO_EARLY_MAX:
                ef = np.nonzero(s1e != 0)[0]
                rec = self._host_begin([f3s[f] for f in ef], [f2s[f] for f in ef], 0, slot=24, background=True, ctx=self._redo_context())
                rec["ids"] = frame_base + ef
                rec["triples"] = None if id_triples is None else [id_triples[f] for f in ef]
                self._host_keep_start(rec)
                self._host_tri2_start(rec, slot=25, background=True)
                early = (ef, rec)
                self.redo_early_status_hits = getattr(self, "redo_early_status_hits", 0) + 1
        if defer and self.GPU_REDO_EARLY and not stage:
            # (re-runs of earlier chunks are under way: the thread looks in on them every 0.2 ms instead of sleeping until this chunk's
            # results arrive — each step of a record is taken the moment its inputs are there)
            while any(item[6] is not None and item[6]["step"] < 3 for item in defer) and not (st["out"].ready() and db.info.ready()):
                self._advance_deferred(defer)
                time.sleep(2e-4)
        res = self._collect(st["out"], F)
        s1, s2 = db.triangulation_status()
        redo = np.nonzero((s1 != 0) | (s2 != 0))[0]
        self.last_declined += len(redo)
        if stage:
            res["stage"] = self._stage_outputs(st)
        self._free_chunk(st)
        res["host_errors"] = {}
        if early is not None:
            if defer is not None and len(redo) == len(early[0]) and np.array_equal(redo, early[0]):
                defer.append((res, redo, frame_base, f3s, f2s, id_triples, early[1]))      # (continues where the early steps left it)
                return res
            self._free_chunk({"db": early[1].get("db"), "side": [early[1]["aux"]]})         # (more frames than the early read knew of: the merged way)
        if len(redo) and defer is not None:
            rec = None
            if self.GPU_REDO_EARLY and not stage and len(redo) <= self.GPU_REDO_EARLY_MAX and len(defer) < 8:
                # (a few frames: their re-run starts now — first triangulations on the pool — and advances while later chunks run)
                rec = self._host_begin([f3s[f] for f in redo], [f2s[f] for f in redo], 0, slot=8 + len(defer), background=True,
                                       ctx=self._redo_context())
                rec["ids"] = frame_base + redo
                rec["triples"] = None if id_triples is None else [id_triples[f] for f in redo]
                self.redo_early_started = getattr(self, "redo_early_started", 0) + 1
            defer.append((res, redo, frame_base, f3s, f2s, id_triples, rec))
            self._advance_deferred(defer)
            return res
        if defer:
            self._advance_deferred(defer)
        if len(redo):
            sub = self._chunk_dev_host([f3s[f] for f in redo], [f2s[f] for f in redo], 0, frame_base + redo,
                                       None if id_triples is None else [id_triples[f] for f in redo], stage)
            for k in ("raw_scale", "height_level", "model", "best_ic", "used", "n_kept", "status"):
                res[k][redo] = sub[k]
            res["host_errors"] = {int(redo[j]): e for j, e in sub["host_errors"].items()}
            if stage:
                for k in ("valid", "tris2", "tri_flags"):
                    for j, f in enumerate(redo):
                        res["stage"][k][f] = sub["stage"][k][j]
        return res

    GPU_REDO_EARLY = True           # a deferred re-run of at most GPU_REDO_EARLY_MAX frames is started at once and advanced while later chunks run
    GPU_REDO_EARLY_MAX = 16

    def _advance_deferred(self, deferred):
        """Started re-runs one step further wherever that step would not wait (see the _host_* steps); their launches go to a context of
        their own (_redo_context) — nothing here blocks."""
        for k, item in enumerate(deferred):
            rec = item[6]
            if rec is None:
                continue
            if rec["step"] == 0 and rec["h1"].ready():
                self._host_keep_start(rec)
            elif rec["step"] == 1 and rec["aux"].ready():
                self._host_tri2_start(rec, slot=16 + k, background=True)
            elif rec["step"] == 2 and rec["h2"].ready():
                self._host_flat_start(rec, 0, rec["ids"], rec["triples"], False)
                self.redo_early_launched = getattr(self, "redo_early_launched", 0) + 1

    def _finish_deferred(self, deferred, stage):
        """The declined frames of every chunk of a call through the host's triangulations, scattered back: the started re-runs each
        finish their remaining steps (launches first, then the results), the others go in one merged batch."""
        if not deferred:
            return
        self._advance_deferred(deferred)
        started = [(k, item) for k, item in enumerate(deferred) if item[6] is not None]
        for k, item in started:
            if item[6]["step"] == 0:
                self._host_keep_start(item[6])
        for k, item in started:
            if item[6]["step"] == 1:
                self._host_tri2_start(item[6], slot=16 + k)
        for k, item in started:
            if item[6]["step"] == 2:
                self._host_flat_start(item[6], 0, item[6]["ids"], item[6]["triples"], stage)
        for k, item in started:
            res, redo = item[0], item[1]
            sub = self._host_collect(item[6], stage)
            for key in ("raw_scale", "height_level", "model", "best_ic", "used", "n_kept", "status"):
                res[key][redo] = sub[key]
            for j, e in sub["host_errors"].items():
                res["host_errors"][int(redo[j])] = e
        deferred = [item[:6] for item in deferred if item[6] is None]
        if not deferred:
            return
        f3_all, f2_all, ids, trs, where = [], [], [], [], []
        for k, (res, redo, frame_base, f3s, f2s, id_triples) in enumerate(deferred):
            for f in redo:
                f3_all.append(f3s[f]); f2_all.append(f2s[f]); ids.append(frame_base + int(f)); where.append((k, int(f)))
                trs.append(None if id_triples is None else id_triples[f])
        sub = self._chunk_dev_host(f3_all, f2_all, 0, np.asarray(ids, dtype=np.int64), None if trs[0] is None else trs, stage)
        for i, (k, f) in enumerate(where):
            res = deferred[k][0]
            for key in ("raw_scale", "height_level", "model", "best_ic", "used", "n_kept", "status"):
                res[key][f] = sub[key][i]
            if i in sub["host_errors"]:
                res["host_errors"][f] = sub["host_errors"][i]
            if stage:
                for key in ("valid", "tris2", "tri_flags"):
                    res["stage"][key][f] = sub["stage"][key][i]

    def _stream_device(self, feature3ds, feature2ds, id_triples, stage, push=True):
        F = len(feature3ds)
        if F == 0:
            return np.zeros(0), np.zeros(0)
        base = self._frame_counter
        self.last_declined = 0
        results, bounds = [], []
        # A frame the device path has no kernel for (_oversized_error) ends the run like every other error (_push_device): the frames
        # before it go through the slew limiter, the window and the sample counter, nothing queued is dropped, then the exception —
        # what the reference's per-frame loop would have left behind.
        oversized = None
        if self.triangulation != "gpu":
            oversized = self._first_oversized(feature2ds, [len(x) for x in feature2ds], 0)
            if oversized is not None:
                F = oversized[0]
        if self.triangulation == "gpu":
            C_ = int(min(self.GPU_CHUNK, max(512, -(-F // 4))))
            mean_pts = max(1, sum(len(x) for x in feature3ds[:64]) // min(F, 64))
            C_ = int(max(512, min(C_, self.GPU_CHUNK_POINTS // mean_pts)))     # (the points cap as it will bite: the short first chunks are fractions of that)
            # short first chunks (C/8, C/4, C/2): the GPU starts after the pack + upload of an eighth of a chunk
            ramp = [int(C_ * x) for x in self.GPU_RAMP_FRACTIONS] if (C_ >= 2048 and F >= 3 * C_) else []
            queue, a = [], 0
            deferred = []                                  # chunks with declined frames: their re-run waits for the call's last chunk
            # MVOSR_TRACE_CHUNKS=1: when this process started and finished each chunk's pack + launches and each collection
            # (self.chunk_trace: (what, chunk, frames, seconds since the call began))
            self.chunk_trace = [] if os.environ.get("MVOSR_TRACE_CHUNKS") else None
            self._trace_t0 = time.perf_counter()
            while a < F:
                b = min(F, a + (ramp[len(bounds)] if len(bounds) < len(ramp) else C_))
                tb = frame_tables(feature3ds[a:b], feature2ds[a:b])          # (sizes from the packer's pointer tables: one C loop)
                lens = tb[2].astype(np.int64) if tb is not None else np.fromiter((len(x) for x in feature3ds[a:b]), dtype=np.int64, count=b - a)
                if oversized is None:
                    oversized = self._first_oversized(feature2ds, lens, a)
                    if oversized is not None:
                        F = oversized[0]                   # (the stream ends before that frame)
                        if a >= F:
                            break
                        b, lens = min(b, F), lens[:F - a]
                        tb = tuple(t[:F - a] for t in tb) if tb is not None else None
                b = min(b, a + max(int(np.searchsorted(np.cumsum(lens), self.GPU_CHUNK_POINTS, side="right")), 1))
                while b - a > 1 and (b - a) * int(lens[:b - a].max()) > 2 * self.GPU_CHUNK_POINTS:     # (workspace = frames x largest frame)
                    b = a + max(1, (b - a) // 2)
                res = self._resident_frames(int(lens[:b - a].max()))
                if b < F and b - a >= 2 * res:                         # whole rounds of the GPU's resident frames: no partly filled last round
                    b = a + ((b - a) // res) * res
                tr = None if id_triples is None else id_triples[a:b]
                self._trace("launch", len(bounds), b - a)
                queue.append((self._chunk_dev_gpu(feature3ds[a:b], feature2ds[a:b], base + a, tr, stage,
                                                  tables=(tuple(t[:b - a] for t in tb) if tb is not None else None),
                                                  early_status=self.GPU_REDO_EARLY and not stage and b == F and a > 0), a, b))
                bounds.append((a, b))
                self._trace("launched", len(bounds) - 1, b - a)
                while len(queue) > self.GPU_PIPELINE:
                    st, pa, pb = queue.pop(0)
                    self._trace("collect", len(results), pb - pa)
                    results.append(self._chunk_dev_finish(st, feature3ds[pa:pb], feature2ds[pa:pb], base + pa,
                                                          None if id_triples is None else id_triples[pa:pb], stage, defer=deferred))
                a = b
            while queue:
                st, pa, pb = queue.pop(0)
                results.append(self._chunk_dev_finish(st, feature3ds[pa:pb], feature2ds[pa:pb], base + pa,
                                                      None if id_triples is None else id_triples[pa:pb], stage, defer=deferred))
            self._finish_deferred(deferred, stage)
        else:
            C_ = 2048
            for a in range(0, F, C_):          # (F: the frames before the first oversized one)
                b = min(F, a + C_)
                bounds.append((a, b))
                results.append(self._chunk_dev_host(feature3ds[a:b], feature2ds[a:b], base + a, None,
                                                    None if id_triples is None else id_triples[a:b], stage))
        if oversized is not None and not results:
            raise self._oversized_error(*oversized)
        cat = lambda k: np.concatenate([r[k] for r in results])
        raw, status, level = cat("raw_scale"), cat("status"), cat("height_level")
        host_errors = {}
        for (a, _), r in zip(bounds, results):
            host_errors.update({a + f: e for f, e in r["host_errors"].items()})
        self.last = {"model": cat("model"), "best_ic": cat("best_ic"), "used": cat("used"), "n_kept": cat("n_kept"),
                     "status": status, "height_level": level, "raw_scale": raw}
        if stage:
            for k in ("valid", "tris2", "tri_flags"):
                self.last[k] = [x for r in results for x in r["stage"][k]]
        if oversized is not None:
            host_errors = dict(host_errors)
            host_errors[F] = self._oversized_error(*oversized)
            raw = np.concatenate([raw, [np.nan]])
            status = np.concatenate([status, np.array([K.ST_ERR_EMPTY], dtype=status.dtype)])
            level = np.concatenate([level, [np.nan]])
        if not push:
            if oversized is not None:
                raise host_errors[F]
            return raw, status, level, host_errors
        return self._push_device(raw, status, level, host_errors)

    # ---- the two halves on their own: what a driver that shards a sequence over GPUs needs (offline.run_sequence_sharded) ----
    def raw_scale_batch(self, feature3ds, feature2ds, frame_base=0):
        """Per-frame half only (no cross-frame state touched): ``(raw_scale[F], status[F], height_level[F], host_errors)``.
        ``frame_base``: position of the block's first frame in the whole sequence — the sample sequence is keyed by the
        frame's position, so a sequence split over ranks draws the triples the single-process run draws."""
        if self.sampling != "device":
            raise ValueError("the sharded driver needs the device sample sequence: ScaleEstimator(..., triangulation='gpu') "
                             "or (..., sampling='device')")
        if len(feature3ds) == 0:
            return np.zeros(0), np.zeros(0, dtype=np.int32), np.zeros(0), {}
        keep, self._frame_counter = self._frame_counter, self._frame_counter + int(frame_base)
        try:
            return self._stream_device(feature3ds, feature2ds, None, False, push=False)
        finally:
            self._frame_counter = keep

    def push_raw_scales(self, raw, status, level=None, host_errors=None):
        """Cross-frame half (slew limiter, window median, raise sites) for raw scales computed elsewhere (other ranks):
        returns ``(scales, stds)`` like ``scale_calculation_batch``."""
        raw = np.asarray(raw, dtype=np.float64)
        status = np.asarray(status, dtype=np.int32)
        level = np.full(len(raw), np.nan) if level is None else np.asarray(level, dtype=np.float64)
        return self._push_device(raw, status, level, host_errors or {})

    def _push_device(self, raw, status, level, host_errors):
        """The cross-frame tail (rescale.py:169-178) on the device over the frames before the first one at which the
        reference raises; then that frame's exception."""
        F = len(raw)
        bad = np.isin(status, (K.ST_ERR_SINGULAR, K.ST_ERR_MASK, K.ST_ERR_EMPTY))
        for f in host_errors:
            bad[f] = True
        n_ok = int(np.argmax(bad)) if bad.any() else F
        ctx = self.ctx
        filtered = np.zeros(0)
        if n_ok:
            # (the recurrence is sequential and the raw scales are on the host already — the statuses decide where the
            # reference raises —: one core walks it in C, mvosr_slew_median_host; mvosr_slew_median is the same on device arrays)
            r = np.ascontiguousarray(raw[:n_ok], dtype=np.float64)
            ap = np.ascontiguousarray(status[:n_ok] == 0, dtype=np.int32)
            pushed, filtered = np.empty(n_ok, np.float64), np.empty(n_ok, np.float64)
            q = np.ascontiguousarray(np.asarray(list(self.scale_queue), dtype=np.float64))
            if 1 <= int(self.window_size) <= 64 and q.size <= 64:
                _lib.check(ctx.lib.mvosr_slew_median_host(_lib.addr(r), _lib.addr(ap), n_ok, SLEW, float(self.scale), int(self.window_size),
                                                          _lib.addr(q) if q.size else None, int(q.size), _lib.addr(pushed),
                                                          _lib.addr(filtered), None), "mvosr_slew_median_host")
                self.scale = float(pushed[-1])                                              # :169-174
                tail = list(self.scale_queue) + list(pushed[max(0, n_ok - self.window_size):])
                self.scale_queue.clear()
                self.scale_queue.extend(tail[-self.window_size:])                           # :175-177
            else:
                # a window the C loop's ring buffer does not hold (the reference takes any: 0 gives the median of an empty deque,
                # nan): the same recurrence, frame by frame, as the reference writes it (:169-178)
                with np.errstate(all="ignore"):
                    for i in range(n_ok):
                        if ap[i]:
                            if r[i] - self.scale > SLEW:
                                self.scale += SLEW
                            elif r[i] - self.scale < -SLEW:
                                self.scale -= SLEW
                            else:
                                self.scale = float(r[i])
                        self.scale_queue.append(self.scale)
                        if len(self.scale_queue) > self.window_size:
                            self.scale_queue.popleft()
                        pushed[i] = self.scale
                        import warnings
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore")
                            filtered[i] = np.median(self.scale_queue)
            self.height_level = level[n_ok - 1]                                             # :92
        self._frame_counter += n_ok + (1 if n_ok < F else 0)
        if n_ok < F:
            if n_ok in host_errors and isinstance(host_errors[n_ok], Exception):
                raise host_errors[n_ok]
            if status[n_ok] == K.ST_ERR_SINGULAR:
                raise np.linalg.LinAlgError("Singular matrix (frame %d of the batch)" % n_ok)  # :79
            if status[n_ok] == K.ST_ERR_MASK:
                raise _lib.MvosrLibraryError("triangulation with an out-of-range vertex id (frame %d of the batch)" % n_ok)
            raise ValueError("frame %d of the batch has no triangles below the vanishing row" % n_ok)
        return filtered, np.ones(F)
