"""Device-side batch objects and launch wrappers over the C ABI (include/mvosr.h).

``DeviceBatch`` uploads a :class:`~mvoscalerecovery_amd.packing.PackedFrames` into HBM once;
``ScaleEngine`` launches the HIP kernels on it.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import os

import numpy as np

from . import _lib
from . import constants as K
from .packing import PackedFrames


def make_params(absolute_reference, camera_pitch=K.CAMERA_PITCH, pitch_threshold_deg=K.PITCH_THRESHOLD_DEG,
                skew_threshold=K.SKEW_THRESHOLD, mode_rel=K.MODE_REL, mode_min=K.MODE_MIN, check_triangle="reference"):
    """cos/sin come from NumPy so the kernels rotate with the very doubles the reference uses
    (/root/reference/src/scale_calculator.py:391-392)."""
    return _lib.Params(float(np.cos(camera_pitch)), float(np.sin(camera_pitch)), float(absolute_reference),
                       float(pitch_threshold_deg), float(skew_threshold), float(mode_rel), int(mode_min),
                       _lib.VOTE_FIXED if check_triangle == "fixed" else _lib.VOTE_REFERENCE)


def exact_mask_of(feat_cnt, everything=False, lazy_last=False):
    """mvosr_batch.exact_mask for a chunk: the frames whose ``height_level`` a LATER step reads get it summed in NumPy's own
    order by the product launch itself — the last frame of the chunk that sets a level (the next chunk, or the caller, may
    read it: a frame with exactly three features below the vanishing row divides by the level an earlier frame left,
    /root/reference/src/scale_calculator.py:263-270,:420-422; a frame that raises leaves the estimator at it) and every frame
    directly followed by such a three-feature frame.  ``everything``: all frames (small re-run batches).  ``lazy_last``: WITHOUT the
    chunk's last level-setting frame — its exact level is computed only if somebody turns out to read it (the next chunk's head, the
    caller: scale_calculator._stream_gpu); on ordinary data that one frame was the exact pass's whole list, i.e. a 23 ms replay of
    Qhull's run per chunk for a number nobody looked at."""
    cnt = np.asarray(feat_cnt)
    if everything:
        return np.ones(len(cnt), dtype=np.uint8)
    m = np.zeros(len(cnt), dtype=np.uint8)
    ok = np.nonzero(cnt > 3)[0]
    if len(ok):
        if not lazy_last:
            m[ok[-1]] = 1
        nxt = ok[ok + 1 < len(cnt)]
        m[nxt[cnt[nxt + 1] == 3]] = 1
    return m


def frame_tables(feature3ds, feature2ds, remap_in_place=False):
    """(data pointers of feature3ds, of feature2ds, rows per frame) as uint64 / uint64 / int32 arrays when every frame is
    a C-contiguous float64 (n,3) / (n,2) pair the C packer can read in place, else ``None``.  Through libmvosr_py.so
    (one C loop over the lists) when it is there.  ``remap_in_place``: the packer will also write feature_remap into the
    feature3d arrays (/root/reference/src/scale_calculator.py:414) — they must then be WRITABLE (a read-only array goes to
    the Python path, which raises ValueError at the assignment as the reference does) and DISTINCT: the packer's threads
    would otherwise read a frame's raw values while another thread remaps the same memory (a batch that holds one array
    object twice is packed in Python, raw values for every occurrence, remapped once per occurrence)."""
    from . import packing
    F = len(feature3ds)
    h = _lib.pyhelper()
    tables = None
    if h is not None and type(feature3ds) is list and type(feature2ds) is list:
        p3, p2, npts = np.empty(F, np.uint64), np.empty(F, np.uint64), np.empty(F, np.int32)
        r = h.mvosr_py_frame_pointers(feature3ds, feature2ds, _lib.addr(p3), _lib.addr(p2), _lib.addr(npts), 1 if remap_in_place else 0)
        tables = (p3, p2, npts) if r == F else None
    elif packing.native_packable(feature3ds, feature2ds, writable=remap_in_place) and F:
        tables = (np.fromiter((a.__array_interface__["data"][0] for a in feature3ds), dtype=np.uint64, count=F),
                  np.fromiter((a.__array_interface__["data"][0] for a in feature2ds), dtype=np.uint64, count=F),
                  np.fromiter((a.shape[0] for a in feature3ds), dtype=np.int32, count=F))
    if tables is not None and remap_in_place and F > 1 and np.unique(tables[0]).size != F:
        return None
    return tables


STANDIN_SEEDED = os.environ.get("MVOSR_STANDIN_SEEDED", "1") == "1"   # the stand-in second triangulation seeded with the first one's rows (Qhull's: any row form seeds)
UPLOAD_PIECES = 4              # pieces a chunk's upload is cut into (pack_upload_native) ...
UPLOAD_PIECE_FRAMES = 256      # ... of at least that many frames


def pack_upload_native(ctx, feature3ds, feature2ds, vanish, remap=None, threads=0, tables=None, lazy_last=False):
    """The batch path's front end without a Python loop over the frames' CONTENTS: the C packer (mvosr_pack_count /
    mvosr_pack_fill, a few host threads) applies the vanishing-row filter (/root/reference/src/scale_calculator.py:252-254)
    and writes the planes x|y|z|v|u straight into page-locked staging memory, which one asynchronous copy moves into a
    device block.  ``remap = (cos, sin)``: feature_remap is applied to the caller's arrays in place on the way (:414).
    Returns ``(PackedFrames without host planes, DeviceBlock)`` for ``DeviceBatch(..., device_triangulation=True,
    uploaded=block)``."""
    from . import packing
    F = len(feature3ds)
    lib = ctx.lib
    if tables is not None:
        p3, p2, npts = tables
    else:
        p3 = np.fromiter((a.__array_interface__["data"][0] for a in feature3ds), dtype=np.uint64, count=F)
        p2 = np.fromiter((a.__array_interface__["data"][0] for a in feature2ds), dtype=np.uint64, count=F)
        npts = np.fromiter((a.shape[0] for a in feature3ds), dtype=np.int32, count=F)
    # ONE pass over the frames: they are laid out by their unfiltered sizes (a few per cent of slack where features lie
    # above the vanishing row) and the packer reports how many it kept
    off, total = packing.pack_layout(npts)
    blk = ctx.block([("feat_off", F, np.int64), ("feat_cnt", F, np.int32), ("x", total, np.float64), ("y", total, np.float64),
                     ("z", total, np.float64), ("v", total, np.float64), ("u", total, np.float64), ("tri_off", F, np.int64),
                     ("exact_mask", F, np.uint8)])
    stage = blk.staging()
    sv = lambda k: stage.view(blk[k].offset, blk[k].shape, blk[k].dtype)
    sv("feat_off")[:] = off
    sv("tri_off")[:] = 2 * off
    cnt_view = sv("feat_cnt")
    base = stage.ptr
    c, s_ = (remap if remap is not None else (1.0, 0.0))
    # A chunk is packed and uploaded in up to four pieces: the copy of a piece's planes runs while the packer fills the next
    # piece, so the chunk's kernels wait for the LAST piece's copy only (one copy after the whole pack: 4.9 ms of packing, then
    # 6.5 ms of PCIe per 4608 frames of 2000 features before the first kernel — it is the first chunks of a call that pay for that)
    n_pieces = max(1, min(UPLOAD_PIECES, F // UPLOAD_PIECE_FRAMES))
    edges = [F * k // n_pieces for k in range(n_pieces + 1)]
    a3, a2, an, ao, ac = _lib.addr(p3), _lib.addr(p2), _lib.addr(npts), _lib.addr(off), _lib.addr(cnt_view)
    for a, b in zip(edges[:-1], edges[1:]):
        _lib.check(lib.mvosr_pack_fill(b - a, a3 + 8 * a, a2 + 8 * a, an + 4 * a, float(vanish), ao + 8 * a,
                                       base + blk["x"].offset, base + blk["y"].offset, base + blk["z"].offset, base + blk["u"].offset,
                                       base + blk["v"].offset, 1 if remap is not None else 0, float(c), float(s_), int(threads),
                                       ac + 4 * a),
                   "mvosr_pack_fill")
        if n_pieces > 1:
            lo, hi = int(off[a]), (int(off[b]) if b < F else int(total))
            blk.commit_ranges(stage, [(blk[k].offset + 8 * lo, 8 * (hi - lo)) for k in ("x", "y", "z", "v", "u")])
    cnt = np.array(cnt_view, dtype=np.int32, copy=True)
    sv("exact_mask")[:] = exact_mask_of(cnt, lazy_last=lazy_last)
    if n_pieces > 1:
        head, tail = blk["feat_off"].offset, blk["tri_off"].offset
        blk.commit_ranges(stage, [(head, blk["x"].offset - head), (tail, blk.nbytes - tail)], last=True)
    else:
        blk.commit(stage)
    pf = PackedFrames(F, off, cnt, None, None, None, None, None, [None] * F, max_feat=int(cnt.max()) if F else 0)
    pf.extra["total_padded"] = total
    return pf, blk


class DeviceBatch:
    """HBM-resident image of a packed batch: ONE device block per upload (features + first triangulation; second
    triangulation + tile index), each filled by one staged asynchronous copy (``_lib.DeviceBlock``)."""

    def __init__(self, ctx: _lib.Context, pf: PackedFrames, with_tri2=True, device_triangulation=False, uploaded=None, exact_all=False,
                 lazy_last=False):
        """``device_triangulation``: both triangulations will be BUILT on the device (:meth:`triangulate`) — the pixel
        column ``u`` travels too, and rows, row counts, vote counters and survivor counts get device buffers that the
        stages hand to each other; nothing of them visits the host."""
        self.ctx = ctx
        self.n_frames = pf.n_frames
        self.max_feat = pf.max_feat
        self._feat_cnt_host = np.ascontiguousarray(pf.feat_cnt, dtype=np.int32)
        self._feat_off_host = np.ascontiguousarray(pf.feat_off, dtype=np.int64)
        self.total_padded = pf.total_padded
        self.n_tri1 = int(pf.tri1_off[-1]) if pf.tri1_off is not None else 0
        self.n_tri2 = int(pf.tri2_off[-1]) if (with_tri2 and pf.tri2_off is not None) else 0
        self.algorithmic_bytes = pf.algorithmic_bytes()
        self.tri2_ids = 0
        self.bufs = {}
        self.blocks = []
        if uploaded is not None:                 # (pack_upload_native: the features are on their way already)
            assert device_triangulation
            self.blocks.append(uploaded)
            self.bufs.update(uploaded.views)
        else:
            arrays = {"feat_off": (pf.feat_off, np.int64), "feat_cnt": (pf.feat_cnt, np.int32),
                      "exact_mask": (exact_mask_of(pf.feat_cnt, exact_all, lazy_last) if pf.n_frames else np.zeros(1, np.uint8), np.uint8)}
            for name in ("x", "y", "z", "v"):
                arrays[name] = (getattr(pf, name), np.float64)
            if pf.tri1_off is not None and not device_triangulation:
                arrays["tri1_off"] = (pf.tri1_off, np.int64)
                arrays["tri1"] = (pf.tri1, np.int32)
            if device_triangulation:
                arrays["u"] = (pf.u, np.float64)
                arrays["tri_off"] = (2 * np.asarray(pf.feat_off, dtype=np.int64), np.int64)     # a frame's rows start at twice its feature offset: room for 2n rows
            self._upload(arrays)
        self.device_triangulation = bool(device_triangulation)
        if device_triangulation:
            F, T = pf.n_frames, 2 * pf.total_padded
            work = ctx.block([("tri1", (T, 3), np.int32), ("tri2", (T, 3), np.int32), ("vote_counters", pf.total_padded, np.int32),
                              ("dt_info", pf.total_padded, np.uint32)])
            info = ctx.block([(k, max(F, 1), np.int32) for k in ("tri1_cnt", "tri2_cnt", "dt1_status", "dt2_status", "n2_expected")])
            self.blocks += [work, info]
            self.info = info
            for k in ("tri1", "tri2", "vote_counters", "dt_info"):
                self.bufs[k] = work[k]
            for k in info.views:
                self.bufs[k] = info[k]
            self.bufs["tri1_off"] = self.bufs["tri2_off"] = self.bufs["tri_off"]
        elif with_tri2 and pf.tri2_off is not None:
            self.set_tri2(pf)
        self._struct = None

    def _queue_early_status(self):
        """The first triangulation's status words on their way to the host right behind that kernel — before the vote, the second
        triangulation and the product kernels of the chunk are even launched (``early_status`` waits for this copy alone)."""
        ctx, F = self.ctx, max(self.n_frames, 1)
        self._early_stage = _lib.PinnedBuffer(ctx, 4 * F)
        _lib.check(ctx.lib.mvosr_memcpy_d2h_kernel(ctx.handle, self._early_stage.ptr, self.bufs["dt1_status"].ptr, 4 * F), "d2h_kernel (early status)")
        self._early_event = ctx.event()                # (a copy by a kernel, not by a copy engine: DeviceBlock.mark_done)
        ctx.record(self._early_event)

    def early_status(self):
        """Host copy of the first triangulation's status (non-zero: declined) as soon as THAT kernel has finished; None when not asked for."""
        stage = getattr(self, "_early_stage", None)
        if stage is None:
            return None
        ctx = self.ctx
        _lib.check(ctx.lib.mvosr_event_sync(ctx.handle, self._early_event), "event_sync")
        s = np.array(stage.view(0, (max(self.n_frames, 1),), np.int32), copy=True)[:self.n_frames]
        stage.free(_lib.MARK_IDLE)
        ctx.lib.mvosr_event_destroy(ctx.handle, self._early_event)
        self._early_stage, self._early_event = None, None
        return s

    def triangulate(self, engine, standin=False, tri1_rows=None, early_status=False):
        """Delaunay #1 -> depth-order vote -> Delaunay #2 over the survivors (/root/reference/src/scale_calculator.py:
        257-267), three launches on the context's stream; rows, counters and counts stay in HBM.  ``standin`` (reference vote
        only; HOT launches without stage outputs, frames that fit the LDS-resident kernels): the second triangulation by the fast
        canonical-row kernel as a stand-in — its rows reach the result only through rounding, and mvosr_scale_batch replaces them
        with SciPy's own (Qhull's replay over its exact pass's list) for exactly the frames in which rounding can decide.
        ``tri1_rows`` (reference vote; the per-frame call): SciPy's rows of the first triangulation, one array per frame, computed
        on the host — one replay of Qhull's run is 20 ms on the device whatever the frame count, SciPy's call 2.6."""
        ctx, lib, b = self.ctx, self.ctx.lib, self.bufs
        assert self.device_triangulation
        self.info.invalidate()
        self.standin = False
        self._struct = None
        if getattr(engine, "check_triangle", "reference") == "reference":
            # the reference's vote reads the ROTATION of every row (:113-115): SciPy's rows themselves, order and rotation, from
            # the kernel that replays Qhull's insertion order (mvosr_delaunay_qhull_batch)
            if tri1_rows is not None:
                assert len(tri1_rows) == self.n_frames
                cnt1 = np.zeros(max(self.n_frames, 1), dtype=np.int32)
                for f, rows in enumerate(tri1_rows):              # (a frame's rows start at twice its feature offset)
                    rows = np.ascontiguousarray(rows, dtype=np.int32)
                    assert len(rows) <= 2 * int(self._feat_cnt_host[f])
                    cnt1[f] = len(rows)
                    if len(rows):
                        _lib.check(lib.mvosr_memcpy_h2d(ctx.handle, b["tri1"].ptr + 12 * 2 * int(self._feat_off_host[f]), _lib.addr(rows), rows.nbytes), "h2d (tri1 rows)")
                b["tri1_cnt"].upload(cnt1)
                b["dt1_status"].upload(np.zeros(max(self.n_frames, 1), dtype=np.int32))
            else:
                _lib.check(lib.mvosr_delaunay_qhull_batch(ctx.handle, self.n_frames, b["feat_off"].ptr, b["feat_cnt"].ptr, b["u"].ptr, b["v"].ptr,
                                                          None, int(self.max_feat), b["tri_off"].ptr, b["tri1"].ptr, b["tri1_cnt"].ptr, None,
                                                          b["dt1_status"].ptr, None), "mvosr_delaunay_qhull_batch (first triangulation)")
                if early_status:
                    self._queue_early_status()
            o = _lib.Outputs()
            o.vote_counters = b["vote_counters"].ptr
            bs = self.struct()
            _lib.check(lib.mvosr_outlier_vote_batch(ctx.handle, C.byref(engine.params), C.byref(bs), C.byref(o), 0), "mvosr_outlier_vote_batch")
            if standin and self.max_feat <= int(lib.mvosr_max_lds_features()) and self.max_feat <= int(lib.mvosr_delaunay_max_points()):
                _lib.check(lib.mvosr_delaunay_batch_ex(ctx.handle, self.n_frames, b["feat_off"].ptr, b["feat_cnt"].ptr, b["u"].ptr, b["v"].ptr,
                                                       b["vote_counters"].ptr, int(self.max_feat), b["tri_off"].ptr, b["tri2"].ptr,
                                                       b["tri2_cnt"].ptr, b["n2_expected"].ptr, b["dt2_status"].ptr,
                                                       *((b["tri_off"].ptr, b["tri1"].ptr, b["tri1_cnt"].ptr) if STANDIN_SEEDED else (None, None, None)),
                                                       None, None), "mvosr_delaunay_batch_ex (stand-in second triangulation)")
                self.standin = True
                self._struct = None
                return
            _lib.check(lib.mvosr_delaunay_qhull_batch(ctx.handle, self.n_frames, b["feat_off"].ptr, b["feat_cnt"].ptr, b["u"].ptr, b["v"].ptr,
                                                      b["vote_counters"].ptr, int(self.max_feat), b["tri_off"].ptr, b["tri2"].ptr,
                                                      b["tri2_cnt"].ptr, b["n2_expected"].ptr, b["dt2_status"].ptr, None),
                       "mvosr_delaunay_qhull_batch (second triangulation)")
            return
        # (the first triangulation leaves per-point facts — rows owned, degree, hull flag, first row — from which the second
        # carries over every star the vote did not touch instead of walking it: mvosr_delaunay_batch_ex)
        _lib.check(lib.mvosr_delaunay_batch_ex(ctx.handle, self.n_frames, b["feat_off"].ptr, b["feat_cnt"].ptr, b["u"].ptr, b["v"].ptr,
                                               None, int(self.max_feat), b["tri_off"].ptr, b["tri1"].ptr, b["tri1_cnt"].ptr, None,
                                               b["dt1_status"].ptr, None, None, None, None, b["dt_info"].ptr),
                   "mvosr_delaunay_batch_ex (first triangulation)")
        if early_status:
            self._queue_early_status()
        o = _lib.Outputs()
        o.vote_counters = b["vote_counters"].ptr
        bs = self.struct()
        _lib.check(lib.mvosr_outlier_vote_batch(ctx.handle, C.byref(engine.params), C.byref(bs), C.byref(o), 0), "mvosr_outlier_vote_batch")
        # (seeded with the first triangulation: its triangles among the survivors are triangles of the second)
        _lib.check(lib.mvosr_delaunay_batch_ex(ctx.handle, self.n_frames, b["feat_off"].ptr, b["feat_cnt"].ptr, b["u"].ptr, b["v"].ptr,
                                               b["vote_counters"].ptr, int(self.max_feat), b["tri_off"].ptr, b["tri2"].ptr,
                                               b["tri2_cnt"].ptr, b["n2_expected"].ptr, b["dt2_status"].ptr,
                                               b["tri_off"].ptr, b["tri1"].ptr, b["tri1_cnt"].ptr, b["dt_info"].ptr, None),
                   "mvosr_delaunay_batch_ex (second triangulation)")

    def triangulation_status(self):
        """Host copies of (first, second) triangulation status per frame (mvosr_dt_status; non-zero: declined)."""
        return self.bufs["dt1_status"].download(), self.bufs["dt2_status"].download()

    def prefetch_info(self):
        self.info.prefetch()

    def mark_info_done(self):
        """See DeviceBlock.mark_done: the statuses are fetched when they are read, behind an event, not by a copy parked on the stream."""
        self.info.mark_done()

    def mark(self, marked=True):
        """See DeviceBlock.mark: call after the last launch of a chunk; engine launches on the batch withdraw it."""
        for b in self.blocks:
            b.mark(marked)
        self._marked = bool(marked)

    def _upload(self, arrays):
        arrays = {k: np.ascontiguousarray(a, dtype=dt) for k, (a, dt) in arrays.items()}
        blk = self.ctx.block([(k, a.shape, a.dtype) for k, a in arrays.items()])
        blk.upload(arrays)
        self.blocks.append(blk)
        for k in arrays:
            self.bufs[k] = blk[k]

    def set_tri2(self, pf: PackedFrames):
        arrays = {"tri2_off": (pf.tri2_off, np.int64), "tri2": (pf.tri2, np.int32)}
        self.n_tri2 = int(pf.tri2_off[-1])
        self.tri2_ids = int(pf.tri2_ids)
        if pf.n2_expected is not None:
            arrays["n2_expected"] = (pf.n2_expected, np.int32)
        if getattr(pf, "tri2_order", None) is not None:
            arrays["tri2_order"] = (pf.tri2_order if pf.tri2_order.size else np.zeros(1, np.int32), np.int32)
        self.tile_w = 0
        if pf.tile_w and pf.tile1_off is not None and pf.tile2_off is not None:
            self.tile_w = int(pf.tile_w)
            arrays["tile_base"] = (pf.tile_base, np.int64)
            arrays["tile1_off"] = (pf.tile1_off, np.int32)
            arrays["tile2_off"] = (pf.tile2_off, np.int32)
            if pf.tile_far is not None:
                arrays["tile_far"] = (pf.tile_far, np.float64)
                arrays["tile_far_off"] = (pf.tile_far_off, np.int64)
        self._upload(arrays)
        self.algorithmic_bytes = pf.algorithmic_bytes()
        self._struct = None

    def struct(self):
        if self._struct is None:
            p = lambda k: (self.bufs[k].ptr if k in self.bufs else None)
            self._struct = _lib.Batch(self.n_frames, p("feat_off"), p("feat_cnt"), p("x"), p("y"), p("z"), p("v"),
                                      p("tri1_off"), p("tri1"), p("tri2_off"), p("tri2"), p("n2_expected"),
                                      self.max_feat, self.tri2_ids, self.total_padded,
                                      getattr(self, "tile_w", 0), 0, p("tile_base"), p("tile1_off"), p("tile2_off"))
            self._struct.tile_far = p("tile_far")
            self._struct.tile_far_off = p("tile_far_off")
            self._struct.tri1_cnt = p("tri1_cnt")
            self._struct.tri2_cnt = p("tri2_cnt")
            self._struct.tri2_order = p("tri2_order")
            self._struct.exact_mask = p("exact_mask")
            if getattr(self, "standin", False):
                self._struct.standin_u = p("u")
                self._struct.standin_keep = p("vote_counters")
                self._struct.standin_rows = p("tri2")
                self._struct.standin_cnt = p("tri2_cnt")
                self._struct.standin_status = p("dt2_status")
            if self.n_frames:               # min_feat + the size classes' counts (ragged batches launch per class)
                mf = self._struct.max_feat
                _lib.check(self.ctx.lib.mvosr_batch_size_hint(_lib.addr(self._feat_cnt_host), self.n_frames,
                                                              C.byref(self._struct)), "mvosr_batch_size_hint")
                self._struct.max_feat = max(mf, self._struct.max_feat)
        return self._struct

    def set_exact_mask(self, mask):
        """Replace the batch's exact mask (``engine.scale_batch(..., masked=True)`` then redoes exactly those frames)."""
        self.bufs["exact_mask"].upload(np.ascontiguousarray(mask, dtype=np.uint8))

    def free(self):
        if getattr(self, "_early_stage", None) is not None:       # (an early status nobody read)
            self._early_stage.free()
            self.ctx.lib.mvosr_event_destroy(self.ctx.handle, self._early_event)
            self._early_stage, self._early_event = None, None
        for b in self.blocks:
            b.free()
        self.blocks = []
        self.bufs = {}


class DeviceOutputs:
    """Output arrays of a launch in ONE device block, read back by one download; ``stage=True`` adds the per-stage
    arrays used by parity tests and by the per-frame drop-in call (selected mask, counters, per-triangle values).
    ``share``: another DeviceOutputs whose per-frame arrays (raw_scale ... counts) this one writes into as well."""

    def __init__(self, ctx, batch: DeviceBatch, counts=True, stage=False, per_triangle=False, hist=False, share=None):
        F = batch.n_frames
        self.ctx = ctx
        spec, zero = [], False
        if share is None:
            spec += [("raw_scale", F, np.float64), ("height", F, np.float64), ("height_level", F, np.float64), ("status", F, np.int32)]
            if counts:
                spec.append(("counts", (F, _lib.N_COUNTS), np.int32))
                zero = True
        if stage:
            spec += [("vote_counters", batch.total_padded, np.int32), ("selected", batch.total_padded, np.uint8)]
            if F == 1:                   # (the per-frame call: the window median of its one raw scale travels with the results)
                spec.append(("filtered", 1, np.float64))
            zero = True
        if per_triangle:
            t2 = max(batch.n_tri2, 1)
            spec += [("tri_normals", (t2, 3), np.float64), ("tri_pitch_deg", t2, np.float64), ("tri_heights", t2, np.float64)]
            zero = True
        if hist:
            spec += [("hist", (F, 2, _lib.HIST_BINS), np.int32), ("stats", (F, 4), np.float64)]
            zero = True
        self.block = ctx.block(spec) if spec else None
        self.shared = share
        self.bufs = dict(self.block.views) if self.block is not None else {}
        if share is not None:
            for k in ("raw_scale", "height", "height_level", "status", "counts"):
                if k in share.bufs:
                    self.bufs[k] = share.bufs[k]
        if zero and self.block is not None:
            self.block.zero()

    def struct(self):
        p = lambda k: (self.bufs[k].ptr if k in self.bufs else None)
        return _lib.Outputs(p("raw_scale"), p("height"), p("height_level"), p("status"), p("counts"),
                            p("vote_counters"), p("selected"), p("tri_normals"), p("tri_pitch_deg"),
                            p("tri_heights"), p("hist"), p("stats"))

    def prefetch(self):
        """Queue the download of the per-frame results behind the launches so far (see DeviceBlock.prefetch)."""
        if self.block is not None:
            self.block.prefetch()
            self.block.mark(True)

    def mark_done(self):
        """See DeviceBlock.mark_done (instead of ``prefetch`` for chunks of a streamed batch)."""
        if self.block is not None:
            self.block.mark_done()
            self.block.mark(True)

    def invalidate(self):
        """A launch is about to write the arrays: host copies are stale."""
        if self.block is not None:
            self.block.invalidate()
            self.block.mark(False)
        if self.shared is not None:
            self.shared.invalidate()

    def get(self, name):
        return self.bufs[name].download()

    def ready(self):
        """True when ``get`` would not wait (the download queued by ``prefetch`` has finished)."""
        return self.block is None or self.block.ready()

    def free(self):
        if self.block is not None:
            self.block.free()
            self.block = None
        self.bufs = {}


class ScaleEngine:
    """Launches the hot-path kernels.  One engine = one context (device + stream) + parameters."""

    def __init__(self, absolute_reference, device=0, ctx=None, **param_kw):
        self.ctx = ctx if ctx is not None else _lib.default_context(device)
        self.lib = self.ctx.lib
        self.check_triangle = param_kw.get("check_triangle", "reference")
        self.params = make_params(absolute_reference, **param_kw)

    def scale_batch(self, batch: DeviceBatch, out: DeviceOutputs, waves=0, first=0, count=0, exact=False, masked=False, hot_only=False):
        """``exact``: every frame of the range in the exact mode; ``masked``: ONLY the frames of the batch's exact mask, in
        the exact mode — the other frames' outputs stay as they are; ``hot_only``: the product kernels only (MVOSR_WAVES_HOT_ONLY):
        frames the exact pass would redo come back with status ``_lib.ST_REDO``."""
        b, o = batch.struct(), out.struct()
        out.invalidate()
        if getattr(batch, "_marked", False):
            batch.mark(False)
        flags = (_lib.WAVES_EXACT if exact else 0) | (_lib.WAVES_EXACT_MASKED if masked else 0) | (_lib.WAVES_HOT_ONLY if hot_only else 0)
        _lib.check(self.lib.mvosr_scale_batch(self.ctx.handle, C.byref(self.params), C.byref(b), C.byref(o),
                                              int(waves) | flags, int(first), int(count)), "mvosr_scale_batch")

    def outlier_vote_batch(self, batch: DeviceBatch, out: DeviceOutputs, waves=0):
        b, o = batch.struct(), out.struct()
        out.invalidate()
        _lib.check(self.lib.mvosr_outlier_vote_batch(self.ctx.handle, C.byref(self.params), C.byref(b), C.byref(o),
                                                     int(waves)), "mvosr_outlier_vote_batch")

    def road_model_batch(self, batch: DeviceBatch, out: DeviceOutputs, height_level=None, waves=0):
        b, o = batch.struct(), out.struct()
        out.invalidate()
        hl = self.ctx.to_device(height_level, np.float64) if height_level is not None else None
        _lib.check(self.lib.mvosr_road_model_batch(self.ctx.handle, C.byref(self.params), C.byref(b),
                                                   hl.ptr if hl is not None else None, C.byref(o), int(waves)),
                   "mvosr_road_model_batch")
        self.ctx.sync()
        if hl is not None:
            hl.free()

    def window_median(self, raw_dev_ptr, n, window, queue=(), out_dev_ptr=None):
        q = np.ascontiguousarray(np.asarray(list(queue), dtype=np.float64))
        _lib.check(self.lib.mvosr_window_median(self.ctx.handle, raw_dev_ptr, int(n), int(window),
                                                _lib.addr(q) if q.size else None, int(q.size), out_dev_ptr),
                   "mvosr_window_median")

    def window_median_blocked(self, blocks_dev_ptr, n, n_blocks, block_stride, window, queue=(), out_dev_ptr=None):
        """The window median over an all-gathered sequence read in place (sharding.GatheredFrames)."""
        q = np.ascontiguousarray(np.asarray(list(queue), dtype=np.float64))
        _lib.check(self.lib.mvosr_window_median_blocked(self.ctx.handle, blocks_dev_ptr, int(n), int(n_blocks), int(block_stride),
                                                        int(window), _lib.addr(q) if q.size else None, int(q.size), out_dev_ptr),
                   "mvosr_window_median_blocked")

    def window_median_host(self, raw, window, queue=()):
        """Convenience: upload a host sequence, filter on the GPU, download."""
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        if raw.size == 0:
            return raw.copy()
        d_in = self.ctx.to_device(raw)
        d_out = self.ctx.empty(raw.shape, np.float64)
        self.window_median(d_in.ptr, raw.size, window, queue, d_out.ptr)
        self.ctx.sync()
        res = d_out.download()
        d_in.free()
        d_out.free()
        return res
