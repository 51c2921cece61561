"""Device-side batch objects and launch wrappers over the C ABI (include/mvosr.h).

``DeviceBatch`` uploads a :class:`~mvoscalerecovery_amd.packing.PackedFrames` into HBM once;
``ScaleEngine`` launches the HIP kernels on it.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from . import constants as K
from .packing import PackedFrames


def make_params(absolute_reference, camera_pitch=K.CAMERA_PITCH, pitch_threshold_deg=K.PITCH_THRESHOLD_DEG,
                skew_threshold=K.SKEW_THRESHOLD, mode_rel=K.MODE_REL, mode_min=K.MODE_MIN):
    """cos/sin come from NumPy so the kernels rotate with the very doubles the reference uses
    (/root/reference/src/scale_calculator.py:391-392)."""
    return _lib.Params(float(np.cos(camera_pitch)), float(np.sin(camera_pitch)), float(absolute_reference),
                       float(pitch_threshold_deg), float(skew_threshold), float(mode_rel), int(mode_min), 0)


class DeviceBatch:
    """HBM-resident image of a packed batch."""

    def __init__(self, ctx: _lib.Context, pf: PackedFrames, with_tri2=True):
        self.ctx = ctx
        self.n_frames = pf.n_frames
        self.max_feat = pf.max_feat
        self._feat_cnt_host = np.ascontiguousarray(pf.feat_cnt, dtype=np.int32)
        self.total_padded = pf.total_padded
        self.n_tri1 = int(pf.tri1_off[-1]) if pf.tri1_off is not None else 0
        self.n_tri2 = int(pf.tri2_off[-1]) if (with_tri2 and pf.tri2_off is not None) else 0
        self.algorithmic_bytes = pf.algorithmic_bytes()
        self.tri2_ids = 0
        self.bufs = {}
        up = self.bufs
        up["feat_off"] = ctx.to_device(pf.feat_off, np.int64)
        up["feat_cnt"] = ctx.to_device(pf.feat_cnt, np.int32)
        for name in ("x", "y", "z", "v"):
            up[name] = ctx.to_device(getattr(pf, name), np.float64)
        if pf.tri1_off is not None:
            up["tri1_off"] = ctx.to_device(pf.tri1_off, np.int64)
            up["tri1"] = ctx.to_device(pf.tri1, np.int32)
        if with_tri2 and pf.tri2_off is not None:
            self.set_tri2(pf)
        self._struct = None

    def set_tri2(self, pf: PackedFrames):
        self.bufs["tri2_off"] = self.ctx.to_device(pf.tri2_off, np.int64)
        self.bufs["tri2"] = self.ctx.to_device(pf.tri2, np.int32)
        self.n_tri2 = int(pf.tri2_off[-1])
        self.tri2_ids = int(pf.tri2_ids)
        if pf.n2_expected is not None:
            self.bufs["n2_expected"] = self.ctx.to_device(pf.n2_expected, np.int32)
        self.tile_w = 0
        if pf.tile_w and pf.tile1_off is not None and pf.tile2_off is not None:
            self.tile_w = int(pf.tile_w)
            self.bufs["tile_base"] = self.ctx.to_device(pf.tile_base, np.int64)
            self.bufs["tile1_off"] = self.ctx.to_device(pf.tile1_off, np.int32)
            self.bufs["tile2_off"] = self.ctx.to_device(pf.tile2_off, np.int32)
            if pf.tile_far is not None:
                self.bufs["tile_far"] = self.ctx.to_device(pf.tile_far, np.float64)
                self.bufs["tile_far_off"] = self.ctx.to_device(pf.tile_far_off, np.int64)
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
            if self.n_frames:               # min_feat + the size classes' counts (ragged batches launch per class)
                mf = self._struct.max_feat
                _lib.check(self.ctx.lib.mvosr_batch_size_hint(self._feat_cnt_host.ctypes.data, self.n_frames,
                                                              C.byref(self._struct)), "mvosr_batch_size_hint")
                self._struct.max_feat = max(mf, self._struct.max_feat)
        return self._struct

    def free(self):
        for b in self.bufs.values():
            b.free()
        self.bufs = {}


class DeviceOutputs:
    """Output arrays of a launch; ``stage=True`` adds the per-stage arrays used by parity tests
    and by the per-frame drop-in call (selected mask, counters, per-triangle values)."""

    def __init__(self, ctx, batch: DeviceBatch, counts=True, stage=False, per_triangle=False, hist=False):
        F = batch.n_frames
        self.ctx = ctx
        self.bufs = {
            "raw_scale": ctx.empty(F, np.float64), "height": ctx.empty(F, np.float64),
            "height_level": ctx.empty(F, np.float64), "status": ctx.empty(F, np.int32),
        }
        if counts:
            self.bufs["counts"] = ctx.zeros((F, _lib.N_COUNTS), np.int32)
        if stage:
            self.bufs["vote_counters"] = ctx.zeros(batch.total_padded, np.int32)
            self.bufs["selected"] = ctx.zeros(batch.total_padded, np.uint8)
        if per_triangle:
            t2 = max(batch.n_tri2, 1)
            self.bufs["tri_normals"] = ctx.zeros((t2, 3), np.float64)
            self.bufs["tri_pitch_deg"] = ctx.zeros(t2, np.float64)
            self.bufs["tri_heights"] = ctx.zeros(t2, np.float64)
        if hist:
            self.bufs["hist"] = ctx.zeros((F, 2, _lib.HIST_BINS), np.int32)
            self.bufs["stats"] = ctx.zeros((F, 4), np.float64)

    def struct(self):
        p = lambda k: (self.bufs[k].ptr if k in self.bufs else None)
        return _lib.Outputs(p("raw_scale"), p("height"), p("height_level"), p("status"), p("counts"),
                            p("vote_counters"), p("selected"), p("tri_normals"), p("tri_pitch_deg"),
                            p("tri_heights"), p("hist"), p("stats"))

    def get(self, name):
        return self.bufs[name].download()

    def free(self):
        for b in self.bufs.values():
            b.free()
        self.bufs = {}


class ScaleEngine:
    """Launches the hot-path kernels.  One engine = one context (device + stream) + parameters."""

    def __init__(self, absolute_reference, device=0, ctx=None, **param_kw):
        self.ctx = ctx if ctx is not None else _lib.default_context(device)
        self.lib = self.ctx.lib
        self.params = make_params(absolute_reference, **param_kw)

    def scale_batch(self, batch: DeviceBatch, out: DeviceOutputs, waves=0, first=0, count=0):
        b, o = batch.struct(), out.struct()
        _lib.check(self.lib.mvosr_scale_batch(self.ctx.handle, C.byref(self.params), C.byref(b), C.byref(o),
                                              int(waves), int(first), int(count)), "mvosr_scale_batch")

    def outlier_vote_batch(self, batch: DeviceBatch, out: DeviceOutputs, waves=0):
        b, o = batch.struct(), out.struct()
        _lib.check(self.lib.mvosr_outlier_vote_batch(self.ctx.handle, C.byref(self.params), C.byref(b), C.byref(o),
                                                     int(waves)), "mvosr_outlier_vote_batch")

    def road_model_batch(self, batch: DeviceBatch, out: DeviceOutputs, height_level=None, waves=0):
        b, o = batch.struct(), out.struct()
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
                                                q.ctypes.data if q.size else None, int(q.size), out_dev_ptr),
                   "mvosr_window_median")

    def window_median_blocked(self, blocks_dev_ptr, n, n_blocks, block_stride, window, queue=(), out_dev_ptr=None):
        """The window median over an all-gathered sequence read in place (sharding.GatheredFrames)."""
        q = np.ascontiguousarray(np.asarray(list(queue), dtype=np.float64))
        _lib.check(self.lib.mvosr_window_median_blocked(self.ctx.handle, blocks_dev_ptr, int(n), int(n_blocks), int(block_stride),
                                                        int(window), q.ctypes.data if q.size else None, int(q.size), out_dev_ptr),
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
