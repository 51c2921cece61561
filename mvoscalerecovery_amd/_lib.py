"""ctypes binding of libmvosr.so (C ABI: include/mvosr.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded every product entry
point raises :class:`MvosrLibraryError`.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C mvoscalerecovery_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

LIB_PATH = os.environ.get("MVOSR_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmvosr.so")   # (override: A/B builds in profiles/)
ABI_VERSION = 12
VOTE_REFERENCE, VOTE_FIXED = 0, 1          # mvosr_params.vote_mode
WAVES_EXACT = 0x100                        # MVOSR_WAVES_EXACT, or-ed into waves_per_frame
WAVES_EXACT_MASKED = 0x200                 # MVOSR_WAVES_EXACT_MASKED: only the frames of mvosr_batch.exact_mask, in the exact mode
WAVES_HOT_ONLY = 0x400                     # MVOSR_WAVES_HOT_ONLY: product kernels only; frames the exact pass would redo come back ST_REDO
ST_REDO = -2                               # MVOSR_ST_REDO
MARK_NOW, MARK_IDLE, MARK_UPLOAD = 1, 2, 3 # mvosr_block_mark
TRI2_SURVIVORS, TRI2_FEATURES = 0, 1      # mvosr_batch.tri2_ids
N_COUNTS = 8
TILE_W = 512                              # MVOSR_TILE_W
HIST_BINS = 169


ERR_ALLOC = -5


class MvosrLibraryError(RuntimeError):
    """libmvosr.so is missing / not loadable / a call into it failed."""


class MvosrAllocError(MvosrLibraryError):
    """MVOSR_ERR_ALLOC: a grow-only device workspace could not be allocated; nothing was launched (the batch paths send the chunk
    through the host's triangulations instead)."""


class Params(C.Structure):
    _fields_ = [("cos_pitch", C.c_double), ("sin_pitch", C.c_double), ("absolute_reference", C.c_double),
                ("pitch_threshold_deg", C.c_double), ("skew_threshold", C.c_double), ("mode_rel", C.c_double),
                ("mode_min", C.c_int32), ("vote_mode", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("n_frames", C.c_int64), ("feat_off", C.c_void_p), ("feat_cnt", C.c_void_p),
                ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("v", C.c_void_p),
                ("tri1_off", C.c_void_p), ("tri1", C.c_void_p), ("tri2_off", C.c_void_p), ("tri2", C.c_void_p),
                ("n2_expected", C.c_void_p), ("max_feat", C.c_int32), ("tri2_ids", C.c_int32),
                ("total_feat", C.c_int64),
                ("tile_w", C.c_int32), ("min_feat", C.c_int32), ("tile_base", C.c_void_p),
                ("tile1_off", C.c_void_p), ("tile2_off", C.c_void_p), ("size_hint", C.c_int32 * 4),
                ("tile_far", C.c_void_p), ("tile_far_off", C.c_void_p),
                ("tri1_cnt", C.c_void_p), ("tri2_cnt", C.c_void_p), ("tri2_order", C.c_void_p), ("exact_mask", C.c_void_p),
                ("standin_u", C.c_void_p), ("standin_keep", C.c_void_p), ("standin_rows", C.c_void_p), ("standin_cnt", C.c_void_p),
                ("standin_status", C.c_void_p)]


class Outputs(C.Structure):
    _fields_ = [("raw_scale", C.c_void_p), ("height", C.c_void_p), ("height_level", C.c_void_p),
                ("status", C.c_void_p), ("counts", C.c_void_p), ("vote_counters", C.c_void_p),
                ("selected", C.c_void_p), ("tri_normals", C.c_void_p), ("tri_pitch_deg", C.c_void_p),
                ("tri_heights", C.c_void_p), ("hist", C.c_void_p), ("stats", C.c_void_p)]


class RescaleParams(C.Structure):
    """mvosr_rescale_params"""
    _fields_ = [("good_bits", C.c_uint32), ("min_valid", C.c_int32), ("loose_deg", C.c_double), ("tight_deg", C.c_double),
                ("height_factor", C.c_double), ("ransac_min_points", C.c_int32), ("n_hyp", C.c_int32),
                ("threshold", C.c_double), ("goal_fraction", C.c_double), ("absolute_reference", C.c_double),
                ("seed", C.c_uint64), ("frame_base", C.c_int64)]


class RescaleOutputs(C.Structure):
    """mvosr_rescale_outputs"""
    _fields_ = [("raw_scale", C.c_void_p), ("height_level", C.c_void_p), ("model", C.c_void_p), ("best_ic", C.c_void_p),
                ("used", C.c_void_p), ("n_kept", C.c_void_p), ("status", C.c_void_p), ("tri_height", C.c_void_p),
                ("tri_flags", C.c_void_p), ("hyp_counts", C.c_void_p)]


# every symbol include/mvosr.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "mvosr_abi_version": (C.c_int, []),
    "mvosr_last_error": (C.c_char_p, []),
    "mvosr_device_count": (C.c_int, []),
    "mvosr_device_numa_node": (C.c_int, [C.c_int]),
    "mvosr_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "mvosr_ctx_destroy": (C.c_int, [_P]),
    "mvosr_ctx_set_stream": (C.c_int, [_P, _P]),
    "mvosr_ctx_stream": (_P, [_P]),
    "mvosr_ctx_sync": (C.c_int, [_P]),
    "mvosr_ctx_reserve": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "mvosr_ctx_workspace_limit": (C.c_int, [_P, C.c_int64]),
    "mvosr_ctx_profile": (C.c_int, [_P, C.c_int]),
    "mvosr_ctx_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mvosr_ctx_device_info": (C.c_int, [_P, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mvosr_malloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "mvosr_free": (C.c_int, [_P, _P]),
    "mvosr_host_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "mvosr_host_free": (C.c_int, [_P, _P]),
    "mvosr_ctx_trim": (C.c_int, [_P]),
    "mvosr_ctx_alloc_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.c_int]),
    "mvosr_memcpy_h2d_async": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "mvosr_upload_fence": (C.c_int, [_P]),
    "mvosr_memcpy_d2h_async": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "mvosr_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "mvosr_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "mvosr_memcpy_d2h_kernel": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "mvosr_memset": (C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    "mvosr_event_create": (C.c_int, [_P, C.POINTER(_P)]),
    "mvosr_event_record": (C.c_int, [_P, _P]),
    "mvosr_event_elapsed_ms": (C.c_int, [_P, _P, _P, C.POINTER(C.c_float)]),
    "mvosr_event_sync": (C.c_int, [_P, _P]),
    "mvosr_event_query": (C.c_int, [_P, _P, C.POINTER(C.c_int)]),
    "mvosr_event_destroy": (C.c_int, [_P, _P]),
    "mvosr_pack_count": (C.c_int, [C.c_int64, _P, _P, C.c_double, _P, C.c_int]),
    "mvosr_pack_fill": (C.c_int, [C.c_int64, _P, _P, _P, C.c_double, _P, _P, _P, _P, _P, _P, C.c_int, C.c_double, C.c_double, C.c_int, _P]),
    "mvosr_block_mark": (C.c_int, [_P, _P, C.c_int]),
    "mvosr_default_params": (None, [C.POINTER(Params), C.c_double]),
    "mvosr_scale_batch": (C.c_int, [_P, C.POINTER(Params), C.POINTER(Batch), C.POINTER(Outputs), C.c_int,
                                    C.c_int64, C.c_int64]),
    "mvosr_batch_size_hint": (C.c_int, [_P, C.c_int64, C.POINTER(Batch)]),
    "mvosr_outlier_vote_batch": (C.c_int, [_P, C.POINTER(Params), C.POINTER(Batch), C.POINTER(Outputs), C.c_int]),
    "mvosr_road_model_batch": (C.c_int, [_P, C.POINTER(Params), C.POINTER(Batch), _P, C.POINTER(Outputs), C.c_int]),
    "mvosr_window_median": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, C.c_int, _P]),
    "mvosr_window_median_blocked": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int64, C.c_int, _P, C.c_int, _P]),
    "mvosr_graph_inliers_batch": (C.c_int, [_P, C.POINTER(Batch), C.c_uint32, _P, _P, _P]),
    "mvosr_flat_selection_batch": (C.c_int, [_P, C.POINTER(Batch), C.c_double, C.c_double, C.c_double, _P, _P, _P, _P, _P,
                                             C.c_int64]),
    "mvosr_graph_keep_batch": (C.c_int, [_P, C.POINTER(Batch), C.c_uint32, C.c_int32, _P, _P, _P, _P]),
    "mvosr_flat_ransac_batch": (C.c_int, [_P, C.POINTER(Batch), _P, C.POINTER(RescaleParams), _P, _P, _P, C.POINTER(RescaleOutputs),
                                          C.c_int64]),
    "mvosr_slew_median": (C.c_int, [_P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_int, _P, C.c_int, _P, _P]),
    "mvosr_slew_median_host": (C.c_int, [_P, _P, C.c_int64, C.c_double, C.c_double, C.c_int, _P, C.c_int, _P, _P, _P]),
    "mvosr_ransac_plane_batch": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_int, C.c_double, C.c_double,
                                           _P, _P, _P, _P]),
    "mvosr_ransac_line_batch": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, C.c_int, C.c_double, C.c_double,
                                          _P, _P, _P, _P]),
    "mvosr_triangle_batch": (C.c_int, [_P, C.POINTER(Batch), C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                       _P, _P, _P]),
    "mvosr_plane_inliers": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, C.c_double, _P]),
    "mvosr_delaunay_batch": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P]),
    "mvosr_delaunay_batch_seeded": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mvosr_delaunay_batch_ex": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mvosr_delaunay_qhull_batch": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "mvosr_delaunay_qhull_max_points": (C.c_int, []),
    "mvosr_qhull_rows_host": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, _P, _P]),
    "mvosr_delaunay_max_points": (C.c_int, []),
    "mvosr_delaunay_lds_points": (C.c_int, []),
    "mvosr_delaunay_frames_per_cu": (C.c_int, [C.c_int]),
    "mvosr_lds_bytes": (C.c_size_t, [C.c_int]),
    "mvosr_max_lds_features": (C.c_int, []),
}

_lib = None
_pyhelper = False


def pyhelper():
    """libmvosr_py.so (csrc/mvosr_pyhelper.c): pointer / size tables of a list of per-frame arrays through the buffer
    protocol in C.  Plumbing only; ``None`` when it is not built (the caller then asks the arrays one by one in Python)."""
    global _pyhelper
    if _pyhelper is False:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmvosr_py.so")
        try:
            h = C.PyDLL(path)
            h.mvosr_py_frame_pointers.restype = C.c_long
            h.mvosr_py_frame_pointers.argtypes = [C.py_object, C.py_object, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            _pyhelper = h
        except (OSError, AttributeError):
            _pyhelper = None
    return _pyhelper


def load():
    """Load libmvosr.so once and set prototypes.  Raises if it is not there — loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise MvosrLibraryError(
            "HIP extension not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mvoscalerecovery_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    # A context owns a compute and an upload stream, the exact device path runs two contexts and a third for re-runs, a host application (torch) has streams of
    # its own: beyond ROCm's default of four hardware queues per process streams SHARE a queue and stop overlapping (measured: the
    # exact path's two contexts 78 k instead of 94 k frames/s inside bench.py).  The variable that lifts the limit is the HOST
    # APPLICATION's (it is read when the HIP runtime initialises and is inherited by child processes): this binding sets it only when
    # asked to — MVOSR_HW_QUEUES=<n> (bench.py opts in with 8, like MVOSR_AFFINITY) — and says so when the request comes too late.
    want = os.environ.get("MVOSR_HW_QUEUES", "").strip()
    if want.isdigit() and int(want) > 0 and "GPU_MAX_HW_QUEUES" not in os.environ:
        import sys
        import warnings
        torch = sys.modules.get("torch")
        if torch is not None and getattr(getattr(torch, "cuda", None), "is_initialized", lambda: False)():
            warnings.warn("mvoscalerecovery_amd: MVOSR_HW_QUEUES=%s asked for after the HIP runtime was initialised (torch.cuda is up): "
                          "GPU_MAX_HW_QUEUES has no effect any more; set it in the environment of the process instead" % want,
                          RuntimeWarning, stacklevel=2)
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = want
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:
        raise MvosrLibraryError("cannot load %s: %s" % (LIB_PATH, exc)) from exc
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise MvosrLibraryError("libmvosr.so lacks symbol %s" % name) from exc
        fn.restype = res
        fn.argtypes = args
    if lib.mvosr_abi_version() != ABI_VERSION:
        raise MvosrLibraryError("libmvosr.so ABI %d != binding ABI %d" % (lib.mvosr_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mvosr_last_error()
        if rc == ERR_ALLOC:
            raise MvosrAllocError("%s failed (%d): %s" % (what or "libmvosr call", rc, (msg or b"").decode()))
        raise MvosrLibraryError("%s failed (%d): %s" % (what or "libmvosr call", rc, (msg or b"").decode()))


def addr(arr):
    """The address of a NumPy array's data (what ``arr.ctypes.data`` returns, without building the ctypes helper object:
    15 us each, a dozen per per-frame call)."""
    return arr.__array_interface__["data"][0]


class DeviceBuffer:
    """A hipMalloc'ed array with NumPy dtype/shape metadata."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = math.prod(self.shape) * self.dtype.itemsize
        p = C.c_void_p()
        check(ctx.lib.mvosr_malloc(ctx.handle, max(self.nbytes, 16), C.byref(p)), "mvosr_malloc")
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes, (arr.shape, self.shape)
        check(self.ctx.lib.mvosr_memcpy_h2d(self.ctx.handle, self.ptr, addr(arr), self.nbytes), "h2d")
        return self

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(self.ctx.lib.mvosr_memcpy_d2h(self.ctx.handle, addr(out), self.ptr, self.nbytes), "d2h")
        return out

    def fill(self, byte=0):
        check(self.ctx.lib.mvosr_memset(self.ctx.handle, self.ptr, byte, self.nbytes), "memset")
        return self

    def free(self):
        if self.ptr:
            self.ctx.lib.mvosr_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedBuffer:
    """Page-locked host memory from the context's caching allocator (mvosr_host_alloc), seen as NumPy arrays."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(max(nbytes, 16))
        p = C.c_void_p()
        check(ctx.lib.mvosr_host_alloc(ctx.handle, self.nbytes, C.byref(p)), "mvosr_host_alloc")
        self.ptr = p.value
        self._raw = (C.c_char * self.nbytes).from_address(self.ptr)

    def view(self, offset, shape, dtype):
        dtype = np.dtype(dtype)
        count = math.prod(shape) if isinstance(shape, (tuple, list)) else int(shape)
        return np.frombuffer(self._raw, dtype=dtype, count=count, offset=int(offset)).reshape(shape)

    def free(self, mark=0):
        """``mark``: MARK_UPLOAD — the buffer was the source of uploads only; MARK_IDLE — nothing queued uses it (its
        download has been waited for).  Without one, its next user waits for everything queued so far."""
        if self.ptr:
            self._raw = None
            if mark:
                self.ctx.lib.mvosr_block_mark(self.ctx.handle, self.ptr, int(mark))
            self.ctx.lib.mvosr_host_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceView:
    """A typed range of a DeviceBlock (what the C ABI sees as one array)."""

    def __init__(self, block, offset, shape, dtype):
        self.block, self.offset = block, int(offset)
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = math.prod(self.shape) * self.dtype.itemsize
        self.ptr = block.ptr + self.offset

    def download(self):
        return self.block.read(self)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes, (arr.shape, self.shape)
        check(self.block.ctx.lib.mvosr_memcpy_h2d(self.block.ctx.handle, self.ptr, addr(arr), self.nbytes), "h2d")
        self.block.invalidate()
        return self

    def free(self):                      # the block owns the memory
        pass


class DeviceBlock:
    """ONE device allocation holding several arrays (256-byte aligned), filled by ONE upload from a page-locked
    staging buffer and read back by ONE download: the per-frame drop-in call and every chunk of the batch path move
    their inputs and outputs as a couple of transfers instead of one allocation and one blocking copy per array."""

    _PLANS = {}
    ALIGN = 256
    STAGE_LIMIT = 1 << 28              # larger uploads go array by array (a staging mirror of that size is not worth its memory)

    def __init__(self, ctx, spec):
        """``spec``: list of (name, shape, dtype)."""
        self.ctx = ctx
        self.views = {}
        # (the per-frame call builds the same few blocks for every frame of a size: the layout is remembered per spec)
        try:
            key = tuple((n_, tuple(sh) if isinstance(sh, (tuple, list)) else sh, dt) for n_, sh, dt in spec)
            cached = self._PLANS.get(key)
        except TypeError:                  # an unhashable dtype spelling: no cache
            key, cached = None, None
        if cached is None:
            size = 0
            plan = []
            for name, shape, dtype in spec:
                size = (size + self.ALIGN - 1) & ~(self.ALIGN - 1)
                shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
                dtype = np.dtype(dtype)
                nb = math.prod(shape) * dtype.itemsize
                plan.append((name, size, shape, dtype))
                size += max(nb, 16)
            cached = (plan, max(size, 16))
            if key is not None:
                if len(self._PLANS) > 256:
                    self._PLANS.clear()
                self._PLANS[key] = cached
        plan, self.nbytes = cached
        p = C.c_void_p()
        check(ctx.lib.mvosr_malloc(ctx.handle, self.nbytes, C.byref(p)), "mvosr_malloc")
        self.ptr = p.value
        for name, off, shape, dtype in plan:
            self.views[name] = DeviceView(self, off, shape, dtype)
        self._mirror = None               # host copy of the whole block (outputs), valid until invalidate()

    def __getitem__(self, name):
        return self.views[name]

    def __contains__(self, name):
        return name in self.views

    def upload(self, arrays):
        """``arrays``: name -> NumPy array for (a subset of) the block's views.  One staged asynchronous copy on the upload
        stream, the compute stream waits for it (no host wait)."""
        ctx = self.ctx
        if not arrays:
            return self
        if self.nbytes > self.STAGE_LIMIT:
            for name, arr in arrays.items():
                v = self.views[name]
                arr = np.ascontiguousarray(arr, dtype=v.dtype)
                assert arr.nbytes == v.nbytes, (name, arr.shape, v.shape)
                check(ctx.lib.mvosr_memcpy_h2d(ctx.handle, v.ptr, addr(arr), v.nbytes), "h2d")
            return self
        lo = min(self.views[n].offset for n in arrays)
        hi = max(self.views[n].offset + self.views[n].nbytes for n in arrays)
        stage = PinnedBuffer(ctx, hi - lo)
        for name, arr in arrays.items():
            v = self.views[name]
            dst = stage.view(v.offset - lo, v.shape, v.dtype)
            np.copyto(dst, np.asarray(arr).reshape(v.shape), casting="same_kind")
        check(ctx.lib.mvosr_memcpy_h2d_async(ctx.handle, self.ptr + lo, stage.ptr, hi - lo), "h2d_async")
        check(ctx.lib.mvosr_upload_fence(ctx.handle), "upload_fence")
        stage.free(MARK_UPLOAD)           # (back to the cache; its next user waits for the copy)
        return self

    def staging(self):
        """A page-locked image of the whole block for the caller to fill in place (``stage.view`` at each view's
        offset); ``commit(stage)`` then uploads it with one asynchronous copy."""
        return PinnedBuffer(self.ctx, self.nbytes)

    def commit(self, stage):
        ctx = self.ctx
        check(ctx.lib.mvosr_memcpy_h2d_async(ctx.handle, self.ptr, stage.ptr, self.nbytes), "h2d_async")
        check(ctx.lib.mvosr_upload_fence(ctx.handle), "upload_fence")
        stage.free(MARK_UPLOAD)
        return self

    def commit_ranges(self, stage, ranges, last=False):
        """Parts of a staging image, as they become ready: ``ranges`` = (offset, bytes) within the block, one asynchronous
        copy each on the upload stream.  ``last``: the compute stream waits for everything uploaded so far and the staging
        buffer goes back to the cache."""
        ctx = self.ctx
        for off, nb in ranges:
            if nb > 0:
                check(ctx.lib.mvosr_memcpy_h2d_async(ctx.handle, self.ptr + int(off), stage.ptr + int(off), int(nb)), "h2d_async")
        if last:
            check(ctx.lib.mvosr_upload_fence(ctx.handle), "upload_fence")
            stage.free(MARK_UPLOAD)
        return self

    def zero(self):
        check(self.ctx.lib.mvosr_memset(self.ctx.handle, self.ptr, 0, self.nbytes), "memset")
        return self

    def mark(self, marked=True):
        """The block's last use is what has been queued SO FAR (mvosr_block_mark): a later ``free`` does not make its next
        user wait for work queued after this point.  ``mark(False)``: the block is used again."""
        if self.ptr:
            check(self.ctx.lib.mvosr_block_mark(self.ctx.handle, self.ptr, 1 if marked else 0), "mvosr_block_mark")
        return self

    def prefetch(self):
        """Queue the download of the whole block behind the work launched so far and mark that point with an event:
        a later ``read`` waits for the event only — not for work queued after it (the next chunk's kernels)."""
        ctx = self.ctx
        if self.nbytes > self.STAGE_LIMIT:
            return self
        self._mirror = None
        self._pf_stage = PinnedBuffer(ctx, self.nbytes)
        check(ctx.lib.mvosr_memcpy_d2h_async(ctx.handle, self._pf_stage.ptr, self.ptr, self.nbytes), "d2h_async")
        if getattr(self, "_pf_event", None) is None:
            self._pf_event = ctx.event()
        ctx.record(self._pf_event)
        return self

    def mark_done(self):
        """Like ``prefetch`` — the block on its way to a page-locked image behind the work launched so far, an event behind the copy —
        but the copy is made by a KERNEL (mvosr_memcpy_d2h_kernel), not by a copy engine.  For the chunks of a streamed batch: a
        ``hipMemcpyAsync`` queued behind long kernels parks the SDMA engine HIP assigns it until they finish, and an upload of another
        stream that lands on the same engine waits with it — the reference-exact batch call ran at 95 or 118 k frames/s by process for
        exactly that (LABNOTES 10.14); a copy issued AFTER the kernels waits the other way round, behind an upload in flight."""
        ctx = self.ctx
        if self.nbytes > self.STAGE_LIMIT:
            return self
        self.invalidate()
        self._pf_stage = PinnedBuffer(ctx, self.nbytes)
        check(ctx.lib.mvosr_memcpy_d2h_kernel(ctx.handle, self._pf_stage.ptr, self.ptr, self.nbytes), "d2h_kernel")
        if getattr(self, "_pf_event", None) is None:
            self._pf_event = ctx.event()
        ctx.record(self._pf_event)
        return self

    def ready(self):
        """True when a ``read`` would not wait: the block's host copy exists, or the download queued by ``prefetch`` / ``mark_done`` has finished."""
        if self._mirror is not None:
            return True
        if getattr(self, "_pf_stage", None) is None:
            return False
        done = C.c_int(0)
        check(self.ctx.lib.mvosr_event_query(self.ctx.handle, self._pf_event, C.byref(done)), "event_query")
        return bool(done.value)

    def invalidate(self):
        self._mirror = None
        if getattr(self, "_pf_stage", None) is not None:
            self._pf_stage.free()
            self._pf_stage = None

    def read(self, view):
        """The view's contents as a NumPy array.  The first read after a launch brings the WHOLE block to the host (one
        copy + one synchronisation); later reads are served from that copy."""
        ctx = self.ctx
        if self.nbytes > self.STAGE_LIMIT:
            out = np.empty(view.shape, dtype=view.dtype)
            check(ctx.lib.mvosr_memcpy_d2h(ctx.handle, addr(out), view.ptr, view.nbytes), "d2h")
            return out
        if self._mirror is None and getattr(self, "_pf_stage", None) is not None:
            check(ctx.lib.mvosr_event_sync(ctx.handle, self._pf_event), "event_sync")
            self._mirror = np.array(self._pf_stage.view(0, (self.nbytes,), np.uint8), copy=True)
            self._pf_stage.free(MARK_IDLE)
            self._pf_stage = None
        if self._mirror is None:
            stage = PinnedBuffer(ctx, self.nbytes)
            check(ctx.lib.mvosr_memcpy_d2h_async(ctx.handle, stage.ptr, self.ptr, self.nbytes), "d2h_async")
            ctx.sync()
            self._mirror = np.array(stage.view(0, (self.nbytes,), np.uint8), copy=True)
            stage.free(MARK_IDLE)
        return np.array(self._mirror[view.offset:view.offset + view.nbytes].view(view.dtype).reshape(view.shape), copy=True)

    def free(self):
        if getattr(self, "_pf_stage", None) is not None:
            self._pf_stage.free()
            self._pf_stage = None
        if getattr(self, "_pf_event", None) is not None:
            self.ctx.lib.mvosr_event_destroy(self.ctx.handle, self._pf_event)
            self._pf_event = None
        if self.ptr:
            self.ctx.lib.mvosr_free(self.ctx.handle, self.ptr)
            self.ptr = None
        self._mirror = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One device + one HIP stream (mvosr_ctx)."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.mvosr_ctx_create(int(device), C.byref(h)), "mvosr_ctx_create")
        self.handle = h
        self.device = int(device)
        name = C.create_string_buffer(128)
        ncu, lds = C.c_int(), C.c_int()
        check(self.lib.mvosr_ctx_device_info(self.handle, name, 128, C.byref(ncu), C.byref(lds)))
        self.name, self.n_cu, self.lds_per_block = name.value.decode(), ncu.value, lds.value
        self.numa_node = int(self.lib.mvosr_device_numa_node(int(device)))
        self.pinned_cpus = pin_thread_to_node(self.numa_node)

    def to_device(self, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        return DeviceBuffer(self, arr.shape, arr.dtype).upload(arr)

    def empty(self, shape, dtype):
        return DeviceBuffer(self, shape, dtype)

    def zeros(self, shape, dtype):
        return DeviceBuffer(self, shape, dtype).fill(0)

    def block(self, spec):
        return DeviceBlock(self, spec)

    def alloc_stats(self):
        """{hip_malloc, hip_free, host_malloc, host_free, cache_hits, cached_device_bytes, cached_host_bytes, live_blocks}"""
        out = (C.c_int64 * 8)()
        check(self.lib.mvosr_ctx_alloc_stats(self.handle, out, 8), "mvosr_ctx_alloc_stats")
        keys = ("hip_malloc", "hip_free", "host_malloc", "host_free", "cache_hits", "cached_device_bytes", "cached_host_bytes", "live_blocks")
        return dict(zip(keys, (int(v) for v in out)))

    def trim(self):
        check(self.lib.mvosr_ctx_trim(self.handle), "mvosr_ctx_trim")

    def workspace_limit(self, nbytes):
        """Cap the triangulation kernels' grow-only workspace (0: none): chunks that would need more take the host's triangulations."""
        check(self.lib.mvosr_ctx_workspace_limit(self.handle, int(nbytes)), "mvosr_ctx_workspace_limit")

    def sync(self):
        check(self.lib.mvosr_ctx_sync(self.handle), "mvosr_ctx_sync")

    def set_stream(self, stream_ptr):
        check(self.lib.mvosr_ctx_set_stream(self.handle, C.c_void_p(stream_ptr)), "mvosr_ctx_set_stream")

    def event(self):
        ev = C.c_void_p()
        check(self.lib.mvosr_event_create(self.handle, C.byref(ev)), "event_create")
        return ev

    def record(self, ev):
        check(self.lib.mvosr_event_record(self.handle, ev), "event_record")

    def elapsed_ms(self, start, stop):
        ms = C.c_float()
        check(self.lib.mvosr_event_elapsed_ms(self.handle, start, stop, C.byref(ms)), "event_elapsed")
        return float(ms.value)

    def profile(self, enable=True):
        check(self.lib.mvosr_ctx_profile(self.handle, 1 if enable else 0), "mvosr_ctx_profile")

    def profile_read(self, call_index):
        a, b = C.c_float(), C.c_float()
        check(self.lib.mvosr_ctx_profile_read(self.handle, int(call_index), C.byref(a), C.byref(b)), "mvosr_ctx_profile_read")
        return float(a.value), float(b.value)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mvosr_ctx_destroy(self.handle)
            self.handle = None


_contexts = {}


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            cpus |= set(range(int(a), int(b or a) + 1))
    return cpus


def pin_thread_to_node(node):
    """Keep the calling thread — and the threads it starts from now on: the packer's pool, the upload helpers — on the CPUs
    of NUMA node ``node`` (the device's, see mvosr_device_numa_node): page-locked staging memory is then allocated next to the
    device's PCIe root port and packed by cores next to it (two-socket host, 32 768 frames of 2000 features end to end:
    502-505 k frames/s there, 460-488 k on the other node, 441-468 k left to the scheduler).  Nothing is done when the node
    is unknown, when the thread is already confined to one node (a launcher's own pinning is respected), or WITHOUT
    MVOSR_AFFINITY=1: the affinity of the calling thread is the host application's business — threads it starts later inherit it
    — so the pinning is opt-in (bench.py opts in and says so).  Returns the CPU set applied, or None."""
    if node is None or node < 0 or os.environ.get("MVOSR_AFFINITY", "0") != "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        mine = os.sched_getaffinity(0)
        local = _cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read()) & mine
        if not local or local == mine:
            return None
        import glob
        for other in glob.glob("/sys/devices/system/node/node[0-9]*/cpulist"):
            cp = _cpulist(open(other).read())
            if mine <= cp:                      # confined to a single node already
                return None
        os.sched_setaffinity(0, local)
        return local
    except (OSError, ValueError):
        return None


def default_context(device=0):
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]
