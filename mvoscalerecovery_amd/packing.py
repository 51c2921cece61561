"""Host-side packing of frames into the HBM layout the kernels read (include/mvosr.h, mvosr_batch).

Per frame the path consumes: the features that pass the vanishing-row filter
(/root/reference/src/scale_calculator.py:252-254), as raw ``x, y, z`` (before feature_remap,
which the kernels fuse into their load) and the pixel row ``v``; the first triangulation
``Delaunay(feature2d).simplices`` (:257-258) and the second one over the features that survive
the depth-order vote (:266-267).  Both triangulations are *inputs* of the GPU path: Qhull runs on
the host (SciPy), exactly where the reference calls it, and its ``simplices`` are carried
verbatim (int32, vertex order inside a row untouched — SURVEY.md fact 4).

Layout: structure of arrays.  ``x|y|z|v`` are float64 planes holding all frames back to back,
each frame's segment starting at an even element (16-byte aligned); ``tri1``/``tri2`` are
(T,3) int32 row-major with per-frame offsets in triangles.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

VANISH = 185        # /root/reference/src/scale_calculator.py:22


def delaunay_simplices(points2d):
    """scipy.spatial.Delaunay(points).simplices — the reference's call (:257-258,:266-267)."""
    from scipy.spatial import Delaunay
    return np.ascontiguousarray(Delaunay(points2d).simplices, dtype=np.int32)


def qhull_rows_host(points2d):
    """``mvosr_qhull_rows_host``: SciPy's ``Delaunay(points).simplices`` — set, order and rotation — by the C replay of Qhull's
    run on the host (csrc/mvosr_qhull_host.c), or ``None`` where it declines (a decision inside a roundoff guard band: the caller
    asks SciPy).  One point set, ~0.2 us per point."""
    from . import _lib
    lib = _lib.load()
    p = np.ascontiguousarray(points2d, dtype=np.float64)
    if p.ndim != 2 or p.shape[1] != 2:
        raise ValueError("points must be (n, 2)")
    n = p.shape[0]
    rows = np.empty((2 * n + 8, 3), dtype=np.int32)
    cnt = np.zeros(1, dtype=np.int32)
    rc = lib.mvosr_qhull_rows_host(_lib.addr(p), n, 2, _lib.addr(rows), rows.shape[0], _lib.addr(cnt), None)
    qhull_rows_host.last_reason = int(rc)
    if rc < 0:
        _lib.check(rc, "mvosr_qhull_rows_host")
    if rc != 0:
        return None
    return rows[:int(cnt[0])]


def delaunay_simplices_fast(points2d):
    """The first triangulation of a per-frame call (:257): the host replay where it accepts the set, SciPy otherwise — the same
    rows either way (selfcheck.py holds the replay to the installed SciPy)."""
    rows = qhull_rows_host(points2d)
    if rows is None:
        delaunay_simplices_fast.declined = getattr(delaunay_simplices_fast, "declined", 0) + 1
        return delaunay_simplices(points2d)
    return rows


def qhull_rows_host_or_none():
    """The host replay (``qhull_rows_host``) as a callable ``points -> rows or None`` for the self-check."""
    return qhull_rows_host


def _delaunay_job(points2d, fast=False, canonical=False):
    try:
        rows = delaunay_simplices_fast(points2d) if fast else delaunay_simplices(points2d)
        return canonical_rows(rows) if canonical else rows
    except Exception as exc:  # QhullError etc.: re-raised in frame order by the caller
        return exc


# ---- the host's Delaunay stage on a persistent process pool -------------------------------------
# Qhull is a few ms per 2000 points and holds the GIL for about a third of that, so the stage scales
# with processes, not threads.  The pool is created once per worker count and kept (forking per call
# costs more than the triangulations); SciPy is imported BEFORE the fork so that the children inherit
# it instead of importing it each on their first job; points and simplices travel through two
# shared-memory segments (files in /dev/shm, grow-only, reused between calls), so that the parent handles a few bytes
# per frame instead of pickling ~70 KB through a pipe — with 64 workers the pipe was the bottleneck.
_pool = None
_pool_size = 0
_shm = {}                 # role -> _Segment owned by this (parent) process
_shm_attached = {}        # worker side: path -> _Segment


def _get_pool(workers):
    global _pool, _pool_size
    if _pool is None or _pool_size != workers:
        if _pool is not None:
            _pool.terminate()
        import multiprocessing as mp
        import scipy.spatial                                   # noqa: F401  (inherited by the forked workers)
        _pool = mp.get_context("fork").Pool(workers)
        _pool_size = workers
    return _pool


def canonical_rows(tri):
    """Rows of a triangulation in the canonical form of ``check_triangle="fixed"``: vertex ids ascending inside a row,
    rows in lexicographic order — a function of the triangle SET alone (what ``mvosr_delaunay_batch`` emits, so that
    SciPy's rows and the device's rows of the same triangulation are the same array)."""
    t = np.sort(np.asarray(tri, dtype=np.int32).reshape(-1, 3), axis=1)
    if t.shape[0] == 0:
        return t
    if t.min() >= 0 and t.max() < (1 << 21):
        # (one 63-bit key per row, sorted, unpacked: a third of np.lexsort's time on 4 000 rows — the host path of a "fixed"-mode
        # estimator canonicalises two triangulations per frame)
        key = np.sort((t[:, 0].astype(np.int64) << 42) | (t[:, 1].astype(np.int64) << 21) | t[:, 2].astype(np.int64))
        out = np.empty((len(key), 3), dtype=np.int32)
        out[:, 0], out[:, 1], out[:, 2] = key >> 42, (key >> 21) & 0x1FFFFF, key & 0x1FFFFF
        return out
    return np.ascontiguousarray(t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))])


def delaunay_gpu(ctx, point_sets, keeps=None, rows="canonical", order_out=False):
    """``rows="qhull"``: ``mvosr_delaunay_qhull_batch`` — SciPy's rows themselves, order and rotation (the kernel replays
    Qhull's insertion order); ``order_out``: also the per-point insertion steps in ``delaunay_gpu.last_order``.

    The device stage for the triangulations (``mvosr_delaunay_batch``): per point set the (T,3) int32
    rows — the triangle set SciPy returns for points in general position, in canonical form (:func:`canonical_rows`) —
    or ``None`` where the kernel declined (duplicate / collinear / cocircular points within its guard bands, fewer than
    3 points): those sets are for the host's Qhull.  ``keeps`` (optional, one int array per set): only the points with
    ``keep >= 0`` are triangulated, numbered by their rank among those.  This host-array form is what tests and small
    callers use; the batch path of ``ScaleEstimator`` keeps points, masks and rows on the device."""
    from . import _lib
    F = len(point_sets)
    if F == 0:
        return []
    cnt = np.array([len(p) for p in point_sets], dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(cnt.astype(np.int64))])
    toff = 2 * off
    total = max(int(off[-1]), 1)
    uv = np.zeros((total, 2), dtype=np.float64)
    kp = np.zeros(total, dtype=np.int32)
    for f, p in enumerate(point_sets):
        if len(p):
            uv[off[f]:off[f + 1]] = np.asarray(p, dtype=np.float64).reshape(-1, 2)
            if keeps is not None:
                kp[off[f]:off[f + 1]] = np.asarray(keeps[f], dtype=np.int32)
    d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
    d_keep = ctx.to_device(kp) if keeps is not None else None
    d_off, d_cnt, d_toff = ctx.to_device(off[:-1].astype(np.int64)), ctx.to_device(cnt), ctx.to_device(toff[:-1].astype(np.int64))
    d_tri = ctx.empty((2 * total, 3), np.int32)
    d_tcnt, d_st, d_used = ctx.zeros(F, np.int32), ctx.zeros(F, np.int32), ctx.zeros(F, np.int32)
    d_ord = None
    if rows == "qhull":
        d_ord = ctx.zeros(total, np.int32) if order_out else None
        _lib.check(ctx.lib.mvosr_delaunay_qhull_batch(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr,
                                                      d_keep.ptr if d_keep is not None else None, int(cnt.max()), d_toff.ptr,
                                                      d_tri.ptr, d_tcnt.ptr, d_used.ptr, d_st.ptr,
                                                      d_ord.ptr if d_ord is not None else None), "mvosr_delaunay_qhull_batch")
    else:
        _lib.check(ctx.lib.mvosr_delaunay_batch(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr,
                                                d_keep.ptr if d_keep is not None else None, int(cnt.max()), d_toff.ptr,
                                                d_tri.ptr, d_tcnt.ptr, d_used.ptr, d_st.ptr), "mvosr_delaunay_batch")
    ctx.sync()
    tri, tcnt, st, used = d_tri.download(), d_tcnt.download(), d_st.download(), d_used.download()
    if d_ord is not None:
        o = d_ord.download()
        delaunay_gpu.last_order = [o[off[f]:off[f + 1]] for f in range(F)]
        d_ord.free()
    for b in (d_u, d_v, d_off, d_cnt, d_toff, d_tri, d_tcnt, d_st, d_used) + ((d_keep,) if d_keep is not None else ()):
        b.free()
    delaunay_gpu.last_status = st                      # (bits 8.. of a declined frame's status say why: mvosr_delaunay.hip)
    delaunay_gpu.last_used = used
    return [np.ascontiguousarray(tri[toff[f]:toff[f] + tcnt[f]]) if st[f] == 0 else None for f in range(F)]


def delaunay_gpu_max_points():
    """Largest point set the device stage takes (its points, grid and rows live in one workgroup's LDS)."""
    from . import _lib
    return int(_lib.load().mvosr_delaunay_max_points())


def delaunay_qhull_max_points():
    """Largest point set ``mvosr_delaunay_qhull_batch`` takes (its facet ids are 16-bit)."""
    from . import _lib
    return int(_lib.load().mvosr_delaunay_qhull_max_points())


def delaunay_gpu_or_host(ctx, point_sets, workers=0, canonical=True):
    """``triangulation="gpu"`` for host-side point sets: :func:`delaunay_gpu` for every set it accepts, SciPy/Qhull (the
    reference's call) for the ones it declines — degenerate inputs, sets too large for its LDS plan — and for sets SciPy
    itself rejects, whose exception is returned in place of the rows (as :func:`delaunay_many` does).  ``canonical``:
    the host's rows are brought to the device stage's row form."""
    out = [None] * len(point_sets)
    cap = delaunay_gpu_max_points()
    small = [f for f, p in enumerate(point_sets) if 3 <= len(p) <= cap]
    if small:
        for f, t in zip(small, delaunay_gpu(ctx, [point_sets[f] for f in small])):
            out[f] = t
    rest = [f for f, t in enumerate(out) if t is None]
    if rest:
        for f, t in zip(rest, delaunay_many([point_sets[f] for f in rest], workers)):
            out[f] = canonical_rows(t) if (canonical and not isinstance(t, Exception)) else t
    delaunay_gpu_or_host.last_host_fraction = len(rest) / max(len(point_sets), 1)
    return out


def start_pool(workers=None):
    """Create the Delaunay worker pool NOW.  The workers are forked, so this belongs before the first GPU call of the
    process (a HIP context, torch.cuda, RCCL all start runtime threads, and a fork taken while one of them holds a
    lock can deadlock the child): ``ScaleEstimator.__init__`` and ``bench.py`` call it before they create the context.
    Returns the number of workers (0: the stage runs in this process)."""
    workers = resolve_workers(workers)
    if workers > 1:
        _get_pool(workers)
        return workers
    return 0


class _Segment:
    """A file in /dev/shm (or the temp dir) mapped into memory: plain mmap, so that neither side
    involves multiprocessing's resource tracker."""

    def __init__(self, path, size=None):
        import mmap
        import os
        self.path = path
        flags = os.O_RDWR | (os.O_CREAT | os.O_EXCL if size is not None else 0)
        fd = os.open(path, flags, 0o600)
        try:
            if size is not None:
                os.ftruncate(fd, size)
            self.size = os.fstat(fd).st_size
            self.buf = mmap.mmap(fd, self.size)
        finally:
            os.close(fd)

    def close(self, unlink=False):
        import os
        try:
            self.buf.close()
        except (BufferError, ValueError):
            pass                                               # a NumPy view is still alive; the mapping goes with it
        if unlink:
            try:
                os.unlink(self.path)
            except OSError:
                pass


_seg_serial = 0


def _shm_segment(role, nbytes):
    """Grow-only segment owned by the parent."""
    global _seg_serial
    import os
    import tempfile
    seg = _shm.get(role)
    if seg is None or seg.size < nbytes:
        if seg is not None:
            seg.close(unlink=True)
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        _seg_serial += 1
        seg = _Segment(os.path.join(base, "mvosr_%d_%s_%d" % (os.getpid(), role, _seg_serial)), max(int(nbytes * 1.25), 1 << 20))
        _shm[role] = seg
    return seg


def _shm_attach(path, keep):
    """Worker side: map a segment by path; mappings of segments the parent has replaced (unlinked) are dropped."""
    import os
    for old in [k for k in _shm_attached if k not in keep and not os.path.exists(k)]:
        _shm_attached.pop(old).close()
    seg = _shm_attached.get(path)
    if seg is None:
        seg = _shm_attached[path] = _Segment(path)
    return seg


def shutdown_pool():
    global _pool, _pool_size
    if _pool is not None:
        # let the workers leave by themselves (close = "no more jobs"): a SIGTERM is caught by signal handlers a
        # preloaded tool may have installed in the forked workers (rocprofv3 then "finalizes" in a process that never
        # touched the GPU and can hang); terminate() only for workers still there after a grace period
        try:
            _pool.close()
            for p in list(getattr(_pool, "_pool", [])):
                p.join(2.0)
        except Exception:
            pass
        _pool.terminate()
        _pool, _pool_size = None, 0
    for role in list(_shm):
        _shm.pop(role).close(unlink=True)


import atexit as _atexit                                           # noqa: E402
_atexit.register(shutdown_pool)


def _delaunay_shm_job(job):
    """Worker: triangulate the listed point sets of the input segment, write the rows to the output segment."""
    in_name, out_name, items = job[:3]
    fast = bool(job[3]) if len(job) > 3 else False
    canonical = bool(job[4]) if len(job) > 4 else False
    pin, pout = _shm_attach(in_name, (in_name, out_name)), _shm_attach(out_name, (in_name, out_name))
    results = []
    for in_off, n, out_off, cap in items:
        pts = np.ndarray((n, 2), dtype=np.float64, buffer=pin.buf, offset=16 * in_off)
        r = _delaunay_job(pts, fast, canonical)
        if isinstance(r, Exception):
            results.append(r)
        elif r.shape[0] > cap:
            results.append(RuntimeError("Delaunay returned %d rows for %d points" % (r.shape[0], n)))
        else:
            np.ndarray((r.shape[0], 3), dtype=np.int32, buffer=pout.buf, offset=12 * out_off)[:] = r
            results.append(int(r.shape[0]))
    return results


def available_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a container
    on a 256-thread host may be limited to 16 — more Delaunay workers than that only get throttled)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                quota, period = int(fq.read()), int(fp.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def resolve_workers(workers):
    """``None`` -> ``MVOSR_DELAUNAY_WORKERS`` if set, else one worker per available CPU (batch paths), shared evenly
    between the ranks of a node (``LOCAL_WORLD_SIZE`` of torch.distributed.run); anything else as given."""
    if workers is not None:
        return int(workers)
    import os
    env = os.environ.get("MVOSR_DELAUNAY_WORKERS")                      # e.g. 0 under a profiler: no forked workers at all
    if env is not None and env.strip().lstrip("-").isdigit():
        return max(0, int(env))
    try:
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        ranks = 1
    return max(1, available_cpus() // ranks)


class _DelaunayHandle:
    """Result of :func:`delaunay_submit`: ``get()`` waits for the workers and returns, per point set, the (T,3) int32
    simplices or the exception SciPy raised for it."""

    def __init__(self, n, done=None, async_result=None, rows=None, out_off=None, canonical=False):
        self.n, self._done, self._async, self._rows, self._out_off = n, done, async_result, rows, out_off
        self.canonical = bool(canonical)         # the rows come back in canonical form already (the workers did it)

    def ready(self):
        """True when ``get()`` would not wait."""
        return self._done is not None or self._async.ready()

    def get(self):
        if self._done is None:
            out, f = [], 0
            for res in self._async.get():
                for r in res:
                    out.append(r if isinstance(r, Exception) else self._rows[self._out_off[f]:self._out_off[f] + r].copy())
                    f += 1
            self._done, self._async, self._rows = out, None, None
        return self._done


class _JoinedHandle:
    """Several handles as one: ``parts`` = [(positions, handle)], the positions of a part's sets in the joined list."""

    def __init__(self, n, parts):
        self.n, self.parts = n, parts
        self.canonical = all(getattr(h, "canonical", False) for _, h in parts) if parts else False

    def ready(self):
        return all(h.ready() for _, h in self.parts)

    def get(self):
        out = [None] * self.n
        for pos, h in self.parts:
            for i, r in zip(pos, h.get()):
                out[int(i)] = r
        return out


def delaunay_submit(point_sets, workers=0, slot=0, fast=False, canonical=False, background=False):
    """Start triangulating many point sets on the process pool and return at once (a handle with ``get()``): the host
    stage that bounds end-to-end throughput (SURVEY.md §7 hard part 1) runs while the caller packs, uploads and
    launches the GPU stages of other chunks.  ``slot`` names the pair of shared-memory segments the call uses — calls
    that are in flight at the same time need different slots.  ``fast``: the C replay of Qhull's run (:func:`qhull_rows_host`) where it
    accepts a set, SciPy otherwise — the same rows (the default estimator's few-frames path; ``triangulation="scipy"`` never asks).
    ``background``: a SINGLE set goes to the pool as well (when there is one) instead of being triangulated here — for callers that
    have other work before they ask for the rows (a declined frame discovered in the middle of a batch call: 2.6 ms of SciPy off the
    thread that packs and launches the following chunks)."""
    n = len(point_sets)
    workers = resolve_workers(workers)
    if not (workers and workers > 1 and (n > 1 or (background and n == 1))):
        return _DelaunayHandle(n, done=[_delaunay_job(p, fast, canonical) for p in point_sets], canonical=canonical)
    pool = _get_pool(int(workers))
    counts = np.array([len(p) for p in point_sets], dtype=np.int64)
    in_off = np.concatenate([[0], np.cumsum(counts)])
    caps = 2 * counts + 8                                       # a planar triangulation has < 2n triangles
    out_off = np.concatenate([[0], np.cumsum(caps)])
    pin = _shm_segment("in%d" % slot, 16 * max(int(in_off[-1]), 1))
    pout = _shm_segment("out%d" % slot, 12 * max(int(out_off[-1]), 1))
    allpts = np.ndarray((int(in_off[-1]), 2), dtype=np.float64, buffer=pin.buf)
    for f, p in enumerate(point_sets):
        allpts[in_off[f]:in_off[f + 1]] = p
    per_job = max(1, min(16, n // (int(workers) * 4)))
    jobs = [(pin.path, pout.path, [(int(in_off[f]), int(counts[f]), int(out_off[f]), int(caps[f]))
                                    for f in range(j, min(n, j + per_job))], bool(fast), bool(canonical)) for j in range(0, n, per_job)]
    rows = np.ndarray((int(out_off[-1]), 3), dtype=np.int32, buffer=pout.buf)
    return _DelaunayHandle(n, async_result=pool.map_async(_delaunay_shm_job, jobs), rows=rows, out_off=out_off, canonical=canonical)


def delaunay_many(point_sets, workers=0):
    """Triangulate many point sets, optionally on the process pool.  Returns per set the (T,3) int32 simplices, or
    the exception SciPy raised for it."""
    return delaunay_submit(point_sets, workers).get()


@dataclass
class PackedFrames:
    """NumPy-side image of an ``mvosr_batch``."""
    n_frames: int
    feat_off: np.ndarray            # int64 [F]
    feat_cnt: np.ndarray            # int32 [F]
    x: np.ndarray                   # float64 [total padded]
    y: np.ndarray
    z: np.ndarray
    v: np.ndarray
    u: np.ndarray                   # host only (Delaunay input); never uploaded
    lower_index: list               # per frame: indices of the kept features in the caller's arrays
    tri1_off: np.ndarray = None     # int64 [F+1]
    tri1: np.ndarray = None         # int32 [T1,3]
    tri2_off: np.ndarray = None
    tri2: np.ndarray = None
    n2_expected: np.ndarray = None  # int32 [F]
    max_feat: int = 0
    tri2_ids: int = 0               # 0: tri2 indexes the survivors (SciPy's numbering), 1: the frame's features
    tri2_order: np.ndarray = None   # int32 [T2]: where the k-th ORIGINAL row of a frame went (layouts that permute tri2's rows)
    tile_w: int = 0                 # tile index of dense frames (mvosr_batch.tile_*): 0 = none
    tile_base: np.ndarray = None    # int64 [F+1]
    tile1_off: np.ndarray = None    # int32 [tile_base[F]]
    tile2_off: np.ndarray = None
    tile_far: np.ndarray = None     # float64: the far rows' vertices, copied out of the planes (mvosr_batch.tile_far)
    tile_far_off: np.ndarray = None # int64 [F+1], in doubles
    extra: dict = field(default_factory=dict)

    @property
    def total_padded(self):
        return int(self.extra.get("total_padded", self.x.shape[0] if self.x is not None else 0))

    def frame_slice(self, f):
        o = int(self.feat_off[f])
        return slice(o, o + int(self.feat_cnt[f]))

    def algorithmic_bytes(self):
        """B = 8*(3N+N) + 12*(T1+T2) + 12 per frame, summed over the batch."""
        n = int(self.feat_cnt.sum())
        t1 = int(self.tri1_off[-1]) if self.tri1_off is not None else 0
        t2 = int(self.tri2_off[-1]) if self.tri2_off is not None else 0
        return 8 * 4 * n + 12 * (t1 + t2) + 12 * self.n_frames


FRAME_ALIGN = 16           # doubles


def pack_features(feature3ds, feature2ds, vanish=VANISH):
    """Apply the vanishing-row filter and lay the survivors out as planes."""
    F = len(feature3ds)
    cnt = np.zeros(F, dtype=np.int32)
    lower_index = []
    for f in range(F):
        f2 = np.asarray(feature2ds[f], dtype=np.float64)
        idx = np.nonzero(f2[:, 1] > vanish)[0] if f2.size else np.zeros(0, dtype=np.int64)   # :252
        lower_index.append(idx)
        cnt[f] = idx.shape[0]
    # every frame's segment starts on a 128-byte line of the planes (16 doubles): a frame's loads then never share a
    # line with its neighbour's, and the 4 KB tile blocks of the dense kernel are line-aligned (the C ABI asks for
    # even offsets only)
    padded = (cnt.astype(np.int64) + (FRAME_ALIGN - 1)) & ~np.int64(FRAME_ALIGN - 1)
    off = np.zeros(F, dtype=np.int64)
    if F:
        off[1:] = np.cumsum(padded)[:-1]
    total = int(padded.sum()) if F else 0
    x = np.zeros(max(total, 2), dtype=np.float64)
    y = np.zeros_like(x)
    z = np.zeros_like(x)
    v = np.zeros_like(x)
    u = np.zeros_like(x)
    for f in range(F):
        n = int(cnt[f])
        if n == 0:
            continue
        idx = lower_index[f]
        f3 = np.asarray(feature3ds[f], dtype=np.float64)
        f2 = np.asarray(feature2ds[f], dtype=np.float64)
        o = int(off[f])
        x[o:o + n] = f3[idx, 0]
        y[o:o + n] = f3[idx, 1]
        z[o:o + n] = f3[idx, 2]
        u[o:o + n] = f2[idx, 0]
        v[o:o + n] = f2[idx, 1]
    return PackedFrames(F, off, cnt, x, y, z, v, u, lower_index, max_feat=int(cnt.max()) if F else 0)


def native_packable(feature3ds, feature2ds, writable=False):
    """Can the C packer (mvosr_pack_*) read these frames in place?  C-contiguous float64 (N,3) / (N,2) NumPy arrays —
    ``writable``: and may it write feature_remap into the feature3d arrays (not into a read-only one)?"""
    f64 = np.dtype(np.float64)
    for a, b in zip(feature3ds, feature2ds):
        if not (type(a) is np.ndarray and type(b) is np.ndarray and a.dtype == f64 and b.dtype == f64 and a.ndim == 2 and b.ndim == 2
                and a.flags.c_contiguous and b.flags.c_contiguous and a.shape[1] == 3 and b.shape[1] == 2 and a.shape[0] == b.shape[0]
                and (a.flags.writeable or not writable)):
            return False
    return True


def pack_layout(cnt):
    """Frame offsets (every frame's segment on a 128-byte line of the planes) and the planes' length for per-frame counts."""
    padded = (cnt.astype(np.int64) + (FRAME_ALIGN - 1)) & ~np.int64(FRAME_ALIGN - 1)
    off = np.zeros(len(cnt), dtype=np.int64)
    if len(cnt):
        off[1:] = np.cumsum(padded)[:-1]
    return off, max(int(padded.sum()) if len(cnt) else 0, 2)


def _pack_tris(tris):
    F = len(tris)
    off = np.zeros(F + 1, dtype=np.int64)
    for f, t in enumerate(tris):
        off[f + 1] = off[f] + (0 if t is None else int(t.shape[0]))
    flat = np.zeros((max(int(off[-1]), 1), 3), dtype=np.int32)
    for f, t in enumerate(tris):
        if t is not None and t.shape[0]:
            flat[off[f]:off[f + 1]] = t
    return off, flat


def submit_tri1(pf: PackedFrames, workers=0, slot=0, fast=False, background=False):
    """Start the first triangulation of every frame (SciPy on the packed (u,v)); finish with :func:`attach_tri1`."""
    pts = []
    for f in range(pf.n_frames):
        s = pf.frame_slice(f)
        pts.append(np.stack([pf.u[s], pf.v[s]], axis=1))
    return delaunay_submit(pts, workers, slot, fast, canonical=bool(pf.extra.get("canonical")), background=background)


def attach_tri1(pf: PackedFrames, tri1s=None, workers=0):
    """First triangulation per frame: given (a list, or the handle :func:`submit_tri1` returned), or SciPy now."""
    if tri1s is None:
        tri1s = submit_tri1(pf, workers)
    done = bool(pf.extra.get("tri1_is_canonical"))
    if isinstance(tri1s, _DelaunayHandle):
        done = done or tri1s.canonical
        tri1s = tri1s.get()
    pf.extra["tri1_errors"] = {f: t for f, t in enumerate(tri1s) if isinstance(t, Exception)}
    tri1s = [None if isinstance(t, Exception) else np.ascontiguousarray(t, dtype=np.int32) for t in tri1s]
    if pf.extra.get("canonical") and not done:  # check_triangle="fixed": rows as a function of the triangle set alone (the workers' job where there are any)
        tri1s = [None if t is None else canonical_rows(t) for t in tri1s]
    pf.tri1_off, pf.tri1 = _pack_tris(tri1s)
    return pf


class _Tri2Handle:
    def __init__(self, todo, handle, n_frames):
        self.todo, self.handle, self.n_frames = todo, handle, n_frames
        self.canonical = bool(getattr(handle, "canonical", False))

    def ready(self):
        return self.handle.ready()

    def get(self):
        tri2s = [np.zeros((0, 3), dtype=np.int32)] * self.n_frames
        for f, t in zip(self.todo, self.handle.get()):
            tri2s[f] = t
        return tri2s


def survivor_points(pf: PackedFrames, valid_masks):
    """Per frame the (u, v) of the features the vote kept, in their ORIGINAL order (what the reference's second
    Delaunay call sees, :264-266), or ``None`` for frames that never make that call (<= 3 features, :263-270)."""
    perms = pf.extra.get("perm") or [None] * pf.n_frames
    pts = []
    for f in range(pf.n_frames):
        s = pf.frame_slice(f)
        m = np.asarray(valid_masks[f], dtype=bool)
        u, v = pf.u[s], pf.v[s]
        if perms[f] is not None:
            inv = np.empty(len(perms[f]), dtype=np.int64)
            inv[perms[f]] = np.arange(len(perms[f]))
            u, v, m = u[inv], v[inv], m[inv]
        pts.append(np.stack([u[m], v[m]], axis=1) if len(m) > 3 else None)
    return pts


def submit_tri2(pf: PackedFrames, valid_masks, workers=0, slot=1, fast=False, background=False):
    """Start the second triangulation of every frame: SciPy over the features with ``valid_masks[f]`` (the vote result
    that came back from the GPU, in the PACKED order).  For frames that :func:`apply_locality_order` permuted, Delaunay
    still runs on the survivors in their original order — the reference's exact call.  Finish with :func:`attach_tri2`."""
    perms = pf.extra.get("perm") or [None] * pf.n_frames
    pts = []
    for f in range(pf.n_frames):
        s = pf.frame_slice(f)
        m = np.asarray(valid_masks[f], dtype=bool)
        u, v = pf.u[s], pf.v[s]
        if perms[f] is not None:
            inv = np.empty(len(perms[f]), dtype=np.int64)
            inv[perms[f]] = np.arange(len(perms[f]))
            u, v, m = u[inv], v[inv], m[inv]
        # <= 3 features below the vanishing row: the reference never makes the second call (:263-270)
        pts.append(np.stack([u[m], v[m]], axis=1) if len(m) > 3 else None)
    todo = [f for f, p in enumerate(pts) if p is not None]
    return _Tri2Handle(todo, delaunay_submit([pts[f] for f in todo], workers, slot, fast, canonical=bool(pf.extra.get("canonical")),
                                             background=background), pf.n_frames)


def attach_tri2(pf: PackedFrames, tri2s=None, valid_masks=None, workers=0, feature_ids=False):
    """Second triangulation per frame: given (numbered over the survivors in the caller's / original
    order, as SciPy returns it; a list or the handle of :func:`submit_tri2`), or SciPy now over the features with
    ``valid_masks[f]``.  Rows of frames that :func:`apply_locality_order` permuted are relabelled.
    ``feature_ids=True`` (needs the masks) renumbers the rows over the frame's packed features instead of over the
    survivors (``mvosr_batch.tri2_ids = MVOSR_TRI2_FEATURES``): dense frames then run without compaction."""
    perms = pf.extra.get("perm") or [None] * pf.n_frames
    if feature_ids and valid_masks is None:
        raise ValueError("feature-numbered tri2 needs the vote masks")
    if tri2s is None:
        assert valid_masks is not None
        tri2s = submit_tri2(pf, valid_masks, workers)
    done2 = False
    if isinstance(tri2s, _Tri2Handle):
        done2 = tri2s.canonical
        tri2s = tri2s.get()
    pf.extra["tri2_errors"] = {f: t for f, t in enumerate(tri2s) if isinstance(t, Exception)}
    tri2s = [None if isinstance(t, Exception) else np.ascontiguousarray(t, dtype=np.int32) for t in tri2s]
    if pf.extra.get("canonical") and not done2:
        tri2s = [None if t is None else canonical_rows(t) for t in tri2s]
    row_src = [None] * pf.n_frames          # row_src[f][i] = index, in the caller's original order, of the row now stored at i
    if valid_masks is not None:
        for f in range(pf.n_frames):
            if perms[f] is not None and tri2s[f] is not None and tri2s[f].shape[0]:
                m_new = np.asarray(valid_masks[f], dtype=bool)
                m_old = np.empty_like(m_new)
                m_old[perms[f]] = m_new
                tri2s[f], row_src[f] = _relabel_tri2(tri2s[f], m_old, perms[f], return_order=True)
    tiled = feature_ids and pf.extra.get("tile1") is not None
    tile2 = []
    orders = None
    if feature_ids:
        orders = []
        for f in range(pf.n_frames):
            n = int(pf.feat_cnt[f])
            offs = np.zeros((n + TILE_W - 1) // TILE_W + 1, dtype=np.int32)
            inv = np.zeros(0, dtype=np.int32)
            if tri2s[f] is not None and tri2s[f].shape[0]:
                survivors = np.nonzero(np.asarray(valid_masks[f], dtype=bool))[0].astype(np.int32)   # increasing: row order is kept
                tri2s[f] = survivors[tri2s[f]]
                t = tri2s[f].shape[0]
                src = row_src[f] if row_src[f] is not None else np.arange(t)
                if tiled:
                    tri2s[f], offs = _tile_sort(tri2s[f], n)
                    src = src[_tile_sort.last_order]
                inv = np.empty(t, dtype=np.int32)
                inv[src] = np.arange(t, dtype=np.int32)                                           # original row k is stored at inv[k]
            tile2.append(offs)
            orders.append(inv)
    pf.tri2_ids = 1 if feature_ids else 0
    pf.tri2_off, pf.tri2 = _pack_tris(tri2s)
    pf.tri2_order = None
    if orders is not None:
        pf.tri2_order = np.ascontiguousarray(np.concatenate(orders) if len(orders) else np.zeros(0, np.int32), dtype=np.int32)
        if pf.tri2_order.shape[0] != int(pf.tri2_off[-1]):
            pf.tri2_order = None
    if tiled:
        pf.extra["tile2"] = tile2
        _finish_tile_index(pf)
    if valid_masks is not None:
        pf.n2_expected = np.array([int(np.count_nonzero(m)) for m in valid_masks], dtype=np.int32)
    return pf


def tile_frames(pf: PackedFrames, repeats: int) -> PackedFrames:
    """Replicate a pool of packed frames ``repeats`` times (bench datasets: a pool of P unique
    frames tiled to the requested size)."""
    F = pf.n_frames
    padded_total = pf.total_padded
    feat_off = np.concatenate([pf.feat_off + r * padded_total for r in range(repeats)])
    out = PackedFrames(F * repeats, feat_off, np.tile(pf.feat_cnt, repeats),
                       np.tile(pf.x, repeats), np.tile(pf.y, repeats), np.tile(pf.z, repeats),
                       np.tile(pf.v, repeats), np.tile(pf.u, repeats), pf.lower_index * repeats,
                       max_feat=pf.max_feat)
    for name in ("tri1", "tri2"):
        off = getattr(pf, name + "_off")
        arr = getattr(pf, name)
        if off is None:
            continue
        t = int(off[-1])
        new_off = np.concatenate([off[:-1] + r * t for r in range(repeats)] + [np.array([repeats * t], dtype=np.int64)])
        setattr(out, name + "_off", new_off)
        setattr(out, name, np.tile(arr[:max(t, 1)], (repeats, 1)))
    if pf.n2_expected is not None:
        out.n2_expected = np.tile(pf.n2_expected, repeats)
    out.tri2_ids = pf.tri2_ids
    if pf.tri2_order is not None:
        out.tri2_order = np.tile(pf.tri2_order[:max(int(pf.tri2_off[-1]), 0)], repeats)
    if pf.tile_w and pf.tile1_off is not None and pf.tile2_off is not None:
        nt = int(pf.tile_base[-1])
        out.tile_w = pf.tile_w
        out.tile_base = np.concatenate([pf.tile_base[:-1] + r * nt for r in range(repeats)] + [np.array([repeats * nt], dtype=np.int64)])
        out.tile1_off = np.tile(pf.tile1_off, repeats)
        out.tile2_off = np.tile(pf.tile2_off, repeats)
        if pf.tile_far is not None:
            nf = int(pf.tile_far_off[-1])
            out.tile_far = np.tile(pf.tile_far[:max(nf, 1)], repeats)
            out.tile_far_off = np.concatenate([pf.tile_far_off[:-1] + r * nf for r in range(repeats)]
                                              + [np.array([repeats * nf], dtype=np.int64)])
    return out



# ---- locality ordering (dense frames) ----------------------------------------------------------
def _morton_key(u, v):
    """Interleave 16-bit quantised pixel coordinates (Z-order)."""
    def spread(a):
        a = a.astype(np.uint64)
        a = (a | (a << np.uint64(8))) & np.uint64(0x00FF00FF)
        a = (a | (a << np.uint64(4))) & np.uint64(0x0F0F0F0F)
        a = (a | (a << np.uint64(2))) & np.uint64(0x33333333)
        a = (a | (a << np.uint64(1))) & np.uint64(0x55555555)
        return a

    def quant(a):
        lo, hi = float(a.min()), float(a.max())
        if not hi > lo:
            return np.zeros(a.shape[0], dtype=np.uint64)
        return np.minimum(((a - lo) * (65535.0 / (hi - lo))).astype(np.uint64), np.uint64(65535))
    return spread(quant(u)) | (spread(quant(v)) << np.uint64(1))


def apply_locality_order(pf: PackedFrames, min_features=0):
    """Reorder a frame's features along a Z-order curve of their pixel coordinates and sort the
    triangle ROWS of tri1 so that consecutive rows touch nearby vertices (tri2 follows in
    :func:`attach_tri2`).  For DENSE frames (whose planes live in global memory)
    neighbouring lanes then gather from the same cache lines instead of 64 different ones: 2.6x
    at N=20000.  (For LDS-resident frames it was measured to hurt: neighbouring lanes then collide
    on the same vote-counter word.)  Frames with fewer than ``min_features`` features are left alone.

    The path's results do not depend on either order: the vote is an integer sum over incident
    triangles, the selection is a set, the histogram is order-free (only the floating-point sums
    behind height_level / mean / std are taken in a different order).  What must NOT change is the
    order of the three vertices inside a row (/root/reference/src/scale_calculator.py:113-115):
    rows are relabelled and permuted as a whole.  ``pf.extra['perm'][f][k]`` = original position
    (among the frame's packed features) of the feature now at k; per-feature outputs of the kernels
    come back in the new order, ``pf.lower_index`` is permuted along so that it still maps a packed
    position to the caller's row.
    """
    perms = []
    for f in range(pf.n_frames):
        n = int(pf.feat_cnt[f])
        if n < max(min_features, 1):
            perms.append(None)
            continue
        sl = pf.frame_slice(f)
        perm = np.argsort(_morton_key(pf.u[sl], pf.v[sl]), kind="stable")
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        for name in ("x", "y", "z", "v", "u"):
            plane = getattr(pf, name)
            plane[sl] = plane[sl][perm]
        if pf.lower_index[f] is not None:
            pf.lower_index[f] = np.asarray(pf.lower_index[f])[perm]
        perms.append(perm)
        if pf.tri1_off is not None:
            a, b = int(pf.tri1_off[f]), int(pf.tri1_off[f + 1])
            if b > a:
                t = inv[pf.tri1[a:b]].astype(np.int32)
                pf.tri1[a:b] = t[np.argsort(t.min(axis=1), kind="stable")]
    pf.extra["perm"] = perms
    return pf


TILE_W = 512                    # MVOSR_TILE_W (include/mvosr.h)


def _tile_sort(tri, n, tile_w=TILE_W):
    """Rows (numbered over the frame's packed features) in the order the tiled kernel walks them: rows whose vertices
    all lie in the tile of their smallest vertex or the next one, sorted by that smallest vertex, then the "far" rows.
    Returns ``(rows, offsets)``; ``offsets[k]`` = first row of tile k for k < ntiles, ``offsets[ntiles]`` = first far row."""
    ntiles = (n + tile_w - 1) // tile_w
    if tri.shape[0] == 0:
        return tri, np.zeros(ntiles + 1, dtype=np.int32)
    lo, hi = tri.min(axis=1), tri.max(axis=1)
    far = hi >= (lo // tile_w + 2) * tile_w
    order = np.lexsort((lo, far))                       # near rows first, by smallest vertex (stable)
    rows = np.ascontiguousarray(tri[order])
    n_near = int(np.count_nonzero(~far))
    offs = np.searchsorted(lo[order][:n_near], np.arange(ntiles) * tile_w, side="left")
    _tile_sort.last_order = order                       # stored row i = original row order[i]
    return rows, np.concatenate([offs, [n_near]]).astype(np.int32)


def _finish_tile_index(pf: PackedFrames):
    t1, t2 = pf.extra.get("tile1"), pf.extra.get("tile2")
    if t1 is None or t2 is None or any(t is None for t in t1) or any(t is None for t in t2):
        return
    lens = np.array([len(t) for t in t1], dtype=np.int64)
    pf.tile_base = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pf.tile1_off = np.concatenate(t1).astype(np.int32)
    pf.tile2_off = np.concatenate(t2).astype(np.int32)
    pf.tile_w = TILE_W
    # the far rows' vertices as one contiguous block per frame: (y, z, v) x 3 per far row of tri1, then (x, y, z) x 3
    # per far row of tri2 — plain copies of the planes' values
    blocks, lens = [], []
    for f in range(pf.n_frames):
        sl = pf.frame_slice(f)
        x, y, z, v = pf.x[sl], pf.y[sl], pf.z[sl], pf.v[sl]
        parts = []
        a, b = int(pf.tri1_off[f]), int(pf.tri1_off[f + 1])
        far = pf.tri1[a + int(t1[f][-1]):b]
        if len(far):
            parts.append(np.stack([y[far], z[far], v[far]], axis=2).reshape(-1))
        a, b = int(pf.tri2_off[f]), int(pf.tri2_off[f + 1])
        far = pf.tri2[a + int(t2[f][-1]):b]
        if len(far):
            parts.append(np.stack([x[far], y[far], z[far]], axis=2).reshape(-1))
        blk = np.concatenate(parts) if parts else np.zeros(0)
        blocks.append(blk)
        lens.append(len(blk))
    pf.tile_far = np.ascontiguousarray(np.concatenate(blocks) if blocks else np.zeros(0), dtype=np.float64)
    if pf.tile_far.size == 0:
        pf.tile_far = np.zeros(1)                                  # (a valid device pointer even when no frame has far rows)
    pf.tile_far_off = np.concatenate([[0], np.cumsum(np.array(lens, dtype=np.int64))]).astype(np.int64)


def apply_tile_order(pf: PackedFrames):
    """Lay every frame of a DENSE batch out for the tiled gather kernel: features sorted along the
    image axis of larger extent — a Delaunay triangle's three vertices are then a few dozen positions apart (at
    N = 20000: median 58, 99th percentile 180; along a Z-order curve the 99th percentile is 6700) —, tri1 relabelled
    and its rows sorted by smallest vertex with the few far rows last, and the tile index built (tri2 follows in
    :func:`attach_tri2`).  The path's results do not depend on feature or row order (the vote is an integer sum over
    incident rows, the selection a set, the histogram order-free); the order of the three vertices INSIDE a row, which
    the vote does depend on (/root/reference/src/scale_calculator.py:113-115), is untouched.  ``pf.extra['perm'][f][k]`` =
    original packed position of the feature now at k; ``pf.lower_index`` is permuted along."""
    perms, tiles = [], []
    for f in range(pf.n_frames):
        n = int(pf.feat_cnt[f])
        sl = pf.frame_slice(f)
        if n == 0:
            perms.append(None)
            tiles.append(np.zeros(1, dtype=np.int32))
            continue
        u, v = pf.u[sl], pf.v[sl]
        key = u if (u.max() - u.min()) >= (v.max() - v.min()) else v
        perm = np.argsort(key, kind="stable")
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        for name in ("x", "y", "z", "v", "u"):
            plane = getattr(pf, name)
            plane[sl] = plane[sl][perm]
        if pf.lower_index[f] is not None:
            pf.lower_index[f] = np.asarray(pf.lower_index[f])[perm]
        perms.append(perm)
        offs = np.zeros((n + TILE_W - 1) // TILE_W + 1, dtype=np.int32)
        if pf.tri1_off is not None:
            a, b = int(pf.tri1_off[f]), int(pf.tri1_off[f + 1])
            if b > a:
                rows, offs = _tile_sort(inv[pf.tri1[a:b]].astype(np.int32), n)
                pf.tri1[a:b] = rows
        tiles.append(offs)
    pf.extra["perm"] = perms
    pf.extra["tile1"] = tiles
    pf.extra["tile2"] = None
    return pf


def _relabel_tri2(tri_old, m_old, perm, return_order=False):
    """Second-triangulation rows numbered over the survivors in ORIGINAL order -> over the survivors
    in the permuted order; rows sorted by their smallest vertex (``return_order``: also the original index of every
    stored row)."""
    m_new = m_old[perm]
    c_old = np.cumsum(m_old) - 1
    c_new = np.cumsum(m_new) - 1
    remap = np.empty(int(m_old.sum()), dtype=np.int64)
    keep_new = np.nonzero(m_new)[0]
    remap[c_old[perm[keep_new]]] = c_new[keep_new]
    t = remap[tri_old].astype(np.int32)
    order = np.argsort(t.min(axis=1), kind="stable")
    return (t[order], order) if return_order else t[order]
