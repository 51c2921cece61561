"""Multi-GPU data parallelism over frames.

The raw scale of a frame depends on that frame only (/root/reference/src/scale_calculator.py:411-422
up to the filter), so a sequence shards into contiguous blocks of frames, one block per rank
(one process per GPU).  The only cross-frame coupling — the window median of
scale_calculator.py:396-400 — is a sliding-window function of the raw sequence, so it runs after
ONE all-gather of the per-rank records (RCCL over xGMI through ``torch.distributed``'s ``nccl``
backend; ``gloo`` in CPU tests).  No halo, no all-reduce, no second collective.

The record of a rank is one allocation ``[raw_scale f64 x cap | height_level f64 x cap | status i32 x cap]``
(20 bytes per frame, ``cap`` = largest shard): the kernels write their per-frame outputs straight into
its three views, the all-gather moves it as bytes, and the window-median kernel reads the gathered
buffer in place (``mvosr_window_median_blocked``) — no packing or unpacking kernel on the step.
"""
from __future__ import annotations

import numpy as np


def partition(n_frames: int, world_size: int, rank: int):
    """Contiguous block of rank ``rank``: the first ``n_frames % world_size`` ranks get one extra
    frame.  Returns ``(start, stop)``."""
    base, extra = divmod(n_frames, world_size)
    start = rank * base + min(rank, extra)
    stop = start + base + (1 if rank < extra else 0)
    return start, stop


def shard_sizes(n_frames: int, world_size: int):
    return [partition(n_frames, world_size, r)[1] - partition(n_frames, world_size, r)[0] for r in range(world_size)]


def record_stride_bytes(cap: int) -> int:
    """Bytes of one rank's record, a multiple of 8 so that every block's raw-scale view stays aligned."""
    return (20 * cap + 7) & ~7


class RankRecord:
    """This rank's per-frame outputs in ONE buffer: ``raw`` / ``level`` (float64) and ``status`` (int32)
    are views into it — hand their ``data_ptr()`` to the kernels, or fill them from host arrays."""

    def __init__(self, cap: int, device):
        import torch
        self.cap = int(cap)
        self.buf = torch.zeros(max(record_stride_bytes(self.cap), 8), dtype=torch.uint8, device=device)
        c = self.cap
        self.raw = self.buf[0:8 * c].view(torch.float64)
        self.level = self.buf[8 * c:16 * c].view(torch.float64)
        self.status = self.buf[16 * c:20 * c].view(torch.int32)

    def fill(self, raw, status, level=None):
        """Host arrays of this rank's block -> the record (the tail up to ``cap`` is padding)."""
        import torch
        n = len(raw)
        dev = self.buf.device
        self.raw[:n] = torch.from_numpy(np.ascontiguousarray(raw, dtype=np.float64)).to(dev)
        self.status[:n] = torch.from_numpy(np.ascontiguousarray(status, dtype=np.int32)).to(dev)
        if level is not None:
            self.level[:n] = torch.from_numpy(np.ascontiguousarray(level, dtype=np.float64)).to(dev)
        if n < self.cap:
            self.raw[n:] = float("nan")
            self.level[n:] = float("nan")
            self.status[n:] = -1
        return self


class GatheredFrames:
    """The all-gathered records of every rank, still in gathered (per-rank blocked) form."""

    def __init__(self, recv, n_frames, world, cap):
        self.recv, self.n_frames, self.world, self.cap = recv, int(n_frames), int(world), int(cap)
        self.stride_bytes = record_stride_bytes(cap)
        self.sizes = shard_sizes(self.n_frames, self.world)

    @property
    def stride_doubles(self):
        return self.stride_bytes // 8

    def _field(self, lo, hi, dtype):
        import torch
        parts = []
        for r, s in enumerate(self.sizes):
            blk = self.recv[r * self.stride_bytes:(r + 1) * self.stride_bytes]
            parts.append(blk[lo * self.cap:hi * self.cap].view(dtype)[:s])
        return torch.cat(parts) if parts else torch.empty(0, dtype=dtype, device=self.recv.device)

    # fresh tensors in sequence order (never views of the cached receive buffer)
    def raw(self):
        import torch
        return self._field(0, 8, torch.float64)

    def level(self):
        import torch
        return self._field(8, 16, torch.float64)

    def status(self):
        import torch
        return self._field(16, 20, torch.int32)


_recv = {}
collectives_issued = 0          # counted so that the bench line / tests can state "one per step"


def all_gather_record(rec: RankRecord, n_frames: int, group=None) -> GatheredFrames:
    """THE collective of the path: all-gather of the ranks' records (bytes).  The receive buffer is cached
    between steps (same device / shard size / world): the returned object reads it in place, so take
    ``raw()`` / ``status()`` / ``level()`` copies before the next gather if you need them to persist."""
    global collectives_issued
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    cap = max(shard_sizes(n_frames, world)) if n_frames else 0
    if cap != rec.cap:
        raise ValueError("record capacity %d != largest shard %d" % (rec.cap, cap))
    key = (str(rec.buf.device), rec.buf.numel(), world)
    if key not in _recv:
        _recv[key] = torch.empty(world * rec.buf.numel(), dtype=torch.uint8, device=rec.buf.device)
    recv = _recv[key]
    if rec.buf.is_cuda and dist.get_backend(group) == "gloo":
        # dry runs of the N-rank path with ranks sharing a GPU (MVOSR_SHARE_GPU; RCCL refuses two ranks on one device):
        # gloo moves host memory, so the record is staged through the host
        host = torch.empty(recv.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(host, rec.buf.cpu(), group=group)
        recv.copy_(host)
    else:
        dist.all_gather_into_tensor(recv, rec.buf, group=group)
    collectives_issued += 1
    return GatheredFrames(recv, n_frames, world, cap)


def gather_and_filter(rec: RankRecord, n_frames, window, median_fn, queue=(), group=None):
    """Gather the records of all ranks (one collective) and apply the window median to the whole sequence.

    ``median_fn(gathered, window, queue) -> filtered tensor`` runs the K4 kernel on the gathered buffer
    in place (product: :func:`make_gpu_median`) — injected so that the CPU (gloo) tests can drive the
    same code with the oracle's filter.  Returns ``(filtered, gathered)``."""
    g = all_gather_record(rec, n_frames, group)
    return median_fn(g, window, queue), g


def make_gpu_median(engine):
    """``median_fn`` backed by the window-median kernel on torch CUDA memory (zero-copy: the kernel reads the
    gathered records / a plain tensor and writes a torch tensor on the stream the engine's context has adopted)."""
    import torch

    def fn(seq, window, queue=()):
        if isinstance(seq, GatheredFrames):
            out = torch.empty(seq.n_frames, dtype=torch.float64, device=seq.recv.device)
            engine.window_median_blocked(seq.recv.data_ptr(), seq.n_frames, seq.world, seq.stride_doubles, window, queue,
                                         out.data_ptr())
            return out
        out = torch.empty_like(seq)
        engine.window_median(seq.data_ptr(), seq.numel(), window, queue, out.data_ptr())
        return out
    return fn


def init_distributed(backend=None):
    """Read RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment (torch.distributed.run)."""
    import os
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("MVOSR_FORCE_DIST") == "1") and not dist.is_initialized():   # FORCE_DIST: a group of one, for diagnostics
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = os.environ.get("MVOSR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            # MVOSR_SHARE_GPU=1: dry runs of the N-rank path on a box with fewer GPUs (gloo only;
            # RCCL refuses two ranks on one device)
            if os.environ.get("MVOSR_SHARE_GPU") == "1":
                local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
