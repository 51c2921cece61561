"""Per-sequence driver around the hot path: the caller contract of
/root/reference/src/main_offline.py:24-93 (replay of a saved ``{motions, move_flags,
feature2ds, feature3ds}`` dict) and of the matching loop in /root/reference/src/main.py:74-147.

The reference's drivers never run on the GPU box (they need cv2 / KITTI images), so this module
is the build's own counterpart of the loop *around* ``ScaleEstimator``: frame order, the
"not moving" skip, the ``N > minimum_feature_for_scale`` gate with repeat-previous-scale, the
initial ``scales=[0]`` / ``error=[100]`` entries and the ``scales[1:]`` slice on save.  The
estimator is injected, so the same loop drives the HIP-backed ``ScaleEstimator`` (per frame or
batched) and, in tests, the CPU oracle.
"""
from __future__ import annotations

import numpy as np

MINIMUM_FEATURE_FOR_SCALE = 100     # /root/reference/src/param.py:37
CAMERA_H = 1.75                     # /root/reference/src/param.py:36


def load_sequence_dict(path):
    """/root/reference/src/main_offline.py:26-32: a pickled dict saved with ``np.save``."""
    data = np.load(path, allow_pickle=True)
    return data.item()


def save_sequence_dict(path, data):
    """/root/reference/src/main.py:149-154."""
    with open(path, "wb") as fh:          # np.save on a file object does not append ".npy"
        np.save(fh, data, allow_pickle=True)


def plan_sequence(data, minimum_feature_for_scale=MINIMUM_FEATURE_FOR_SCALE):
    """Classify every frame the way main_offline.py:57-88 does, without touching the estimator.

    Returns an int8 array: 0 = not moving (scale 0, error 0; :64-68), 1 = processed by the
    estimator (:73-83), 2 = too few features, repeat previous scale/error (:84-86).
    """
    flags = data["move_flags"]
    kinds = np.zeros(len(flags), dtype=np.int8)
    for i, mv in enumerate(flags):
        if not mv:
            kinds[i] = 0
        elif np.asarray(data["feature3ds"][i]).shape[0] > minimum_feature_for_scale:
            kinds[i] = 1
        else:
            kinds[i] = 2
    return kinds


def assemble_outputs(kinds, est_scales, est_stds):
    """Merge the estimator's outputs for the processed frames back into the per-frame lists,
    main_offline.py:46-47,64-68,82-86.  Returns (scales, error) INCLUDING the initial entry."""
    scales, error = [0], [100]
    it = 0
    for k in kinds:
        if k == 0:
            scales.append(0)
            error.append(0)
        elif k == 1:
            scales.append(est_scales[it])
            error.append(est_stds[it])
            it += 1
        else:
            scales.append(scales[-1])
            error.append(error[-1])
    return scales, error


def run_sequence(data, estimator, minimum_feature_for_scale=MINIMUM_FEATURE_FOR_SCALE):
    """Frame-at-a-time replay (main_offline.py:57-88).  ``estimator`` exposes the reference
    call surface ``initial_estimation`` / ``scale_calculation``.  Returns a dict with
    ``scales`` (already sliced ``[1:]`` like the save at :90), ``error`` and ``pitchs``."""
    kinds = plan_sequence(data, minimum_feature_for_scale)
    est_scales, est_stds, pitchs = [], [], []
    for i, k in enumerate(kinds):
        if k != 1:
            continue
        motion = np.asarray(data["motions"][i], dtype=np.float64)
        pitchs.append(estimator.initial_estimation(motion[3:12:4].reshape(-1)))
        scale, std = estimator.scale_calculation(np.array(data["feature3ds"][i], dtype=np.float64),
                                                 np.array(data["feature2ds"][i], dtype=np.float64))
        est_scales.append(scale)
        est_stds.append(std)
    scales, error = assemble_outputs(kinds, est_scales, est_stds)
    return {"scales": np.array(scales[1:], dtype=np.float64), "error": np.array(error, dtype=np.float64),
            "pitchs": np.array(pitchs, dtype=np.float64), "kinds": kinds}


def run_sequence_batched(data, estimator, minimum_feature_for_scale=MINIMUM_FEATURE_FOR_SCALE, **batch_kw):
    """Same outputs as :func:`run_sequence`, but all processed frames go through ONE call of the
    estimator's ``scale_calculation_batch`` (frames stay in order, so the window median sees the
    same sequence).  This is the throughput path for ``main_offline``-shaped replays.  ``batch_kw`` goes to that call
    (``id_triples`` of ``rescale.ScaleEstimator``: one entry per PROCESSED frame)."""
    kinds = plan_sequence(data, minimum_feature_for_scale)
    idx = [i for i, k in enumerate(kinds) if k == 1]
    pitchs = [estimator.initial_estimation(np.asarray(data["motions"][i], dtype=np.float64)[3:12:4].reshape(-1))
              for i in idx]
    f3 = [np.asarray(data["feature3ds"][i], dtype=np.float64) for i in idx]
    f2 = [np.asarray(data["feature2ds"][i], dtype=np.float64) for i in idx]
    est_scales, est_stds = estimator.scale_calculation_batch(f3, f2, **batch_kw)
    scales, error = assemble_outputs(kinds, list(est_scales), list(est_stds))
    return {"scales": np.array(scales[1:], dtype=np.float64), "error": np.array(error, dtype=np.float64),
            "pitchs": np.array(pitchs, dtype=np.float64), "kinds": kinds}


# ---- the same replay sharded over the GPUs of a node (SURVEY.md §8e, config C4 in driver form) ------
ST_HOST_QHULL = 100        # codes carried in the gathered status array for frames whose Delaunay call raised
ST_HOST_OTHER = 101


def run_sequence_sharded(data, estimator, group=None, minimum_feature_for_scale=MINIMUM_FEATURE_FOR_SCALE):
    """:func:`run_sequence_batched` with the processed frames split into contiguous blocks, one per rank
    of ``torch.distributed`` (one process per GPU).  Every rank runs the per-frame half on its block
    (``estimator.raw_scale_batch``: host Delaunay on its own CPUs, kernels on its GPU), ONE all-gather
    reassembles the ranks' ``(raw_scale, height_level, status)`` records for the whole sequence, and every rank applies the
    cross-frame half (``estimator.push_raw_scales``: window median, raise sites) to it — so all ranks
    return the same dict, equal to the single-process result.  Without an initialised process group it
    runs as one rank."""
    import torch
    import torch.distributed as dist
    from . import sharding
    kinds = plan_sequence(data, minimum_feature_for_scale)
    idx = [i for i, k in enumerate(kinds) if k == 1]
    pitchs = [estimator.initial_estimation(np.asarray(data["motions"][i], dtype=np.float64)[3:12:4].reshape(-1))
              for i in idx]
    multi = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    start, stop = sharding.partition(len(idx), world, rank)
    mine = idx[start:stop]
    f3 = [np.asarray(data["feature3ds"][i], dtype=np.float64) for i in mine]
    f2 = [np.asarray(data["feature2ds"][i], dtype=np.float64) for i in mine]
    import inspect
    if "frame_base" in inspect.signature(estimator.raw_scale_batch).parameters:      # (rescale.ScaleEstimator: its sample
        raw, status, level, host_errors = estimator.raw_scale_batch(f3, f2, frame_base=start)   # sequence is keyed by the frame's position)
    else:
        raw, status, level, host_errors = estimator.raw_scale_batch(f3, f2)
    status = np.array(status, dtype=np.int32)
    for f, exc in host_errors.items():
        status[f] = ST_HOST_QHULL if type(exc).__name__ == "QhullError" else ST_HOST_OTHER
    if world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        cap = max(sharding.shard_sizes(len(idx), world))
        rec = sharding.RankRecord(cap, dev).fill(raw, status, level)
        g = sharding.all_gather_record(rec, len(idx), group)                  # the ONE collective of the path
        raw, status, level = g.raw().cpu().numpy(), g.status().cpu().numpy(), g.level().cpu().numpy()
    errors = {}
    for f in (int(v) for v in np.nonzero(status >= ST_HOST_QHULL)[0]):
        if start <= f < stop:
            errors[f] = host_errors[f - start]                                # the exception itself
        elif status[f] == ST_HOST_QHULL:
            from scipy.spatial import QhullError
            errors[f] = QhullError("Delaunay failed for processed frame %d (raised on another rank)" % f)
        else:
            errors[f] = RuntimeError("host stage failed for processed frame %d (raised on another rank)" % f)
    est_scales, est_stds = estimator.push_raw_scales(raw, np.where(status >= ST_HOST_QHULL, 0, status), level, errors)
    scales, error = assemble_outputs(kinds, list(est_scales), list(est_stds))
    return {"scales": np.array(scales[1:], dtype=np.float64), "error": np.array(error, dtype=np.float64),
            "pitchs": np.array(pitchs, dtype=np.float64), "kinds": kinds}


def save_outputs(res_addr, tag, scales, motions):
    """/root/reference/src/main_offline.py:90-93: ``<res_addr>scales.txt<tag>`` and ``<res_addr>path.txt<tag>``
    (the integrated trajectory, one 3x4 pose per line).  Returns the poses."""
    np.savetxt(res_addr + 'scales.txt' + tag, scales)
    poses = get_path(np.array(motions, dtype=np.float64), np.asarray(scales, dtype=np.float64))
    np.savetxt(res_addr + 'path.txt' + tag, poses)
    return poses


# ---- pose integration -----------------------------------------------------
def motion2pose(motions):
    """/root/reference/src/main_offline.py:100-111 (= script/transformation.py:10-21): chain the
    relative 3x4 motions into absolute poses, first row identity."""
    motions = np.asarray(motions, dtype=np.float64)
    n = motions.shape[0]
    poses = np.zeros((n + 1, 12))
    pose = np.eye(4)
    poses[0] = pose[:3].reshape(-1)
    step = np.eye(4)
    for i in range(n):
        step[:3, :] = motions[i].reshape(3, 4)
        pose = pose @ step
        poses[i + 1] = pose[:3].reshape(-1)
    return poses


def get_path(motions, scales):
    """/root/reference/src/main_offline.py:114-119: scale each relative translation, integrate.
    (The reference scales ``motions`` in place; this works on a copy.)"""
    m = np.array(motions, dtype=np.float64, copy=True)
    m[:, 3:12:4] = m[:, 3:12:4] * np.asarray(scales, dtype=np.float64)[:, None]
    return motion2pose(m)
