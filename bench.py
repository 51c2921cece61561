#!/usr/bin/env python3
"""Headline benchmark: frames/s of scale recovery on synthetic KITTI-shaped flow.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--features 2000|20000] [--workload c2|kitti]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment: this process starts the N ranks itself (N child
processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before anything here touches a GPU), relays rank 0's
JSON line and exits non-zero if any rank fails.  A box with fewer than N GPUs is refused unless ``--share-gpu``
(dry run of the N-rank path: ranks share the devices round-robin, gloo instead of RCCL; marked in the line).

One "step" = one pass of the hot path over one batch of F frames per GPU, resident in HBM:
mvosr_scale_batch = the scale kernel (feature_remap -> depth-order vote on tri1 -> compaction -> per-triangle plane
normal / pitch / height on tri2 -> height_level -> selected points; one workgroup per frame) + its exact pass over the
(normally empty) list of frames whose result hangs on the last bits of height_level + the road-model kernel
(histogram / modes / skew -> height -> raw scale; one wavefront per frame); then (N > 1) ONE all-gather of the ranks'
(raw_scale, height_level, status) records over RCCL and the window-median kernel reading the gathered buffer in
place.  Workloads (BASELINE.json configs): ``--features 2000`` = configs[1]/[3] (C2/C4: ~4000 triangles per frame),
``--features 20000`` = configs[4] (C5, dense frames: the gather kernel), ``--workload kitti`` = the size
distribution of configs[2] (C3: 300-1500 features per frame, ragged).  Both Delaunay triangulations are precomputed
on the host (they are inputs of the GPU path, like the optical flow itself); `e2e` in the line says what the whole
drop-in call reaches with them included.  Weak scaling: every rank owns F frames.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      algorithmic bytes per launch (8*(3N+N)+12*(T1+T2)+12 per frame) / average duration of
                the dominant kernel, measured with HIP events recorded on its launch stream around that kernel
                (mvosr_ctx_profile), against the 8 TB/s HBM peak; `step_*` gives the same for all kernels of the step;
  cpu_baseline  the CPU oracle (vectorised NumPy port of the reference) on one core of this box, on a bounded sample of
                the same frames, triangulations supplied; `cpu_baseline_reference_shaped` the loop-faithful flavour
                (per-triangle Python loops, as the reference is written); `cpu_baseline_all_cores` the vectorised one
                on every CPU the process may use (N = 1 only);
  e2e           frames/s of ScaleEstimator.scale_calculation_batch on a bounded sample, host Delaunay x2, packing and
                uploads included, with the number of host CPUs it used;
  e2e_gpu_triangulation   the same call with both triangulations built on the device (triangulation="gpu",
                check_triangle="fixed": a declared deviation, bit-equal to the fixed-mode oracle), plus the Delaunay kernel
                alone and the allocation counters of the timed call;
  e2e_gpu_exact   the same call with check_triangle="reference": SciPy/Qhull's own rows built on the device by replaying Qhull's
                insertion order — the reference's result, bit for bit, with nothing on the host but packing;
  e2e_rescale   the estimator the reference's drivers really import (rescale.ScaleEstimator, /root/reference/src/main.py:20),
                device-resident (Delaunay x2, GraphChecker vote, flat_selection + RANSAC plane; slew limiter + window median: a C
                loop on the host) from per-frame arrays; no declared deviation;
  latency       per-frame latency of the drop-in scale_calculation call in the reference's loop shape.
"""
from __future__ import annotations

import os as _os
_os.environ.setdefault("MVOSR_HW_QUEUES", "8"); _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")    # this process owns more than four streams (torch's, two contexts with an upload stream each): ROCm's default of four hardware queues would make them share
_os.environ.setdefault("MVOSR_AFFINITY", "1")     # the end-to-end legs pin this process to the CPUs of the device's NUMA node (opt-in in the library)
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 measured copy
ABS_REF = 1.75                  # param.camera_h
WINDOW = 5                      # main.py:55


# ---------------------------------------------------------------------------------------------------------------
# self-launch (no GPU call may precede this: torch.cuda.device_count() does not initialise the runtime)
# ---------------------------------------------------------------------------------------------------------------
def gpu_count_without_runtime():
    """GPUs of this box WITHOUT touching the HIP runtime (the parent of the self-launched ranks must stay GPU-free): the
    KFD topology's nodes with SIMDs (CPUs are nodes too, with simd_count 0), cut down by a *_VISIBLE_DEVICES list."""
    import glob
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            with open(path) as fh:
                for ln in fh:
                    if ln.startswith("simd_count"):
                        n += 1 if int(ln.split()[1]) > 0 else 0
                        break
        except (OSError, ValueError):
            pass
    return n


def _tail(path, lines=30):
    try:
        with open(path, errors="replace") as fh:
            return "".join(fh.readlines()[-lines:])
    except OSError:
        return ""


def launch_ranks(args, argv):
    """Start the N ranks as child processes (each a fresh interpreter: nothing here has touched a GPU), every rank's
    stdout / stderr in files of a temporary directory; poll them all — a rank that dies before the first collective would
    otherwise leave rank 0 waiting in the all-gather for ever — and on the first failure (or the timeout) stop the rest and
    print the tails of what the ranks wrote."""
    import shutil
    import socket
    import tempfile
    n_dev = gpu_count_without_runtime()
    if n_dev is None:
        import torch
        n_dev = torch.cuda.device_count()          # (does not initialise the runtime on this image)
    if n_dev < args.gpus and not args.share_gpu:
        print("bench.py: --gpus %d but this box has %d GPU(s); refusing to run (use --share-gpu for a dry run of the "
              "%d-rank path on fewer devices)" % (args.gpus, n_dev, args.gpus), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    logdir = tempfile.mkdtemp(prefix="mvosr_bench_")
    procs, files = [], []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if args.share_gpu:
            env["MVOSR_SHARE_GPU"] = "1"
            env["MVOSR_DIST_BACKEND"] = "gloo"
        fo = open(os.path.join(logdir, "rank%d.out" % rank), "w")
        fe = open(os.path.join(logdir, "rank%d.err" % rank), "w")
        files += [fo, fe]
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=fo, stderr=fe))
    deadline = time.time() + args.launch_timeout
    codes = [None] * args.gpus
    failed = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed is None:
                    failed = "rank %d exited with code %s" % (r, codes[r])
        if failed is None and time.time() > deadline:
            failed = "timeout after %d s" % args.launch_timeout
        if failed is not None:
            break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        codes = [p.poll() for p in procs]
    for f in files:
        f.close()
    out0 = ""
    try:
        with open(os.path.join(logdir, "rank0.out")) as fh:
            out0 = fh.read()
    except OSError:
        pass
    if failed is not None or any(codes):
        print("bench.py: %s; rank exit codes %s; logs in %s" % (failed or "a rank failed", codes, logdir), file=sys.stderr)
        for r in range(args.gpus):
            if codes[r] not in (0,):
                print("---- rank %d (exit %s) stderr tail\n%s---- rank %d stdout tail\n%s" % (
                    r, codes[r], _tail(os.path.join(logdir, "rank%d.err" % r)), r, _tail(os.path.join(logdir, "rank%d.out" % r), 10)), file=sys.stderr)
        sys.stdout.write(out0)
        return 1
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if len(lines) != 1:
        print("bench.py: expected one JSON line from rank 0, got %d (logs in %s)" % (len(lines), logdir), file=sys.stderr)
        return 1
    shutil.rmtree(logdir, ignore_errors=True)
    print(lines[0])
    return 0


# ---------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------
TILE_ABOVE = [1 << 30]


SNAP = [0.0, 1.0]     # --workload gridded: pixel coordinates rounded to this grid (0.25 px) in this fraction of the frames (--snap-fraction)


def _synth(synth, i, n, base_seed):
    """A synthetic frame of the workload; ``--workload gridded``: its pixel coordinates snapped to a quarter-pixel grid — collinear and
    cocircular sites by construction (what a detector with sub-pixel refinement to 1/4 px hands over): the regime in which Qhull
    merges facets and the replay DECLINES to the host's SciPy."""
    f3, f2 = synth.synth_frame(i, n, base_seed=base_seed)
    if SNAP[0] > 0.0 and ((i * 2654435761) % (1 << 32)) / float(1 << 32) < SNAP[1]:
        f2 = np.ascontiguousarray(np.round(f2 / SNAP[0]) * SNAP[0])
    return f3, f2


def frame_sizes(args, pool):
    if args.workload == "kitti":
        # configs[2]: per-frame feature counts after the VO's masks, a few hundred to ~1500 (C3)
        rng = np.random.default_rng(4541)
        return [int(v) for v in rng.integers(300, 1501, pool)]
    return [args.features] * pool


def build_pool(ctx, engine, sizes, seed):
    """P unique synthetic frames, both triangulations (SciPy on the host; the vote mask that the
    second triangulation is built on comes from the GPU vote kernel)."""
    from mvoscalerecovery_amd import packing, synth
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs
    pool = len(sizes)
    frames = [_synth(synth, i, sizes[i], base_seed=seed) for i in range(pool)]
    pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames])
    t0 = time.perf_counter()
    packing.attach_tri1(pf, None, None)
    t_del1 = time.perf_counter() - t0
    cap = min(int(ctx.lib.mvosr_max_lds_features()), TILE_ABOVE[0])
    dense = pf.max_feat > cap
    t_tile = 0.0
    if dense:            # dense frames: the tiled layout (what ScaleEstimator.scale_calculation_batch does)
        t0 = time.perf_counter()
        packing.apply_tile_order(pf)
        t_tile = time.perf_counter() - t0
    db = DeviceBatch(ctx, pf, with_tri2=False)
    out = DeviceOutputs(ctx, db, counts=True, stage=True)
    engine.outlier_vote_batch(db, out)
    ctx.sync()
    counters = out.get("vote_counters")
    masks = [counters[pf.frame_slice(f)] >= 0 for f in range(pool)]
    out.free()
    db.free()
    t0 = time.perf_counter()
    packing.attach_tri2(pf, None, masks, None, feature_ids=dense)    # dense: rows numbered over the features (no compaction)
    build_pool.host_layout_ms_per_frame = t_tile * 1e3 / pool        # (sort, relabel, tile index of the dense layout: one host thread)
    t_del2 = time.perf_counter() - t0
    workers = max(1, packing.resolve_workers(None))
    return frames, pf, masks, (t_del1 + t_del2) * workers / pool       # CPU-seconds of Delaunay per frame


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1).  The sample of the all-cores leg is built and its workers are forked BEFORE the GPU
# runtime starts (a fork with live GPU runtime threads can deadlock the children).
# ---------------------------------------------------------------------------------------------------------------
_ALL_CORES_JOBS = None          # inherited by the forked workers


def _all_cores_job(i):
    from oracle import scale_oracle as so
    f3, f2, t1, t2 = _ALL_CORES_JOBS[i % len(_ALL_CORES_JOBS)]
    return so.frame_raw_scale(f3.copy(), f2, ABS_REF, t1, t2, keep=False).raw_scale


def prepare_cpu_legs(sizes, seed, sample=24):
    """CPU-only sample of the workload (triangulations by the oracle itself) and the forked all-cores pool."""
    global _ALL_CORES_JOBS
    import multiprocessing as mp
    from mvoscalerecovery_amd import packing, synth
    from oracle import scale_oracle as so
    jobs = []
    for i in range(min(sample, len(sizes))):
        f3, f2 = _synth(synth, i, sizes[i], base_seed=seed)
        r = so.frame_raw_scale(f3, f2, ABS_REF)
        jobs.append((f3, f2, np.ascontiguousarray(r.tri1, dtype=np.int32), np.ascontiguousarray(r.tri2, dtype=np.int32), r.raw_scale, r.status))
    _ALL_CORES_JOBS = [j[:4] for j in jobs]
    workers = packing.available_cpus()
    pool = mp.get_context("fork").Pool(workers)
    pool.map(_all_cores_job, range(workers * 4))                           # imports + first touches
    return jobs, pool, workers


def cpu_single_core(jobs, budget_s, loops=False):
    from oracle import scale_oracle as so
    from oracle import scale_oracle_loops as sl
    done, t_used, mism = 0, 0.0, 0
    while t_used < budget_s:
        f3, f2, t1, t2, want_raw, want_st = jobs[done % len(jobs)]
        t0 = time.perf_counter()
        if loops:
            raw, st = sl.frame_raw_scale(f3, f2, ABS_REF, t1, t2)[:2]
        else:
            r = so.frame_raw_scale(f3, f2, ABS_REF, t1, t2, keep=False)
            raw, st = r.raw_scale, r.status
        t_used += time.perf_counter() - t0
        mism += 0 if (st == want_st and ((np.isnan(raw) and np.isnan(want_raw)) or raw == want_raw)) else 1
        done += 1
    return done / t_used, done, mism


def cpu_all_cores(pool, workers, per_worker_s, single_core_fps):
    n = max(workers * 50, int(workers * per_worker_s * single_core_fps))
    t0 = time.perf_counter()
    pool.map(_all_cores_job, range(n), chunksize=25)
    dt = time.perf_counter() - t0
    pool.close()
    return n / dt, n


def e2e_leg(args, device, sizes, seed, budget_frames):
    """The drop-in batch call end to end (host Delaunay x2 on the worker pool, packing, uploads, kernels)."""
    from mvoscalerecovery_amd import packing, synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    n = min(budget_frames, 4096)
    frames = [_synth(synth, 100000 + i, sizes[i % len(sizes)], base_seed=seed) for i in range(n)]
    est = ScaleEstimator(ABS_REF, window_size=WINDOW, device=device, mutate_inputs=False, triangulation="scipy")
    nw = min(64, max(8, n // 8))
    est.scale_calculation_batch([f[0] for f in frames[:nw]], [f[1] for f in frames[:nw]])         # warm-up (pool, workspaces)
    t0 = time.perf_counter()
    est.scale_calculation_batch([f[0] for f in frames], [f[1] for f in frames])
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "frames": n, "host_cpus": packing.resolve_workers(None),
            "what": "ScaleEstimator.scale_calculation_batch: vanishing-row filter, packing, SciPy Delaunay x2 on the host "
                    "process pool (overlapped with the GPU stages chunk by chunk), uploads, kernels, window median"}


_E2E_FRAMES = {}


def _e2e_frames(sizes, seed, n_frames):
    """The end-to-end legs' input: a list of n_frames per-frame arrays drawn from a pool of DISTINCT synthetic frames larger
    than the host's last-level cache (4096 frames of 2000 features = 330 MB), so that the packer streams its input from
    DRAM as it would for a real sequence; the pool is repeated to n_frames (its size is reported as distinct_frames)."""
    from mvoscalerecovery_amd import synth
    key = (tuple(sizes[:8]), len(sizes), seed, n_frames)
    if key not in _E2E_FRAMES:
        npool = min(n_frames, 4096 if max(sizes) <= 6000 else 16)
        pool = [_synth(synth, 200000 + i, sizes[i % len(sizes)], base_seed=seed) for i in range(npool)]
        _E2E_FRAMES.clear()
        _E2E_FRAMES[key] = (pool, [pool[i % npool][0] for i in range(n_frames)], [pool[i % npool][1] for i in range(n_frames)])
    return _E2E_FRAMES[key]


def e2e_rescale_leg(args, device, sizes, seed, n_frames):
    """The estimator the reference's drivers really import (/root/reference/src/main.py:20: rescale.ScaleEstimator), end to
    end from per-frame arrays, device-resident: C packer -> one upload per chunk -> Delaunay #1 -> GraphChecker vote ->
    Delaunay #2 -> flat_selection + RANSAC plane (one kernel) -> slew limiter + window median.  No declared deviation:
    the vote and the kept triangle set do not depend on the row form, and the reference's RANSAC is unseeded."""
    from mvoscalerecovery_amd.rescale import ScaleEstimator
    pool, f3s, f2s = _e2e_frames(sizes, seed, n_frames)
    est = ScaleEstimator(ABS_REF, window_size=WINDOW, device=device, triangulation="gpu", delaunay_workers=0, ransac_seed=2024)
    for _ in range(2):
        est.scale_calculation_batch(f3s, f2s)
    ctx = est.ctx
    a0 = ctx.alloc_stats()
    times = []
    for _ in range(3):                                                        # (the median of three timed calls: one call is ~90 ms of host + GPU pipeline)
        t0 = time.perf_counter()
        scales, _ = est.scale_calculation_batch(f3s, f2s)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    a1 = ctx.alloc_stats()
    st = est.last["status"]
    return {"value": n_frames / dt, "unit": "frames/s", "frames": n_frames, "distinct_frames": len(pool),
            "timed_calls_frames_per_s": [n_frames / t for t in times],
            "declined": int(est.last_declined), "frames_with_plane": int((st == 0).sum()),
            "scale_median": float(np.median(scales)),
            "hip_malloc_calls_in_timed_call": a1["hip_malloc"] - a0["hip_malloc"],
            "hip_host_malloc_calls_in_timed_call": a1["host_malloc"] - a0["host_malloc"],
            "what": "rescale.ScaleEstimator(triangulation='gpu').scale_calculation_batch on a list of per-frame arrays: the "
                    "estimator /root/reference/src/main.py:20 imports, every stage on the device (Delaunay x2, GraphChecker vote, "
                    "flat_selection + 100-hypothesis RANSAC plane on the device; slew limiter + window median in C on the host); deterministic stages "
                    "without a deviation; the sampler spends an iteration on a triple naming one vertex twice, as the reference does (zero inliers)"}


def _selfcheck_record(est):
    """What guards the reference-exact device path on this box: the installed SciPy's version and the first-use comparison of the Qhull
    replay (device kernel and host C form) with it (mvoscalerecovery_amd/selfcheck.py)."""
    sc = getattr(est, "qhull_selfcheck", None)
    if not sc:
        return None
    rec = {"ok": bool(sc.get("ok")), "skipped": bool(sc.get("skipped")), "scipy": sc.get("scipy"), "numpy": sc.get("numpy"),
           "replayed": sc.get("replayed"), "runs": est.triangulation}
    for side in ("device", "host"):
        if side in sc:
            rec[side] = {k: sc[side][k] for k in ("sets", "compared", "declined")}
            rec[side]["different"] = len(sc[side]["different"])
    return rec


def e2e_gpu_leg(args, device, sizes, seed, n_frames, exact=False):
    """The batch call end to end with BOTH TRIANGULATIONS BUILT ON THE DEVICE (triangulation="gpu", which selects
    check_triangle="fixed": DESIGN.md §3.5): C packer -> one upload per chunk -> Delaunay #1 -> vote -> Delaunay #2 ->
    scale kernel -> road model -> results; rows, masks and counts never leave HBM.  Also: the device triangulation alone
    (mvosr_delaunay_batch on resident point sets), and the alloc counters of a steady-state call."""
    from mvoscalerecovery_amd import _lib, packing, synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    pool, f3s, f2s = _e2e_frames(sizes, seed, n_frames)
    npool = len(pool)
    # (the exact leg keeps the Delaunay worker pool — forked in main() before the GPU runtime started —: frames the replay declines are
    # triangulated there, under the GPU's queued chunks)
    est = ScaleEstimator(ABS_REF, window_size=WINDOW, device=device, mutate_inputs=False, triangulation="gpu",
                         delaunay_workers=None if (exact or SNAP[0] > 0.0) else 0, check_triangle="reference" if exact else "fixed")
    for _ in range(2):                                                        # warm-up: kernels, allocator caches (same sizes as the timed call)
        est.scale_calculation_batch(f3s, f2s)
    ctx = est.engine.ctx
    a0 = ctx.alloc_stats()
    times = []
    for _ in range(3):                                                        # (the median of three timed calls)
        t0 = time.perf_counter()
        est.scale_calculation_batch(f3s, f2s)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    a1 = ctx.alloc_stats()
    # the triangulation kernel alone on resident point sets of the workload's size
    n = int(max(sizes))
    F = 4096 if n <= 6000 else 256
    cnt = np.full(F, n, dtype=np.int32)
    off = np.arange(F, dtype=np.int64) * n
    uv = np.concatenate([pool[i % min(64, npool)][1][:n] for i in range(F)])
    if uv.shape[0] == F * n:
        d_u, d_v = ctx.to_device(np.ascontiguousarray(uv[:, 0])), ctx.to_device(np.ascontiguousarray(uv[:, 1]))
        d_off, d_cnt, d_toff = ctx.to_device(off), ctx.to_device(cnt), ctx.to_device(2 * off)
        d_tri = ctx.empty((2 * F * n, 3), np.int32)
        d_tcnt, d_st = ctx.zeros(F, np.int32), ctx.zeros(F, np.int32)
        if exact:
            launch = lambda: _lib.check(ctx.lib.mvosr_delaunay_qhull_batch(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, n, d_toff.ptr,
                                                                           d_tri.ptr, d_tcnt.ptr, None, d_st.ptr, None), "mvosr_delaunay_qhull_batch")
        else:
            launch = lambda: _lib.check(ctx.lib.mvosr_delaunay_batch(ctx.handle, F, d_off.ptr, d_cnt.ptr, d_u.ptr, d_v.ptr, None, n, d_toff.ptr,
                                                                     d_tri.ptr, d_tcnt.ptr, None, d_st.ptr), "mvosr_delaunay_batch")
        launch(); ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(3):
            launch()
        ctx.record(e1)
        dt_ms = ctx.elapsed_ms(e0, e1) / 3
        dt_alone = {"sets_per_s": F / dt_ms * 1e3, "points_per_set": n, "sets": F, "kernel_ms": dt_ms,
                    "declined": int((d_st.download() != 0).sum())}
        for b in (d_u, d_v, d_off, d_cnt, d_toff, d_tri, d_tcnt, d_st):
            b.free()
    else:
        dt_alone = None
    roof = None
    if exact and dt_alone is not None and n == 2000:
        # The default path's dominant kernel priced like the headline's (VERDICT r5 #2a).  Its roof is VECTOR INSTRUCTION ISSUE, not HBM
        # (a chain of dependent insertions; HBM at ~9 % of the peak): achieved = the VALU wave-instructions one set costs (rocprofv3
        # SQ_INSTS_VALU of this kernel on these sizes: profiles/qhull_counters.json, re-collected per round) x the sets/s measured HERE
        # with HIP events; peak = 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per 64-lane instruction.
        try:
            qc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "qhull_counters.json")))
            peak = 256 * 4 * 2.4e9 / 4.0
            ach = qc["valu_wave_instructions_per_set"] * dt_alone["sets_per_s"]
            roof = {"bound": "valu_issue", "achieved": ach, "peak": peak, "unit": "wave-instructions/s", "frac": ach / peak,
                    "kernel": "qhull_rows_kernel<unsigned short, 64, 64, false>", "kernel_ms": dt_alone["kernel_ms"], "sets_per_launch": F,
                    "valu_per_insertion": qc["valu_wave_instructions_per_set"] / qc["insertions_per_set"],
                    "salu_per_insertion": qc["salu_instructions_per_set"] / qc["insertions_per_set"],
                    "hbm_bytes_per_set": qc["fetch_bytes_per_set"] + qc["write_bytes_per_set"],
                    "hbm_frac_of_8TBps": (qc["fetch_bytes_per_set"] + qc["write_bytes_per_set"]) * dt_alone["sets_per_s"] / 8e12,
                    "counters_from": "profiles/qhull_counters.json (%s)" % qc.get("tag")}
        except Exception as exc:                                        # noqa: BLE001
            roof = {"error": "%s: %s" % (type(exc).__name__, exc)}
    return {"value": n_frames / dt, "unit": "frames/s", "frames": n_frames, "distinct_frames": npool,
            "timed_calls_frames_per_s": [n_frames / t for t in times], "roofline": roof,
            "declined_total": int(est.declined_total), "declined_fraction": float(est.declined_total) / max(n_frames, 1),
            "qhull_selfcheck": _selfcheck_record(est), "delaunay_kernel": dt_alone,
            "hip_malloc_calls_in_timed_call": a1["hip_malloc"] - a0["hip_malloc"], "hip_host_malloc_calls_in_timed_call": a1["host_malloc"] - a0["host_malloc"],
            "what": ("ScaleEstimator(triangulation='gpu', check_triangle='reference').scale_calculation_batch on a list of per-frame arrays: "
                     "the C packer, one upload per chunk, Qhull's rows #1 (insertion order replayed on the device) / the reference's vote / "
                     "second triangulation as a stand-in (canonical rows) with Qhull's own rows for the frames of the exact pass / scale "
                     "kernel / road model on the device, window median; NO declared deviation: bit-equal to the reference "
                     "(tests/golden/seq4541.npz through this path)") if exact else
                    ("ScaleEstimator(triangulation='gpu').scale_calculation_batch on a list of per-frame arrays: vanishing-row filter + "
                     "packing by the C packer into page-locked memory, one upload per chunk, Delaunay #1 / vote / Delaunay #2 / scale "
                     "kernel / road model on the device, window median; a declared deviation from the reference (check_triangle='fixed'), "
                     "bit-equal to the fixed-mode oracle")}


def e2e_sharded_leg(args, device, rank, world, sizes, seed, per_rank):
    """offline.run_sequence_sharded end to end on `world` ranks (VERDICT r4 #4/#6a): a main_offline-shaped sequence of
    per_rank * world frames given as per-frame arrays; every rank packs, uploads and computes its contiguous block, ONE
    all-gather reassembles the (raw scale, level, status) records, every rank applies the cross-frame half.  Timed between
    barriers, the slowest rank counts.  All three estimators."""
    import torch
    import torch.distributed as dist
    from mvoscalerecovery_amd import offline, synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    from mvoscalerecovery_amd.rescale import ScaleEstimator as RescaleEstimator
    multi = dist.is_available() and dist.is_initialized()
    out = {}
    pool = [_synth(synth, 400000 + i, sizes[i % len(sizes)], base_seed=seed) for i in range(min(1024, per_rank))]
    motion = np.array([1, 0, 0, 0.01, 0, 1, 0, -0.02, 0, 0, 1, 0.9997], dtype=np.float64)
    for name, make, frames_rank in (
            ("scale_fixed", lambda: ScaleEstimator(ABS_REF, window_size=WINDOW, device=device, mutate_inputs=False, triangulation="gpu", delaunay_workers=0), per_rank),
            ("scale_exact", lambda: ScaleEstimator(ABS_REF, window_size=WINDOW, device=device, mutate_inputs=False, triangulation="gpu",
                                                   check_triangle="reference", delaunay_workers=0), max(per_rank // 2, 1)),
            ("rescale", lambda: RescaleEstimator(ABS_REF, window_size=WINDOW, device=device, delaunay_workers=0, triangulation="gpu", ransac_seed=2024), per_rank)):
        total = frames_rank * world
        data = {"motions": [motion] * total, "move_flags": [True] * total,
                "feature3ds": [pool[i % len(pool)][0] for i in range(total)], "feature2ds": [pool[i % len(pool)][1] for i in range(total)]}
        est = make()
        offline.run_sequence_sharded(data, est)                        # warm-up: kernels, allocator caches, RCCL
        times = []
        for _ in range(2):
            est = make()
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = offline.run_sequence_sharded(data, est)
            torch.cuda.synchronize()
            if multi:
                dist.barrier()
            dt = time.perf_counter() - t0
            if multi:
                t = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", device) if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            times.append(dt)
        out[name] = {"value": total / min(times), "unit": "frames/s", "frames_total": total, "frames_per_rank": frames_rank,
                     "scales_finite": int(np.isfinite(res["scales"]).sum())}
    out["what"] = ("offline.run_sequence_sharded over per-frame arrays on %d rank(s): contiguous blocks of frames per rank, pack + upload + "
                   "device triangulations + kernels on each rank's GPU, one all-gather of the 20-byte records, window median on every rank; "
                   "scale_fixed = check_triangle='fixed' (declared deviation), scale_exact = the reference's result (Qhull's rows on the device), "
                   "rescale = the estimator main.py imports" % world)
    return out


def latency_leg(args, device, sizes, seed, frames=100):
    """Per-frame latency of the drop-in call in the reference's loop shape (/root/reference/src/main.py:110-113): one
    scale_calculation per frame — with SciPy's triangulations (the default, bit-exact path) and with the device's."""
    from mvoscalerecovery_amd import synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    fr = [_synth(synth, 300000 + i, sizes[i % len(sizes)], base_seed=seed) for i in range(frames)]
    out = {}
    for name, kw in (("scipy", {"triangulation": "scipy"}), ("gpu", {"triangulation": "gpu"}), ("gpu_exact", {"triangulation": "gpu", "check_triangle": "reference"})):
        est = ScaleEstimator(ABS_REF, window_size=WINDOW, device=device, delaunay_workers=0, **kw)
        for f3, f2 in fr[:5]:
            est.scale_calculation(f3.copy(), f2)
        a0 = est.engine.ctx.alloc_stats()
        t = []
        for f3, f2 in fr:
            a = f3.copy()
            t0 = time.perf_counter()
            est.scale_calculation(a, f2)
            t.append(time.perf_counter() - t0)
        a1 = est.engine.ctx.alloc_stats()
        t = np.array(t) * 1e3
        out[name] = {"median_ms": float(np.median(t)), "p90_ms": float(np.percentile(t, 90)),
                     "hip_malloc_calls": a1["hip_malloc"] - a0["hip_malloc"], "hip_host_malloc_calls": a1["host_malloc"] - a0["host_malloc"]}
        if name == "gpu_exact":       # (frames the one-SciPy-call path handed back to the host's path; exact levels it computed on demand)
            from mvoscalerecovery_amd import packing as _pk
            out[name]["qhull_selfcheck"] = _selfcheck_record(est)
            out[name]["host_replay"] = bool(getattr(est, "_host_replay", False))
            out[name]["host_replay_declined_to_scipy"] = int(getattr(_pk.delaunay_simplices_fast, "declined", 0))
            out[name]["redone_on_host"] = int(getattr(est, "single_fast_redone", 0))
            out[name]["levels_on_demand"] = int(getattr(est, "single_fast_levels", 0))
    # the estimator the reference's drivers import, device-resident (one frame per call: /root/reference/src/main.py:113)
    from mvoscalerecovery_amd.rescale import ScaleEstimator as RescaleEstimator
    est = RescaleEstimator(ABS_REF, window_size=WINDOW, device=device, delaunay_workers=0, triangulation="gpu", ransac_seed=2024)
    for f3, f2 in fr[:5]:
        est.scale_calculation(f3, f2)
    a0 = est.ctx.alloc_stats()
    t = []
    for f3, f2 in fr:
        t0 = time.perf_counter()
        est.scale_calculation(f3, f2)
        t.append(time.perf_counter() - t0)
    a1 = est.ctx.alloc_stats()
    t = np.array(t) * 1e3
    out["rescale_gpu"] = {"median_ms": float(np.median(t)), "p90_ms": float(np.percentile(t, 90)),
                          "hip_malloc_calls": a1["hip_malloc"] - a0["hip_malloc"], "hip_host_malloc_calls": a1["host_malloc"] - a0["host_malloc"]}
    out["what"] = "ScaleEstimator.scale_calculation per frame (stage outputs, flat_feature), %d frames after 5 warm-up calls" % frames
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=0, help="frames per step per GPU (0: 65536, or 3072 for dense frames)")
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--workload", choices=("c2", "kitti", "gridded"), default="c2",
                    help="c2: every frame has --features features (configs[1]); kitti: 300-1500 per frame (configs[2]'s sizes); "
                         "gridded: c2's frames with pixel coordinates rounded to 1/4 px (sites in degenerate position: the decline path)")
    ap.add_argument("--snap-grid", type=float, default=0.25,
                    help="--workload gridded: the grid the pixel coordinates are rounded to, in pixels (0.25; 0.0625 = a detector refining to 1/16 px)")
    ap.add_argument("--snap-fraction", type=float, default=1.0,
                    help="--workload gridded: the share of the frames whose coordinates are snapped (1.0: all; 0.005: the decline rate "
                         "ordinary data showed before round 5's restatements — the cost of a FEW declined frames per chunk)")
    ap.add_argument("--pool", type=int, default=0, help="unique synthetic frames tiled to --frames (0: 1024, or 32 for dense frames)")
    ap.add_argument("--waves", type=int, default=0, help="wavefronts per frame (0 = auto)")
    ap.add_argument("--share-gpu", action="store_true", help="dry run: more ranks than GPUs (gloo, devices shared round-robin)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile-above", type=int, default=0,
                    help="diagnostic: frames with more features than this take the tiled layout / kernel (0: only frames beyond the LDS capacity)")
    ap.add_argument("--no-tiles", action="store_true", help="diagnostic: dense frames without the tile index (the two-sweep gather kernel)")
    ap.add_argument("--no-far-table", action="store_true", help="diagnostic: dense frames without the far rows' vertex table (the kernel gathers)")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--e2e-sharded", action="store_true", help="also at N = 1: the sharded end-to-end leg (runs by itself for N > 1)")
    ap.add_argument("--c4", action="store_true",
                    help="BASELINE configs[3] literally: --total-frames (default 1 000 000) frames SPLIT over the ranks in contiguous, "
                         "possibly ragged blocks (strong scaling) instead of --frames per rank (weak scaling)")
    ap.add_argument("--total-frames", type=int, default=0, help="with --c4: frames of the whole job (0: 1 000 000)")
    ap.add_argument("--launch-timeout", type=int, default=3600, help="seconds the self-launched ranks may take")
    ap.add_argument("--alias-pool", action="store_true",
                    help="diagnostic: every tile reads the feature planes of the SAME pool frames (cache-resident: 41 %% "
                         "less HBM traffic); the JSON line is marked and is not a benchmark result")
    args = ap.parse_args()
    if args.tile_above > 0:
        TILE_ABOVE[0] = args.tile_above
    if args.workload == "gridded":
        SNAP[0], SNAP[1] = float(args.snap_grid), float(args.snap_fraction)

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        print("bench.py: WORLD_SIZE=%s but --gpus %d" % (env_world, args.gpus), file=sys.stderr)
        sys.exit(2)

    dense_cfg = args.features > 6000 and args.workload in ("c2", "gridded")
    pool_n = args.pool or (32 if dense_cfg else 1024)
    frames_req = args.frames or (3072 if dense_cfg else 65536)
    pool_n = min(pool_n, frames_req)
    sizes = frame_sizes(args, pool_n)
    rank_env = int(os.environ.get("RANK", "0"))
    want_cpu = args.gpus == 1 and rank_env == 0 and not args.no_cpu_baseline

    # ---- everything that forks, before the GPU runtime exists -------------------------------------------------
    from mvoscalerecovery_amd import packing
    cpu_jobs = cpu_pool = None
    if want_cpu:
        cpu_jobs, cpu_pool, cpu_workers = prepare_cpu_legs(sizes, 2024, sample=8 if dense_cfg else 24)
    packing.start_pool(None)

    import torch
    import torch.distributed as dist
    from mvoscalerecovery_amd import _lib, sharding
    from mvoscalerecovery_amd.engine import ScaleEngine

    rank, local, world = sharding.init_distributed()
    n_gpus = world
    if os.environ.get("MVOSR_BENCH_FAIL_RANK") == str(rank) and world > 1:      # (test hook of the launcher's failure handling)
        raise RuntimeError("injected failure on rank %d" % rank)
    if os.environ.get("MVOSR_SHARE_GPU") == "1":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    ctx = _lib.Context(local)                       # raises if libmvosr.so / the GPU is missing
    stream = torch.cuda.Stream(device=local)        # a real (non-null) stream shared by torch/RCCL and the kernels
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    engine = ScaleEngine(ABS_REF, ctx=ctx)

    c4_total = (args.total_frames or 1000000) if args.c4 else 0
    if args.c4:
        a_, b_ = sharding.partition(c4_total, world, rank)
        frames_req = b_ - a_                                       # this rank's block of the job (ragged: the first total % world ranks hold one more)
        repeats = max(1, -(-frames_req // pool_n))
        F = frames_req
    else:
        repeats = max(1, frames_req // pool_n)
        F = pool_n * repeats
    frames, pf_pool, masks, delaunay_cpu_s = build_pool(ctx, engine, sizes, seed=2024)
    dense = pf_pool.max_feat > min(int(ctx.lib.mvosr_max_lds_features()), TILE_ABOVE[0])
    # The pool is uploaded once and replicated in HBM on the device (torch.repeat): F frames at distinct
    # addresses (F * 156 KB >> the 256 MB Infinity Cache) without building them on the host.
    dev = torch.device("cuda", local)
    pool_pad = pf_pool.total_padded
    t1p, t2p = int(pf_pool.tri1_off[-1]), int(pf_pool.tri2_off[-1])
    keep = {}                                           # tensors that own the batch's device memory

    def rep(name, arr, dtype):
        keep[name] = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)).to(dev).repeat(repeats)
        return keep[name].data_ptr()

    def offs(name, per_pool, stride, closing):
        if args.alias_pool and not closing:
            o = np.concatenate([per_pool for r in range(repeats)])      # feature planes only (tri*_off are CSR)
        else:
            o = np.concatenate([per_pool + r * stride for r in range(repeats)] + ([np.array([repeats * stride], np.int64)] if closing else []))
        keep[name] = torch.from_numpy(o.astype(np.int64)).to(dev)
        return keep[name].data_ptr()

    bstruct = _lib.Batch(F, offs("feat_off", pf_pool.feat_off, pool_pad, False), rep("feat_cnt", pf_pool.feat_cnt, np.int32),
                         rep("x", pf_pool.x, np.float64), rep("y", pf_pool.y, np.float64), rep("z", pf_pool.z, np.float64),
                         rep("v", pf_pool.v, np.float64),
                         offs("tri1_off", pf_pool.tri1_off[:-1], t1p, True), rep("tri1", pf_pool.tri1[:t1p].reshape(-1), np.int32),
                         offs("tri2_off", pf_pool.tri2_off[:-1], t2p, True), rep("tri2", pf_pool.tri2[:t2p].reshape(-1), np.int32),
                         rep("n2", pf_pool.n2_expected, np.int32), pf_pool.max_feat, int(pf_pool.tri2_ids), pool_pad * repeats)
    cnt_host = np.tile(np.ascontiguousarray(pf_pool.feat_cnt, dtype=np.int32), repeats)[:F]
    if args.c4:
        # the job's frame g is pool frame g % pool_n, whichever rank holds it: this rank's frame k is the job's frame
        # a_ + k.  Per-frame metadata is laid out accordingly (explicit row counts: the offsets are no CSR array then)
        if dense:
            print("bench.py: --c4 is configs[3] (LDS-resident frames); dense frames run with --features 20000 --gpus N", file=sys.stderr)
            sys.exit(2)
        k_ = np.arange(repeats * pool_n, dtype=np.int64)
        src, copy = (k_ + a_) % pool_n, k_ // pool_n

        def up(name, arr, dtype):
            keep[name] = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype)).to(dev)
            return keep[name].data_ptr()
        t1o, t2o = np.asarray(pf_pool.tri1_off, dtype=np.int64), np.asarray(pf_pool.tri2_off, dtype=np.int64)
        bstruct.feat_off = up("c4_feat_off", np.asarray(pf_pool.feat_off, dtype=np.int64)[src] + copy * pool_pad, np.int64)
        bstruct.feat_cnt = up("c4_feat_cnt", np.asarray(pf_pool.feat_cnt)[src], np.int32)
        bstruct.tri1_off = up("c4_tri1_off", t1o[src] + copy * t1p, np.int64)
        bstruct.tri2_off = up("c4_tri2_off", t2o[src] + copy * t2p, np.int64)
        bstruct.tri1_cnt = up("c4_tri1_cnt", (t1o[1:] - t1o[:-1])[src], np.int32)
        bstruct.tri2_cnt = up("c4_tri2_cnt", (t2o[1:] - t2o[:-1])[src], np.int32)
        bstruct.n2_expected = up("c4_n2", np.asarray(pf_pool.n2_expected)[src], np.int32)
        cnt_host = np.ascontiguousarray(np.asarray(pf_pool.feat_cnt, dtype=np.int32)[src][:F])
    _lib.check(ctx.lib.mvosr_batch_size_hint(cnt_host.ctypes.data, F, C.byref(bstruct)), "mvosr_batch_size_hint")
    if pf_pool.tile_w and not args.no_tiles:
        nt = int(pf_pool.tile_base[-1])
        bstruct.tile_w = int(pf_pool.tile_w)
        bstruct.tile_base = offs("tile_base", pf_pool.tile_base[:-1], nt, True)
        bstruct.tile1_off = rep("tile1_off", pf_pool.tile1_off, np.int32)
        bstruct.tile2_off = rep("tile2_off", pf_pool.tile2_off, np.int32)
        if pf_pool.tile_far is not None and not args.no_far_table:
            nf = int(pf_pool.tile_far_off[-1])
            bstruct.tile_far = rep("tile_far", pf_pool.tile_far[:max(nf, 1)], np.float64)
            bstruct.tile_far_off = offs("tile_far_off", pf_pool.tile_far_off[:-1], nf, True)
    bytes_per_launch = pf_pool.algorithmic_bytes() * repeats if not args.c4 else int(pf_pool.algorithmic_bytes() * (F / pool_n))
    n_mean = float(pf_pool.feat_cnt.mean())
    t1_mean = float(pf_pool.tri1_off[-1]) / pool_n
    t2_mean = float(pf_pool.tri2_off[-1]) / pool_n

    # the kernels write this rank's outputs straight into its record; the N-rank step all-gathers the record
    total_frames = c4_total if args.c4 else F * n_gpus
    rec = sharding.RankRecord(max(sharding.shard_sizes(total_frames, n_gpus)) if args.c4 else F, dev)
    height = torch.empty(F, dtype=torch.float64, device=dev)
    outs = _lib.Outputs(rec.raw.data_ptr(), height.data_ptr(), rec.level.data_ptr(), rec.status.data_ptr(),
                        None, None, None, None, None, None, None, None)
    _lib.check(ctx.lib.mvosr_ctx_reserve(ctx.handle, F, pool_pad * repeats), "mvosr_ctx_reserve")    # no hipMalloc inside the timed steps
    median = sharding.make_gpu_median(engine)
    force_gather = bool(os.environ.get("MVOSR_BENCH_FORCE_GATHER")) and dist.is_initialized()   # diagnostic: the N>1 step on one rank
    gathered = n_gpus > 1 or force_gather

    last_gather = [None]

    def step(ev_pair=None):
        if ev_pair is not None:
            ctx.record(ev_pair[0])
        _lib.check(ctx.lib.mvosr_scale_batch(ctx.handle, C.byref(engine.params), C.byref(bstruct), C.byref(outs),
                                             args.waves, 0, 0), "mvosr_scale_batch")
        if ev_pair is not None:
            ctx.record(ev_pair[1])
        if gathered:
            filtered, last_gather[0] = sharding.gather_and_filter(rec, total_frames, WINDOW, median)
        else:
            filtered = median(rec.raw, WINDOW)
        return filtered

    def barrier():
        if n_gpus > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.profile(True)                    # HIP events around each of the step's kernels, on the launch stream
    events = [(ctx.event(), ctx.event()) for _ in range(args.steps)]
    c0 = sharding.collectives_issued
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        filtered = step(events[k])
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    collectives_per_step = (sharding.collectives_issued - c0) / max(args.steps, 1)
    if n_gpus > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms_avg = float(np.mean([ctx.elapsed_ms(a, b) for a, b in events]))       # all kernels of the step
    prof = [ctx.profile_read(k) for k in range(min(args.steps, 64))]
    kernel_ms_avg = float(np.mean([p[0] for p in prof]))                               # scale kernel (+ its exact pass)
    road_ms_avg = float(np.mean([p[1] for p in prof]))                                 # road_model_kernel
    ctx.profile(False)

    # ---- probe: the same K steps alternating between TWO streams (two contexts: a workspace and a record each),
    # so that the road model of step k — serial behind its scale kernel on one stream — can run under the scale kernel of step
    # k+1.  Reported beside the headline (whose config is "single stream"), never in its place.
    two_streams = None
    if n_gpus == 1 and not args.c4 and not dense and not os.environ.get("MVOSR_BENCH_NO_TWO_STREAMS"):
        try:
            ctx2 = _lib.Context(local)
            stream2 = torch.cuda.Stream(device=local)
            ctx2.set_stream(stream2.cuda_stream)
            engine2 = ScaleEngine(ABS_REF, ctx=ctx2)
            rec2 = sharding.RankRecord(F, dev)
            height2 = torch.empty(F, dtype=torch.float64, device=dev)
            outs2 = _lib.Outputs(rec2.raw.data_ptr(), height2.data_ptr(), rec2.level.data_ptr(), rec2.status.data_ptr(),
                                 None, None, None, None, None, None, None, None)
            _lib.check(ctx2.lib.mvosr_ctx_reserve(ctx2.handle, F, pool_pad * repeats), "mvosr_ctx_reserve")
            median2 = sharding.make_gpu_median(engine2)
            sets = ((ctx, engine, outs, rec, median, stream), (ctx2, engine2, outs2, rec2, median2, stream2))

            def step2(k):
                c_, e_, o_, r_, m_, s_ = sets[k & 1]
                with torch.cuda.stream(s_):
                    _lib.check(c_.lib.mvosr_scale_batch(c_.handle, C.byref(e_.params), C.byref(bstruct), C.byref(o_), args.waves, 0, 0), "mvosr_scale_batch")
                    return m_(r_.raw, WINDOW)
            for k in range(4):
                step2(k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                step2(k)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            same = bool(torch.equal(rec.raw.nan_to_num(nan=-1.0), rec2.raw.nan_to_num(nan=-1.0)))
            two_streams = {"ms_per_step": dt2 / args.steps * 1e3, "value": F * args.steps / dt2, "unit": "frames/s",
                           "step_frac": bytes_per_launch / (dt2 / args.steps) / 1e9 / HBM_PEAK_GBPS, "raw_scales_equal": same,
                           "what": "the same %d steps, alternating between two streams (a context, workspace and record each): the road model of "
                                   "step k may run under the scale kernel of step k+1; wall clock per step against the 8 TB/s x algorithmic bytes" % args.steps}
            torch.cuda.set_stream(stream)
            del rec2, height2
            ctx2.close()
        except Exception as exc:                                        # noqa: BLE001
            two_streams = {"error": "%s: %s" % (type(exc).__name__, exc)}
    gpu_raw = rec.raw[:pool_n].cpu().numpy()
    gpu_status = rec.status[:pool_n].cpu().numpy()
    st_all = rec.status.cpu().numpy()
    import zlib
    raw_crc = zlib.crc32(rec.raw.cpu().numpy().tobytes())       # same workload => same number, whatever the build
    if args.c4:                                                   # ... and, for the split job, however it is split: the whole sequence
        whole_raw = last_gather[0].raw() if last_gather[0] is not None else rec.raw[:F]
        whole_st = last_gather[0].status() if last_gather[0] is not None else rec.status[:F]
        raw_crc = zlib.crc32(whole_raw.cpu().numpy().tobytes())
        st_all = whole_st.cpu().numpy()

    sharded = None
    if not args.no_e2e and not args.c4 and pf_pool.max_feat <= 4000 and (n_gpus > 1 or args.e2e_sharded):
        # every rank takes part (the leg has a collective); this context's streams go first (see below)
        torch.cuda.synchronize()
        keep.clear()
        del rec, height
        ctx.close()
        try:
            sharded = e2e_sharded_leg(args, local, rank, n_gpus, sizes, 2024, 16384 if F >= 16384 else 4096)
        except Exception as exc:                                        # noqa: BLE001
            sharded = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if rank == 0:
        value = total_frames * args.steps / elapsed
        achieved = bytes_per_launch / (kernel_ms_avg * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get("entries", {}).get("%s_%d" % (args.workload, args.features)) or (tj if tj.get("features") == args.features and args.workload == "c2" else None)
                if ent and ent.get("frames"):
                    traffic = ent["hbm_bytes_per_frame"] * F          # per launch, like `achieved`
                    traffic_src = "profiles/traffic.json (rocprofv3 --pmc passes of this command; not measured in this run)"
            except Exception:
                traffic = None
        kname = "scale_frames_kernel"
        if dense:
            kname = ("scale_frames_tiled_kernel" if (pf_pool.tile_w and not args.no_tiles) else
                     "scale_frames_dense_feat_kernel" if pf_pool.tri2_ids else "scale_frames_dense_kernel")
        wl = ("synthetic %d-feature / ~%d-triangle frames (T1~%d, T2~%d)" % (args.features, round(t1_mean + t2_mean), round(t1_mean), round(t2_mean))
              if args.workload == "c2" else
              "synthetic %d-feature frames, %.1f %% of them with pixel coordinates rounded to 1/4 px (collinear / cocircular sites: Qhull merges "
              "facets, the device triangulations decline such frames to the host's SciPy; T1~%d, T2~%d)" % (args.features, 100 * SNAP[1], round(t1_mean), round(t2_mean))
              if args.workload == "gridded" else
              "synthetic KITTI-sized frames, 300-1500 features each (mean %.0f; T1~%d, T2~%d)" % (n_mean, round(t1_mean), round(t2_mean)))
        line = {
            "metric": "frames/sec scale-recovery, KITTI-00 flow (~2k feats/frame), 1/2/4/8 GPU",
            "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.c4 else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl + ", %d frames per step per GPU (pool of %d unique frames tiled in HBM), GPU stages only: "
                                     "both Delaunay triangulations precomputed on the host (see e2e), single stream" % (F, pool_n),
                       "frames_per_step_per_gpu": F, "frames_per_step_total": total_frames,
                       "split": ("configs[3]: %d frames split over %d rank(s) in contiguous blocks (rank 0 holds %d)" % (total_frames, n_gpus, F)
                                 if args.c4 else "weak scaling: every rank owns its own frames"),
                       "features_per_frame": n_mean, "tri1_per_frame": t1_mean,
                       "tri2_per_frame": t2_mean, "window": WINDOW, "parallelism": "frames sharded x%d" % n_gpus,
                       "waves_per_frame": args.waves if args.waves else "auto"},
            "world_size": dist.get_world_size() if dist.is_initialized() else 1,
            "backend": (dist.get_backend() if dist.is_initialized() else None),
            "collectives_per_step": collectives_per_step,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "kernel_ms_avg": kernel_ms_avg,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "algorithmic_bytes_per_frame": bytes_per_launch / F,
                         "frames_per_s_kernel_only": F / (kernel_ms_avg * 1e-3),
                         "road_model_kernel_ms_avg": road_ms_avg, "step_kernels_ms_avg": step_ms_avg,
                         "step_achieved": bytes_per_launch / (step_ms_avg * 1e-3) / 1e9,
                         "step_frac": bytes_per_launch / (step_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            "raw_scale_crc32": raw_crc,
            "status_histogram": {str(k): int(v) for k, v in zip(*np.unique(st_all, return_counts=True))},
            "host_delaunay_cpu_ms_per_frame": delaunay_cpu_s * 1e3,
        }
        if os.environ.get("MVOSR_SHARE_GPU") == "1":
            line["diagnostic"] = "share-gpu dry run: %d ranks on %d device(s), gloo — NOT a scaling result" % (n_gpus, torch.cuda.device_count())
        if args.alias_pool:
            line["diagnostic"] = "alias-pool: NOT a benchmark result"
        if want_cpu:
            # the GPU's results for the sample frames (pool frames 0..len(jobs)-1 are the same seeds)
            mism_gpu = 0
            for i, j in enumerate(cpu_jobs):
                ok = gpu_status[i] == j[5] and ((np.isnan(gpu_raw[i]) and np.isnan(j[4])) or gpu_raw[i] == j[4])
                mism_gpu += 0 if ok else 1
            line["parity_mismatches_vs_oracle"] = mism_gpu
            fps, sample_n, _ = cpu_single_core(cpu_jobs, 10.0)
            line["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": "%d runs over %d frames of the same workload, triangulations supplied (vectorised NumPy "
                                              "oracle, one thread; host Delaunay %.1f CPU-ms/frame not included)"
                                              % (sample_n, len(cpu_jobs), delaunay_cpu_s * 1e3)}
            # the secondary legs must not cost the headline line: a failure is reported in its place
            try:
                fps_l, sample_l, mism_l = cpu_single_core(cpu_jobs, 10.0, loops=True)
                line["cpu_baseline_reference_shaped"] = {
                    "value": fps_l, "unit": "frames/s", "cores": 1, "kind": "port",
                    "sample": "%d runs over the same frames; loop-faithful flavour of the oracle (one Python iteration per triangle "
                              "in find_outliers / feature_selection_by_tri, np.matrix(...).I per triangle, as "
                              "scale_calculator.py:151-167,228-229 are written), triangulations supplied; %d mismatches vs the vectorised oracle"
                              % (sample_l, mism_l)}
            except Exception as exc:                                    # noqa: BLE001
                line["cpu_baseline_reference_shaped"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                allv, alln = cpu_all_cores(cpu_pool, cpu_workers, 8.0, fps)
                line["cpu_baseline_all_cores"] = {"value": allv, "unit": "frames/s", "cores": cpu_workers, "kind": "port",
                                                  "sample": "%d frames dealt from the same sample to %d processes (the CPUs this process "
                                                            "may use: affinity and cgroup quota), triangulations supplied" % (alln, cpu_workers)}
            except Exception as exc:                                    # noqa: BLE001
                line["cpu_baseline_all_cores"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if dense:
            line["host_tile_layout_ms_per_frame"] = getattr(build_pool, "host_layout_ms_per_frame", None)
        if n_gpus == 1 and not args.no_e2e:
            # The secondary legs run in their own context.  This one's two streams go first: with torch's two and the legs'
            # two the process would own six — more than ROCm's four hardware queues per process (GPU_MAX_HW_QUEUES), and
            # streams that share a queue serialise: the end-to-end leg's upload stream stopped overlapping its kernels
            # (measured: 172 k instead of 240 k frames/s with a second, idle context alive).
            torch.cuda.synchronize()
            keep.clear()
            ctx.close()
        if sharded is not None:
            line["e2e_sharded"] = sharded
        if two_streams is not None:
            line["two_streams"] = two_streams
        if n_gpus == 1 and not args.no_e2e and dense:
            # dense frames end to end: host Qhull (~140 CPU-ms per call and frame) bounds it; the tile layout is included
            try:
                line["e2e"] = e2e_leg(args, local, sizes, 2024, 96)
            except Exception as exc:                                    # noqa: BLE001
                line["e2e"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                line["e2e_gpu_triangulation"] = e2e_gpu_leg(args, local, sizes, 2024, 1024)
            except Exception as exc:                                    # noqa: BLE001
                line["e2e_gpu_triangulation"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                line["e2e_gpu_exact"] = e2e_gpu_leg(args, local, sizes, 2024, 1024, exact=True)
            except Exception as exc:                                    # noqa: BLE001
                line["e2e_gpu_exact"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        if n_gpus == 1 and not args.no_e2e and not dense and args.workload == "gridded":
            # the decline path: a quarter of these frames goes to the host's SciPy (worker pool), the rest stays on the device
            for key, kw in (("e2e_gpu_exact", {"exact": True}), ("e2e_gpu_triangulation", {})):
                try:
                    line[key] = e2e_gpu_leg(args, local, sizes, 2024, 4096 if SNAP[1] > 0.2 else 16384, **kw)
                except Exception as exc:                                    # noqa: BLE001
                    line[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                line["e2e"] = e2e_leg(args, local, sizes, 2024, 2048)
            except Exception as exc:                                    # noqa: BLE001
                line["e2e"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        elif n_gpus == 1 and not args.no_e2e and not dense:
            try:
                line["e2e"] = e2e_leg(args, local, sizes, 2024, 4096)
            except Exception as exc:                                    # noqa: BLE001
                line["e2e"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                line["e2e_gpu_triangulation"] = e2e_gpu_leg(args, local, sizes, 2024, 32768)
            except Exception as exc:                                    # noqa: BLE001
                line["e2e_gpu_triangulation"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                line["e2e_gpu_exact"] = e2e_gpu_leg(args, local, sizes, 2024, 16384, exact=True)
            except Exception as exc:                                    # noqa: BLE001
                line["e2e_gpu_exact"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                line["e2e_rescale"] = e2e_rescale_leg(args, local, sizes, 2024, 32768)
            except Exception as exc:                                    # noqa: BLE001
                line["e2e_rescale"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                line["latency"] = latency_leg(args, local, sizes, 2024)
            except Exception as exc:                                    # noqa: BLE001
                line["latency"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        print(json.dumps(line))
        sys.stdout.flush()
    if n_gpus > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
