#!/usr/bin/env python3
"""Headline benchmark: frames/s of scale recovery on synthetic KITTI-shaped flow.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--features 2000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of F frames per GPU, resident in HBM:
mvosr_scale_batch = the scale kernel (feature_remap -> depth-order vote on tri1 -> compaction ->
per-triangle plane normal / pitch / height on tri2 -> height_level -> selected points; one
workgroup per frame) + the road-model kernel (histogram / modes / skew -> height -> raw scale; one
wavefront per frame), then (N>1) one RCCL all-gather of the raw scales + statuses, then the
window-median kernel over the gathered sequence.  Workload = BASELINE.json configs[1]: 2000 features /
~4000 triangles per frame (T1~3981 + T2~3780), both Delaunay triangulations precomputed on the
host (they are inputs of the GPU path, like the optical flow itself).  Weak scaling: every rank
owns F frames.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     algorithmic bytes per launch (8*(3N+N)+12*(T1+T2)+12 per frame)
               / average duration of the dominant kernel (scale_frames_kernel), measured with HIP
               events recorded on its launch stream around that kernel (mvosr_ctx_profile),
               against the 8 TB/s HBM peak; `step_*` gives the same for both kernels of the step;
  cpu_baseline the CPU oracle (NumPy port of the reference) timed on this box's host, one
               core, on a bounded sample of the same frames (N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 measured copy
ABS_REF = 1.75                  # param.camera_h
WINDOW = 5                      # main.py:55


def build_pool(ctx, engine, n_features, pool, seed):
    """P unique synthetic frames, both triangulations (SciPy on the host; the vote mask that the
    second triangulation is built on comes from the GPU vote kernel)."""
    from mvoscalerecovery_amd import packing, synth
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs
    frames = [synth.synth_frame(i, n_features, base_seed=seed) for i in range(pool)]
    pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames])
    t0 = time.perf_counter()
    packing.attach_tri1(pf)
    t_del1 = time.perf_counter() - t0
    cap = int(ctx.lib.mvosr_max_lds_features())
    if n_features > cap:            # dense frames: Z-order layout (what ScaleEstimator.scale_calculation_batch does)
        packing.apply_locality_order(pf, min_features=cap + 1)
    db = DeviceBatch(ctx, pf, with_tri2=False)
    out = DeviceOutputs(ctx, db, counts=True, stage=True)
    engine.outlier_vote_batch(db, out)
    ctx.sync()
    counters = out.get("vote_counters")
    masks = [counters[pf.frame_slice(f)] >= 0 for f in range(pool)]
    out.free()
    db.free()
    t0 = time.perf_counter()
    packing.attach_tri2(pf, None, masks, feature_ids=n_features > cap)    # dense: rows numbered over the features (no compaction)
    t_del2 = time.perf_counter() - t0
    return frames, pf, masks, (t_del1 + t_del2) / pool


def cpu_baseline(frames, pf, gpu_raw, gpu_status, budget_s=12.0):
    """Time the oracle (one core) on the pool frames with the triangulations supplied, i.e. the
    same work the GPU kernel does; also checks the GPU results against it."""
    from oracle import scale_oracle as so
    done, t_used = 0, 0.0
    mismatches = 0
    reordered = any(p is not None for p in (pf.extra.get("perm") or []))
    tris = {}
    i = 0
    P = len(frames)
    while t_used < budget_s and done < 4 * P:
        f = i % P
        if reordered:                      # dense frames were re-laid out for the GPU: the oracle triangulates itself (untimed)
            if f not in tris:
                r0 = so.frame_raw_scale(frames[f][0], frames[f][1], ABS_REF)
                tris[f] = (r0.tri1, r0.tri2)
            tri1, tri2 = tris[f]
        else:
            tri1 = pf.tri1[pf.tri1_off[f]:pf.tri1_off[f + 1]]
            tri2 = pf.tri2[pf.tri2_off[f]:pf.tri2_off[f + 1]]
        t0 = time.perf_counter()
        r = so.frame_raw_scale(frames[f][0], frames[f][1], ABS_REF, tri1, tri2, keep=False)
        t_used += time.perf_counter() - t0
        if i < P:
            same = (r.status == gpu_status[f]) and ((np.isnan(r.raw_scale) and np.isnan(gpu_raw[f])) or r.raw_scale == gpu_raw[f])
            mismatches += 0 if same else 1
        done += 1
        i += 1
    return done / t_used, done, mismatches


_ALL_CORES_JOBS = None          # inherited by the forked workers of cpu_baseline_all_cores


def _all_cores_job(i):
    from oracle import scale_oracle as so
    f3, f2, t1, t2 = _ALL_CORES_JOBS[i % len(_ALL_CORES_JOBS)]
    return so.frame_raw_scale(f3.copy(), f2, ABS_REF, t1, t2).raw_scale


def cpu_baseline_all_cores(frames, pf, per_worker=1500):
    """The same oracle on every CPU the process may use (cgroup quota respected), one process per CPU,
    frames dealt round-robin from the pool, triangulations supplied.  Skipped for re-laid-out (dense) pools."""
    global _ALL_CORES_JOBS
    import multiprocessing as mp
    from mvoscalerecovery_amd import packing
    if any(p is not None for p in (pf.extra.get("perm") or [])) or pf.tri2_ids:
        return None
    workers = packing.available_cpus()
    _ALL_CORES_JOBS = [(frames[i][0], frames[i][1], pf.tri1[pf.tri1_off[i]:pf.tri1_off[i + 1]], pf.tri2[pf.tri2_off[i]:pf.tri2_off[i + 1]])
                       for i in range(len(frames))]
    n = workers * per_worker
    with mp.get_context("fork").Pool(workers) as pool:
        pool.map(_all_cores_job, range(workers * 4))                       # imports + first touches
        t0 = time.perf_counter()
        pool.map(_all_cores_job, range(n), chunksize=25)
        dt = time.perf_counter() - t0
    return n / dt, n, workers


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=65536, help="frames per step per GPU")
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--pool", type=int, default=128, help="unique synthetic frames tiled to --frames")
    ap.add_argument("--waves", type=int, default=0, help="wavefronts per frame (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--alias-pool", action="store_true",
                    help="diagnostic: every tile reads the feature planes of the SAME pool frames (cache-resident: 41 %% "
                         "less HBM traffic); the JSON line is marked and is not a benchmark result")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mvoscalerecovery_amd import _lib, packing, sharding
    from mvoscalerecovery_amd.engine import DeviceBatch, ScaleEngine

    rank, local, world = sharding.init_distributed()
    if world != args.gpus:
        if rank == 0:
            print("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus), file=sys.stderr)
    n_gpus = world
    torch.cuda.set_device(local)
    ctx = _lib.Context(local)                       # raises if libmvosr.so / the GPU is missing
    stream = torch.cuda.Stream(device=local)        # a real (non-null) stream shared by torch/RCCL and the kernels
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    engine = ScaleEngine(ABS_REF, ctx=ctx)

    pool = min(args.pool, args.frames)
    repeats = max(1, args.frames // pool)
    F = pool * repeats
    frames, pf_pool, masks, delaunay_s = build_pool(ctx, engine, args.features, pool, seed=2024)
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
    bytes_per_launch = pf_pool.algorithmic_bytes() * repeats
    n_mean = float(pf_pool.feat_cnt.mean())
    t1_mean = float(pf_pool.tri1_off[-1]) / pool
    t2_mean = float(pf_pool.tri2_off[-1]) / pool

    raw = torch.empty(F, dtype=torch.float64, device=dev)
    height = torch.empty(F, dtype=torch.float64, device=dev)
    level = torch.empty(F, dtype=torch.float64, device=dev)
    status = torch.empty(F, dtype=torch.int32, device=dev)
    outs = _lib.Outputs(raw.data_ptr(), height.data_ptr(), level.data_ptr(), status.data_ptr(),
                        None, None, None, None, None, None, None, None)
    median = sharding.make_gpu_median(engine)
    total_frames = F * n_gpus

    force_gather = bool(os.environ.get("MVOSR_BENCH_FORCE_GATHER")) and dist.is_initialized()   # diagnostic: the N>1 step on one rank

    def step(ev_pair=None):
        if ev_pair is not None:
            ctx.record(ev_pair[0])
        _lib.check(ctx.lib.mvosr_scale_batch(ctx.handle, C.byref(engine.params), C.byref(bstruct), C.byref(outs),
                                             args.waves, 0, 0), "mvosr_scale_batch")
        if ev_pair is not None:
            ctx.record(ev_pair[1])
        if n_gpus > 1 or force_gather:
            filtered, _, _ = sharding.gather_and_filter(raw, status, total_frames, WINDOW, median)
        else:
            filtered = median(raw, WINDOW)
        return filtered

    def barrier():
        if n_gpus > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.profile(True)                    # HIP events around each of the step's two kernels, on the launch stream
    events = [(ctx.event(), ctx.event()) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        filtered = step(events[k])
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if n_gpus > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    step_ms_avg = float(np.mean([ctx.elapsed_ms(a, b) for a, b in events]))       # both kernels of the step
    prof = [ctx.profile_read(k) for k in range(min(args.steps, 64))]
    kernel_ms_avg = float(np.mean([p[0] for p in prof]))                               # scale_frames_kernel
    road_ms_avg = float(np.mean([p[1] for p in prof]))                                 # road_model_kernel
    ctx.profile(False)

    gpu_raw = raw[:pool].cpu().numpy()
    gpu_status = status[:pool].cpu().numpy()
    st_all = status.cpu().numpy()

    if rank == 0:
        value = total_frames * args.steps / elapsed
        achieved = bytes_per_launch / (kernel_ms_avg * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("frames") and tj.get("features") == args.features:
                    traffic = tj["hbm_bytes_per_frame"] * F          # per launch, like `achieved`
            except Exception:
                traffic = None
        line = {
            "metric": "frames/sec scale-recovery, KITTI-00 flow (~2k feats/frame), 1/2/4/8 GPU",
            "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic %d-feature / ~%d-triangle frames (T1~%d, T2~%d), %d frames per step per GPU "
                                   "(pool of %d unique frames tiled in HBM), precomputed Delaunay x2, single stream"
                                   % (args.features, round(t1_mean + t2_mean), round(t1_mean), round(t2_mean), F, pool),
                       "frames_per_step_per_gpu": F, "features_per_frame": n_mean, "tri1_per_frame": t1_mean,
                       "tri2_per_frame": t2_mean, "window": WINDOW, "parallelism": "frames sharded x%d" % n_gpus,
                       "waves_per_frame": args.waves if args.waves else "auto"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": ("scale_frames_dense_kernel" if args.features > ctx.lib.mvosr_max_lds_features() else "scale_frames_kernel"),
                         "kernel_ms_avg": kernel_ms_avg,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "algorithmic_bytes_per_frame": bytes_per_launch / F,
                         "frames_per_s_kernel_only": F / (kernel_ms_avg * 1e-3),
                         "road_model_kernel_ms_avg": road_ms_avg, "step_kernels_ms_avg": step_ms_avg,
                         "step_achieved": bytes_per_launch / (step_ms_avg * 1e-3) / 1e9,
                         "step_frac": bytes_per_launch / (step_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            "status_histogram": {str(k): int(v) for k, v in zip(*np.unique(st_all, return_counts=True))},
            "host_delaunay_ms_per_frame": delaunay_s * 1e3,
        }
        if args.alias_pool:
            line["diagnostic"] = "alias-pool: NOT a benchmark result"
        if n_gpus == 1 and not args.no_cpu_baseline:
            fps, sample_n, mism = cpu_baseline(frames, pf_pool, gpu_raw, gpu_status)
            line["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": "%d frames of the same %d-feature pool, triangulations supplied "
                                              "(NumPy oracle, one thread; host Delaunay %.1f ms/frame not included)"
                                              % (sample_n, args.features, delaunay_s * 1e3)}
            line["parity_mismatches_vs_oracle"] = mism
            allc = cpu_baseline_all_cores(frames, pf_pool)
            if allc is not None:
                line["cpu_baseline_all_cores"] = {"value": allc[0], "unit": "frames/s", "cores": allc[2], "kind": "port",
                                                  "sample": "%d frames dealt from the same pool to %d processes (the CPUs this "
                                                            "process may use: affinity and cgroup quota), triangulations supplied"
                                                            % (allc[1], allc[2])}
        print(json.dumps(line))
    if n_gpus > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
