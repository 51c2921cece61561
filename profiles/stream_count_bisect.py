#!/usr/bin/env python3
"""Diagnostic: what made the end-to-end leg 30 % slower inside bench.py than in a process of its own — the batch call of
ScaleEstimator(triangulation="gpu") five times in a row after a setup step:
    python profiles/stream_count_bisect.py plain | torchmem | ctx2 | fork | spawnpool | tm_ctx | tm_ctx_noadopt | tm_fork | ctx_fork | all
Result (profiles/r03_stream_count.txt): torch's streams AND a second mvosr context together — more than ROCm's four hardware
queues per process; either alone is harmless."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
import numpy as np
from mvoscalerecovery_amd import _lib, synth
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
keep = []
if mode in ("torchmem", "all", "tm_ctx", "tm_fork", "tm_ctx_noadopt"):
    import torch
    torch.cuda.set_device(0); st = torch.cuda.Stream(device=0); torch.cuda.set_stream(st)
    keep.append(torch.zeros(10 * (1 << 27), dtype=torch.float64, device="cuda"))     # 10 GB
if mode in ("ctx2", "all", "tm_ctx", "ctx_fork", "tm_ctx_noadopt"):
    c2 = _lib.Context(0); keep.append(c2)
    if mode in ("all", "tm_ctx"):
        c2.set_stream(st.cuda_stream)
    c2.profile(True); c2.profile(False)
if mode in ("fork", "all", "tm_fork", "ctx_fork"):
    _lib.default_context(0)                       # GPU initialised, THEN a worker pool is forked (what build_pool does)
    import multiprocessing as mp
    pool = mp.get_context("fork").Pool(16); keep.append(pool)
    pool.map(abs, range(64))
if mode in ("spawnpool",):
    from mvoscalerecovery_amd import packing
    _lib.default_context(0)
    fr = [synth.synth_frame(i, 500, base_seed=1) for i in range(64)]
    pf = packing.pack_features([f[0] for f in fr], [f[1] for f in fr]); packing.attach_tri1(pf)
F, N = 32768, 2000
pool_f = [synth.synth_frame(i, N, base_seed=2024) for i in range(256)]
f3, f2 = [pool_f[i % 256][0] for i in range(F)], [pool_f[i % 256][1] for i in range(F)]
est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", delaunay_workers=0)
out = []
for k in range(5):
    t0 = time.perf_counter(); est.scale_calculation_batch(f3, f2); out.append(F / (time.perf_counter() - t0))
print("%-10s %s" % (mode, " ".join("%.0f" % v for v in out)))
