#!/usr/bin/env python3
"""How many frames of a chunk does the exact pass redo, and which?  (the exact path's tail is the list replay of those frames: 23 ms
however few).  (1) One chunk through the product kernels alone (MVOSR_WAVES_HOT_ONLY): the frames that come back MVOSR_ST_REDO are the
ones the kernels themselves put on the list.  (2) With a diagnostic build (profiles/ab_build.sh ablate -DMVOSR_ABLATE; MVOSR_LIB_PATH):
the list of the chunk's real launch, read back from the context's workspace.   python profiles/redo_list_census.py [frames] [features]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import _lib, synth                                  # noqa: E402
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator               # noqa: E402
from mvoscalerecovery_amd.engine import exact_mask_of                          # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
frames = [synth.synth_frame(200000 + i, N, base_seed=2024) for i in range(F)]
est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=0)
f3s, f2s = [f[0] for f in frames], [f[1] for f in frames]
st = est._chunk_gpu(f3s, f2s, False, hot_only=True)
raw, status, level, counts, errs = est._chunk_gpu_finish(st, f3s, f2s)
redo = int((status == _lib.ST_REDO).sum())
hist = {int(k): int(v) for k, v in zip(*np.unique(status, return_counts=True))}
print("%d frames of %d features through the product kernels alone: %d come back MVOSR_ST_REDO (%.2f %%); statuses %s; exact mask adds %d"
      % (F, N, redo, 100.0 * redo / F, hist, int(exact_mask_of(np.full(F, N)).sum())))
lib = est.engine.ctx.lib
if hasattr(lib, "mvosr_debug_redo_list"):
    for lazy in (False, True):
        st = est._chunk_gpu(f3s, f2s, False, lazy_last=lazy)
        out = np.zeros(64, dtype=np.int32)
        rc = lib.mvosr_debug_redo_list(est.engine.ctx.handle, C.c_int64(F), C.c_void_p(out.ctypes.data), 64)
        est._chunk_gpu_finish(st, f3s, f2s)
        print("real launch, lazy_last=%s: rc %d, the exact pass's list holds %d frame(s): %s" % (lazy, rc, out[0], out[1:1 + min(int(out[0]), 16)].tolist()))
import time
for lazy in (False, True, False, True):
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        st = est._chunk_gpu(f3s, f2s, False, lazy_last=lazy)
        est._chunk_gpu_finish(st, f3s, f2s)
        t.append(time.perf_counter() - t0)
    print("one chunk of %d frames, lazy_last=%s: %.1f ms" % (F, lazy, 1e3 * sorted(t)[1]))
