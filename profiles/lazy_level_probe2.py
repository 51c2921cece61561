import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
F, N = 16384, 2000
pool = [synth.synth_frame(i, N, base_seed=2024) for i in range(2048)]
f3, f2 = [pool[i % len(pool)][0] for i in range(F)], [pool[i % len(pool)][1] for i in range(F)]
for two in (True, False):
    for lazy in (True, False):
        est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=0)
        est.GPU_EXACT_LAZY_LEVEL = lazy
        est.GPU_EXACT_TWO_CONTEXTS = two
        real = est._chunk_gpu_finish
        seen = []
        def spy(st, a, b, **kw):
            lib = st["engine"].ctx.lib
            if hasattr(lib, "mvosr_debug_redo_list") and st.get("gpu"):
                out = np.zeros(8, dtype=np.int32)
                lib.mvosr_debug_redo_list(st["engine"].ctx.handle, C.c_int64(len(a)), C.c_void_p(out.ctypes.data), 8)
                seen.append((len(a), int(out[0]), out[1:1 + min(int(out[0]), 4)].tolist()))
            return real(st, a, b, **kw)
        est._chunk_gpu_finish = spy
        est.scale_calculation_batch(f3, f2)
        seen.clear()
        t0 = time.perf_counter(); est.scale_calculation_batch(f3, f2); dt = time.perf_counter() - t0
        print("two_contexts %s lazy %s: %.1f ms; per chunk (frames, list length, list): %s" % (two, lazy, 1e3 * dt, seen), flush=True)
