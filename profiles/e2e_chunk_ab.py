import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.rescale import ScaleEstimator as R
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator as S
F, N = 32768, 2000
pool = [synth.synth_frame(i, N, base_seed=2024) for i in range(4096)]
f3 = [pool[i % 4096][0] for i in range(F)]; f2 = [pool[i % 4096][1] for i in range(F)]
CASES = [(int(x.split(":")[0]), int(x.split(":")[1])) for x in (sys.argv[1] if len(sys.argv) > 1 else "10000000:2,8192000:2,6000000:2,7000000:2,4096000:2,8192000:3").split(",")]
if len(sys.argv) > 2 and sys.argv[2] == "kitti":
    rng = np.random.default_rng(1)
    pool = [synth.synth_frame(i, int(rng.integers(300, 1501)), base_seed=2024) for i in range(4096)]
    f3 = [pool[i % 4096][0] for i in range(F)]; f2 = [pool[i % 4096][1] for i in range(F)]
RAMPS = [tuple(float(v) for v in r.split("/")) for r in os.environ.get("MVOSR_AB_RAMPS", "0.125/0.25/0.5").split(",")]
from mvoscalerecovery_amd import engine
PIECES = [int(x) for x in os.environ.get("MVOSR_AB_PIECES", "4").split(",")]
for which in ("rescale", "scale"):
    for pts, pipe, ramp, pieces in [(p_, q_, r_, u_) for p_, q_ in CASES for r_ in RAMPS for u_ in PIECES]:
        engine.UPLOAD_PIECES = pieces
        est = R(1.75, window_size=5, triangulation="gpu", delaunay_workers=0, ransac_seed=1) if which == "rescale" else S(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", delaunay_workers=0)
        est.GPU_CHUNK_POINTS, est.GPU_PIPELINE, est.GPU_RAMP_FRACTIONS = pts, pipe, ramp
        est.scale_calculation_batch(f3, f2)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); est.scale_calculation_batch(f3, f2); ts.append(time.perf_counter() - t0)
        print(which, "chunk points", pts, "pipeline", pipe, "ramp", ramp, "upload pieces", pieces, "-> %.0f k frames/s (best of 3), median %.0f" % (F / min(ts) / 1e3, F / sorted(ts)[1] / 1e3), flush=True)
