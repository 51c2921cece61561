"""Soak: many seeded frames of varied size and noise through the GPU path and the oracle, every raw scale,
status and selection count compared.  python profiles/soak.py [frames] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import constants as K, synth                    # noqa: E402
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator          # noqa: E402
from oracle import scale_oracle as so                                      # noqa: E402

if __name__ == "__main__":
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    frames = []
    for i in range(F):
        sizes = [120, 300, 700, 1100, 1600, 2000, 2048, 2400, 3000] + ([4000, 7000, 12000] if os.environ.get("SOAK_DENSE") else [])
        n = int(rng.choice(sizes))
        frames.append(synth.synth_frame(i, n, base_seed=100000 * seed, sigma=float(rng.choice([0.0, 0.002, 0.01, 0.03, 0.08])),
                                        upper_fraction=float(rng.choice([0.0, 0.1, 0.4]))))
    est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="scipy")
    t0 = time.perf_counter()
    if os.environ.get("SOAK_ONE_LAUNCH"):
        # triangulations from the oracle, every frame in ONE launch: a ragged batch of >= 2048 frames is then launched
        # per size class (the streaming path works in chunks of 512, which never are)
        pre = [so.frame_raw_scale(f3.copy(), f2, 1.75) for f3, f2 in frames]
        ok = [i for i, r in enumerate(pre) if r.tri1 is not None and r.tri2 is not None and len(r.tri2)]
        frames = [frames[i] for i in ok]
        F = len(frames)
        raw, status, level, errs = est.raw_scale_batch([f[0] for f in frames], [f[1] for f in frames],
                                                       [pre[i].tri1 for i in ok], [pre[i].tri2 for i in ok])
    else:
        raw, status, level, errs = est.raw_scale_batch([f[0] for f in frames], [f[1] for f in frames])
    t_gpu = time.perf_counter() - t0
    counts = est.last_counts
    bad = 0
    hist = {}
    t0 = time.perf_counter()
    for i, (f3, f2) in enumerate(frames):
        if i in errs:
            continue
        r = so.frame_raw_scale(f3.copy(), f2, 1.75)
        hist[r.status] = hist.get(r.status, 0) + 1
        same = r.status == status[i]
        if same and not (np.isnan(r.raw_scale) and np.isnan(raw[i])):
            if r.status in (so.ST_NO_FLAT, so.ST_LEVEL):
                same = abs(raw[i] - r.raw_scale) <= 1e-13 * abs(r.raw_scale)
            else:
                same = raw[i] == r.raw_scale
        if same and r.status < so.ST_ERR_LEFT and r.sel is not None:
            same = counts[i, K.CNT_SELECTED] == len(r.sel.selected_ids) and counts[i, K.CNT_VALID] == int(r.valid.sum())
        if not same:
            bad += 1
            print("MISMATCH frame", i, "n", len(f3), "gpu", status[i], raw[i], "oracle", r.status, r.raw_scale)
    print("soak: %d frames, %d host errors, %d mismatches; statuses %s; gpu path %.1f s, oracle %.1f s"
          % (F, len(errs), bad, dict(sorted(hist.items())), t_gpu, time.perf_counter() - t0))
    sys.exit(1 if bad else 0)
