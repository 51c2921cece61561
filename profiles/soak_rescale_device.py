"""Soak of the DEVICE-RESIDENT rescale path: batches of seeded frames (sizes 120-3000, three noise levels, frames with many
features above the vanishing row, duplicate pixels that the device triangulation declines) through
rescale.ScaleEstimator(triangulation="gpu") and through the oracle's restatement with the same counter-based sample sequence
(SciPy triangulations in canonical row form): vote masks, point lists, inlier counts and consumed hypotheses must be EQUAL,
planes / levels / scales within 1e-9.   python profiles/soak_rescale_device.py [batches] [frames per batch]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import rescale, synth                          # noqa: E402
from oracle import rescale_oracle as ro                                   # noqa: E402

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    rng = np.random.default_rng(12)
    gpu = rescale.ScaleEstimator(1.75, window_size=5, triangulation="gpu", ransac_seed=77, delaunay_workers=4)
    gpu.GPU_CHUNK = int(os.environ.get("SOAK_CHUNK", "24"))       # (SOAK_CHUNK=8192 with 600 frames per batch: chunks of 512 frames and more take the triangulation kernel's small-frame and arena-out instantiations)
    ora = ro.OracleRescaleEstimator(1.75, window_size=5, device_seed=77)
    bad, worst, declined, frames_total = 0, 0.0, 0, 0
    for b in range(B):
        frames = []
        for i in range(F):
            n = int(rng.choice([120, 150, 400, 800, 1300, 2000, 3000]))
            f3, f2 = synth.synth_frame(1000 * b + i, n, base_seed=555, sigma=float(rng.choice([0.002, 0.01, 0.03])),
                                       upper_fraction=float(rng.choice([0.0, 0.1, 0.4])))
            if rng.random() < 0.05:                       # duplicate pixels: declined by the device triangulation, redone on the host
                f2 = f2.copy(); f3 = f3.copy()
                j, k = rng.integers(0, n, 2)
                f2[j] = f2[k]; f3[j] = f3[k]
            frames.append((f3, f2))
        got, _ = gpu.scale_calculation_batch([f[0] for f in frames], [f[1] for f in frames], stage=True)
        declined += gpu.last_declined
        L = gpu.last
        for i, (f3, f2) in enumerate(frames):
            try:
                want, _ = ora.scale_calculation(f3.copy(), f2.copy())
            except Exception as exc:                      # noqa: BLE001
                print("oracle raised at", b, i, type(exc).__name__)
                raise
            frames_total += 1
            why = []
            if not np.array_equal(L["valid"][i], ora.last["valid"]):
                why.append("vote mask")
            ids = L["tris2"][i][(L["tri_flags"][i] & 4) != 0].reshape(-1)
            if not np.array_equal(ids, ora.last["flat"].ids):
                why.append("point list (%d vs %d entries)" % (len(ids), len(ora.last["flat"].ids)))
            if "model" in ora.last:
                if int(L["status"][i]) != 0:
                    why.append("status %d" % int(L["status"][i]))
                if int(L["best_ic"][i]) != ora.last["best_ic"] or int(L["used"][i]) != ora.last["used"]:
                    why.append("best_ic %d vs %d, used %d vs %d" % (int(L["best_ic"][i]), ora.last["best_ic"], int(L["used"][i]), ora.last["used"]))
            elif int(L["status"][i]) != 11:
                why.append("status %d, oracle has no plane" % int(L["status"][i]))
            ok = not why
            rel = abs(got[i] - want) / abs(want)
            worst = max(worst, rel)
            if not ok or rel > 1e-9:
                bad += 1
                print("MISMATCH batch", b, "frame", i, len(f3), got[i], want, rel, "; ".join(why))
    print("device-resident rescale soak: %d frames, %d mismatches, %d triangulations declined to the host, worst relative scale difference %.2e"
          % (frames_total, bad, declined, worst))
    sys.exit(1 if bad else 0)
