"""Soak of the device-resident path: batches of random frames through ScaleEstimator(triangulation="gpu") (Delaunay #1 -> vote
-> seeded Delaunay #2 -> scale kernel -> road model, in chunks) against ScaleEstimator(triangulation="scipy",
check_triangle="fixed") on the same frames (host Qhull rows, the same kernels) — scales and stds must be identical arrays —
and against the loop-faithful NumPy oracle on a sample.
    python profiles/soak_gpu_path.py [seconds] [exact]
With "exact": the reference's own vote on Qhull's rows built on the device (triangulation="gpu", check_triangle="reference") against
the host-SciPy default and the reference-mode oracle — the path without a declared deviation."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from mvoscalerecovery_amd import synth
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
import scale_oracle as so

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
EXACT = len(sys.argv) > 2 and sys.argv[2] == "exact"
MODE = "reference" if EXACT else "fixed"
rng = np.random.default_rng(777)
t_end = time.time() + budget
batches = frames = bad = oracle_checked = oracle_bad = declined = 0
while time.time() < t_end:
    big = rng.uniform() < 0.35          # a chunk that fills the GPU: the Delaunay kernel's two- / four-wavefront instantiations
    if big:
        F = int(rng.integers(520, 1400))
        sizes = rng.integers(60, int(rng.choice([450, 1100, 1500, 2100, 3200])), F)
    else:
        F = int(rng.integers(40, 400))
        sizes = rng.integers(150, 2600, F)
    seed = int(rng.integers(1 << 30))
    fr = [synth.synth_frame(i, int(n), base_seed=seed, upper_fraction=float(rng.uniform(0.0, 0.3))) for i, n in enumerate(sizes)]
    f3, f2 = [f[0] for f in fr], [f[1] for f in fr]
    g = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", check_triangle=MODE, delaunay_workers=0)
    g.GPU_CHUNK = 2048 if big else int(rng.integers(16, 256))
    if EXACT:
        g.GPU_EXACT_CHUNK = int(rng.choice([64, 300, 16384]))
        if rng.uniform() < 0.5:
            g.GPU_MIN_CHUNK = 16        # (chunk boundaries everywhere in half of the batches)
    h = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="scipy", check_triangle=MODE, delaunay_workers=8)
    try:
        sg = g.scale_calculation_batch(f3, f2); eg = None
    except Exception as exc:          # noqa: BLE001
        sg, eg = None, type(exc).__name__
    try:
        sh = h.scale_calculation_batch(f3, f2); eh = None
    except Exception as exc:          # noqa: BLE001
        sh, eh = None, type(exc).__name__
    ok = eg == eh and (sg is None or (np.array_equal(sg[0], sh[0], equal_nan=True) and np.array_equal(sg[1], sh[1])))
    ok = ok and np.array_equal(np.asarray(g.last_raw_scale), np.asarray(h.last_raw_scale), equal_nan=True)
    if eg is None:                  # the level the estimator is left with (round 6: computed when read, for the exact path's batches)
        lg, lh = getattr(g, "height_level", None), getattr(h, "height_level", None)
        ok = ok and (lg == lh or (lg is not None and lh is not None and np.isnan(lg) and np.isnan(lh)))

    bad += 0 if ok else 1
    if not ok:
        print("MISMATCH batch seed=%d F=%d chunk=%d: %s / %s" % (seed, F, g.GPU_CHUNK, eg, eh))
    declined += int(g.last_declined)
    if batches % 10 == 0 and sg is not None:                  # the oracle on the first frames of every tenth batch
        o = so.OracleScaleEstimator(1.75, window_size=5, check_triangle=MODE)
        for i in range(min(12, F)):
            s, sd = o.scale_calculation(f3[i].copy(), f2[i].copy())
            oracle_checked += 1
            if s != sg[0][i] or sd != sg[1][i]:
                oracle_bad += 1
    batches += 1; frames += F
print(("check_triangle=%s: " % MODE) + "batches %d, frames %d: %d batches differ from the host-triangulated run; oracle sample %d frames, %d differ; declined in last chunks %d"
      % (batches, frames, bad, oracle_checked, oracle_bad, declined))
