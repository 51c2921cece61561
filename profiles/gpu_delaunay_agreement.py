#!/usr/bin/env python3
"""How often the scales of the DECLARED DEVIATION — triangulation="gpu", check_triangle="fixed": the order-invariant vote
on the device's triangulations — equal the reference's own, and what the device stage buys end to end.
(Against its own oracle, Oracle(check_triangle="fixed"), that mode is bit-equal on every frame: tests/test_gpu_dropin.py.)
Also: the same fixed vote on SciPy's triangulations (identical to "gpu" by construction), and "gpu" with the reference's
flag pattern (unpinned: Qhull's row rotation is not reproducible).

    python profiles/gpu_delaunay_agreement.py  >  profiles/r03_gpu_delaunay.json

Sequences: the 4541-frame main_offline-shaped golden (tests/golden/seq4541.npz: the reference's raw and filtered scales)
and the 400 adversarial frames of tests/golden/frame_fuzz.npz."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_npz                                              # noqa: E402
from mvoscalerecovery_amd import offline, packing, synth, constants as K  # noqa: E402
from mvoscalerecovery_amd.scale_calculator import ScaleEstimator          # noqa: E402

packing.start_pool(None)
out = {}
z, meta = load_npz("seq4541.npz")
data = synth.synth_sequence_dict(meta["n_frames"], base_seed=meta["seed"], **meta["kw"])
for mode, kw in (("scipy", {"triangulation": "scipy"}), ("gpu", {"triangulation": "gpu"}), ("scipy_fixed", {"triangulation": "scipy", "check_triangle": "fixed"}),
                 ("gpu_reference_pattern", {"triangulation": "gpu", "check_triangle": "reference"})):
    est = ScaleEstimator(meta["abs_ref"], window_size=meta["window"], mutate_inputs=False, **kw)
    offline.run_sequence_batched(data, ScaleEstimator(meta["abs_ref"], window_size=meta["window"], mutate_inputs=False, **kw))     # warm-up
    t0 = time.perf_counter()
    res = offline.run_sequence_batched(data, est)
    dt = time.perf_counter() - t0
    raw = est.last_raw_scale
    same_raw = (raw == z["raw_scales"]) | (np.isnan(raw) & np.isnan(z["raw_scales"]))
    same_f = (res["scales"] == z["scales"]) | (np.isnan(res["scales"]) & np.isnan(z["scales"]))
    rel = np.abs(raw - z["raw_scales"]) / np.abs(z["raw_scales"])
    out["seq4541_" + mode] = {
        "processed_frames": int(len(raw)), "raw_scale_equal_fraction": float(np.mean(same_raw)),
        "filtered_scale_equal_fraction": float(np.mean(same_f)),
        "raw_scale_rel_diff_median_of_differing": float(np.median(rel[~same_raw])) if (~same_raw).any() else 0.0,
        "raw_scale_rel_diff_max": float(np.nanmax(rel)), "raw_scale_within_1e-4_fraction": float(np.mean(rel <= 1e-4)),
        "status_equal_fraction": None, "seconds": dt, "frames_per_s_end_to_end": float(len(raw) / dt),
        "check_triangle": est.check_triangle, "triangulation": est.triangulation,
        "declined_by_the_device_stage_last_chunk": int(est.last_declined)}
zf = np.load(os.path.join(ROOT, "tests", "golden", "frame_fuzz.npz"))
names = list(zf["exception_names"])
agree = tot = exc_agree = 0
for i in range(len(zf["scale"])):
    f3, f2 = synth.fuzz_frame(i, int(zf["seed"]))
    est = ScaleEstimator(1.75, window_size=5, triangulation="gpu")
    want_exc = names[zf["raised"][i] - 1] if zf["raised"][i] else None
    try:
        s, sd = est.scale_calculation(f3.copy(), f2.copy())
        got_exc = None
    except Exception as exc:                                               # noqa: BLE001
        got_exc, s = type(exc).__name__, None
    tot += 1
    exc_agree += got_exc == want_exc
    if want_exc is None and got_exc is None:
        agree += (s == zf["scale"][i]) or (np.isnan(s) and np.isnan(zf["scale"][i]))
    elif want_exc == got_exc:
        agree += 1
out["frame_fuzz_gpu"] = {"frames": tot, "same_outcome_fraction": agree / tot, "same_exception_or_none_fraction": exc_agree / tot}
# the device stage alone on bench-shaped frames
frames = [synth.synth_frame(i, 2000, base_seed=2024)[1] for i in range(1024)]
ctx = est.engine.ctx
packing.delaunay_gpu(ctx, frames[:64])
t0 = time.perf_counter(); got = packing.delaunay_gpu(ctx, frames); dt = time.perf_counter() - t0
out["delaunay_stage"] = {"point_sets": len(frames), "points_each": 2000, "declined": int(sum(g is None for g in got)),
                         "sets_per_s_incl_transfers": len(frames) / dt}
t0 = time.perf_counter(); packing.delaunay_many(frames[:256], None); dt = time.perf_counter() - t0
out["delaunay_stage"]["scipy_pool_sets_per_s"] = 256 / dt
out["delaunay_stage"]["host_cpus"] = packing.resolve_workers(None)
print(json.dumps(out, indent=1))
