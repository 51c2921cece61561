"""ONE equivalence test over the estimator's mode matrix (DESIGN.md §4, table "construction x call shape"): every way of constructing
``scale_calculator.ScaleEstimator`` x every shape of call runs the same frames — the 440 adversarial frames of
tests/golden/frame_fuzz*.npz (the reference's own results, unpatched and with check_triangle patched) and 40 ordinary frames — and
must give the reference's numbers for its vote mode, whichever kernels, triangulation source and host path the combination selects.
Needs a real MI355X:  python -m pytest tests -m gpu"""
import os

import numpy as np
import pytest

from gpu_helpers import _oracle

pytestmark = pytest.mark.gpu

# construction -> (constructor keywords, the vote mode it must reproduce)
CONSTRUCTIONS = {
    "default": ({}, "reference"),                                                  # /root/reference/src/main.py:55 as written
    "gpu+reference": ({"triangulation": "gpu", "check_triangle": "reference"}, "reference"),
    "scipy": ({"triangulation": "scipy"}, "reference"),
    "gpu": ({"triangulation": "gpu"}, "fixed"),                                    # the declared-deviation speed mode
    "scipy+fixed": ({"triangulation": "scipy", "check_triangle": "fixed"}, "fixed"),
}
# call shape -> how the frames are fed
SHAPES = ("per_frame", "small_batches", "one_batch", "streamed_chunks", "two_contexts_off", "standin_off", "sharded_driver_halves")


def _fuzz(mode):
    from mvoscalerecovery_amd import synth
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "frame_fuzz.npz" if mode == "reference" else "frame_fuzz_fixed.npz"))
    idx = [i for i in range(len(z["scale"])) if not z["raised"][i]]
    return [synth.fuzz_frame(i, int(z["seed"])) for i in idx], np.array([z["scale"][i] for i in idx])


def _ordinary():
    from mvoscalerecovery_amd import synth
    return [synth.synth_frame(i, 250 + 47 * i, base_seed=8086, upper_fraction=0.1) for i in range(40)]


def _raw(est, shape, frames):
    """Raw scales and statuses of `frames` through one call shape (no cross-frame state: raw_scale_batch / fresh windows)."""
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    f3s, f2s = [f[0] for f in frames], [f[1] for f in frames]
    if shape == "per_frame":
        raw, status = [], []
        for f3, f2 in frames:
            r, s, _, err = est.scale_calculation_batch([f3.copy()], [f2], _single=True, _raw_only=True)
            assert not err
            raw.append(r[0]); status.append(s[0])
        return np.array(raw), np.array(status)
    if shape == "small_batches":                      # a handful of frames per call: below the device replay's break-even
        out = [est.raw_scale_batch([a.copy() for a in f3s[k:k + 5]], f2s[k:k + 5]) for k in range(0, len(frames), 5)]
    elif shape == "sharded_driver_halves":            # what offline.run_sequence_sharded does on two ranks: two blocks, then one record
        h = len(frames) // 2
        out = [est.raw_scale_batch([a.copy() for a in f3s[:h]], f2s[:h]), est.raw_scale_batch([a.copy() for a in f3s[h:]], f2s[h:])]
    else:
        if shape == "streamed_chunks":                # chunk boundaries everywhere
            est.PIPELINE_CHUNK = 32; est.GPU_MIN_CHUNK = 32; est.GPU_CHUNK = 64; est.GPU_EXACT_CHUNK = 64
        if shape == "two_contexts_off":
            est.GPU_EXACT_TWO_CONTEXTS = False; est.GPU_REDO_CONTEXT = False
        if shape == "standin_off":
            est.GPU_EXACT_STANDIN = False; est.GPU_EXACT_HOST_REPLAY = False; est._host_replay = False
        out = [est.raw_scale_batch([a.copy() for a in f3s], f2s)]
    assert all(not o[3] for o in out)
    return np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out])


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("construction", list(CONSTRUCTIONS))
def test_mode_matrix_equivalence(gpu, construction, shape):
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    so = _oracle()
    kw, mode = CONSTRUCTIONS[construction]
    est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=0, **kw)
    assert est.check_triangle == mode
    fuzz, want = _fuzz(mode)
    raw, status = _raw(est, shape, fuzz)
    same = (raw == want) | (np.isnan(raw) & np.isnan(want))
    assert same.all(), (construction, shape, np.nonzero(~same)[0][:8], raw[~same][:4], want[~same][:4])
    # ordinary frames: the oracle in the construction's vote mode (SciPy's rows), scale and status
    frames = _ordinary()
    est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=0, **kw)
    raw, status = _raw(est, shape, frames)
    for k, (f3, f2) in enumerate(frames):
        r = so.frame_raw_scale(f3, f2, 1.75, check_triangle=mode)
        assert raw[k] == r.raw_scale and status[k] == r.status, (construction, shape, k, raw[k], r.raw_scale)


def test_mode_matrix_is_the_documented_one():
    """DESIGN.md §4's table lists exactly these constructions and call shapes (so the document cannot drift from the test)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "DESIGN.md")).read()
    for name in list(CONSTRUCTIONS) + list(SHAPES):
        assert "`%s`" % name in text, name
