"""Helpers shared by the -m gpu parity tests (tests/test_gpu_*.py): packing, the oracle's per-frame results, the fused launch and
the frame-by-frame comparison.  Test infrastructure."""
import os

import numpy as np
import pytest

from conftest import load_json, load_npz


def _oracle():
    from oracle import scale_oracle as so
    return so


def _pack(frames, tri1s=None, tri2s=None, masks=None, feature_ids=False):
    from mvoscalerecovery_amd import packing
    pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames])
    packing.attach_tri1(pf, tri1s)
    if tri2s is not None:
        packing.attach_tri2(pf, tri2s, masks, feature_ids=feature_ids)
    return pf


def _oracle_frames(frames, abs_ref=1.75):
    so = _oracle()
    return [so.frame_raw_scale(f3, f2, abs_ref) for f3, f2 in frames]


def _run_fused(gpu, frames, oracle_res, waves=0, abs_ref=1.75, stage=True, per_triangle=False, hist=True, feature_ids=False):
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    pf = _pack(frames, [r.tri1 for r in oracle_res], [r.tri2 for r in oracle_res], [r.valid for r in oracle_res], feature_ids)
    eng = ScaleEngine(abs_ref, ctx=gpu)
    db = DeviceBatch(gpu, pf)
    out = DeviceOutputs(gpu, db, counts=True, stage=stage, per_triangle=per_triangle, hist=hist)
    eng.scale_batch(db, out, waves=waves)
    gpu.sync()
    res = {k: out.get(k) for k in out.bufs}
    out.free()
    db.free()
    return pf, res


def _assert_frame_equal(so, r, res, pf, f, check_stage=True):
    from mvoscalerecovery_amd import constants as K
    sl = pf.frame_slice(f)
    assert res["status"][f] == r.status, (f, res["status"][f], r.status)
    if check_stage:
        assert np.array_equal(res["vote_counters"][sl], r.counters), f
        nv = int(r.valid.sum())
        assert res["counts"][f, K.CNT_VALID] == nv
        sel = np.nonzero(res["selected"][sl][:nv])[0]
        if r.status != so.ST_ERR_SINGULAR:
            assert np.array_equal(sel, r.sel.selected_ids), f
            assert res["counts"][f, K.CNT_TRI_PITCH] == int(r.sel.valid_pitch.sum())
            assert res["counts"][f, K.CNT_TRI_VALID] == int(r.sel.tri_valid.sum())
    if np.isnan(r.height_level):
        assert np.isnan(res["height_level"][f])
    elif "selected" in res or r.status in (so.ST_NO_FLAT, so.ST_LEVEL):
        # with stage outputs (EXACT kernel mode), and whenever the level is the frame's result, height_level is
        # np.mean's own double: the steep triangles' heights summed in NumPy's pairwise order (:239-240)
        assert res["height_level"][f] == r.height_level, (f, res["height_level"][f], r.height_level)
    else:
        # product mode: the sweep's fixed-order sum (same value to ~1e-15; no decision can depend on the difference:
        # frames in which one could are redone in EXACT mode)
        assert abs(res["height_level"][f] - r.height_level) <= 1e-13 * abs(r.height_level), f
    for name, want in (("height", r.height), ("raw_scale", r.raw_scale)):
        got = res[name][f]
        assert (np.isnan(got) and np.isnan(want)) or got == want, (f, name, got, want)
    if r.road is not None and "hist" in res:
        assert np.array_equal(res["hist"][f, 0], r.road.hist_raw), f
        assert np.array_equal(res["hist"][f, 1], r.road.hist), f
        assert res["counts"][f, K.CNT_KEPT] == r.road.n_kept
        assert res["counts"][f, K.CNT_MODES] == r.road.n_modes
        assert res["counts"][f, K.CNT_MODE_LEFT] == r.road.mode_left
        assert res["counts"][f, K.CNT_MODE_RIGHT] == r.road.mode_right
        if not np.isnan(r.road.skew):
            np.testing.assert_allclose(res["stats"][f, :3], [r.road.mean, r.road.std, r.road.skew], rtol=1e-12)


def _pack_tiled(frames, ores):
    """Dense batch in the layout of packing.apply_tile_order (what ScaleEstimator.scale_calculation_batch builds)."""
    from mvoscalerecovery_amd import packing
    pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames])
    packing.attach_tri1(pf, [r.tri1 for r in ores])
    packing.apply_tile_order(pf)
    masks = [np.asarray(r.valid)[pf.extra["perm"][f]] for f, r in enumerate(ores)]
    packing.attach_tri2(pf, [r.tri2 for r in ores], masks, feature_ids=True)
    assert pf.tile_w == 512 and pf.tri2_ids == 1
    return pf


# ---------------------------------------------------------------- the `rescale` variant (f2/f4)
def _ransac_triples(seed, call, n, h=100):
    rng = np.random.default_rng([seed, call])
    return np.stack([rng.choice(n, 3, replace=False) for _ in range(h)]).astype(np.int32)


def _rescale_frames(sizes, base_seed=2468):
    from mvoscalerecovery_amd import synth
    return [synth.synth_frame(i, n, base_seed=base_seed, upper_fraction=0.1) for i, n in enumerate(sizes)]


def _check_rescale_device_against_oracle(est, ref, frames, batch):
    """est: product estimator in a device-sampling mode with stage outputs; ref: OracleRescaleEstimator(device_seed=...)."""
    want, snap = [], []
    for f3, f2 in frames:
        want.append(ref.scale_calculation(f3, f2)[0])
        snap.append(dict(valid=ref.last["valid"].copy(), ids=ref.last["flat"].ids.copy(), level=ref.last["flat"].height_level,
                         model=None if "model" not in ref.last else np.array(ref.last["model"]),
                         best_ic=ref.last.get("best_ic"), used=ref.last.get("used")))
    if batch:
        got, sd = est.scale_calculation_batch([f[0] for f in frames], [f[1] for f in frames], stage=True)
        lasts = [est.last] * len(frames)
        idx = list(range(len(frames)))
    else:
        est.stage_outputs = True
        got, lasts, idx = [], [], []
        for f3, f2 in frames:
            got.append(est.scale_calculation(f3, f2)[0])
            lasts.append(dict(est.last))
            idx.append(0)
    for i, (w, sn) in enumerate(zip(want, snap)):
        L, j = lasts[i], idx[i]
        assert np.array_equal(L["valid"][j], sn["valid"]), i                                  # graph.py:35
        ids = L["tris2"][j][(L["tri_flags"][j] & 4) != 0].reshape(-1)
        assert np.array_equal(ids, sn["ids"]), i                                              # rescale.py:101 (canonical rows both sides)
        np.testing.assert_allclose(L["height_level"][j], sn["level"], rtol=1e-9)
        if sn["model"] is not None:
            assert int(L["status"][j]) == 0
            m_ref = sn["model"] if sn["model"][1] >= 0 else -sn["model"]
            np.testing.assert_allclose(L["model"][j], m_ref, rtol=1e-7, atol=1e-11)
            assert int(L["best_ic"][j]) == sn["best_ic"] and int(L["used"][j]) == sn["used"], i
        else:
            assert int(L["status"][j]) == 11
        assert abs(got[i] - w) <= 1e-9 * abs(w), (i, got[i], w)
    assert list(est.scale_queue) == pytest.approx(list(ref.scale_queue), rel=1e-9)


def _device_count():
    from mvoscalerecovery_amd import _lib
    return int(_lib.load().mvosr_device_count())
