"""The CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Runs without a GPU."""
import os

import numpy as np
import pytest

from conftest import load_json, load_npz
from oracle import scale_oracle as so


# ---------------------------------------------------------------- known answers
def test_kat_ctor_and_remap():
    kat = load_json("kat.json")
    assert kat["ctor"]["camera_pitch"] == so.CAMERA_PITCH
    assert kat["ctor"]["vanish"] == so.VANISH
    assert kat["ctor"]["default_window"] == 6
    est = so.OracleScaleEstimator(1.75)
    assert est.window_size == kat["ctor"]["default_window"]
    p = np.array(kat["remap"]["in"])
    assert np.array_equal(so.remap(p), np.array(kat["remap"]["out"]))
    assert np.array_equal(p, np.array(kat["remap"]["in"]))        # oracle does not mutate


def test_kat_initial_estimation():
    kat = load_json("kat.json")
    est = so.OracleScaleEstimator(1.75)
    assert est.initial_estimation(np.array(kat["initial_estimation"]["in"])) == kat["initial_estimation"]["out"]
    assert len(est.motion_queue) == 1


def test_kat_check_triangle_flags():
    """The quirk: the (0,2) pair marks vertices 0 and 1."""
    kat = load_json("kat.json")
    for case in kat["check_triangle"]:
        v = np.array(case["v"])
        d = np.array(case["d"])
        counters = so.outlier_votes(v, d, np.array([[0, 1, 2]]))
        flag = counters == 0          # start 1, flagged -> 0, not flagged -> 2
        assert flag.tolist() == case["flag"], case


def test_fixed_check_triangle_mode_is_order_invariant():
    """check_triangle="fixed" (the declared deviation of SURVEY.md §8 f1: `b > 0` marks vertices 0 and 2): per pair test
    the two vertices of the pair are flagged; on the reference's own known-answer cases the fixed pattern differs from
    the reference's exactly where b > 0; and a frame's result does not change when rows are rotated, reflected or
    permuted — in both flavours of the oracle — while the reference's pattern does change."""
    from oracle import scale_oracle_loops as sl
    from mvoscalerecovery_amd import synth
    kat = load_json("kat.json")
    for case in kat["check_triangle"]:
        v, d = np.array(case["v"], dtype=float), np.array(case["d"], dtype=float)
        a, b, c = (v[0] - v[1]) * (d[0] - d[1]) > 0, (v[0] - v[2]) * (d[0] - d[2]) > 0, (v[1] - v[2]) * (d[1] - d[2]) > 0
        want = [a or b, a or c, b or c]
        assert (so.outlier_votes(v, d, np.array([[0, 1, 2]]), "fixed") == 0).tolist() == want
        assert sl.check_triangle(v, d, "fixed").tolist() == want
        if not b:
            assert want == case["flag"]
    rng = np.random.default_rng(5)
    f3, f2 = synth.synth_frame(2, 700, base_seed=77, upper_fraction=0.1)
    base = so.frame_raw_scale(f3, f2, 1.75, check_triangle="fixed")
    low = so.lower_mask(f2)
    t1 = so.delaunay(f2[low])
    differs = False
    for trial in range(3):
        t = np.stack([np.roll(r, rng.integers(3)) if rng.integers(2) else r[::-1] for r in t1])[rng.permutation(len(t1))]
        r = so.frame_raw_scale(f3, f2, 1.75, tri1=t, check_triangle="fixed")
        assert np.array_equal(r.counters, base.counters) and r.raw_scale == base.raw_scale and r.height_level == base.height_level
        assert np.array_equal(r.tri1, base.tri1) and np.array_equal(r.tri2, base.tri2)          # canonical rows
        differs |= not np.array_equal(so.frame_raw_scale(f3, f2, 1.75, tri1=t).counters, so.frame_raw_scale(f3, f2, 1.75).counters)
    assert differs                                                    # the reference's pattern IS order-dependent
    lo = sl.frame_raw_scale(f3, f2, 1.75, check_triangle_mode="fixed")
    assert lo[0] == base.raw_scale and lo[2] == base.height_level and np.array_equal(lo[3], base.counters)
    est = so.OracleScaleEstimator(1.75, window_size=5, check_triangle="fixed")
    assert est.scale_calculation(f3, f2)[0] == base.raw_scale
    with pytest.raises(ValueError):
        so.OracleScaleEstimator(1.75, check_triangle="other")


def test_kat_three_triangles():
    kat = load_json("kat.json")["three_triangles"]
    sel = so.tri_select(np.array(kat["pts"]), np.array(kat["tri"]))
    assert sel.selected_ids.tolist() == kat["ids"]
    assert sel.height_level == kat["height_level"]
    np.testing.assert_allclose(sel.normals[0], [0, 1.25, 0], atol=1e-12)
    np.testing.assert_allclose(sel.pitch_deg, [-90, 0, 90], atol=1e-9)


def test_kat_scale_filtering():
    kat = load_json("kat.json")["scale_filtering"]
    for w, rec in kat.items():
        out, q = so.window_median(rec["in"], int(w))
        assert out.tolist() == rec["out"]
        assert len(q) == min(int(w), len(rec["in"]))
        # carried-in queue == one long run
        out2a, q2 = so.window_median(rec["in"][:4], int(w))
        out2b, _ = so.window_median(rec["in"][4:], int(w), q2)
        assert np.concatenate([out2a, out2b]).tolist() == rec["out"]


# ---------------------------------------------------------------- road model edge cases
def test_road_cases():
    cases = load_json("road_cases.json")
    assert len(cases) >= 50
    seen = set()
    for name, c in cases.items():
        rm = so.road_model(np.array(c["y"], dtype=np.float64), c["height_level"])
        seen.add(rm.status)
        if c["raises"] is not None:
            assert c["raises"] == "IndexError", name
            assert rm.status in (so.ST_ERR_LEFT, so.ST_ERR_RIGHT), name
            with pytest.raises(IndexError):
                so.raise_for_status(rm.status)
        else:
            assert rm.status in (so.ST_MODE, so.ST_RIGHT, so.ST_MEDIAN, so.ST_LEVEL), name
            assert rm.height == c["height"], (name, rm.height, c["height"])
    assert {so.ST_MODE, so.ST_RIGHT, so.ST_MEDIAN, so.ST_LEVEL, so.ST_ERR_LEFT, so.ST_ERR_RIGHT} <= seen


def test_road_fuzz_golden():
    """800 generated lists (synth.road_fuzz_list) against what the reference returned for them
    (tests/golden/road_fuzz.npz, made by make_golden.py --sets road_fuzz): heights bit-equal, IndexError
    where the reference raised."""
    from mvoscalerecovery_amd import synth
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "road_fuzz.npz"))
    seed = int(z["seed"])
    statuses = set()
    for i in range(len(z["heights"])):
        y = synth.road_fuzz_list(i, seed)
        assert len(y) == z["sizes"][i] and float(np.sum(y)) == z["sums"][i], i
        rm = so.road_model(y, 0.5 + 0.001 * i)
        statuses.add(rm.status)
        if z["raises"][i]:
            assert rm.status in (so.ST_ERR_LEFT, so.ST_ERR_RIGHT), i
        else:
            assert rm.status <= so.ST_LEVEL and rm.height == z["heights"][i], (i, rm.status, rm.height, z["heights"][i])
    assert {so.ST_MODE, so.ST_RIGHT, so.ST_MEDIAN, so.ST_LEVEL, so.ST_ERR_LEFT, so.ST_ERR_RIGHT} <= statuses


def test_frame_fuzz_golden():
    """400 small adversarial frames (synth.fuzz_frame: duplicate pixels, tied depths, few pixel rows,
    extreme depth scales, ground-only / wall scenes, a handful of points, negative heights) against what
    the reference's scale_calculation returned or raised for them (tests/golden/frame_fuzz.npz)."""
    from mvoscalerecovery_amd import synth
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "frame_fuzz.npz"))
    names = list(z["exception_names"])
    assert {"IndexError", "QhullError"} <= set(names)
    for i in range(len(z["scale"])):
        f3, f2 = synth.fuzz_frame(i, int(z["seed"]))
        assert float(np.sum(f3)) + float(np.sum(f2)) == z["sums"][i], i
        est = so.OracleScaleEstimator(1.75, window_size=5)
        want_exc = names[z["raised"][i] - 1] if z["raised"][i] else None
        try:
            s, sd = est.scale_calculation(f3.copy(), f2.copy())
            got_exc = None
        except Exception as exc:  # noqa: BLE001 - the type is what is compared
            got_exc = type(exc).__name__
        assert got_exc == want_exc, (i, got_exc, want_exc)
        if want_exc is None:
            assert sd == z["std"][i], i
            assert (np.isnan(s) and np.isnan(z["scale"][i])) or s == z["scale"][i], (i, s, z["scale"][i])


def test_histogram_restatement_equals_numpy():
    rng = np.random.default_rng(0)
    for _ in range(20):
        y = np.concatenate([rng.uniform(-1, 18, 500), np.round(rng.uniform(0, 17, 200), 1), so.BIN_EDGES])
        ref, _ = np.histogram(y, bins=so.BIN_EDGES)
        assert np.array_equal(so.histogram_170(y), ref)


def test_frame_cases():
    cases = load_json("frame_cases.json")
    for name, c in cases.items():
        f3, f2 = np.array(c["f3"]), np.array(c["f2"])
        est = so.OracleScaleEstimator(1.75, window_size=5)
        if c["raises"] is not None:
            with pytest.raises(Exception) as ei:
                est.scale_calculation(f3, f2)
            assert type(ei.value).__name__ == c["raises"], name
            continue
        s, sd = est.scale_calculation(f3, f2)
        assert sd == c["std"], name
        if np.isnan(c["scale"]):
            assert np.isnan(s), name
        else:
            assert s == c["scale"], name
        if np.isnan(c["height_level"]):
            assert np.isnan(est.height_level)
        else:
            assert est.height_level == c["height_level"]
        if c["n_flat"] is None:
            assert est.last.status == so.ST_NO_FLAT
        else:
            assert len(est.flat_feature) == c["n_flat"]


# ---------------------------------------------------------------- per-stage goldens
def test_stage_goldens(stages):
    assert len(stages) >= 16
    for g in stages:
        f3 = so.remap(g["f3"])
        assert np.array_equal(f3[:, 1:3], g["remapped_yz"])
        low = so.lower_mask(g["f2"])
        f3l, f2l = f3[low], g["f2"][low]
        # SciPy here must give the triangulation the fixture holds (same image => same Qhull)
        assert np.array_equal(so.delaunay(f2l), g["tri1"])
        counters = so.outlier_votes(f2l[:, 1], f3l[:, 2], g["tri1"])
        valid = so.votes_valid(counters)
        assert np.array_equal(valid, g["valid"])
        f3v = f3l[valid]
        assert np.array_equal(so.delaunay(f2l[valid]), g["tri2"])
        sel = so.tri_select(f3v, g["tri2"])
        assert np.array_equal(sel.selected_ids, g["selected_ids"])
        assert sel.height_level == float(g["height_level"])
        if g["per_triangle"]:
            np.testing.assert_array_equal(sel.normals_len, g["normals_len"])
            np.testing.assert_array_equal(-sel.normals[:, 1] / sel.normals_len, g["neg_unit_ny"])
            np.testing.assert_array_equal(sel.pitch_deg, g["pitch_rad"] * 180 / np.pi)
            np.testing.assert_array_equal(sel.heights, g["tri_heights"])
        rm = so.road_model(f3v[sel.selected_ids, 1], sel.height_level)
        assert np.array_equal(rm.hist_raw, g["hist_raw"])
        assert rm.n_kept == int(g["n_kept"])
        assert rm.n_modes == int(g["n_modes"])
        assert rm.height == float(g["height"])
        if "skew" in g:
            assert rm.skew == float(g["skew"])
            assert np.array_equal(so.local_min_flags(rm.hist), g["local_min"])
        res = so.frame_raw_scale(g["f3"], g["f2"], g["abs_ref"], g["tri1"], g["tri2"])
        assert res.raw_scale == float(g["scale_first_call"])
        assert res.std == float(g["std"])


def test_tri_row_order_does_not_matter_vertex_order_does(stages):
    """permuting triangle rows changes nothing; permuting vertices inside
    triangles changes the vote."""
    g = stages[0]
    f3 = so.remap(g["f3"])
    low = so.lower_mask(g["f2"])
    v, z = g["f2"][low, 1], f3[low, 2]
    base = so.outlier_votes(v, z, g["tri1"])
    rng = np.random.default_rng(1)
    assert np.array_equal(base, so.outlier_votes(v, z, g["tri1"][rng.permutation(len(g["tri1"]))]))
    rolled = np.roll(g["tri1"], 1, axis=1)
    assert not np.array_equal(base, so.outlier_votes(v, z, rolled))


def test_dense_golden():
    from mvoscalerecovery_amd import synth
    z, meta = load_npz("dense.npz")
    f3, f2 = synth.synth_frame(meta["frame_idx"], meta["n"], base_seed=meta["seed"])
    assert synth.checksum(f3, f2) == meta["crc"]
    tri1, tri2 = z["tri1"].astype(np.int32), z["tri2"].astype(np.int32)
    res = so.frame_raw_scale(f3, f2, meta["abs_ref"], tri1, tri2)
    assert np.array_equal(res.valid, z["valid"])
    assert np.array_equal(res.sel.selected_ids, z["selected_ids"])
    assert res.height_level == float(z["height_level"])
    assert res.height == float(z["height"])
    assert np.array_equal(res.road.hist_raw, z["hist_raw"])


# ---------------------------------------------------------------- whole-sequence goldens
def _run_oracle_sequence(name, **est_kw):
    from mvoscalerecovery_amd import synth, offline
    z, meta = load_npz(name)
    data = synth.synth_sequence_dict(meta["n_frames"], base_seed=meta["seed"], **meta["kw"])
    crc = 0
    for a3, a2 in zip(data["feature3ds"], data["feature2ds"]):
        if len(a3):
            crc = synth.checksum(np.array([crc], dtype=np.int64), a3, a2)
    assert crc == meta["crc"], "synthetic sequence drifted from the fixture"
    est = so.OracleScaleEstimator(meta["abs_ref"], window_size=meta["window"], **est_kw)
    raws = []
    real = est.scale_filtering
    est.scale_filtering = lambda s: (raws.append(s), real(s))[1]
    res = offline.run_sequence(data, est)
    return z, res, np.array(raws)


def test_seq200_golden():
    """Config C1 shape: 200 frames through the main.py / main_offline.py loop."""
    z, res, raws = _run_oracle_sequence("seq200.npz")
    assert np.array_equal(res["kinds"], z["kinds"])
    np.testing.assert_array_equal(raws, z["raw_scales"])
    np.testing.assert_array_equal(res["scales"], z["scales"])
    np.testing.assert_array_equal(res["error"], z["error"])
    np.testing.assert_array_equal(res["pitchs"], z["pitchs"])


def test_seq4541_golden():
    """Config C3 shape: 4541-frame main_offline replay (ragged N, not-moving and too-few frames)."""
    z, res, raws = _run_oracle_sequence("seq4541.npz")
    assert np.array_equal(res["kinds"], z["kinds"])
    np.testing.assert_array_equal(raws, z["raw_scales"])
    np.testing.assert_array_equal(res["scales"], z["scales"])
    np.testing.assert_array_equal(res["error"], z["error"])


# ---------------------------------------------------------------- the declared deviation, pinned to the reference's code
def test_fixed_mode_stage_goldens_from_the_patched_reference():
    """check_triangle="fixed" against the REFERENCE run with that one line patched (tests/golden/make_golden.py:
    fixed_reference — /root/reference/src/scale_calculator.py:113-115 marking vertices 0 and 2, rows in canonical form):
    vote masks, selected ids, height_level, histogram, modes, height and scale of the 20 stage frames."""
    from mvoscalerecovery_amd import synth
    z, meta = load_npz("stages_fixed.npz")
    for k, fr in enumerate(meta["frames"]):
        f3, f2 = synth.synth_frame(fr["frame_idx"], fr["n"], base_seed=fr["seed"], upper_fraction=fr["upper_fraction"])
        assert synth.checksum(f3, f2) == fr["crc"]
        g = {name[len("f%d_" % k):]: z[name] for name in z.files if name.startswith("f%d_" % k)}
        r = so.frame_raw_scale(f3, f2, meta["abs_ref"], check_triangle="fixed")
        assert np.array_equal(r.valid, g["valid"]), k
        assert np.array_equal(r.sel.selected_ids, g["selected_ids"]), k
        assert r.height_level == float(g["height_level"]) and r.height == float(g["height"]), k
        assert r.raw_scale == float(g["scale_first_call"]) and r.std == float(g["std"]), k
        assert np.array_equal(r.road.hist_raw, g["hist_raw"]) and r.road.n_kept == int(g["n_kept"]) and r.road.n_modes == int(g["n_modes"])
        if "skew" in g:
            assert r.road.skew == float(g["skew"])


def test_fixed_mode_seq4541_from_the_patched_reference():
    """Config C3's 4541-frame sequence in the fixed mode: every raw and filtered scale equals the patched reference's."""
    z, res, raws = _run_oracle_sequence("seq4541_fixed.npz", check_triangle="fixed")
    assert np.array_equal(res["kinds"], z["kinds"])
    np.testing.assert_array_equal(raws, z["raw_scales"])
    np.testing.assert_array_equal(res["scales"], z["scales"])
    np.testing.assert_array_equal(res["error"], z["error"])
    ref, _ = load_npz("seq4541.npz")
    agree = float(np.mean(z["raw_scales"] == ref["raw_scales"]))
    assert 0.90 < agree < 0.97, agree            # the deviation itself, reference vs patched reference (profiles/fixed_mode_accuracy.md)


def test_fixed_mode_frame_fuzz_from_the_patched_reference():
    """The 400 adversarial frames: same return value or exception type as the patched reference."""
    from mvoscalerecovery_amd import synth
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "frame_fuzz_fixed.npz"))
    names = list(z["exception_names"])
    for i in range(len(z["scale"])):
        f3, f2 = synth.fuzz_frame(i, int(z["seed"]))
        assert float(np.sum(f3)) + float(np.sum(f2)) == z["sums"][i], i
        est = so.OracleScaleEstimator(1.75, window_size=5, check_triangle="fixed")
        want_exc = names[z["raised"][i] - 1] if z["raised"][i] else None
        try:
            s, sd = est.scale_calculation(f3.copy(), f2.copy())
            got_exc = None
        except Exception as exc:  # noqa: BLE001 - the type is what is compared
            got_exc = type(exc).__name__
        assert got_exc == want_exc, (i, got_exc, want_exc)
        if want_exc is None:
            assert sd == z["std"][i], i
            assert (np.isnan(s) and np.isnan(z["scale"][i])) or s == z["scale"][i], (i, s, z["scale"][i])


def test_level_at_zero_family_golden():
    """Fuzz kind 10 (steep triangles of both signs, height_level within 1e-6 of zero, flat triangles within 1e-13 of it)
    against the reference's own stage methods (tests/golden/level_zero.npz; level_zero_fixed.npz for the patched
    reference): the level to the last bit, the selected ids."""
    from mvoscalerecovery_amd import synth
    for name, mode in (("level_zero.npz", "reference"), ("level_zero_fixed.npz", "fixed")):
        z = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
        for k in range(int(z["count"])):
            f3, f2 = synth.fuzz_frame(int(z["first"]) + k, int(z["seed"]))
            assert float(np.sum(f3)) + float(np.sum(f2)) == float(z["f%d_sum" % k]), k
            r = so.frame_raw_scale(f3, f2, 1.75, check_triangle=mode)
            assert np.array_equal(r.valid, z["f%d_valid" % k]), (name, k)
            assert r.height_level == float(z["f%d_height_level" % k]), (name, k, r.height_level)
            assert abs(r.height_level) < 1e-6
            assert np.array_equal(r.sel.selected_ids, z["f%d_selected_ids" % k]), (name, k)
            h, flat = r.sel.heights, r.sel.valid_pitch
            assert (h[~flat] > 0).any() and (h[~flat] < 0).any() and (np.abs(h[flat] - r.height_level) < 1e-12).sum() > 20


def test_too_few_lower_features_branch():
    """scale_calculator.py:263-270: exactly three features below the vanishing row -> no second triangulation,
    the scale is absolute_reference / (height_level of an earlier frame), std 100; AttributeError on a fresh estimator."""
    from mvoscalerecovery_amd import synth
    g = load_json("too_few.json")
    frames = synth.too_few_sequence(g["seed"], g["n_frames"])
    assert synth.checksum(*[a for fr in frames for a in fr]) == g["crc"]
    assert g["first_frame_raises"] == "AttributeError"
    with pytest.raises(AttributeError):
        so.OracleScaleEstimator(g["abs_ref"], window_size=g["window"]).scale_calculation(*frames[0])
    est = so.OracleScaleEstimator(g["abs_ref"], window_size=g["window"])
    for k, (f3, f2) in enumerate(frames[1:]):
        s, sd = est.scale_calculation(f3, f2)
        assert s == g["scales"][k] and sd == g["stds"][k], k
        assert est.height_level == g["height_levels"][k], k
        assert (est.flat_feature is None) == g["flat_none"][k], k
    assert 100.0 in g["stds"]


def test_loop_faithful_flavour_against_the_same_goldens(stages):
    """oracle/scale_oracle_loops.py (per-triangle Python loops, as the reference is written — the "reference-shaped"
    CPU baseline of bench.py) against the reference's stage goldens and a slice of the frame fuzz set."""
    from oracle import scale_oracle_loops as sl
    from mvoscalerecovery_amd import synth
    for g in stages[4:10]:
        raw, status, level, counters, selected = sl.frame_raw_scale(g["f3"], g["f2"], g["abs_ref"], g["tri1"], g["tri2"])
        assert np.array_equal(counters >= 0, g["valid"])
        assert np.array_equal(selected, g["selected_ids"])
        assert level == float(g["height_level"])
        assert raw == float(g["scale_first_call"])
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "frame_fuzz.npz"))
    names = list(z["exception_names"])
    checked = 0
    for i in range(0, len(z["scale"]), 7):
        if z["raised"][i] and names[z["raised"][i] - 1] in ("QhullError", "ValueError", "AttributeError"):
            continue
        f3, f2 = synth.fuzz_frame(i, int(z["seed"]))
        raw, status, level, _, _ = sl.frame_raw_scale(f3, f2, 1.75)
        if z["raised"][i]:
            assert status in (so.ST_ERR_LEFT, so.ST_ERR_RIGHT, so.ST_ERR_SINGULAR), i
        else:
            assert (np.isnan(raw) and np.isnan(z["scale"][i])) or raw == z["scale"][i], (i, raw, z["scale"][i])
        checked += 1
    assert checked >= 40


def test_road_long_lists():
    """Lists longer than np.add.reduce's 8192-element buffer: the oracle calls NumPy itself, so this pins the fixture."""
    from mvoscalerecovery_amd import synth
    g = load_json("road_long.json")
    for k, c in enumerate(g["cases"]):
        y = synth.road_long_list(k, g["seed"])
        assert len(y) == c["n"] and float(np.sum(y)) == c["sum"]
        rm = so.road_model(y, 0.7)
        assert rm.height == c["height"] and rm.skew == c["skew"], (k, rm.height, rm.skew)


def test_driver_loop_against_the_references_own_main_offline(tmp_path):
    """tests/golden/seq200_main_offline.npz holds what /root/reference/src/main_offline.py ITSELF wrote (scales.txt,
    path.txt) for the synthetic 200-frame dict: the build's driver loop (offline.run_sequence + save_outputs) around
    the oracle estimator must write the same files — gates, repeat-previous-scale, scales[1:], pose integration."""
    import zlib
    from mvoscalerecovery_amd import offline, synth
    z, meta = load_npz("seq200_main_offline.npz")
    data = synth.synth_sequence_dict(meta["n_frames"], base_seed=meta["seed"], **meta["kw"])
    res = offline.run_sequence(data, so.OracleScaleEstimator(meta["abs_ref"], window_size=meta["window"]))
    np.testing.assert_array_equal(res["scales"], z["scales"])
    poses = offline.save_outputs(str(tmp_path) + "/synth_result_", ".golden", res["scales"], data["motions"])
    np.testing.assert_array_equal(np.loadtxt(str(tmp_path) + "/synth_result_path.txt.golden"), z["path"])
    assert zlib.crc32(open(str(tmp_path) + "/synth_result_scales.txt.golden").read().encode()) == int(z["scales_txt_crc"])
    # the same golden as the spy-based seq200 fixture (which drives the reference estimator through offline.run_sequence)
    z2, _ = load_npz("seq200.npz")
    np.testing.assert_array_equal(z["scales"], z2["scales"])
