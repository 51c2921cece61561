"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors of the reference —
the kernels one by one and fused, through the C ABI: stage goldens of the reference, dense / tiled variants, road model, window median, seeded batches, guards.  Needs a real MI355X:  python -m pytest tests -m gpu

Constructions say which path they mean: ``triangulation="scipy"`` is the host-SciPy baseline every device path is compared with; a
construction without the keyword IS the shipped default (triangulation "gpu" with the reference's vote)."""
import os

import numpy as np
import pytest

from conftest import load_json, load_npz
from gpu_helpers import _assert_frame_equal, _oracle, _oracle_frames, _pack, _pack_tiled, _run_fused

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- library surface
def test_library_loaded_and_device(gpu):
    from mvoscalerecovery_amd import _lib
    lib = _lib.load()
    assert lib.mvosr_abi_version() == _lib.ABI_VERSION
    assert lib.mvosr_device_count() >= 1
    assert gpu.n_cu >= 200
    assert lib.mvosr_max_lds_features() >= 6000
    assert lib.mvosr_lds_bytes(2000) <= 160 * 1024 // 3      # three 2000-feature frames per CU


def test_status_codes_match_oracle():
    from mvoscalerecovery_amd import constants as K
    so = _oracle()
    for name in ("ST_MODE", "ST_RIGHT", "ST_MEDIAN", "ST_LEVEL", "ST_NO_FLAT", "ST_ERR_LEFT", "ST_ERR_RIGHT",
                 "ST_ERR_SINGULAR", "ST_ERR_MASK", "ST_ERR_EMPTY"):
        assert getattr(K, name) == getattr(so, name)


# ---------------------------------------------------------------- golden vectors of the reference
def test_stage_goldens_fused(gpu, stages):
    """Every per-stage golden of the reference, through the fused kernel with stage outputs."""
    so = _oracle()
    frames = [(g["f3"], g["f2"]) for g in stages]
    ores = [so.frame_raw_scale(g["f3"], g["f2"], g["abs_ref"], g["tri1"], g["tri2"]) for g in stages]
    for waves in (0, 4, 8, 16):
        pf, res = _run_fused(gpu, frames, ores, waves=waves, per_triangle=True)
        for f, g in enumerate(stages):
            sl = pf.frame_slice(f)
            assert np.array_equal(res["vote_counters"][sl] >= 0, g["valid"])
            nv = int(g["valid"].sum())
            assert np.array_equal(np.nonzero(res["selected"][sl][:nv])[0], g["selected_ids"])
            assert res["height"][f] == float(g["height"])
            assert res["raw_scale"][f] == float(g["scale_first_call"])
            assert np.array_equal(res["hist"][f, 0], g["hist_raw"])
            assert res["counts"][f, 4] == int(g["n_kept"])
            assert res["counts"][f, 5] == int(g["n_modes"])
            assert res["height_level"][f] == float(g["height_level"])
            if "skew" in g:
                np.testing.assert_allclose(res["stats"][f, 2], float(g["skew"]), rtol=1e-12)
            if g["per_triangle"]:
                t = slice(int(pf.tri2_off[f]), int(pf.tri2_off[f + 1]))
                # mean height of 3 vertices: same three additions and one division -> bit-exact
                assert np.array_equal(res["tri_heights"][t], g["tri_heights"])
                n = res["tri_normals"][t]
                nlen = np.sqrt((n * n).sum(1))
                # LU solve vs LAPACK inverse: agreement to rounding amplified by the triangle's
                # conditioning (points ~10 m away, sides ~0.1 m)
                np.testing.assert_allclose(nlen, g["normals_len"], rtol=1e-9)
                pitch = g["pitch_rad"] * 180 / np.pi
                np.testing.assert_allclose(res["tri_pitch_deg"][t], pitch, rtol=0, atol=1e-6)
                assert np.array_equal(res["tri_pitch_deg"][t] < -80, pitch < -80)
            _assert_frame_equal(so, ores[f], res, pf, f)


def test_one_wave_per_frame_small_frames(gpu, stages):
    """The one-frame-per-wavefront variant on the frames small enough for it."""
    so = _oracle()
    small = [g for g in stages if g["f3"].shape[0] <= 512]      # capacity of the one-wave variant
    assert len(small) >= 5
    frames = [(g["f3"], g["f2"]) for g in small]
    ores = [so.frame_raw_scale(g["f3"], g["f2"], g["abs_ref"], g["tri1"], g["tri2"]) for g in small]
    pf, res = _run_fused(gpu, frames, ores, waves=1)
    for f, g in enumerate(small):
        assert res["height"][f] == float(g["height"])
        _assert_frame_equal(so, ores[f], res, pf, f)


def test_dense_golden_and_lds_refusal(gpu):
    """Config C5 shape, N=20000 (T1~40000): does not fit LDS in fp64.  The dense variant (planes in
    a global workspace, gathers through L2) must reproduce the reference's golden; the LDS-resident
    variant, when asked for explicitly, must refuse rather than corrupt."""
    from mvoscalerecovery_amd import _lib, synth
    so = _oracle()
    z, meta = load_npz("dense.npz")
    f3, f2 = synth.synth_frame(meta["frame_idx"], meta["n"], base_seed=meta["seed"])
    r = so.frame_raw_scale(f3, f2, meta["abs_ref"], z["tri1"].astype(np.int32), z["tri2"].astype(np.int32))
    with pytest.raises(_lib.MvosrLibraryError, match="LDS|features"):
        _run_fused(gpu, [(f3, f2)], [r], waves=8)
    pf, res = _run_fused(gpu, [(f3, f2)], [r], per_triangle=True)
    sl = pf.frame_slice(0)
    assert np.array_equal(res["vote_counters"][sl] >= 0, z["valid"])
    nv = int(z["valid"].sum())
    assert np.array_equal(np.nonzero(res["selected"][sl][:nv])[0], z["selected_ids"])
    assert res["height"][0] == float(z["height"])
    assert np.array_equal(res["hist"][0, 0], z["hist_raw"])
    _assert_frame_equal(so, r, res, pf, 0)
    # the same frame with the second triangulation numbered over the features (no compaction)
    pf_f, res_f = _run_fused(gpu, [(f3, f2)], [r], per_triangle=True, feature_ids=True)
    _assert_frame_equal(so, r, res_f, pf_f, 0)
    for k in ("raw_scale", "height", "status", "selected", "counts", "tri_pitch_deg", "tri_heights", "tri_normals", "hist"):
        assert np.array_equal(res[k], res_f[k], equal_nan=True), k
    # ... and a row that names a feature the vote dropped is an error, not a silent use of that feature
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    pf_b = _pack([(f3, f2)], [r.tri1], [r.tri2], [r.valid], feature_ids=True)
    pf_b.tri2[5, 1] = int(np.nonzero(~r.valid)[0][0])
    db = DeviceBatch(gpu, pf_b)
    out = DeviceOutputs(gpu, db)
    ScaleEngine(meta["abs_ref"], ctx=gpu).scale_batch(db, out)
    gpu.sync()
    assert out.get("status")[0] == so.ST_ERR_MASK
    out.free()
    db.free()


@pytest.mark.parametrize("n,count", [(500, 6), (7000, 4), (20000, 3), (40000, 1)])
def test_dense_seeded_batches(gpu, n, count):
    """40000 features: ~80000 triangles, more than 64 per thread (second flag word of phase_select).
    500 features: the LDS-resident kernel against the gather variant that feature-numbered rows select."""
    from mvoscalerecovery_amd import synth
    so = _oracle()
    frames = [synth.synth_frame(i, n, base_seed=9000 + n, upper_fraction=0.05 * (i % 2)) for i in range(count)]
    ores = _oracle_frames(frames)
    pf, res = _run_fused(gpu, frames, ores)
    for f in range(count):
        _assert_frame_equal(so, ores[f], res, pf, f)
    pf2, res2 = _run_fused(gpu, frames, ores, stage=False, hist=False)
    for k in ("raw_scale", "height", "status"):
        assert np.array_equal(res[k], res2[k], equal_nan=True)
    for f in range(count):
        _assert_frame_equal(so, ores[f], res2, pf2, f, check_stage=False)      # product mode: height_level to 1e-13
    pf4, res4 = _run_fused(gpu, frames, ores, stage=False, hist=False, feature_ids=True)
    for f in range(count):
        _assert_frame_equal(so, ores[f], res4, pf4, f, check_stage=False)
    # second triangulation renumbered over the features (no compaction in the kernel): same results
    pf3, res3 = _run_fused(gpu, frames, ores, feature_ids=True)
    assert pf3.tri2_ids == 1
    for f in range(count):
        _assert_frame_equal(so, ores[f], res3, pf3, f)
    for k in ("raw_scale", "height", "status", "selected", "vote_counters", "counts"):
        assert np.array_equal(res[k], res3[k], equal_nan=True), k


def test_dense_tiled_kernel(gpu):
    """The traffic-lean dense variant (tile index, two-tile LDS ring, per-vertex flat-height keys instead of a second
    sweep) against the oracle: statuses and raw scales exact, height_level to rounding (product mode), vote and
    pitch counts exact; a mixed batch with a small and an LDS-sized frame; results independent of the variant."""
    from mvoscalerecovery_amd import synth, constants as K
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    so = _oracle()
    sizes = (20000, 7000, 300, 2500, 21000, 9000)
    frames = [synth.synth_frame(i, n, base_seed=4242, upper_fraction=0.05 * (i % 2)) for i, n in enumerate(sizes)]
    ores = _oracle_frames(frames)
    pf = _pack_tiled(frames, ores)
    eng = ScaleEngine(1.75, ctx=gpu)
    db = DeviceBatch(gpu, pf)
    out = DeviceOutputs(gpu, db, counts=True)
    eng.scale_batch(db, out)
    gpu.sync()
    res = {k: out.get(k) for k in out.bufs}
    out.free()
    for f, r in enumerate(ores):
        assert res["status"][f] == r.status, (f, res["status"][f], r.status)
        assert res["raw_scale"][f] == r.raw_scale and res["height"][f] == r.height, f
        assert abs(res["height_level"][f] - r.height_level) <= 1e-13 * abs(r.height_level), f
        assert res["counts"][f, K.CNT_VALID] == int(r.valid.sum())
        assert res["counts"][f, K.CNT_TRI_PITCH] == int(r.sel.valid_pitch.sum())
        assert res["counts"][f, K.CNT_SELECTED] == len(r.sel.selected_ids)
        assert res["counts"][f, K.CNT_KEPT] == r.road.n_kept and res["counts"][f, K.CNT_MODES] == r.road.n_modes
    # stage outputs select the two-sweep kernel in EXACT mode on the same layout: identical results, and the level is
    # NumPy's pairwise sum over the rows in their ORIGINAL order (mvosr_batch.tri2_order): the reference's double
    out2 = DeviceOutputs(gpu, db, counts=True, stage=True)
    eng.scale_batch(db, out2)
    gpu.sync()
    res2 = {k: out2.get(k) for k in out2.bufs}
    out2.free()
    for k in ("raw_scale", "height", "status"):
        assert np.array_equal(res[k], res2[k], equal_nan=True), k
    for f, r in enumerate(ores):
        assert res2["height_level"][f] == r.height_level, f
    # two launches are bit-identical (fixed summation order, order-free atomics)
    out3 = DeviceOutputs(gpu, db, counts=True)
    eng.scale_batch(db, out3)
    gpu.sync()
    for k in ("raw_scale", "height", "height_level", "status", "counts"):
        assert np.array_equal(res[k], out3.get(k), equal_nan=True), k
    out3.free()
    # the far rows' vertices come from the packer's table (mvosr_batch.tile_far); without it the kernel gathers them
    # from the planes: the same results.  A table of the wrong length for a frame refuses that frame.
    assert pf.tile_far is not None and int(pf.tile_far_off[-1]) > 0 and int(pf.tile_far_off[-1]) % 9 == 0
    st = db.struct()
    far_ptr, far_off = st.tile_far, st.tile_far_off
    st.tile_far, st.tile_far_off = None, None
    out4 = DeviceOutputs(gpu, db, counts=True)
    eng.scale_batch(db, out4)
    gpu.sync()
    for k in ("raw_scale", "height", "height_level", "status", "counts"):
        assert np.array_equal(res[k], out4.get(k), equal_nan=True), k
    out4.free()
    bad_off = pf.tile_far_off.copy()
    bad_off[-1] += 9                                    # the last frame seems to have one far row more than its index says
    buf = gpu.to_device(bad_off, np.int64)
    st.tile_far, st.tile_far_off = far_ptr, buf.ptr
    out5 = DeviceOutputs(gpu, db, counts=True)
    eng.scale_batch(db, out5)
    gpu.sync()
    st5 = out5.get("status")
    out5.free(); buf.free()
    assert st5[-1] == K.ST_ERR_MASK and all(st5[f] == ores[f].status for f in range(len(ores) - 1))
    db.free()


def test_level_at_zero_family_through_every_kernel(gpu):
    """Fuzz kind 10 — steep triangles above and below y' = 0, height_level within 1e-6 of zero, every flat triangle within
    1e-13 of it: the case a guard band relative to |level| alone would miss — through the product (HOT + its exact pass)
    variants with 1/4/8/16 wavefronts, the EXACT variant, the tiled dense kernel and the device-triangulation path,
    against the reference's own selection (tests/golden/level_zero.npz, level_zero_fixed.npz): selected counts, the
    selected mask and the level to the last bit where the variant promises it."""
    from mvoscalerecovery_amd import synth, constants as K
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    so = _oracle()
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "level_zero.npz"))
    n = int(z["count"])
    frames = [synth.fuzz_frame(int(z["first"]) + k, int(z["seed"])) for k in range(n)]
    ores = _oracle_frames(frames)
    want_sel = [len(z["f%d_selected_ids" % k]) for k in range(n)]
    eng = ScaleEngine(1.75, ctx=gpu)
    for waves in (0, 1, 4, 8, 16):
        for stage in (False, True):
            pf, res = _run_fused(gpu, frames, ores, waves=waves, stage=stage, hist=False)
            for k in range(n):
                assert res["status"][k] == ores[k].status, (waves, stage, k)
                assert res["counts"][k, K.CNT_SELECTED] == want_sel[k], (waves, stage, k, res["counts"][k, K.CNT_SELECTED], want_sel[k])
                if stage:
                    assert res["height_level"][k] == float(z["f%d_height_level" % k]), (waves, k)
                    sel = np.nonzero(res["selected"][pf.frame_slice(k)][:int(ores[k].valid.sum())])[0]
                    assert np.array_equal(sel, z["f%d_selected_ids" % k]), (waves, k)
    pf = _pack_tiled(frames, ores)                       # the tiled dense kernel on the same frames
    db = DeviceBatch(gpu, pf)
    out = DeviceOutputs(gpu, db, counts=True)
    eng.scale_batch(db, out)
    gpu.sync()
    st, cnt = out.get("status"), out.get("counts")
    out.free(); db.free()
    for k in range(n):
        assert st[k] == ores[k].status and cnt[k, K.CNT_SELECTED] == want_sel[k], ("tiled", k)
    zf = np.load(os.path.join(os.path.dirname(__file__), "golden", "level_zero_fixed.npz"))
    est = ScaleEstimator(1.75, window_size=5, triangulation="gpu", mutate_inputs=False)
    raw, status, level, _ = est.raw_scale_batch([f[0] for f in frames], [f[1] for f in frames])
    for k in range(n):
        assert est.last_counts[k, K.CNT_SELECTED] == len(zf["f%d_selected_ids" % k]), ("gpu triangulation", k)
        assert status[k] == K.ST_ERR_LEFT


def test_dense_tiled_kernel_refuses_a_bad_index(gpu):
    """The tile index is input too: a non-monotone index, or one that walks a row after the tile of its smallest vertex
    has left the ring, gives MVOSR_ST_ERR_MASK for that frame (and only that frame)."""
    from mvoscalerecovery_amd import synth, constants as K
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    frames = [synth.synth_frame(i, 8000, base_seed=777) for i in range(3)]
    ores = _oracle_frames(frames)
    eng = ScaleEngine(1.75, ctx=gpu)
    for kind in ("shift", "order"):
        pf = _pack_tiled(frames, ores)
        a = int(pf.tile_base[1])
        if kind == "shift":
            pf.tile1_off[a + 3:a + 9] -= 700          # rows walked with a later tile than the one their smallest vertex is in
        else:
            pf.tile2_off[a + 5] = pf.tile2_off[a + 4] - 1
        db = DeviceBatch(gpu, pf)
        out = DeviceOutputs(gpu, db, counts=True)
        eng.scale_batch(db, out)
        gpu.sync()
        st = out.get("status")
        out.free(); db.free()
        assert st[0] == ores[0].status and st[2] == ores[2].status and st[1] == K.ST_ERR_MASK, (kind, st)


def test_fixed_vote_mode_kernels_equal_oracle(gpu):
    """check_triangle="fixed" (mvosr_params.vote_mode = MVOSR_VOTE_FIXED): the order-invariant vote through every kernel
    family (1/4/8/16 wavefronts per frame, the dense two-sweep and tiled kernels) against the oracle's fixed mode — and
    the result does not change when rows are rotated / permuted (what the reference's pattern is sensitive to)."""
    from mvoscalerecovery_amd import packing, synth
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    so = _oracle()
    rng = np.random.default_rng(11)
    frames = [synth.synth_frame(i, n, base_seed=8080, upper_fraction=0.1) for i, n in enumerate((250, 900, 2000, 2600))]
    ores = [so.frame_raw_scale(f3, f2, 1.75, check_triangle="fixed") for f3, f2 in frames]
    ref_mode = [so.frame_raw_scale(f3, f2, 1.75) for f3, f2 in frames]
    assert any(not np.array_equal(a.counters, b.counters) for a, b in zip(ores, ref_mode))     # the two patterns do differ
    eng = ScaleEngine(1.75, ctx=gpu, check_triangle="fixed")
    for shuffle in (False, True):
        t1s, t2s = [], []
        for r in ores:
            t1, t2 = r.tri1.copy(), r.tri2.copy()
            if shuffle:
                t1 = np.stack([np.roll(row, rng.integers(3)) for row in t1])[rng.permutation(len(t1))]
            t1s.append(t1.astype(np.int32)); t2s.append(t2.astype(np.int32))
        for waves in (0, 16):
            pf = _pack(frames, t1s, t2s, [r.valid for r in ores])
            db = DeviceBatch(gpu, pf)
            out = DeviceOutputs(gpu, db, counts=True, stage=True)
            eng.scale_batch(db, out, waves=waves)
            gpu.sync()
            raw, st, lvl, cnt = out.get("raw_scale"), out.get("status"), out.get("height_level"), out.get("vote_counters")
            for f, r in enumerate(ores):
                assert st[f] == r.status and raw[f] == r.raw_scale, (shuffle, waves, f, st[f], r.status)
                assert np.array_equal(cnt[pf.frame_slice(f)], r.counters), (shuffle, waves, f)
                if not shuffle:
                    assert lvl[f] == r.height_level, (waves, f)
            out.free(); db.free()
    # dense kernels (tiled and two-sweep) in fixed mode
    dframes = [synth.synth_frame(i, 7000, base_seed=31) for i in range(2)]
    dres = [so.frame_raw_scale(f3, f2, 1.75, check_triangle="fixed") for f3, f2 in dframes]
    pf = _pack_tiled(dframes, dres)
    db = DeviceBatch(gpu, pf)
    out = DeviceOutputs(gpu, db, counts=True)
    eng.scale_batch(db, out)
    gpu.sync()
    raw, st = out.get("raw_scale"), out.get("status")
    for f, r in enumerate(dres):
        assert st[f] == r.status and raw[f] == r.raw_scale, (f, st[f], r.status)
    out.free(); db.free()


def test_road_cases_kernel(gpu):
    """K3 alone on the reference's road-model edge cases (tests/golden/road_cases.json)."""
    from mvoscalerecovery_amd import packing
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    so = _oracle()
    cases = load_json("road_cases.json")
    names = sorted(cases)
    F = len(names)
    cnt = np.array([len(cases[n]["y"]) for n in names], dtype=np.int32)
    padded = (cnt.astype(np.int64) + 1) & ~np.int64(1)
    off = np.concatenate([[0], np.cumsum(padded)[:-1]]).astype(np.int64)
    y = np.zeros(int(padded.sum()), dtype=np.float64)
    for i, n in enumerate(names):
        y[off[i]:off[i] + cnt[i]] = cases[n]["y"]
    pf = packing.PackedFrames(F, off, cnt, y.copy(), y, y.copy(), y.copy(), y.copy(), [None] * F, max_feat=int(cnt.max()))
    hl = np.array([cases[n]["height_level"] for n in names])
    for waves in (1, 4, 8):
        eng = ScaleEngine(1.75, ctx=gpu)
        db = DeviceBatch(gpu, pf, with_tri2=False)
        out = DeviceOutputs(gpu, db, counts=True, hist=True)
        eng.road_model_batch(db, out, hl, waves=waves)
        st, h = out.get("status"), out.get("height")
        counts, hist = out.get("counts"), out.get("hist")
        for i, n in enumerate(names):
            rm = so.road_model(np.array(cases[n]["y"], dtype=np.float64), cases[n]["height_level"])
            assert st[i] == rm.status, (n, st[i], rm.status)
            assert np.array_equal(hist[i, 0], rm.hist_raw), n
            assert counts[i, 4] == rm.n_kept, n
            if cases[n]["raises"]:
                assert st[i] in (so.ST_ERR_LEFT, so.ST_ERR_RIGHT)
            else:
                assert h[i] == cases[n]["height"], (n, h[i], cases[n]["height"])
        out.free()
        db.free()


def test_road_fuzz_kernel(gpu):
    """K3 alone on the 800 generated lists of tests/golden/road_fuzz.npz: status and height against
    the reference's outputs, histogram / kept count / modes against the oracle."""
    from mvoscalerecovery_amd import packing, synth
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    so = _oracle()
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "road_fuzz.npz"))
    seed, F = int(z["seed"]), len(z["heights"])
    lists = [synth.road_fuzz_list(i, seed) for i in range(F)]
    cnt = np.array([len(y) for y in lists], dtype=np.int32)
    padded = (cnt.astype(np.int64) + 1) & ~np.int64(1)
    off = np.concatenate([[0], np.cumsum(padded)[:-1]]).astype(np.int64)
    y = np.zeros(int(padded.sum()), dtype=np.float64)
    for i, v in enumerate(lists):
        y[off[i]:off[i] + cnt[i]] = v
    pf = packing.PackedFrames(F, off, cnt, y.copy(), y, y.copy(), y.copy(), y.copy(), [None] * F, max_feat=int(cnt.max()))
    hl = 0.5 + 0.001 * np.arange(F)
    eng = ScaleEngine(1.75, ctx=gpu)
    db = DeviceBatch(gpu, pf, with_tri2=False)
    # with the statistics output the sums are made in NumPy's own order: mean / std / skew bit-equal
    out = DeviceOutputs(gpu, db, counts=True, hist=True)
    eng.road_model_batch(db, out, hl)
    st, h = out.get("status"), out.get("height")
    counts, hist, stats = out.get("counts"), out.get("hist"), out.get("stats")
    # without it the wavefront's own summation order decides, NumPy's only where the two could differ
    out2 = DeviceOutputs(gpu, db, counts=True)
    eng.road_model_batch(db, out2, hl)
    st2, h2 = out2.get("status"), out2.get("height")
    n_stats = 0
    for i in range(F):
        rm = so.road_model(lists[i], hl[i])
        assert st[i] == rm.status and st2[i] == rm.status, (i, st[i], st2[i], rm.status)
        assert np.array_equal(hist[i, 0], rm.hist_raw), i
        assert counts[i, 4] == rm.n_kept, i
        if z["raises"][i]:
            assert st[i] in (so.ST_ERR_LEFT, so.ST_ERR_RIGHT), i
        else:
            assert h[i] == z["heights"][i] and h2[i] == z["heights"][i], (i, h[i], h2[i], z["heights"][i])
        if rm.status in (so.ST_MODE, so.ST_RIGHT):
            n_stats += 1
            assert np.array_equal(stats[i, :3], [rm.mean, rm.std, rm.skew], equal_nan=True), (i, stats[i], rm.mean, rm.std, rm.skew)
    assert n_stats > 300
    out.free()
    out2.free()
    db.free()


def test_road_long_lists_kernel(gpu):
    """Lists of 8193-40000 values: mean / std / skew in NumPy's order means NumPy's 8192-element reduce chunks, each
    pairwise-summed (np_pairwise_sum_cold) — bit-equal skew, and the reference's height."""
    from mvoscalerecovery_amd import packing, synth
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    g = load_json("road_long.json")
    lists = [synth.road_long_list(k, g["seed"]) for k in range(len(g["cases"]))]
    F = len(lists)
    cnt = np.array([len(y) for y in lists], dtype=np.int32)
    padded = (cnt.astype(np.int64) + 1) & ~np.int64(1)
    off = np.concatenate([[0], np.cumsum(padded)[:-1]]).astype(np.int64)
    y = np.zeros(int(padded.sum()), dtype=np.float64)
    for i, v in enumerate(lists):
        y[off[i]:off[i] + cnt[i]] = v
    pf = packing.PackedFrames(F, off, cnt, y.copy(), y, y.copy(), y.copy(), y.copy(), [None] * F, max_feat=int(cnt.max()))
    eng = ScaleEngine(1.75, ctx=gpu)
    db = DeviceBatch(gpu, pf, with_tri2=False)
    out = DeviceOutputs(gpu, db, counts=True, hist=True)
    eng.road_model_batch(db, out, np.full(F, 0.7))
    h, stats = out.get("height"), out.get("stats")
    out.free(); db.free()
    for k, c in enumerate(g["cases"]):
        assert h[k] == c["height"], (k, h[k], c["height"])
        assert stats[k, 2] == c["skew"], (k, stats[k, 2], c["skew"])


def test_window_median_kernel(gpu):
    from mvoscalerecovery_amd.engine import ScaleEngine
    so = _oracle()
    eng = ScaleEngine(1.75, ctx=gpu)
    kat = load_json("kat.json")["scale_filtering"]
    for w, rec in kat.items():
        assert eng.window_median_host(rec["in"], int(w)).tolist() == rec["out"]
    rng = np.random.default_rng(3)
    raw = rng.uniform(0.5, 3.0, 5000)
    raw[100] = np.nan
    for w in (1, 2, 5, 6, 11, 64):
        q = list(rng.uniform(0.5, 3.0, min(w, 3)))
        want, _ = so.window_median(raw, w, q)
        got = eng.window_median_host(raw, w, q)
        assert np.array_equal(got, want, equal_nan=True)


# ---------------------------------------------------------------- seeded batches vs the oracle
@pytest.mark.parametrize("n,count,waves", [(2000, 24, 8), (2000, 8, 16), (700, 24, 4), (250, 48, 1), (4000, 6, 8), (6000, 3, 16)])
def test_seeded_batches(gpu, n, count, waves):
    from mvoscalerecovery_amd import synth
    so = _oracle()
    frames = [synth.synth_frame(i, n, base_seed=7000 + n, upper_fraction=0.05 * (i % 3)) for i in range(count)]
    ores = _oracle_frames(frames)
    pf, res = _run_fused(gpu, frames, ores, waves=waves)
    for f in range(count):
        _assert_frame_equal(so, ores[f], res, pf, f)
    # product path (no stage outputs, fast pitch test) gives the same results
    pf2, res2 = _run_fused(gpu, frames, ores, waves=waves, stage=False, hist=False)
    for k in ("raw_scale", "height", "status"):
        assert np.array_equal(res[k], res2[k], equal_nan=True)
    np.testing.assert_allclose(res2["height_level"], res["height_level"], rtol=1e-13)     # (fixed-order sum vs NumPy's order)
    for f in range(count):
        _assert_frame_equal(so, ores[f], res2, pf2, f, check_stage=False)
    assert np.array_equal(res["counts"], res2["counts"])


def test_ragged_batch_and_determinism(gpu):
    """Ragged frame sizes in one launch; two launches of the same batch are bit-identical."""
    from mvoscalerecovery_amd import synth
    so = _oracle()
    rng = np.random.default_rng(11)
    frames = [synth.synth_frame(i, int(rng.integers(101, 2300)), base_seed=31337, upper_fraction=0.1) for i in range(40)]
    ores = _oracle_frames(frames)
    pf, res = _run_fused(gpu, frames, ores)
    for f in range(len(frames)):
        _assert_frame_equal(so, ores[f], res, pf, f)
    _, res_b = _run_fused(gpu, frames, ores)
    for k in res:
        assert np.array_equal(res[k], res_b[k], equal_nan=True), k


def test_size_class_dispatch(gpu):
    """A ragged batch of >= 2048 frames in product mode is launched per size class (1 / 4 / 8 wavefronts per frame,
    each class with its own LDS request): every frame equals the oracle, the whole result equals the single-variant
    launch bit for bit, frames that need the EXACT pass (nothing selected) and frames the kernel never sweeps
    (no features) included; a sub-range launch (first_frame / n_launch) classifies only its own frames."""
    from mvoscalerecovery_amd import packing, synth, constants as K
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    rng = np.random.default_rng(5)
    sizes = [int(v) for v in rng.integers(60, 1500, 44)] + [320, 321, 1024, 1025]
    frames = [synth.synth_frame(i, n, base_seed=777, upper_fraction=0.05 * (i % 3)) for i, n in enumerate(sizes)]
    cases = load_json("frame_cases.json")
    for name in ("wall_none_selected", "all_flat_nan_level", "five_points"):
        if name in cases:
            frames.append((np.array(cases[name]["f3"]), np.array(cases[name]["f2"])))
    ores = _oracle_frames(frames)
    pool = len(frames)
    pf_pool = _pack(frames, [r.tri1 for r in ores], [r.tri2 for r in ores], [r.valid for r in ores])
    repeats = 48
    pf = packing.tile_frames(pf_pool, repeats)
    assert pf.n_frames >= 2048 and pf.max_feat > 1024 and int(pf.feat_cnt.min()) <= 320
    eng = ScaleEngine(1.75, ctx=gpu)
    db = DeviceBatch(gpu, pf)

    def run(waves, first=0, n=0):
        out = DeviceOutputs(gpu, db, counts=True)
        for k in ("raw_scale", "height", "height_level"):
            out.bufs[k].upload(np.full(pf.n_frames, -7.0))
        eng.scale_batch(db, out, waves=waves, first=first, count=n)
        gpu.sync()
        r = {k: out.get(k) for k in ("raw_scale", "height", "height_level", "status", "counts")}
        out.free()
        return r

    by_class = run(0)
    single = run(8)
    for k in ("raw_scale", "height", "status", "counts"):
        assert np.array_equal(by_class[k], single[k], equal_nan=True), k
    # (product mode: the level is each variant's own fixed-order sum unless it is the result — then it is exact)
    np.testing.assert_allclose(by_class["height_level"], single["height_level"], rtol=1e-13, equal_nan=True)
    for f in range(pool):
        for r in (0, repeats - 1):
            g = r * pool + f
            assert by_class["status"][g] == ores[f].status, (f, r)
            if ores[f].status not in K.ERROR_STATUSES:
                assert by_class["raw_scale"][g] == ores[f].raw_scale or (np.isnan(by_class["raw_scale"][g]) and np.isnan(ores[f].raw_scale)), (f, r)
    # the class counts are a hint: without it (every class launched over the whole range) and with one that
    # understates every class (the overflow goes through the EXACT pass) the results are the same
    st = db.struct()
    hint = list(st.size_hint)
    assert hint[3] != 0 and sum(hint[:3]) == pf.n_frames and int(st.min_feat) == int(pf.feat_cnt.min())
    for variant in ((0, 0, 0, 0), (hint[0] // 2, hint[1] // 3, 7, hint[3])):
        for k in range(4):
            st.size_hint[k] = variant[k]
        other = run(0)
        for k in ("raw_scale", "height", "status", "counts"):
            assert np.array_equal(by_class[k], other[k], equal_nan=True), (variant, k)
    for k in range(4):
        st.size_hint[k] = hint[k]
    # ... and so is min_feat: a header that overstates the smallest frame (the smallest class is then not launched)
    # loses no frame — the classification sends them through the EXACT pass
    true_min = int(st.min_feat)
    st.min_feat = 700
    other = run(0)
    for k in ("raw_scale", "height", "status", "counts"):
        assert np.array_equal(by_class[k], other[k], equal_nan=True), ("min_feat", k)
    st.min_feat = true_min
    # a sub-range: frames outside it keep the sentinel
    first, n = 3 * pool + 5, 2100
    part = run(0, first, n)
    assert np.array_equal(part["raw_scale"][first:first + n], by_class["raw_scale"][first:first + n], equal_nan=True)
    assert np.array_equal(part["status"][first:first + n], by_class["status"][first:first + n])
    assert np.all(part["raw_scale"][:first] == -7.0) and np.all(part["raw_scale"][first + n:] == -7.0)
    db.free()


def test_mask_mismatch_and_bad_index_are_flagged(gpu):
    from mvoscalerecovery_amd import constants as K, synth
    frames = [synth.synth_frame(i, 500, base_seed=5) for i in range(3)]
    ores = _oracle_frames(frames)
    # frame 1: tri2 built on a different mask; frame 2: a vertex id out of range
    bad_mask = ores[1].valid.copy()
    bad_mask[np.nonzero(bad_mask)[0][0]] = False
    ores[1].valid = bad_mask
    ores[2].tri2 = ores[2].tri2.copy()
    ores[2].tri2[5, 1] = 100000
    pf, res = _run_fused(gpu, frames, ores, stage=False, hist=False)
    assert res["status"][0] <= K.ST_LEVEL
    assert res["status"][1] == K.ST_ERR_MASK
    assert res["status"][2] == K.ST_ERR_MASK


def test_row_count_guards(gpu):
    """Input that is not a triangulation of the frame's points must not corrupt silently (the C ABI takes device
    pointers, so the host cannot check): more first-triangulation rows than a 16-bit vote counter can absorb
    (32765), or more second-triangulation rows than the per-thread flag words can name -> MVOSR_ST_ERR_MASK."""
    from mvoscalerecovery_amd import synth, constants as K
    so = _oracle()
    frames = [synth.synth_frame(i, 700, base_seed=616) for i in range(3)]
    ores = _oracle_frames(frames)
    base = [o.status for o in ores]
    # frame 1: tri1 repeated up to 33000 rows (every vertex far beyond +-32765 votes is possible)
    t1 = [o.tri1 for o in ores]
    t1[1] = np.tile(t1[1], (33000 // len(t1[1]) + 1, 1))[:33000]
    # frame 2: tri2 repeated beyond 64 flags x 256 threads (4 wavefronts per frame)
    t2 = [o.tri2 for o in ores]
    t2[2] = np.tile(t2[2], (17000 // len(t2[2]) + 1, 1))[:17000]
    from mvoscalerecovery_amd import packing
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames])
    packing.attach_tri1(pf, t1)
    packing.attach_tri2(pf, t2, [o.valid for o in ores])
    eng = ScaleEngine(1.75, ctx=gpu)
    for stage in (False, True):
        db = DeviceBatch(gpu, pf)
        out = DeviceOutputs(gpu, db, counts=True, stage=stage)
        eng.scale_batch(db, out, waves=4)
        gpu.sync()
        st = out.get("status")
        out.free(); db.free()
        assert st[0] == base[0] and st[1] == K.ST_ERR_MASK and st[2] == K.ST_ERR_MASK, (stage, st)


def test_frame_larger_than_batch_header(gpu):
    """mvosr_batch.max_feat sizes the launch's LDS and picks its variant: a frame with more features than it states
    (possible only through the C ABI, where the counts live in device memory) is refused with MVOSR_ST_ERR_MASK —
    the frames around it are processed as usual."""
    from mvoscalerecovery_amd import synth, constants as K
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    frames = [synth.synth_frame(i, n, base_seed=99) for i, n in enumerate((300, 700, 350))]
    ores = _oracle_frames(frames)
    pf = _pack(frames, [r.tri1 for r in ores], [r.tri2 for r in ores], [r.valid for r in ores])
    eng = ScaleEngine(1.75, ctx=gpu)
    db = DeviceBatch(gpu, pf)
    st = db.struct()
    st.max_feat = 400
    for k in range(4):
        st.size_hint[k] = 0
    out = DeviceOutputs(gpu, db, counts=True)
    eng.scale_batch(db, out)
    gpu.sync()
    status, raw = out.get("status"), out.get("raw_scale")
    out.free(); db.free()
    assert status[1] == K.ST_ERR_MASK and np.isnan(raw[1])
    for f in (0, 2):
        assert status[f] == ores[f].status and raw[f] == ores[f].raw_scale, f


def test_dense_fan_vote_counter_range(gpu):
    """A fan: one centre vertex in every row.  With 30000 rows its 16-bit counter holds the exact vote; with 40000 it
    would wrap into the neighbouring feature's half — the dense vote sees the update that crosses the end and flags
    the frame (MVOSR_ST_ERR_MASK) instead of returning a corrupted mask."""
    from mvoscalerecovery_amd import packing, constants as K
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    so = _oracle()
    eng = ScaleEngine(1.75, ctx=gpu, camera_pitch=0.0)
    for n_ring, expect_flag in ((30000, False), (40000, True)):
        n = n_ring + 1
        ang = np.linspace(0.0, 2 * np.pi, n_ring, endpoint=False)
        u = np.concatenate([[600.0], 600.0 + 300.0 * np.cos(ang)])
        v = np.concatenate([[300.0], 300.0 + 80.0 * np.sin(ang)])
        z = np.concatenate([[10.0], np.full(n_ring, 10.0)])          # equal depths: every product is 0 -> nobody is flagged
        f3 = np.stack([np.zeros(n), np.zeros(n), z], axis=1)
        f2 = np.stack([u, v], axis=1)
        ring = 1 + np.arange(n_ring)
        tri = np.stack([np.zeros(n_ring, dtype=np.int64), ring, 1 + (np.arange(n_ring) + 1) % n_ring], axis=1).astype(np.int32)
        pf = packing.pack_features([f3], [f2], vanish=-1.0)
        packing.attach_tri1(pf, [tri])
        db = DeviceBatch(gpu, pf, with_tri2=False)
        out = DeviceOutputs(gpu, db, counts=True, stage=True)
        eng.outlier_vote_batch(db, out)
        gpu.sync()
        st, counters = out.get("status")[0], out.get("vote_counters")[pf.frame_slice(0)]
        out.free(); db.free()
        if expect_flag:
            assert st == K.ST_ERR_MASK
        else:
            assert st == 0
            assert np.array_equal(counters, so.outlier_votes(v, z, tri))
            assert counters[0] == 1 + n_ring


def test_singular_triangle_status(gpu):
    """Two identical vertices in one triangle: LAPACK reports a zero pivot (LinAlgError, :229)."""
    from mvoscalerecovery_amd import constants as K, synth
    so = _oracle()
    f3, f2 = synth.synth_frame(0, 300, base_seed=8)
    r = so.frame_raw_scale(f3, f2, 1.75)
    # make two surviving features 3-D identical (pixels stay distinct so Delaunay is unchanged)
    low = np.nonzero(r.lower)[0][np.nonzero(r.valid)[0]]
    a, b = low[r.tri2[0, 0]], low[r.tri2[0, 1]]
    f3 = f3.copy()
    f3[b] = f3[a]
    r2 = so.frame_raw_scale(f3, f2, 1.75, r.tri1, r.tri2)
    if r2.status == so.ST_ERR_SINGULAR and np.array_equal(r2.valid, r.valid):
        pf, res = _run_fused(gpu, [(f3, f2)], [r2], stage=False, hist=False)
        assert res["status"][0] == K.ST_ERR_SINGULAR


def test_triangle_batch_golden(gpu):
    """Row a12: the legacy per-triangle batch vs what /root/reference/src/triangle_batch.py printed."""
    from mvoscalerecovery_amd import synth, triangle_batch
    from oracle import triangle_batch_oracle as tbo
    g = load_json("triangle_batch.json")
    pts = []
    for fr in g["frames"]:
        f3, f2 = synth.synth_frame(fr["frame_idx"], fr["n"], base_seed=fr["seed"])
        pts.append(np.stack([f2[:, 0], f2[:, 1], f3[:, 2]], axis=1))
    h, counts, status = triangle_batch.camera_heights(pts)
    assert np.all(status == 0)
    # same kept sets (integer counts exact); the means are sums in a different order than NumPy's
    for i, p in enumerate(pts):
        want, n_kept, n_clip = tbo.camera_height(p)
        assert (counts[i, 0], counts[i, 1]) == (n_kept, n_clip), i
        assert want == g["heights"][i]
    np.testing.assert_allclose(h, g["heights"], rtol=1e-13)


def test_road_norm_helpers(gpu):
    """Row a11: get_pitch_ransac / get_inliers on the GPU vs the oracle's restatement."""
    from mvoscalerecovery_amd import estimate_road_norm as ern
    from oracle import rescale_oracle as ro
    rng = np.random.default_rng(4)
    pts = np.stack([rng.uniform(-5, 5, 800), 0.8 + 0.02 * rng.uniform(-5, 5, 800) + rng.normal(0, 0.002, 800),
                    rng.uniform(4, 40, 800)], axis=1)
    pts[::9, 1] += rng.uniform(0.05, 0.5, pts[::9].shape[0])
    triples = np.stack([rng.choice(800, 3, replace=False) for _ in range(30)]).astype(np.int32)
    m, ic = ern.get_pitch_ransac(pts, 30, 0.005, triples=triples)
    m_ref, ic_ref, used = ro.run_ransac(pts, triples, 0.005)
    m_ref = m_ref if m_ref[1] >= 0 else -m_ref
    assert ic == ic_ref
    np.testing.assert_allclose(m, m_ref, rtol=1e-9, atol=1e-13)
    mask = ern.get_inliers(m_ref, pts, 0.01)
    assert np.array_equal(mask, np.abs(pts @ m_ref[:3] + m_ref[3]) < 0.01)


# ---------------------------------------------------------------- full-size properties (BASELINE configs[1])
def test_road_norm_helpers_reference_golden(gpu):
    """estimate_road_norm.py's helpers and ScaleEstimator.road_model_calculation_ransac against the reference's own
    outputs (tests/golden/road_norm.json, sample sequences replayed): motion helpers incl. the np.matrix return type,
    plane RANSAC -> height / pitch / inliers, line RANSAC -> model (up to the SVD's sign) and inlier count."""
    import json
    from mvoscalerecovery_amd import estimate_road_norm as ern
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "road_norm.json")))
    for c in g["motion"]:
        t = np.array(c["t"])
        n = ern.get_norm_svd(t)
        assert type(n).__name__ == c["norm_type"] and list(n.shape) == c["norm_shape"]
        np.testing.assert_allclose(np.asarray(n).reshape(-1), c["norm"], rtol=0, atol=1e-14)
        assert abs(ern.get_pitch_svd(t) - c["pitch_svd"]) <= 1e-14
        assert ern.get_pitch(t) == c["pitch"]
    est = ScaleEstimator(1.75, window_size=5, triangulation="scipy")
    for c in g["planes"]:
        pts = np.array(c["pts"])
        h, pitch, inl = est.road_model_calculation_ransac(pts, triples=np.array(c["triples"], dtype=np.int32))
        assert abs(h - c["height"]) <= 1e-9 * abs(c["height"]) and abs(pitch - c["pitch"]) <= 1e-9
        assert inl.shape[0] == c["n_inliers"] and abs(float(np.sum(inl)) - c["inlier_sum"]) <= 1e-9 * abs(c["inlier_sum"])
    for c in g["lines"]:
        m, ic = ern.get_pitch_line_ransac(np.array(c["xy"]), 40, 0.01, pairs=np.array(c["pairs"], dtype=np.int32))
        ref = np.array(c["model"])
        assert ic == c["best_ic"]
        assert min(np.abs(m - ref).max(), np.abs(m + ref).max()) <= 1e-9 and m[1] >= 0


def test_full_size_properties(gpu):
    """16 384 frames x 2000 features (the bench workload): results cannot be compared frame by frame
    with the CPU oracle in seconds, so size-independent properties are checked instead:
    (1) every tiled copy of a pool frame gives bit-identical outputs (a checksum of checksums);
    (2) the pool frames themselves equal the oracle; (3) permuting the ROWS of both triangulations
    changes nothing; (4) reversing the frame order reverses the outputs;
    (5) two launches are bit-identical; (6) the window median of the 16 384 raw scales equals the
    oracle's sliding median."""
    import zlib
    from mvoscalerecovery_amd import packing, synth
    from mvoscalerecovery_amd.engine import DeviceBatch, DeviceOutputs, ScaleEngine
    so = _oracle()
    pool, repeats = 32, 512
    frames = [synth.synth_frame(i, 2000, base_seed=2024) for i in range(pool)]
    ores = _oracle_frames(frames)
    eng = ScaleEngine(1.75, ctx=gpu)

    def run(pf):
        db = DeviceBatch(gpu, pf)
        out = DeviceOutputs(gpu, db, counts=True)
        eng.scale_batch(db, out)
        gpu.sync()
        r = {k: out.get(k) for k in ("raw_scale", "height", "height_level", "status", "counts")}
        out.free()
        db.free()
        return r

    pf_pool = _pack(frames, [r.tri1 for r in ores], [r.tri2 for r in ores], [r.valid for r in ores])
    pf = packing.tile_frames(pf_pool, repeats)
    assert pf.n_frames == 16384
    res = run(pf)
    # (2) pool == oracle
    for f in range(pool):
        assert res["status"][f] == ores[f].status and res["raw_scale"][f] == ores[f].raw_scale, f
    # (1) checksum of checksums over the copies
    def crc(r, sl, level=False):
        c = 0
        for k in ("raw_scale", "height", "status", "counts") + (("height_level",) if level else ()):
            c = zlib.crc32(np.ascontiguousarray(r[k][sl]).tobytes(), c)
        return c
    sums = {crc(res, slice(r * pool, (r + 1) * pool)) for r in range(repeats)}
    assert len(sums) == 1
    # (height_level: the product mode's own fixed-order sum — except for a batch's LAST frame, which the launch finishes
    # in the exact mode for whoever reads it next (mvosr_batch.exact_mask): equal to rounding across the copies)
    for r in range(1, repeats):
        np.testing.assert_allclose(res["height_level"][r * pool:(r + 1) * pool], res["height_level"][:pool], rtol=1e-13)
    # (5) determinism
    res_b = run(pf)
    assert crc(res, slice(None), level=True) == crc(res_b, slice(None), level=True)
    # (3) triangle rows permuted (vertex order inside rows untouched)
    rng = np.random.default_rng(0)
    t1 = [r.tri1[rng.permutation(len(r.tri1))] for r in ores]
    t2 = [r.tri2[rng.permutation(len(r.tri2))] for r in ores]
    pf_perm = packing.tile_frames(_pack(frames, t1, t2, [r.valid for r in ores]), repeats)
    res_p = run(pf_perm)
    for k in ("raw_scale", "height", "status", "counts"):
        assert np.array_equal(res[k], res_p[k], equal_nan=True), k
    np.testing.assert_allclose(res_p["height_level"], res["height_level"], rtol=1e-13)     # a sum in another order
    # (4) frame order reversed
    rev = list(range(pool))[::-1]
    pf_rev = _pack([frames[i] for i in rev], [ores[i].tri1 for i in rev], [ores[i].tri2 for i in rev], [ores[i].valid for i in rev])
    res_r = run(pf_rev)
    for k in ("raw_scale", "height", "status"):
        assert np.array_equal(res_r[k], res[k][:pool][::-1], equal_nan=True), k
    np.testing.assert_allclose(res_r["height_level"], res["height_level"][:pool][::-1], rtol=1e-13)   # (the batch's last frame: exact mode)
    # (6) window median over the whole sequence
    want, _ = so.window_median(res["raw_scale"], 5)
    assert np.array_equal(eng.window_median_host(res["raw_scale"], 5), want)
