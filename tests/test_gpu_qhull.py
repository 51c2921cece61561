"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors of the reference —
qhull_rows_kernel (SURVEY §8 f1 without the declared deviation): SciPy's rows themselves on the device and the reference-exact estimator on top of them.  Needs a real MI355X:  python -m pytest tests -m gpu

Constructions say which path they mean: ``triangulation="scipy"`` is the host-SciPy baseline every device path is compared with; a
construction without the keyword IS the shipped default (triangulation "gpu" with the reference's vote)."""
import os

import numpy as np
import pytest

from conftest import load_json, load_npz
from gpu_helpers import _oracle

pytestmark = pytest.mark.gpu


def test_qhull_rows_kernel_equals_scipy_rows(gpu):
    """mvosr_delaunay_qhull_batch: the rows are scipy.spatial.Delaunay(points).simplices — same rows, same ORDER, same
    ROTATION (what /root/reference/src/scale_calculator.py:105-119 reads) — on frames of 5 to 4000 points, on the survivors of
    a mask (ids = ranks), run to run; degenerate sets are declined, never mis-triangulated; the insertion order the kernel
    reports equals the CPU restatement's (oracle/qhull_rows.py)."""
    from scipy.spatial import Delaunay
    from mvoscalerecovery_amd import packing, synth
    from oracle.qhull_rows import QhullDelaunay2D
    rng = np.random.default_rng(23)
    sizes = [2000, 1500, 700, 300, 120, 40, 9, 5, 4000, 2300] + [int(x) for x in rng.integers(100, 2200, 30)]
    sets = [synth.synth_frame(i, n, base_seed=515)[1] for i, n in enumerate(sizes)]
    sets.append(rng.uniform(0, 1, (1800, 2)) * [1241.0, 376.0])
    sets.append(np.concatenate([rng.uniform(0, 100, (500, 2)), rng.uniform(40, 41, (500, 2))]))      # a dense cluster in a sparse field
    got = packing.delaunay_gpu(gpu, sets, rows="qhull", order_out=True)
    declined = 0
    for k, (pts, tri) in enumerate(zip(sets, got)):
        if tri is None:
            declined += 1
            continue
        ref = Delaunay(pts).simplices
        assert tri.shape == ref.shape and np.array_equal(tri, ref), (k, len(pts))
    assert declined <= 2, declined
    q = QhullDelaunay2D(sets[3])
    order = packing.delaunay_gpu.last_order[3]
    assert [int(p) for p in np.argsort(order, kind="stable") if order[p] > 0] == [p for p in q.order[4:] if p < len(sets[3])]
    keep = np.where(rng.uniform(size=len(sets[0])) < 0.93, 3, -2).astype(np.int32)
    t2 = packing.delaunay_gpu(gpu, [sets[0]], [keep], rows="qhull")[0]
    assert np.array_equal(t2, Delaunay(sets[0][keep >= 0]).simplices)
    assert int(packing.delaunay_gpu.last_used[0]) == int((keep >= 0).sum())
    again = packing.delaunay_gpu(gpu, sets[:4], rows="qhull")
    for x, y in zip(got[:4], again):
        assert np.array_equal(x, y)
    grid = np.stack(np.meshgrid(np.arange(20.0), np.arange(15.0)), axis=-1).reshape(-1, 2)
    dup = sets[3].copy(); dup[10] = dup[200]
    line = np.stack([np.arange(50.0), 2.0 * np.arange(50.0)], axis=1)
    assert all(t is None for t in packing.delaunay_gpu(gpu, [grid, dup, line, sets[3][:2]], rows="qhull"))
    # what trackers hand over: float32-rounded positions, bucketed detections, clusters: exact; sites snapped to a pixel grid:
    # exact or declined — a row that is emitted is SciPy's row
    must, may = [], []
    for seed in range(6):
        must.append(synth.synth_frame(seed, 800, base_seed=5)[1].astype(np.float32).astype(np.float64))
        r = np.random.default_rng(seed)
        gx, gy = np.meshgrid(np.arange(0, 1241, 30), np.arange(186, 376, 30))
        c = np.stack([gx.ravel(), gy.ravel()], 1).astype(float)
        pts = np.concatenate([c + r.uniform(0, 30, c.shape), c + r.uniform(0, 30, c.shape)])
        must.append(pts[r.random(len(pts)) < 0.8])
        cen = r.uniform([0, 186], [1241, 376], (12, 2))
        must.append(np.concatenate([k + r.normal(0, 8, (60, 2)) for k in cen] + [r.uniform([0, 186], [1241, 376], (200, 2))]))
        may.append(np.unique(np.round(synth.synth_frame(seed, 500, base_seed=5)[1]), axis=0))
        may.append(np.unique(np.round(synth.synth_frame(seed, 800, base_seed=5)[1] * 4) / 4, axis=0))
    out = packing.delaunay_gpu(gpu, must + may, rows="qhull")
    for k, (pts, tri) in enumerate(zip(must + may, out)):
        assert tri is not None or k >= len(must), k
        if tri is not None:
            assert np.array_equal(tri, Delaunay(pts).simplices), k


def test_seq4541_golden_device_triangulation_reference_exact(gpu, monkeypatch):
    """Config C3 with BOTH triangulations built on the device and the reference's own vote: triangulation="gpu",
    check_triangle="reference" — no declared deviation.  Every raw and filtered scale of the 4541-frame main_offline-shaped
    sequence equals the reference's (north_star: 1e-4; here bit-equal), and almost no frame needs the host's Qhull."""
    from mvoscalerecovery_amd import offline, synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    z, meta = load_npz("seq4541.npz")
    data = synth.synth_sequence_dict(meta["n_frames"], base_seed=meta["seed"], **meta["kw"])
    est = ScaleEstimator(meta["abs_ref"], window_size=meta["window"], mutate_inputs=False, triangulation="gpu",
                         check_triangle="reference", delaunay_workers=0)
    from mvoscalerecovery_amd import packing
    host_frames = []
    real_attach = packing.attach_tri1
    monkeypatch.setattr(packing, "attach_tri1", lambda pf, *a, **k: (host_frames.append(pf.n_frames), real_attach(pf, *a, **k))[1])
    res = offline.run_sequence_batched(data, est)
    assert sum(host_frames) <= 0.02 * meta["n_frames"]                 # (only declined frames may visit the host's SciPy)
    assert np.array_equal(res["kinds"], z["kinds"])
    assert np.array_equal(est.last_raw_scale, z["raw_scales"], equal_nan=True)
    np.testing.assert_array_equal(res["scales"], z["scales"])
    np.testing.assert_array_equal(res["error"], z["error"])
    assert est.last_declined <= 0.02 * meta["n_frames"]


def test_stage_goldens_device_triangulation_reference_exact(gpu):
    """The 20 stage frames of the UNPATCHED reference (tests/golden/stages.npz) through triangulation="gpu",
    check_triangle="reference": scale, std, height_level and the selected road points per frame — per-frame calls and one
    batch call."""
    from mvoscalerecovery_amd import synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    z, meta = load_npz("stages.npz")
    f3s, f2s = [], []
    for k, fr in enumerate(meta["frames"]):
        f3, f2 = synth.synth_frame(fr["frame_idx"], fr["n"], base_seed=fr["seed"], upper_fraction=fr["upper_fraction"])
        est = ScaleEstimator(meta["abs_ref"], window_size=5, triangulation="gpu", check_triangle="reference", mutate_inputs=False,
                             delaunay_workers=0)
        s, sd = est.scale_calculation(f3, f2)
        assert s == float(z["f%d_scale_first_call" % k]) and sd == float(z["f%d_std" % k]), k
        assert est.height_level == float(z["f%d_height_level" % k]), k
        assert len(est.flat_feature) == len(z["f%d_selected_ids" % k]), k
        f3s.append(f3)
        f2s.append(f2)
    est = ScaleEstimator(meta["abs_ref"], window_size=5, triangulation="gpu", check_triangle="reference", mutate_inputs=False,
                         delaunay_workers=0)
    raw, status, level, _ = est.raw_scale_batch(f3s, f2s)
    assert raw.tolist() == [float(z["f%d_scale_first_call" % k]) for k in range(len(f3s))]


def test_dense_frames_reference_exact_on_the_device(gpu, monkeypatch):
    """BASELINE configs[4]'s shape through the exact device path: 20 000-point sets are beyond 16-bit facet ids (7 n facets per
    run) — qhull_rows_kernel<uint32_t>: rows == SciPy's rows (20 000 and 9 000 points, and the survivors of a mask); the
    reference's N = 20 000 golden (tests/golden/dense.npz) with both triangulations built on the device and the reference's vote:
    raw scale, status and selected count as the reference's."""
    from scipy.spatial import Delaunay
    from mvoscalerecovery_amd import constants as K, packing, synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    z, meta = load_npz("dense.npz")
    f3, f2 = synth.synth_frame(meta["frame_idx"], meta["n"], base_seed=meta["seed"])
    low = f2[f2[:, 1] > 185]
    sets = [low, synth.synth_frame(5, 9000, base_seed=808)[1]]
    got = packing.delaunay_gpu(gpu, sets, rows="qhull")
    for pts, tri in zip(sets, got):
        assert tri is not None, int(packing.delaunay_gpu.last_status[0]) >> 8
        assert np.array_equal(tri, Delaunay(pts).simplices)
    assert np.array_equal(got[0], z["tri1"].astype(np.int32))                       # ... which is the golden's first triangulation
    keep = np.where(z["valid"], 1, -1).astype(np.int32)
    t2 = packing.delaunay_gpu(gpu, [low], [keep], rows="qhull")[0]
    assert np.array_equal(t2, z["tri2"].astype(np.int32))                           # and, over the vote's survivors, its second
    est = ScaleEstimator(meta["abs_ref"], window_size=5, mutate_inputs=False, triangulation="gpu", check_triangle="reference", delaunay_workers=0)
    est.GPU_EXACT_FORCE_DEVICE = True                                                 # (four frames: the host replay would be quicker)
    host_calls = []
    real_attach = packing.attach_tri1
    monkeypatch.setattr(packing, "attach_tri1", lambda pf, *a, **k: (host_calls.append(pf.n_frames), real_attach(pf, *a, **k))[1])
    raw, status, level, _ = est.raw_scale_batch([f3] * 4, [f2] * 4)                 # (four frames: above the host's break-even)
    assert est.last_declined == 0 and not host_calls                                # no frame went through the host's SciPy
    for k in range(4):
        assert raw[k] == float(z["scale_first_call"]) and status[k] in (K.ST_MODE, K.ST_RIGHT), (k, raw[k], status[k])
        assert est.last_counts[k, K.CNT_SELECTED] == len(z["selected_ids"])


@pytest.mark.parametrize("standin", [True, False])
def test_frame_fuzz_batched_device_triangulation_reference_exact(gpu, standin):
    """The 440 adversarial frames (levels within 1e-13 of a flat triangle, levels that ARE the result, every raise site's
    neighbourhood) in one batch through triangulation="gpu", check_triangle="reference": raw scales bit-equal to the reference's,
    statuses the oracle's, the levels that decide bit-equal — with the second triangulation as a stand-in (SciPy's own rows built
    only for the frames of the exact pass) and with Qhull's replay for every frame."""
    from mvoscalerecovery_amd import constants as K, synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    so = _oracle()
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "frame_fuzz.npz"))
    idx = [i for i in range(len(z["scale"])) if not z["raised"][i]]
    frames = [synth.fuzz_frame(i, int(z["seed"])) for i in idx]
    est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", check_triangle="reference", delaunay_workers=0)
    est.GPU_EXACT_STANDIN = standin
    est.GPU_EXACT_CHUNK = 128
    raw, status, level, errors = est.raw_scale_batch([f[0] for f in frames], [f[1] for f in frames])
    assert not errors
    seen = set()
    for k, i in enumerate(idx):
        want = z["scale"][i]
        assert (np.isnan(raw[k]) and np.isnan(want)) or raw[k] == want, (i, raw[k], want, status[k])
        r = so.frame_raw_scale(frames[k][0], frames[k][1], 1.75)
        assert status[k] == r.status, (i, status[k], r.status)
        if status[k] in (K.ST_NO_FLAT, K.ST_LEVEL) and not np.isnan(r.height_level):
            assert level[k] == r.height_level, i
        seen.add(int(status[k]))
    assert {K.ST_MODE, K.ST_RIGHT, K.ST_MEDIAN, K.ST_NO_FLAT} <= seen


def test_exact_path_two_contexts_equal_one(gpu):
    """A call of the exact device path large enough for its chunks to alternate between two contexts (>= 8 192 frames) gives the
    arrays the one-context run gives — raw scales, statuses, levels, filtered scales — and the host-SciPy run's on a sample."""
    from mvoscalerecovery_amd import synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    pool = [synth.synth_frame(i, 250 + (37 * i) % 400, base_seed=4242, upper_fraction=0.1) for i in range(257)]
    F = 9300
    f3s, f2s = [pool[i % 257][0] for i in range(F)], [pool[i % 257][1] for i in range(F)]
    res = []
    for two in (True, False):
        est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, triangulation="gpu", check_triangle="reference", delaunay_workers=0)
        est.GPU_EXACT_TWO_CONTEXTS = two
        s, sd = est.scale_calculation_batch(f3s, f2s)
        res.append((np.asarray(s), np.asarray(sd), np.asarray(est.last_raw_scale), np.asarray(est.last_status)))
        assert (getattr(est, "_engine2", None) is not None) == two
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b, equal_nan=True)
    host = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=0, triangulation="scipy")
    raw_h, st_h, _, _ = host.raw_scale_batch(f3s[:257], f2s[:257])
    assert np.array_equal(res[0][2][:257], raw_h, equal_nan=True) and np.array_equal(res[0][3][:257], st_h)


def test_qhull_rows_kernel_hostile_inputs_are_declined(gpu):
    """NaN / infinite / huge / identical / collinear sites, a mask that keeps fewer than three points: the kernel declines (status
    != 0, no rows) — it neither hangs nor writes outside its frame — and the sets around them are untouched."""
    from scipy.spatial import Delaunay
    from mvoscalerecovery_amd import packing, synth
    good = synth.synth_frame(1, 300, base_seed=99)[1]
    nan = good.copy(); nan[17, 0] = np.nan
    inf = good.copy(); inf[40, 1] = np.inf
    huge = good.copy() * 1e200
    same = np.tile(good[:1], (50, 1))
    line = np.stack([np.linspace(0, 1000, 80), np.linspace(200, 300, 80)], axis=1)
    allnan = np.full((30, 2), np.nan)
    sets = [good, nan, good, inf, huge, same, line, allnan, good]
    got = packing.delaunay_gpu(gpu, sets, rows="qhull")
    ref = Delaunay(good).simplices
    for k in (0, 2, 8):
        assert got[k] is not None and np.array_equal(got[k], ref), k
    for k in (1, 3, 5, 6, 7):
        assert got[k] is None, k
    assert got[4] is None or np.array_equal(got[4], Delaunay(huge).simplices)          # (scaled copies: either answer is fine, a wrong one is not)
    keep = np.full(len(good), -1, np.int32); keep[:2] = 1
    assert packing.delaunay_gpu(gpu, [good], [keep], rows="qhull")[0] is None


def test_lazy_level_of_a_chunks_last_frame(gpu):
    """Round 6: in a batch of the default estimator a chunk's last level-setting frame is no longer on the exact mask (it was the
    exact pass's whole list on ordinary data: one frame's replay of Qhull's run, 23 ms at the end of every chunk).  Its level in
    NumPy's order is produced when it is read: by a three-feature frame at the head of the next chunk (finished inside the call), by
    the frame before a raising frame, or by whoever reads ``est.height_level`` after the call — bit-equal to the oracle's each time."""
    from mvoscalerecovery_amd import synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    so = _oracle()

    def run(frames, chunk=16):
        est = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=0)
        est.GPU_EXACT_CHUNK, est.GPU_MIN_CHUNK, est.GPU_EXACT_FORCE_DEVICE = chunk, 1, True
        ref = so.OracleScaleEstimator(1.75, window_size=5)
        want = [ref.scale_calculation(f3.copy(), f2.copy()) for f3, f2 in frames]
        got = est.scale_calculation_batch([f[0] for f in frames], [f[1] for f in frames])
        assert [w[0] for w in want] == list(got[0]) and [w[1] for w in want] == list(got[1])
        return est, ref

    frames = [synth.synth_frame(i, 300 + 17 * i, base_seed=2718, upper_fraction=0.1) for i in range(48)]
    # (1) ordinary frames, three chunks: nothing finished inside the call, the estimator's level exact when read
    est, ref = run(frames)
    assert getattr(est, "lazy_levels_finished", 0) == 0 and est.__dict__.get("_level_thunk") is not None
    assert est.height_level == ref.height_level and est.lazy_levels_on_read == 1
    assert est.height_level == ref.height_level and est.lazy_levels_on_read == 1            # (computed once)
    # (2) a three-feature frame at the head of the second and the third chunk reads the level of the chunk before
    few = [(f3, f2.copy()) for f3, f2 in frames]
    for f in (16, 32, 33):
        low = np.nonzero(few[f][1][:, 1] > 185)[0]
        few[f][1][low[3:], 1] = 100.0
    est, ref = run(few)
    assert est.lazy_levels_finished == 2
    assert est.height_level == ref.height_level
    # (3) ... and the call's LAST frames: the level they read is the last setter's, which is on the mask (followed by a three-feature frame)
    tail = [(f3, f2.copy()) for f3, f2 in frames]
    for f in (46, 47):
        low = np.nonzero(tail[f][1][:, 1] > 185)[0]
        tail[f][1][low[3:], 1] = 100.0
    est, ref = run(tail)
    assert est.height_level == ref.height_level and getattr(est, "lazy_levels_on_read", 0) == 0
    # (4) the eager form gives the same numbers
    est2 = ScaleEstimator(1.75, window_size=5, mutate_inputs=False, delaunay_workers=0)
    est2.GPU_EXACT_CHUNK, est2.GPU_MIN_CHUNK, est2.GPU_EXACT_FORCE_DEVICE, est2.GPU_EXACT_LAZY_LEVEL = 16, 1, True, False
    a = est2.scale_calculation_batch([f[0] for f in few], [f[1] for f in few])
    est3, _ = run(few)
    b = est3.scale_calculation_batch([f[0] for f in few], [f[1] for f in few])
    assert est2.height_level == est3.height_level
