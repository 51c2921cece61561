"""The first-use self-check of the Qhull replay against the installed SciPy (mvoscalerecovery_amd/selfcheck.py): the
reference's Delaunay is whatever Qhull its SciPy bundles (/root/reference/src/scale_calculator.py:12,257,266)."""
import warnings

import numpy as np
import pytest


def _rows(seed, n):
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(Delaunay(rng.uniform(0, 100, (n, 2))).simplices, dtype=np.int32)


def test_compare_rows_decision_logic():
    """The decision on plain arrays: identical rows pass; a rotated row, a permuted row order, a different triangle set fail;
    declined sets are neutral unless they are the majority."""
    from mvoscalerecovery_amd import selfcheck
    ref = [_rows(s, 30 + 10 * s) for s in range(6)]
    ok, d = selfcheck.compare_rows([r.copy() for r in ref], ref)
    assert ok and d["compared"] == 6 and not d["different"]
    rot = [r.copy() for r in ref]
    rot[2] = np.roll(rot[2], 1, axis=1)                       # same triangles, every row rotated
    ok, d = selfcheck.compare_rows(rot, ref)
    assert not ok and d["different"] == [{"set": 2, "rows_replay": len(ref[2]), "rows_scipy": len(ref[2]), "same_triangle_set": True}]
    perm = [r.copy() for r in ref]
    perm[0] = perm[0][::-1].copy()                            # same rows, another order
    assert not selfcheck.compare_rows(perm, ref)[0]
    other = [r.copy() for r in ref]
    other[5] = other[5][:-1]
    ok, d = selfcheck.compare_rows(other, ref)
    assert not ok and d["different"][0]["same_triangle_set"] is False
    some = [r.copy() for r in ref]
    some[1] = None                                            # the replay declined one set, SciPy raised on another: still a check
    refs = list(ref)
    refs[4] = RuntimeError("QhullError")
    ok, d = selfcheck.compare_rows(some, refs)
    assert ok and d["declined"] == 2 and d["compared"] == 4
    assert not selfcheck.compare_rows([None] * 4 + [ref[4].copy(), ref[5].copy()], ref)[0]     # a replay that declines most sets checks nothing
    assert not selfcheck.compare_rows([None] * 6, ref)[0]


def test_check_sets_are_fixed_and_env_switch(monkeypatch):
    from mvoscalerecovery_amd import selfcheck
    a, b = selfcheck.check_point_sets(), selfcheck.check_point_sets()
    assert len(a) == len(selfcheck.CHECK_SETS) == 8
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert [len(x) for x in a] == [n for _, n, _ in selfcheck.CHECK_SETS]
    assert min(len(x) for x in a) == 40 and max(len(x) for x in a) == 2000
    assert selfcheck.enabled()
    monkeypatch.setenv("MVOSR_QHULL_SELFCHECK", "0")
    assert not selfcheck.enabled()


def test_host_replay_passes_the_check_sets():
    """The C replay on the host (libmvosr_py.so) on the self-check's own sets, against the installed SciPy — what the
    estimator's constructor verifies on the GPU box, without the device half."""
    from mvoscalerecovery_amd import packing, selfcheck
    replay = packing.qhull_rows_host_or_none()
    if replay is None:
        pytest.skip("libmvosr_py.so without the host replay")
    sets = selfcheck.check_point_sets()
    ok, d = selfcheck.compare_rows([replay(p) for p in sets], [packing.delaunay_simplices(p) for p in sets])
    assert ok and d["compared"] >= 7, d


@pytest.mark.gpu
def test_selfcheck_passes_on_this_box_and_is_cached(gpu, monkeypatch):
    from mvoscalerecovery_amd import selfcheck
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    monkeypatch.delenv("MVOSR_TRIANGULATION", raising=False)
    res = selfcheck.run(gpu, force=True)
    assert res["ok"] and not res["skipped"] and res["device"]["compared"] >= 7 and not res["device"]["different"], res
    est = ScaleEstimator(1.75, window_size=5, delaunay_workers=0)
    assert (est.triangulation, est.check_triangle) == ("gpu", "reference") and est.qhull_selfcheck["ok"]
    calls = []
    from mvoscalerecovery_amd import packing
    real = packing.delaunay_gpu
    monkeypatch.setattr(packing, "delaunay_gpu", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    ScaleEstimator(1.75, window_size=5, delaunay_workers=0)
    assert not calls                                          # once per process and device
    monkeypatch.setenv("MVOSR_QHULL_SELFCHECK", "0")
    assert selfcheck.run(gpu, force=True)["skipped"]
    monkeypatch.delenv("MVOSR_QHULL_SELFCHECK")
    selfcheck.run(gpu, force=True)


@pytest.mark.gpu
def test_estimator_falls_back_when_the_installed_scipy_rotates_its_rows(gpu, monkeypatch):
    """A SciPy whose Qhull emits the same triangles with other rotations (here: scipy.spatial.Delaunay patched to roll its rows):
    the self-check sees it, warns once naming both versions, and the default estimator runs the host path — identical, batch and
    per-frame call, to triangulation="scipy" on this (patched) box, which is what the reference would compute here."""
    import scipy.spatial
    from mvoscalerecovery_amd import selfcheck, synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    monkeypatch.delenv("MVOSR_TRIANGULATION", raising=False)
    real = scipy.spatial.Delaunay

    class Rolled:
        def __init__(self, pts, *a, **k):
            self.simplices = np.ascontiguousarray(np.roll(real(pts, *a, **k).simplices, 1, axis=1))
    frames = [synth.synth_frame(i, 500 + 37 * i, base_seed=4242, upper_fraction=0.1) for i in range(24)]
    honest = ScaleEstimator(1.75, window_size=5, delaunay_workers=0, triangulation="scipy")
    h, _ = honest.scale_calculation_batch([f[0].copy() for f in frames], [f[1] for f in frames])
    monkeypatch.setattr(scipy.spatial, "Delaunay", Rolled)
    try:
        selfcheck.LAST.clear()
        with pytest.warns(RuntimeWarning, match="different rows on the self-check sets"):
            est = ScaleEstimator(1.75, window_size=5, delaunay_workers=0)
        assert est.triangulation == "scipy" and est.check_triangle == "reference" and not est.qhull_selfcheck["ok"]
        assert all(d["same_triangle_set"] for d in est.qhull_selfcheck["device"]["different"])
        with warnings.catch_warnings():
            warnings.simplefilter("error")                    # (cached: no second warning)
            est2 = ScaleEstimator(1.75, window_size=5, delaunay_workers=0)
        assert est2.triangulation == "scipy"
        ref = ScaleEstimator(1.75, window_size=5, delaunay_workers=0, triangulation="scipy")
        s, sd = est.scale_calculation_batch([f[0].copy() for f in frames], [f[1] for f in frames])
        r, rd = ref.scale_calculation_batch([f[0].copy() for f in frames], [f[1] for f in frames])
        assert np.array_equal(s, r) and np.array_equal(sd, rd)
        assert not np.array_equal(s, h)                       # (the rotation matters: :113-115)
        f3, f2 = synth.synth_frame(99, 900, base_seed=4242)
        assert est.scale_calculation(f3.copy(), f2) == ref.scale_calculation(f3.copy(), f2)
    finally:
        monkeypatch.setattr(scipy.spatial, "Delaunay", real)
        selfcheck.LAST.clear()
        assert selfcheck.run(gpu, force=True)["ok"]
