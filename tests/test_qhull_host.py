"""mvosr_qhull_rows_host (csrc/mvosr_qhull_host.c): the replay of Qhull's run on the HOST — what the per-frame call of the default
estimator uses for its first triangulation instead of scipy.spatial.Delaunay (/root/reference/src/scale_calculator.py:257).  No
GPU needed: the function is plain C inside libmvosr.so.  Checked against SciPy itself (rows: set, order, rotation) and against the
restatement in oracle/qhull_rows.py (same accept / decline decision)."""
import numpy as np
import pytest

from mvoscalerecovery_amd import packing, synth
from oracle.qhull_rows import Declined, QhullDelaunay2D


def _sets():
    out = []
    rng = np.random.default_rng(20260)
    for i, n in enumerate((5, 9, 17, 40, 77, 150, 300, 600, 900, 1200, 1500, 2000, 2300)):
        out.append(("frame %d" % n, synth.synth_frame(i, n, base_seed=8800)[1]))
        out.append(("uniform %d" % n, np.stack([rng.uniform(0, 1241, n), rng.uniform(186, 376, n)], axis=1)))
    for n in (60, 700, 1800):
        p = np.stack([rng.uniform(0, 1241, n), rng.uniform(186, 376, n)], axis=1)
        out.append(("float32 %d" % n, p.astype(np.float32).astype(np.float64)))
        out.append(("gaussian %d" % n, np.stack([rng.normal(600, 150, n), rng.normal(280, 40, n)], axis=1)))
        out.append(("clustered %d" % n, np.concatenate([rng.normal((300 + 200 * k, 250), 12, (n // 4, 2)) for k in range(4)])))
    return out


def test_rows_equal_scipy_and_oracle_decisions():
    compared = declined = 0
    for name, p in _sets():
        p = np.ascontiguousarray(p)
        rows = packing.qhull_rows_host(p)
        try:
            want = QhullDelaunay2D(p).simplices()
        except Declined:
            want = None
        assert (rows is None) == (want is None), (name, packing.qhull_rows_host.last_reason)
        if rows is None:
            declined += 1
            continue
        compared += 1
        assert rows.dtype == np.int32 and np.array_equal(rows, want), name
        assert np.array_equal(rows, packing.delaunay_simplices(p)), name
    assert compared >= 32 and declined <= 3, (compared, declined)


def test_mask_survivors_and_strided_input():
    """The second triangulation's shape of input (the survivors of a vote: :264-266) and a view with a stride."""
    rng = np.random.default_rng(4)
    for n in (400, 1700):
        f2 = synth.synth_frame(n, n, base_seed=313)[1]
        keep = rng.uniform(size=n) < 0.93
        p = np.ascontiguousarray(f2[keep])
        assert np.array_equal(packing.qhull_rows_host(p), packing.delaunay_simplices(p))
        wide = np.zeros((n, 5))
        wide[:, 1:3] = f2
        assert np.array_equal(packing.qhull_rows_host(wide[:, 1:3]), packing.delaunay_simplices(f2))      # (copied to contiguous)


def test_hostile_inputs_are_declined_never_wrong():
    """Sets outside general position: declined (the caller asks SciPy), or — where the replay goes through — SciPy's rows."""
    rng = np.random.default_rng(9)
    grid = np.stack(np.meshgrid(np.arange(12.0) * 7 + 200, np.arange(9.0) * 5 + 190), axis=-1).reshape(-1, 2)
    dup = np.concatenate([rng.uniform(200, 900, (50, 2)), rng.uniform(200, 900, (50, 2))[:10].repeat(2, axis=0)])
    line = np.stack([np.linspace(0, 1000, 40), np.linspace(190, 370, 40)], axis=1)
    quarter = np.round(np.stack([rng.uniform(0, 1241, 500), rng.uniform(186, 376, 500)], axis=1) * 4) / 4
    circle = np.stack([600 + 100 * np.cos(np.arange(24) * np.pi / 12), 280 + 100 * np.sin(np.arange(24) * np.pi / 12)], axis=1)
    for name, p in (("grid", grid), ("duplicates", dup), ("collinear", line), ("quarter pixel", quarter), ("cocircular", circle)):
        rows = packing.qhull_rows_host(np.ascontiguousarray(p))
        if rows is not None:
            assert np.array_equal(rows, packing.delaunay_simplices(p)), name
        else:
            assert packing.qhull_rows_host.last_reason > 0, name
    for p in (np.zeros((0, 2)), np.zeros((2, 2)), np.array([[0.0, 0.0], [1.0, 0.0], [np.nan, 1.0]]), np.full((10, 2), 3.0),
              np.array([[0.0, 0.0], [1e200, 1.0], [1.0, 1e200], [5.0, 5.0]])):
        assert packing.qhull_rows_host(p) is None and packing.qhull_rows_host.last_reason > 0
    with pytest.raises(ValueError):
        packing.qhull_rows_host(np.zeros((4, 3)))
    # delaunay_simplices_fast: SciPy's rows either way
    assert np.array_equal(packing.delaunay_simplices_fast(quarter), packing.delaunay_simplices(quarter))


def test_soak_against_scipy():
    """600 point sets of 30-2300 points (frames, uniform, float32-rounded): every accepted set equals SciPy's rows."""
    rng = np.random.default_rng(77)
    differ = declined = 0
    for i in range(600):
        n = int(rng.integers(30, 2301)) if i % 5 else int(rng.integers(30, 200))
        kind = i % 3
        if kind == 0:
            p = synth.synth_frame(i, n, base_seed=40000)[1]
        else:
            p = np.stack([rng.uniform(0, 1241, n), rng.uniform(186, 376, n)], axis=1)
            if kind == 2:
                p = p.astype(np.float32).astype(np.float64)
        rows = packing.qhull_rows_host(np.ascontiguousarray(p))
        if rows is None:
            declined += 1
        elif not np.array_equal(rows, packing.delaunay_simplices(p)):
            differ += 1
    assert differ == 0 and declined <= 6, (differ, declined)


def test_threads_share_nothing():
    """Per-thread workspaces: eight threads replaying different sets at once give what one thread gives."""
    from concurrent.futures import ThreadPoolExecutor
    sets = [np.ascontiguousarray(synth.synth_frame(i, 300 + 200 * (i % 8), base_seed=515)[1]) for i in range(64)]
    want = [packing.qhull_rows_host(p) for p in sets]
    with ThreadPoolExecutor(8) as ex:
        got = list(ex.map(packing.qhull_rows_host, sets * 3))
    for k, g in enumerate(got):
        w = want[k % 64]
        assert (g is None) == (w is None) and (g is None or np.array_equal(g, w))


def test_sanitizers(tmp_path):
    """The C replay under AddressSanitizer + UBSan on 3 000 point sets, hostile ones included (tests/native/qhull_host_sanitize.c):
    no report, no out-of-range row, a too-small rows buffer declined."""
    import os
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "san")
    build = subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                            os.path.join(root, "tests", "native", "qhull_host_sanitize.c"),
                            os.path.join(root, "mvoscalerecovery_amd", "csrc", "mvosr_qhull_host.c"), "-lm", "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("no sanitizer runtime: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert run.returncode == 0 and "sanitizer run: ok" in run.stdout and "errors 0" in run.stdout, run.stdout[-500:] + run.stderr[-3000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
