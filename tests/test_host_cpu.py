"""CPU-side tests: the C-ABI library loads and exports every symbol the header declares, host
packing, the driver loop, and that the product path fails loudly without a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_npz


def _header_functions():
    src = open(os.path.join(ROOT, "include", "mvosr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mvosr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from mvoscalerecovery_amd import _lib
    lib = _lib.load()
    declared = _header_functions()
    assert len(declared) >= 20
    assert sorted(_lib.SYMBOLS) == declared          # binding table == header
    for name in declared:
        assert hasattr(lib, name), name
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (mvosr_[a-z0-9_]+)", nm))
    assert set(declared) <= exported
    assert lib.mvosr_abi_version() == _lib.ABI_VERSION


def test_shipped_library_has_no_ablation_switch():
    """The ablation hook of the profiling builds (env MVOSR_DEBUG_SKIP: bits that switch sweeps off) exists only under
    -DMVOSR_ABLATE: the product binary does not even contain the variable's name, nor the stamp hooks."""
    from mvoscalerecovery_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"MVOSR_DEBUG_SKIP" not in blob
    assert b"MVOSR_QH_GROUP" not in blob          # (the packed variants of qhull_rows_kernel: A/B builds only)
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "mvosr_debug" not in nm


def test_canonical_rows_and_delaunay_limits():
    """packing.canonical_rows (the row form of check_triangle="fixed" and of the device triangulation) equals the
    oracle's; the device stage's size limit is a host-side query."""
    from mvoscalerecovery_amd import packing
    from oracle import scale_oracle as so
    rng = np.random.default_rng(0)
    t = rng.integers(0, 50, (40, 3)).astype(np.int32)
    a, b = packing.canonical_rows(t), so.canonical_rows(t)
    assert np.array_equal(a, b) and a.dtype == np.int32
    assert np.all(a[:, 0] <= a[:, 1]) and np.all(a[:, 1] <= a[:, 2])
    assert packing.canonical_rows(np.zeros((0, 3), np.int32)).shape == (0, 3)
    assert 4000 <= packing.delaunay_gpu_max_points() < 65536


def test_c_packer_equals_python_packer():
    """mvosr_pack_count / mvosr_pack_fill (host threads, no GPU): the vanishing-row filter, the plane layout and the
    in-place feature_remap of /root/reference/src/scale_calculator.py:252-254,390-394,414 — against packing.pack_features
    and the oracle's remap, bit for bit; the CPython helper's pointer tables; frames it must refuse."""
    import ctypes as C
    from mvoscalerecovery_amd import _lib, engine, packing, synth
    from oracle import scale_oracle as so
    lib = _lib.load()
    frames = [synth.synth_frame(i, n, base_seed=12, upper_fraction=0.3) for i, n in enumerate((500, 3, 0, 1200, 77))]
    frames[2] = (np.zeros((0, 3)), np.zeros((0, 2)))
    f3s, f2s = [f[0].copy() for f in frames], [f[1].copy() for f in frames]
    tb = engine.frame_tables(f3s, f2s)
    assert tb is not None
    p3, p2, npts = tb
    assert npts.tolist() == [len(a) for a in f3s]
    assert all(int(p3[i]) == f3s[i].ctypes.data and int(p2[i]) == f2s[i].ctypes.data for i in range(len(f3s)) if len(f3s[i]))
    F = len(f3s)
    cnt = np.zeros(F, np.int32)
    assert lib.mvosr_pack_count(F, p2.ctypes.data, npts.ctypes.data, 185.0, cnt.ctypes.data, 3) == 0
    ref = packing.pack_features(f3s, f2s, 185)
    assert np.array_equal(cnt, ref.feat_cnt)
    off, total = packing.pack_layout(npts)                  # (laid out by unfiltered sizes, as the batch path does)
    planes = {k: np.full(total, np.nan) for k in "xyzuv"}
    cnt2 = np.zeros(F, np.int32)
    c, s_ = float(np.cos(so.CAMERA_PITCH)), float(np.sin(so.CAMERA_PITCH))
    want_remapped = [so.remap(a) for a in f3s]
    assert lib.mvosr_pack_fill(F, p3.ctypes.data, p2.ctypes.data, npts.ctypes.data, 185.0, off.ctypes.data,
                               planes["x"].ctypes.data, planes["y"].ctypes.data, planes["z"].ctypes.data, planes["u"].ctypes.data,
                               planes["v"].ctypes.data, 1, c, s_, 2, cnt2.ctypes.data) == 0
    assert np.array_equal(cnt2, ref.feat_cnt)
    for f in range(F):
        n, a, b = int(cnt[f]), int(off[f]), ref.frame_slice(f)
        for k in "xyzuv":
            assert np.array_equal(planes[k][a:a + n], getattr(ref, k)[b]), (f, k)        # raw values: the kernels remap at load
        assert np.array_equal(f3s[f], want_remapped[f]), f                                  # the caller's arrays: remapped in place (:414)
    # what the C packer cannot read in place is refused (the batch path then packs in Python)
    assert engine.frame_tables([f3s[0].astype(np.float32)], [f2s[0]]) is None
    assert engine.frame_tables([f3s[0][::2]], [f2s[0][::2]]) is None
    assert engine.frame_tables([f3s[0]], [f2s[3]]) is None
    # ... and what it must not WRITE to: a read-only feature3d when the batch remaps in place (the reference raises
    # ValueError at /root/reference/src/scale_calculator.py:393 — so does the Python path the caller falls back to), and a
    # batch that holds one array object twice (two packer threads on the same memory); without the remap both are fine
    ro = f3s[0].copy()
    ro.flags.writeable = False
    frozen = np.frombuffer(f3s[3].tobytes(), dtype=np.float64).reshape(-1, 3)
    for lst in ([ro], [frozen]):
        assert engine.frame_tables(lst, [f2s[0] if lst[0] is ro else f2s[3]]) is not None
        assert engine.frame_tables(lst, [f2s[0] if lst[0] is ro else f2s[3]], remap_in_place=True) is None
    assert engine.frame_tables([f3s[0], f3s[3], f3s[0]], [f2s[0], f2s[3], f2s[0]], remap_in_place=True) is None
    assert engine.frame_tables([f3s[0], f3s[3], f3s[0]], [f2s[0], f2s[3], f2s[0]]) is not None
    assert not packing.native_packable([ro], [f2s[0]], writable=True) and packing.native_packable([ro], [f2s[0]])


def test_c_packer_wide_path_and_thread_pool():
    """The packer without the in-place remap takes eight features per step where the CPU has AVX-512 (and the same scalar loop
    where not): every size 0..40 and a ragged batch, rows at / above / below the vanishing row, NaN and infinite pixel rows
    (`>` is false for NaN, as in NumPy), against packing.pack_features bit for bit; many calls in a row and calls from two
    threads at once through the library's thread pool."""
    import threading
    from mvoscalerecovery_amd import _lib, engine, packing
    lib = _lib.load()
    rng = np.random.default_rng(5)
    sizes = list(range(0, 41)) + [int(v) for v in rng.integers(41, 3000, 30)]
    f3s = [np.ascontiguousarray(rng.normal(size=(n, 3))) for n in sizes]
    f2s = []
    for n in sizes:
        a = np.ascontiguousarray(rng.uniform(100, 300, size=(n, 2)))
        if n:
            a[rng.integers(0, n, max(1, n // 7)), 1] = 185.0               # exactly at the vanishing row: dropped
            a[rng.integers(0, n, max(1, n // 9)), 1] = np.nan
            a[rng.integers(0, n, max(1, n // 11)), 1] = np.inf
            a[rng.integers(0, n, max(1, n // 13)), 1] = -np.inf
        f2s.append(a)
    keep3 = [a.copy() for a in f3s]
    ref = packing.pack_features(f3s, f2s, 185)
    p3, p2, npts = engine.frame_tables(f3s, f2s)
    F = len(sizes)
    off, total = packing.pack_layout(npts)

    def one(threads):
        planes = {k: np.full(total + 8, -7.0) for k in "xyzuv"}
        cnt = np.zeros(F, np.int32)
        assert lib.mvosr_pack_fill(F, p3.ctypes.data, p2.ctypes.data, npts.ctypes.data, 185.0, off.ctypes.data,
                                   planes["x"].ctypes.data, planes["y"].ctypes.data, planes["z"].ctypes.data, planes["u"].ctypes.data,
                                   planes["v"].ctypes.data, 0, 1.0, 0.0, threads, cnt.ctypes.data) == 0
        assert np.array_equal(cnt, ref.feat_cnt)
        for f in range(F):
            n, a, b = int(cnt[f]), int(off[f]), ref.frame_slice(f)
            for k in "xyzuv":
                assert np.array_equal(planes[k][a:a + n], getattr(ref, k)[b], equal_nan=True), (f, k)
        for k in "xyzuv":
            assert np.all(planes[k][total:] == -7.0)             # nothing past the last frame's slot

    for threads in (1, 2, 5, 16, 0):
        one(threads)
    for _ in range(20):
        one(4)
    errs = []
    def hammer():
        try:
            for _ in range(10):
                one(3)
        except BaseException as e:                               # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=hammer) for _ in range(3)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    assert all(np.array_equal(a, b) for a, b in zip(f3s, keep3))  # (no remap: the caller's arrays are untouched)
    # a forked child has none of the pool's threads: the library starts a pool of its own there
    import os
    pid = os.fork()
    if pid == 0:
        try:
            one(4)
            os._exit(0)
        except BaseException:                                    # noqa: BLE001
            os._exit(1)
    assert os.waitpid(pid, 0)[1] == 0


def test_pin_thread_to_node_respects_what_it_finds():
    """_lib.pin_thread_to_node (the binding keeps a process that opens a context on the device's NUMA node): unknown node, no
    opt-in (MVOSR_AFFINITY=1), a thread already confined to one node and a node without CPUs of ours leave the affinity alone; a real node
    narrows it to that node's CPUs (and the test puts it back)."""
    import glob
    import os
    from mvoscalerecovery_amd import _lib
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_getaffinity")
    before = os.sched_getaffinity(0)
    try:
        assert _lib.pin_thread_to_node(-1) is None and _lib.pin_thread_to_node(None) is None
        assert _lib.pin_thread_to_node(4096) is None                       # no such node
        os.environ.pop("MVOSR_AFFINITY", None)
        assert _lib.pin_thread_to_node(0) is None                          # opt-in: nothing without MVOSR_AFFINITY=1
        os.environ["MVOSR_AFFINITY"] = "0"
        assert _lib.pin_thread_to_node(0) is None
        os.environ["MVOSR_AFFINITY"] = "1"
        assert os.sched_getaffinity(0) == before
        nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*/cpulist"))
        got = _lib.pin_thread_to_node(0) if nodes else None
        if got is not None:                                                # more than one node, and we were on several
            assert got < before and got == os.sched_getaffinity(0) and got <= _lib._cpulist(open(nodes[0]).read())
            assert _lib.pin_thread_to_node(0) is None                      # confined to one node now: left alone
        else:
            assert os.sched_getaffinity(0) == before
        assert _lib._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    finally:
        os.environ.pop("MVOSR_AFFINITY", None)
        os.sched_setaffinity(0, before)


def test_delaunay_residency_query():
    """mvosr_delaunay_frames_per_cu (host-side: what the batch paths round their chunks with): 8, 4, 3, 2, 1 frames per CU,
    never more for a larger frame; one beyond the LDS-resident sizes."""
    from mvoscalerecovery_amd import _lib
    lib = _lib.load()
    top = int(lib.mvosr_delaunay_lds_points())
    vals = [int(lib.mvosr_delaunay_frames_per_cu(n)) for n in range(3, top + 200, 7)]
    assert set(vals) <= {8, 4, 3, 2, 1} and vals[0] == 8 and vals[-1] == 1
    assert all(a >= b for a, b in zip(vals, vals[1:]))
    assert int(lib.mvosr_delaunay_frames_per_cu(2000)) == 3 and int(lib.mvosr_delaunay_frames_per_cu(0)) == 8
    assert int(lib.mvosr_delaunay_frames_per_cu(top + 1)) == 1 and int(lib.mvosr_delaunay_frames_per_cu(20000)) == 1


def test_slew_median_host_equals_the_references_recurrence():
    """mvosr_slew_median_host (no GPU): the slew limiter and window median of /root/reference/src/rescale.py:169-178 written
    out in Python — jumps beyond +-0.3, frames without a plane, a NaN scale that sticks, a carried-in queue."""
    from collections import deque
    from mvoscalerecovery_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(8)
    for n in (1, 5, 63, 64, 65, 1000):
        raw = rng.uniform(0.5, 3.5, n)
        raw[rng.random(n) < 0.1] += 5.0
        apply = (rng.random(n) > 0.2).astype(np.int32)
        raw[apply == 0] = np.nan
        if n == 1000:
            raw[700], apply[700] = np.nan, 1                      # a NaN plane: the reference's comparisons fail, the scale becomes NaN
        q_in, s, w = [1.25, 1.5], 1.5, 5
        q, want_p, want_f = deque(q_in), [], []
        for i in range(n):
            if apply[i]:
                if raw[i] - s > 0.3:
                    s += 0.3
                elif raw[i] - s < -0.3:
                    s -= 0.3
                else:
                    s = raw[i]
            q.append(s)
            if len(q) > w:
                q.popleft()
            want_p.append(s)
            with np.errstate(all="ignore"):
                want_f.append(np.median(q))
        p, f, qa, s_out = np.empty(n), np.empty(n), np.array(q_in), np.zeros(1)
        _lib.check(lib.mvosr_slew_median_host(_lib.addr(raw), _lib.addr(apply), n, 0.3, 1.5, w, _lib.addr(qa), 2, _lib.addr(p), _lib.addr(f),
                                              _lib.addr(s_out)))
        assert np.array_equal(p, np.array(want_p), equal_nan=True) and np.array_equal(f, np.array(want_f), equal_nan=True), n
        assert (np.isnan(s_out[0]) and np.isnan(want_p[-1])) or s_out[0] == want_p[-1]


def test_lds_plan_three_frames_per_cu():
    from mvoscalerecovery_amd import _lib
    lib = _lib.load()
    assert 26 * 2000 < lib.mvosr_lds_bytes(2000) < 54 * 1024        # 26 B per feature + ~1.5 KB
    assert 3 * lib.mvosr_lds_bytes(2000) <= 160 * 1024
    assert lib.mvosr_lds_bytes(300) < 10 * 1024


def test_ctypes_structs_match_the_header(tmp_path):
    """The ctypes mirrors of mvosr_params / mvosr_batch / mvosr_outputs have the header's size and field offsets
    (checked with the C compiler: the structs grew twice in round 2)."""
    import ctypes as C
    from mvoscalerecovery_amd import _lib
    fields = {"mvosr_params": [n for n, _ in _lib.Params._fields_], "mvosr_batch": [n for n, _ in _lib.Batch._fields_],
              "mvosr_outputs": [n for n, _ in _lib.Outputs._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "mvosr.h"', 'int main(void) {']
    for st, names in fields.items():
        src.append('printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for n in names:
            src.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, n, st, n))
    src += ['printf("abi %d\\n", MVOSR_ABI_VERSION);', 'return 0; }']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    assert int(got["abi"]) == _lib.ABI_VERSION
    for st, cls in (("mvosr_params", _lib.Params), ("mvosr_batch", _lib.Batch), ("mvosr_outputs", _lib.Outputs)):
        assert int(got[st]) == C.sizeof(cls), st
        for n, _ in cls._fields_:
            assert int(got["%s.%s" % (st, n)]) == getattr(cls, n).offset, (st, n)


def test_batch_size_hint_is_host_only():
    """mvosr_batch_size_hint needs no GPU: min / max / class counts of a batch from the host's copy of feat_cnt
    (classes: one wavefront per frame up to 320 features, four up to 1024, eight or sixteen above)."""
    import ctypes as C
    from mvoscalerecovery_amd import _lib
    lib = _lib.load()
    cnt = np.array([0, 5, 320, 321, 1024, 1025, 1500, 300, 2000], dtype=np.int32)
    b = _lib.Batch()
    assert lib.mvosr_batch_size_hint(cnt.ctypes.data, len(cnt), C.byref(b)) == 0
    assert (b.min_feat, b.max_feat) == (0, 2000)
    assert list(b.size_hint)[:3] == [4, 2, 3] and b.size_hint[3] != 0
    assert lib.mvosr_batch_size_hint(None, 3, C.byref(b)) != 0                     # null counts: refused, error text set
    assert b"batch_size_hint" in lib.mvosr_last_error()
    bad = np.array([3, -1], dtype=np.int32)
    assert lib.mvosr_batch_size_hint(bad.ctypes.data, 2, C.byref(b)) != 0


def test_tile_layout_far_table_is_a_copy_of_the_planes():
    """packing.apply_tile_order + attach_tri2(feature_ids=True): the far rows' vertex table holds, per far row, the
    planes' own values of its three vertices — (y, z, v) for tri1, (x, y, z) for tri2 — and every frame starts on a
    128-byte line of the planes."""
    from scipy.spatial import Delaunay
    from mvoscalerecovery_amd import packing, synth
    frames = [synth.synth_frame(i, n, base_seed=5) for i, n in enumerate((3000, 700, 2500))]
    pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames])
    assert all(int(o) % 16 == 0 for o in pf.feat_off)
    packing.attach_tri1(pf, None)
    packing.apply_tile_order(pf)
    masks = [np.ones(int(n), dtype=bool) for n in pf.feat_cnt]
    tri2 = [Delaunay(np.stack([pf.u[pf.frame_slice(f)], pf.v[pf.frame_slice(f)]], axis=1)).simplices.astype(np.int32) for f in range(3)]
    packing.attach_tri2(pf, tri2, masks, feature_ids=True)
    assert pf.tile_far is not None and len(pf.tile_far_off) == 4
    for f in range(3):
        sl = pf.frame_slice(f)
        x, y, z, v = pf.x[sl], pf.y[sl], pf.z[sl], pf.v[sl]
        nt = len(pf.tile1_off[int(pf.tile_base[f]):int(pf.tile_base[f + 1])]) - 1
        far1 = pf.tri1[int(pf.tri1_off[f]) + int(pf.tile1_off[int(pf.tile_base[f]) + nt]):int(pf.tri1_off[f + 1])]
        far2 = pf.tri2[int(pf.tri2_off[f]) + int(pf.tile2_off[int(pf.tile_base[f]) + nt]):int(pf.tri2_off[f + 1])]
        blk = pf.tile_far[int(pf.tile_far_off[f]):int(pf.tile_far_off[f + 1])]
        assert len(blk) == 9 * (len(far1) + len(far2))
        want = np.concatenate([np.stack([y[far1], z[far1], v[far1]], axis=2).reshape(-1),
                               np.stack([x[far2], y[far2], z[far2]], axis=2).reshape(-1)]) if len(blk) else np.zeros(0)
        assert np.array_equal(blk, want)
    assert int(pf.tile_far_off[-1]) > 0                     # (the 3000-feature frame has hull slivers across tiles)


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: constructing the estimator without a device raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mvoscalerecovery_amd import _lib
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    with pytest.raises(_lib.MvosrLibraryError, match="no HIP device|device"):
        ScaleEstimator(1.75, window_size=5)


def test_height_level_attribute_is_exact_on_read():
    """``ScaleEstimator.height_level`` (/root/reference/src/scale_calculator.py:217,:241 leave it on the estimator): a plain
    attribute to its users — AttributeError before the first frame, what was assigned afterwards — whose value may be PENDING
    after a per-frame call of the reference-exact path (known in the kernel's summation order only): the pending computation
    runs once, when the attribute is read, and a later assignment replaces it unread.  (Host logic only: no device.)"""
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    est = ScaleEstimator.__new__(ScaleEstimator)
    with pytest.raises(AttributeError, match="height_level"):
        est.height_level
    assert getattr(est, "height_level", None) is None and not hasattr(est, "height_level")
    est.height_level = 1.5
    assert est.height_level == 1.5 and hasattr(est, "height_level")
    calls = []
    est.__dict__["_level_thunk"] = lambda: (calls.append(1), 2.25)[1]
    assert est.height_level == 2.25 and est.height_level == 2.25 and calls == [1]
    est.__dict__["_level_thunk"] = lambda: (calls.append(2), 9.0)[1]
    est.height_level = 3.0                                  # the next frame's level: the pending one is never computed
    assert est.height_level == 3.0 and calls == [1]


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mvoscalerecovery_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), fn


def test_pack_features_layout():
    from mvoscalerecovery_amd import packing, synth
    frames = [synth.synth_frame(i, 101 + 50 * i, base_seed=1, upper_fraction=0.2) for i in range(5)]
    pf = packing.pack_features([f[0] for f in frames], [f[1] for f in frames])
    assert pf.n_frames == 5
    assert np.all(pf.feat_off % 2 == 0)
    for f, (f3, f2) in enumerate(frames):
        low = f2[:, 1] > 185
        sl = pf.frame_slice(f)
        assert pf.feat_cnt[f] == low.sum()
        assert np.array_equal(pf.x[sl], f3[low, 0])
        assert np.array_equal(pf.y[sl], f3[low, 1])
        assert np.array_equal(pf.z[sl], f3[low, 2])
        assert np.array_equal(pf.v[sl], f2[low, 1])
        assert np.array_equal(pf.lower_index[f], np.nonzero(low)[0])
    packing.attach_tri1(pf)
    assert pf.tri1_off[-1] == pf.tri1.shape[0]
    assert pf.tri1.dtype == np.int32
    for f in range(5):
        t = pf.tri1[pf.tri1_off[f]:pf.tri1_off[f + 1]]
        assert t.max() < pf.feat_cnt[f]
    tiled = packing.tile_frames(pf, 3)
    assert tiled.n_frames == 15
    for r in range(3):
        for f in range(5):
            a, b = pf.frame_slice(f), tiled.frame_slice(r * 5 + f)
            assert np.array_equal(pf.y[a], tiled.y[b])
            ta = pf.tri1[pf.tri1_off[f]:pf.tri1_off[f + 1]]
            tb = tiled.tri1[tiled.tri1_off[r * 5 + f]:tiled.tri1_off[r * 5 + f + 1]]
            assert np.array_equal(ta, tb)


def test_algorithmic_bytes_formula():
    """8*(3N+N) + 12*(T1+T2) + 12 -> 157,132 B at N=2000/T1=3981/T2=3779."""
    from mvoscalerecovery_amd import packing
    pf = packing.PackedFrames(1, np.zeros(1, np.int64), np.array([2000], np.int32), *([np.zeros(2000)] * 5), [None])
    pf.tri1_off = np.array([0, 3981])
    pf.tri2_off = np.array([0, 3779])
    assert pf.algorithmic_bytes() == 157132


def test_driver_loop_gates_with_injected_estimator():
    """main_offline.py:57-88 semantics with a scripted estimator."""
    from mvoscalerecovery_amd import offline

    class Fake:
        def __init__(self):
            self.k = 0

        def initial_estimation(self, t):
            return float(t[1])

        def scale_calculation(self, f3, f2):
            self.k += 1
            return 10.0 + self.k, 1

        def scale_calculation_batch(self, f3s, f2s):
            return [11.0 + i for i in range(len(f3s))], [1] * len(f3s)

    def feats(n):
        return np.zeros((n, 3)), np.zeros((n, 2))
    motions = [np.arange(12.0)] * 6
    f3 = [feats(150)[0], [], feats(100)[0], feats(101)[0], feats(5)[0], feats(300)[0]]
    f2 = [feats(150)[1], [], feats(100)[1], feats(101)[1], feats(5)[1], feats(300)[1]]
    data = {"motions": motions, "move_flags": [True, False, True, True, True, True], "feature3ds": f3, "feature2ds": f2}
    res = offline.run_sequence(data, Fake())
    assert res["kinds"].tolist() == [1, 0, 2, 1, 2, 1]
    assert res["scales"].tolist() == [11.0, 0.0, 0.0, 12.0, 12.0, 13.0]
    assert res["error"].tolist() == [100.0, 1.0, 0.0, 0.0, 1.0, 1.0, 1.0]
    assert res["pitchs"].tolist() == [7.0, 7.0, 7.0]
    resb = offline.run_sequence_batched(data, Fake())
    assert resb["scales"].tolist() == res["scales"].tolist()
    assert resb["error"].tolist() == res["error"].tolist()


def test_pose_integration_matches_reference_formulas():
    from mvoscalerecovery_amd import offline
    rng = np.random.default_rng(0)
    n = 6
    motions = []
    for _ in range(n):
        a = rng.normal(0, 0.05)
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        m = np.zeros((3, 4))
        m[:, :3] = R
        m[:, 3] = rng.normal(0, 1, 3)
        motions.append(m.reshape(-1))
    motions = np.array(motions)
    scales = rng.uniform(0.5, 2, n)
    poses = offline.get_path(motions, scales)
    assert poses.shape == (n + 1, 12)
    assert poses[0].tolist() == [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]
    T = np.eye(4)
    for i in range(n):
        M = np.eye(4)
        M[:3] = motions[i].reshape(3, 4)
        M[:3, 3] *= scales[i]
        T = T @ M
        np.testing.assert_allclose(poses[i + 1], T[:3].reshape(-1), atol=1e-14)


def test_sequence_dict_roundtrip(tmp_path):
    from mvoscalerecovery_amd import offline, synth
    data = synth.synth_sequence_dict(8, base_seed=3, n_lo=110, n_hi=130)
    p = str(tmp_path / "seq_result.npy.test")
    offline.save_sequence_dict(p, data)
    back = offline.load_sequence_dict(p)
    assert list(back) == ["motions", "move_flags", "feature2ds", "feature3ds"]
    assert np.array_equal(back["feature3ds"][0], data["feature3ds"][0])


def test_delaunay_pool_equals_in_process():
    """The shared-memory process pool returns SciPy's rows verbatim, in order, exceptions included —
    also right after its segments had to grow (workers must re-map, not read a stale mapping)."""
    from mvoscalerecovery_amd import packing
    rng = np.random.default_rng(5)
    sets = [np.column_stack([rng.uniform(0, 1241, n), rng.uniform(185, 376, n)]) for n in rng.integers(20, 900, 60)]
    sets[7] = sets[7][:2]                                        # too few points: QhullError
    want = packing.delaunay_many(sets, 0)
    try:
        small = packing.delaunay_many(sets[:6], 3)                # creates small segments
        big = packing.delaunay_many([np.tile(s, (1, 1)) for s in sets] * 3, 3)      # forces them to grow
        again = packing.delaunay_many(sets, 3)
    finally:
        packing.shutdown_pool()
    for got, ref in ((small, want[:6]), (big, want * 3), (again, want)):
        assert len(got) == len(ref)
        for g, r in zip(got, ref):
            if isinstance(r, Exception):
                assert type(g) is type(r)
            else:
                assert g.dtype == np.int32 and np.array_equal(g, r)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("mvosr_%d_" % os.getpid())]


def test_available_cpus_is_sane():
    from mvoscalerecovery_amd import packing
    n = packing.available_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert packing.resolve_workers(None) == n and packing.resolve_workers(0) == 0 and packing.resolve_workers(5) == 5
    os.environ["LOCAL_WORLD_SIZE"] = "4"
    try:
        assert packing.resolve_workers(None) == max(1, n // 4)
    finally:
        del os.environ["LOCAL_WORLD_SIZE"]


def _rows_as_pixels(pf, f, tri, ids_are_features, valid=None):
    """Every triangle row as its three (u, v) pixel pairs, in vertex order (what must survive any relabelling)."""
    sl = pf.frame_slice(f)
    u, v = pf.u[sl], pf.v[sl]
    if not ids_are_features:
        keep = np.nonzero(valid)[0]
        u, v = u[keep], v[keep]
    return np.stack([u[tri], v[tri]], axis=-1)


def test_relabellings_keep_every_triangle():
    """packing's two relabellings of the second triangulation — over the Z-ordered survivors (dense frames)
    and over the frame's features (mvosr_batch.tri2_ids = 1) — describe exactly SciPy's triangles: same
    pixel coordinates per row and per vertex position, rows only permuted."""
    from scipy.spatial import Delaunay
    from mvoscalerecovery_amd import packing, synth
    rng = np.random.default_rng(11)
    frames = [synth.synth_frame(i, n, base_seed=31, upper_fraction=0.1) for i, n in enumerate((400, 900, 150))]
    f3s, f2s = [f[0] for f in frames], [f[1] for f in frames]

    def canon(rows):                       # rows as a sorted list of byte strings: order of rows is free
        return sorted(r.tobytes() for r in np.ascontiguousarray(rows))

    for locality in (False, True):
        for feature_ids in (False, True):
            pf = packing.pack_features(f3s, f2s)
            packing.attach_tri1(pf)
            plain = packing.pack_features(f3s, f2s)                      # the untouched layout, for reference
            packing.attach_tri1(plain)
            if locality:
                packing.apply_locality_order(pf, min_features=1)
            masks = [rng.uniform(0, 1, int(c)) > 0.07 for c in pf.feat_cnt]            # in the PACKED order
            packing.attach_tri2(pf, None, masks, feature_ids=feature_ids)
            assert pf.tri2_ids == int(feature_ids)
            for f in range(pf.n_frames):
                sl = pf.frame_slice(f)
                # first triangulation: same triangles as SciPy's on the caller's order
                t1 = pf.tri1[pf.tri1_off[f]:pf.tri1_off[f + 1]]
                t1_ref = plain.tri1[plain.tri1_off[f]:plain.tri1_off[f + 1]]
                assert canon(_rows_as_pixels(pf, f, t1, True)) == canon(_rows_as_pixels(plain, f, t1_ref, True))
                # second triangulation: SciPy on the survivors taken in the CALLER's order
                perm = (pf.extra.get("perm") or [None] * pf.n_frames)[f]
                m = masks[f]
                m_orig = m.copy()
                if perm is not None:
                    m_orig = np.empty_like(m)
                    m_orig[perm] = m
                psl = plain.frame_slice(f)
                pts = np.stack([plain.u[psl][m_orig], plain.v[psl][m_orig]], axis=1)
                want = pts[Delaunay(pts).simplices]
                t2 = pf.tri2[pf.tri2_off[f]:pf.tri2_off[f + 1]]
                got = _rows_as_pixels(pf, f, t2, feature_ids, m)
                assert canon(got) == canon(want), (locality, feature_ids, f)
                if feature_ids:
                    assert np.all(m[t2])                                 # every vertex is a survivor
                assert pf.n2_expected[f] == int(m.sum())


def test_canonical_rows_key_sort_equals_lexsort():
    """packing.canonical_rows (round 6: one 63-bit key per row) against the plain np.sort + np.lexsort form, including ids beyond the
    key's 21 bits (the fallback) and an empty triangulation."""
    from mvoscalerecovery_amd import packing
    rng = np.random.default_rng(5)

    def plain(t):
        t = np.sort(np.asarray(t, dtype=np.int32).reshape(-1, 3), axis=1)
        return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))] if len(t) else t
    for hi in (50, 4000, (1 << 21) - 1, 1 << 22):
        t = rng.integers(0, hi + 1, (3000, 3)).astype(np.int32)
        got = packing.canonical_rows(t)
        assert got.dtype == np.int32 and got.flags.c_contiguous and np.array_equal(got, plain(t)), hi
    assert packing.canonical_rows(np.zeros((0, 3), np.int32)).shape == (0, 3)


def test_exact_mask_lazy_last():
    """engine.exact_mask_of: the chunk's last frame with more than three features below the vanishing row is on the mask, and every
    frame directly followed by a three-feature frame; lazy_last drops only the former (round 6)."""
    from mvoscalerecovery_amd.engine import exact_mask_of
    cnt = np.array([900, 3, 800, 700, 3, 3, 600, 500])
    assert exact_mask_of(cnt).tolist() == [1, 0, 0, 1, 0, 0, 0, 1]
    assert exact_mask_of(cnt, lazy_last=True).tolist() == [1, 0, 0, 1, 0, 0, 0, 0]
    tail = np.array([900, 800, 3])
    assert exact_mask_of(tail).tolist() == exact_mask_of(tail, lazy_last=True).tolist() == [0, 1, 0]      # (held by the other rule)
    assert exact_mask_of(np.array([3, 3]), lazy_last=True).sum() == 0 and exact_mask_of(cnt, everything=True).all()


def test_delaunay_submit_fast_and_canonical_inline():
    """packing.delaunay_submit(workers=0, fast=True, canonical=True): the host replay where it accepts a set, SciPy where it declines,
    rows in canonical form from the job itself (the handle says so, attach_* then skips its own pass)."""
    from mvoscalerecovery_amd import packing, synth
    sets = [synth.synth_frame(i, 200 + 150 * i, base_seed=99)[1] for i in range(5)]
    sets[2] = np.round(sets[2] * 4) / 4                                   # (declined by the replay: SciPy's rows)
    sets.append(np.stack([np.linspace(0, 100, 20), np.linspace(190, 300, 20)], axis=1))          # collinear: SciPy raises
    h = packing.delaunay_submit(sets, 0, fast=True, canonical=True)
    assert h.canonical
    rows = h.get()
    for p, r in zip(sets[:5], rows[:5]):
        assert np.array_equal(r, packing.canonical_rows(packing.delaunay_simplices(p)))
    assert isinstance(rows[5], Exception)
    plain = packing.delaunay_submit(sets[:5], 0, fast=True).get()
    for p, r in zip(sets[:5], plain):
        assert np.array_equal(r, packing.delaunay_simplices(p))


def test_delaunay_submit_background_single_set():
    """packing.delaunay_submit(background=True): ONE set goes to the worker pool too (a declined frame found in the middle of a batch call:
    its SciPy call off the thread that launches the next chunks); the handle says when get() would not wait; same rows as the inline call."""
    import time
    from mvoscalerecovery_amd import packing, synth
    pts = synth.synth_frame(3, 500, base_seed=12)[1]
    inline = packing.delaunay_submit([pts], 2)
    assert inline.ready() and inline._async is None                       # (a single set without the flag: triangulated in the call)
    h = packing.delaunay_submit([pts], 2, slot=9, background=True)
    assert h._async is not None
    t0 = time.time()
    while not h.ready() and time.time() - t0 < 30:
        time.sleep(0.005)
    assert h.ready()
    assert np.array_equal(h.get()[0], inline.get()[0]) and np.array_equal(h.get()[0], packing.delaunay_simplices(pts))
    assert packing.delaunay_submit([pts], 0, background=True).ready()      # (no pool: inline whatever the flag)


def test_joined_delaunay_handle():
    """packing._JoinedHandle: two submissions seen as one list in the caller's order (the last chunk's early-found declined frames and the
    rest of its frames to redo: scale_calculator._chunk_gpu_finish)."""
    from mvoscalerecovery_amd import packing, synth
    sets = [synth.synth_frame(i, 120 + 40 * i, base_seed=5)[1] for i in range(5)]
    a = packing.delaunay_submit([sets[3], sets[0]], 0)
    b = packing.delaunay_submit([sets[1], sets[2], sets[4]], 0, canonical=False)
    j = packing._JoinedHandle(5, [([3, 0], a), ([1, 2, 4], b)])
    assert j.ready() and not j.canonical
    rows = j.get()
    for p, r in zip(sets, rows):
        assert np.array_equal(r, packing.delaunay_simplices(p))
    assert packing._JoinedHandle(2, [([0], packing.delaunay_submit([sets[0]], 0, canonical=True)),
                                     ([1], packing.delaunay_submit([sets[1]], 0, canonical=True))]).canonical
