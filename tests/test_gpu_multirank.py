"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors of the reference —
the N-rank paths (SURVEY §8 e): RCCL at world size 1, gloo ranks sharing this GPU, bench.py's contract and dry runs.  Needs a real MI355X:  python -m pytest tests -m gpu

Constructions say which path they mean: ``triangulation="scipy"`` is the host-SciPy baseline every device path is compared with; a
construction without the keyword IS the shipped default (triangulation "gpu" with the reference's vote)."""
import os

import numpy as np
import pytest

from conftest import load_json, load_npz
from gpu_helpers import _device_count

pytestmark = pytest.mark.gpu


def test_rccl_gather_and_gpu_median_world1(gpu, tmp_path):
    """The multi-GPU step on one GPU: torch.distributed `nccl` (= RCCL) group of size 1, all-gather of
    device tensors, window-median kernel on the stream torch and the context share."""
    import subprocess
    import sys
    import textwrap
    from conftest import ROOT
    script = tmp_path / "w1.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        import numpy as np
        import torch
        import torch.distributed as dist
        sys.path.insert(0, %r)
        from mvoscalerecovery_amd import _lib, sharding
        from mvoscalerecovery_amd.engine import ScaleEngine
        from oracle import scale_oracle as so
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        ctx = _lib.Context(0)
        stream = torch.cuda.Stream(device=0)
        torch.cuda.set_stream(stream)
        ctx.set_stream(stream.cuda_stream)
        eng = ScaleEngine(1.75, ctx=ctx)
        rng = np.random.default_rng(1)
        raw = rng.uniform(0.5, 3.0, 10001)
        st = rng.integers(0, 5, 10001).astype(np.int32)
        lvl = rng.uniform(-1.0, 1.0, 10001)
        rec = sharding.RankRecord(10001, torch.device("cuda", 0)).fill(raw, st, lvl)
        filt, g = sharding.gather_and_filter(rec, 10001, 5, sharding.make_gpu_median(eng), queue=[2.0])
        torch.cuda.synchronize()
        want, _ = so.window_median(raw, 5, [2.0])
        assert np.array_equal(filt.cpu().numpy(), want)
        assert np.array_equal(g.raw().cpu().numpy(), raw) and np.array_equal(g.status().cpu().numpy(), st)
        assert np.array_equal(g.level().cpu().numpy(), lvl)
        dist.destroy_process_group()
        print("world1 ok")
    """ % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29733")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "world1 ok" in p.stdout, p.stdout + p.stderr


def test_one_million_frames_two_ranks_gather_and_median(gpu, tmp_path):
    """Config C4's cross-rank half at full size: 1 000 000 frames, two ranks (both on this GPU: MVOSR_SHARE_GPU, gloo
    staging through the host), each rank's record all-gathered with ONE collective and the window-median kernel
    reading the gathered buffer in place; checked against NumPy's sliding median of the whole sequence."""
    import subprocess
    import sys
    import textwrap
    from conftest import ROOT
    script = tmp_path / "w2.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        import numpy as np
        import torch
        import torch.distributed as dist
        sys.path.insert(0, %r)
        from mvoscalerecovery_amd import _lib, sharding
        from mvoscalerecovery_amd.engine import ScaleEngine
        from oracle import scale_oracle as so
        rank, local, world = sharding.init_distributed("gloo")
        assert world == 2
        n = 1000001                                   # ragged: rank 0 holds one frame more
        rng = np.random.default_rng(2)
        raw_all = rng.uniform(0.5, 3.0, n)
        raw_all[[5, 500000, 999999]] = np.nan
        st_all = rng.integers(0, 5, n).astype(np.int32)
        lvl_all = rng.uniform(-1.0, 1.0, n)
        a, b = sharding.partition(n, world, rank)
        dev = torch.device("cuda", local)
        ctx = _lib.Context(local)
        stream = torch.cuda.Stream(device=local)
        torch.cuda.set_stream(stream)
        ctx.set_stream(stream.cuda_stream)
        eng = ScaleEngine(1.75, ctx=ctx)
        rec = sharding.RankRecord(max(sharding.shard_sizes(n, world)), dev).fill(raw_all[a:b], st_all[a:b], lvl_all[a:b])
        c0 = sharding.collectives_issued
        filt, g = sharding.gather_and_filter(rec, n, 5, sharding.make_gpu_median(eng), queue=[2.0, 1.0])
        torch.cuda.synchronize()
        assert sharding.collectives_issued == c0 + 1
        got = filt.cpu().numpy()
        head, _ = so.window_median(raw_all[:16], 5, [2.0, 1.0])
        assert np.array_equal(got[:16], head, equal_nan=True)
        win = np.lib.stride_tricks.sliding_window_view(raw_all, 5)
        want = np.median(win, axis=1)                 # (NaN propagates, like np.median of the deque)
        assert np.array_equal(got[4:], want, equal_nan=True)
        assert np.array_equal(g.status().cpu().numpy(), st_all) and np.array_equal(g.level().cpu().numpy(), lvl_all)
        dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % ROOT))
    port = 29900 + os.getpid() % 90
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MVOSR_SHARE_GPU="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out


def test_sharded_sequence_driver_two_ranks_one_gpu(gpu, tmp_path):
    """Config C4 in driver form: offline.run_sequence_sharded with the real ScaleEstimator on two ranks
    (gloo, both on this GPU: MVOSR_SHARE_GPU) reproduces the reference's 200-frame golden on every rank;
    without a process group the same call is the single-rank replay."""
    import subprocess
    import sys
    import textwrap
    from conftest import ROOT
    from mvoscalerecovery_amd import offline, synth
    from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
    z, meta = load_npz("seq200.npz")
    data = synth.synth_sequence_dict(meta["n_frames"], base_seed=meta["seed"], **meta["kw"])
    res = offline.run_sequence_sharded(data, ScaleEstimator(meta["abs_ref"], window_size=meta["window"], mutate_inputs=False, triangulation="scipy",
                                                            delaunay_workers=4))
    np.testing.assert_array_equal(res["scales"], z["scales"])
    np.testing.assert_array_equal(res["error"], z["error"])
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %(root)r)
        sys.path.insert(0, os.path.join(%(root)r, "tests"))
        import numpy as np
        import torch.distributed as dist
        from conftest import load_npz
        from mvoscalerecovery_amd import offline, sharding, synth
        from mvoscalerecovery_amd.scale_calculator import ScaleEstimator
        rank, local, world = sharding.init_distributed("gloo")
        z, meta = load_npz("seq200.npz")
        data = synth.synth_sequence_dict(meta["n_frames"], base_seed=meta["seed"], **meta["kw"])
        est = ScaleEstimator(meta["abs_ref"], window_size=meta["window"], mutate_inputs=False, delaunay_workers=4, triangulation="scipy")
        res = offline.run_sequence_sharded(data, est)
        assert np.array_equal(res["scales"], z["scales"]) and np.array_equal(res["error"], z["error"])
        assert np.array_equal(res["pitchs"], z["pitchs"]) and np.array_equal(res["kinds"], z["kinds"])
        dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """) % {"root": ROOT})
    port = 29800 + os.getpid() % 150
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MVOSR_SHARE_GPU="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for rank, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out


def test_bench_line_contract(gpu):
    """bench.py prints ONE JSON line with the driver's keys, a roofline block whose numbers are consistent with each
    other, and a cpu_baseline block; a second run with --gpus 2 on this 1-GPU box refuses loudly (no silent single rank)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--frames", "8192", "--pool", "64", "--steps", "3", "--warmup", "1",
                        "--no-e2e"], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "frames/s" and d["dtype"] == "f64"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert 0.05 < r["frac"] < 1.0 and r["kernel_ms_avg"] < d["ms_per_step"] * 1.05
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "frames/s"
    if _device_count() == 1:
        p2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--frames", "4096", "--pool", "32", "--steps", "1",
                             "--warmup", "1", "--no-e2e", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
        assert p2.returncode != 0 and not [ln for ln in p2.stdout.splitlines() if ln.startswith("{")]


def test_bench_multi_rank_dry_runs(gpu):
    """The N-rank paths of bench.py as dry runs on this box (--share-gpu: gloo, ranks share the device): BASELINE
    configs[3] literally (--c4: a fixed number of frames SPLIT over the ranks, ragged blocks, one collective per step,
    "strong"), configs[4]'s dense frames on two ranks, and a rank that fails: the launcher reports it and exits non-zero
    instead of leaving rank 0 in the all-gather."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--share-gpu", "--steps", "2", "--warmup", "1", "--no-e2e", "--no-cpu-baseline"]
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--c4", "--total-frames", "10001", "--pool", "64"] + common,
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 3 and d["world_size"] == 3 and d["collectives_per_step"] == 1.0
    assert d["config"]["frames_per_step_total"] == 10001 and d["config"]["frames_per_step_per_gpu"] == 3334
    assert abs(d["value"] - 10001 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    single = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--c4", "--total-frames", "10001", "--pool", "64",
                             "--steps", "2", "--warmup", "1", "--no-e2e", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert single.returncode == 0, single.stderr[-3000:]
    d1 = json.loads([ln for ln in single.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["raw_scale_crc32"] == d["raw_scale_crc32"] and d1["status_histogram"] == d["status_histogram"]     # the same job, however it is split
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--features", "20000", "--frames", "16", "--pool", "4"] + common,
                       capture_output=True, text=True, env=env, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["collectives_per_step"] == 1.0 and "tiled" in d["roofline"]["kernel"]
    env_bad = dict(env, MVOSR_BENCH_FAIL_RANK="1")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--frames", "2048", "--pool", "32", "--launch-timeout", "300"] + common,
                       capture_output=True, text=True, env=env_bad, timeout=900)
    assert p.returncode != 0 and "rank 1" in p.stderr and "injected failure" in p.stderr, p.stderr[-2000:]


def test_bench_n_rank_step_over_rccl_at_world_size_one(gpu):
    """The exact N-rank step of bench.py — scale kernels into the rank's record, ONE RCCL all-gather of the records, the
    window median over the gathered buffer read in place — at world size 1 over the real backend (MVOSR_BENCH_FORCE_GATHER):
    backend nccl (= RCCL), one collective per step, and the same raw scales as the ungathered single-GPU run; --c4
    (BASELINE configs[3]: a fixed job split over the ranks) the same way."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MVOSR_BENCH_FORCE_GATHER", "MVOSR_FORCE_DIST"):
        base.pop(k, None)
    common = ["--steps", "3", "--warmup", "1", "--no-e2e", "--no-cpu-baseline", "--pool", "64"]

    def run(extra, gathered):
        env = dict(base)
        if gathered:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MVOSR_BENCH_FORCE_GATHER="1",
                       MVOSR_FORCE_DIST="1")
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + extra + common, capture_output=True, text=True,
                           env=env, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])

    for extra in (["--frames", "4096"], ["--c4", "--total-frames", "10001"]):
        g, s = run(extra, True), run(extra, False)
        assert g["backend"] == "nccl" and g["world_size"] == 1 and g["collectives_per_step"] == 1.0, (g["backend"], g["collectives_per_step"])
        assert s["collectives_per_step"] == 0.0
        assert g["raw_scale_crc32"] == s["raw_scale_crc32"] and g["status_histogram"] == s["status_histogram"]


def test_bench_e2e_sharded_leg(gpu):
    """bench.py's `e2e_sharded` leg (VERDICT r4 #6a): offline.run_sequence_sharded from per-frame arrays with all three
    estimators — two ranks sharing this GPU (gloo), and one rank with --e2e-sharded; every scale finite, a number per estimator."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--frames", "2048", "--pool", "64"]
    for extra, world in ((["--gpus", "2", "--share-gpu"], 2), (["--e2e-sharded"], 1)):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra + common, capture_output=True, text=True, env=env, timeout=1500)
        assert p.returncode == 0, p.stderr[-3000:]
        d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
        sh = d["e2e_sharded"]
        assert "error" not in sh, sh
        for name in ("scale_fixed", "scale_exact", "rescale"):
            assert sh[name]["frames_total"] == sh[name]["frames_per_rank"] * world and sh[name]["scales_finite"] == sh[name]["frames_total"]
            assert sh[name]["value"] > 1000.0
