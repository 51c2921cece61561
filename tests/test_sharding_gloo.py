"""Multi-GPU path on CPU: frame partition + all-gather + window median with world sizes 2 and 8 (gloo).

The gather/filter code is the product's (mvoscalerecovery_amd/sharding.py); the median function is
injected — here the oracle's (the GPU kernel is checked against the same oracle in the gpu tests)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT


def test_partition_covers_everything():
    from mvoscalerecovery_amd import sharding
    for n in (0, 1, 7, 8, 9, 4541, 1000000):
        for w in (1, 2, 3, 4, 8):
            blocks = [sharding.partition(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for a, b in zip(blocks, blocks[1:]):
                assert a[1] == b[0]
            sizes = sharding.shard_sizes(n, w)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %(root)r)
    from mvoscalerecovery_amd import sharding
    from oracle import scale_oracle as so

    rank, local, world = sharding.init_distributed(backend="gloo")
    assert world == %(world)d
    n = %(n)d
    rng = np.random.default_rng(5)
    raw_all = rng.uniform(0.5, 3.0, n)
    raw_all[17] = np.nan
    st_all = rng.integers(0, 5, n).astype(np.int32)
    a, b = sharding.partition(n, world, rank)

    lvl_all = rng.uniform(-1.0, 1.0, n)

    def fast_median(raw, window, queue):
        # so.window_median, vectorised for the million-frame case: the first `window` outputs from the oracle's own loop
        # (the carried-in queue drains there), the rest as medians of full windows
        raw = np.asarray(raw)
        head, _ = so.window_median(raw[:window], window, queue)
        if len(raw) <= window:
            return head
        win = np.lib.stride_tricks.sliding_window_view(raw, window)[1:]
        return np.concatenate([head, np.median(win, axis=1)])

    chk = rng.uniform(0.5, 3.0, 300)
    assert np.array_equal(fast_median(chk, 5, [1.0, 2.0]), so.window_median(chk, 5, [1.0, 2.0])[0])

    def median_fn(g, window, queue=()):
        return torch.from_numpy(fast_median(g.raw().numpy(), window, queue))

    cap = max(sharding.shard_sizes(n, world))
    rec = sharding.RankRecord(cap, "cpu").fill(raw_all[a:b], st_all[a:b], lvl_all[a:b])
    before = sharding.collectives_issued
    filt, g = sharding.gather_and_filter(rec, n, 5, median_fn, queue=[1.0, 2.0])
    assert sharding.collectives_issued == before + 1          # ONE collective per step
    raw, st, lvl = g.raw(), g.status(), g.level()
    # the fields are fresh tensors, not views of the cached receive buffer: a second gather must not change them
    rec2 = sharding.RankRecord(cap, "cpu").fill(lvl_all[a:b], st_all[a:b], raw_all[a:b])
    sharding.all_gather_record(rec2, n)
    want = fast_median(raw_all, 5, [1.0, 2.0])
    assert g.cap == cap and sum(g.sizes) == n and g.sizes == sharding.shard_sizes(n, world)
    assert np.array_equal(raw.numpy(), raw_all, equal_nan=True)
    assert np.array_equal(st.numpy(), st_all)
    assert np.array_equal(lvl.numpy(), lvl_all)
    assert np.array_equal(filt.numpy(), want, equal_nan=True)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def _run_ranks(script, world, port, timeout):
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out


@pytest.mark.parametrize("world,n", [(2, 1000), (2, 1001), (8, 64), (8, 1000001)])
def test_gather_and_filter(tmp_path, world, n):
    """One all-gather of the ranks' records + the window median over the whole sequence, equal and ragged shards; at the
    target world size (8) with 1 000 001 frames the record capacity (125 001) differs from the shard size on seven ranks."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "n": n, "world": world})
    _run_ranks(script, world, 29600 + (os.getpid() + n + world) % 300, 400)


WORKER_SEQ = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np
    import torch.distributed as dist
    from mvoscalerecovery_amd import offline, sharding, synth
    from oracle import scale_oracle as so

    class OracleAdapter:
        # the two halves the sharded driver needs, on the CPU oracle
        def __init__(self):
            self.queue = []
        def initial_estimation(self, motion_t):
            return so.OracleScaleEstimator(1.75, window_size=5).initial_estimation(motion_t)
        def raw_scale_batch(self, f3s, f2s):
            rs = [so.frame_raw_scale(a.copy(), b, 1.75) for a, b in zip(f3s, f2s)]
            return (np.array([r.raw_scale for r in rs]), np.array([r.status for r in rs], dtype=np.int32),
                    np.array([r.height_level for r in rs]), {})
        def push_raw_scales(self, raw, status, level=None, host_errors=None):
            assert not host_errors and np.all(status < so.ST_ERR_LEFT)
            out, self.queue = so.window_median(np.asarray(raw), 5, self.queue)
            return out, np.where(status == so.ST_NO_FLAT, 100, 1).astype(float)

    rank, local, world = sharding.init_distributed("gloo")
    data = synth.synth_sequence_dict(%(n)d, base_seed=99, n_lo=120, n_hi=260)
    got = offline.run_sequence_sharded(data, OracleAdapter())
    want = offline.run_sequence(data, so.OracleScaleEstimator(1.75, window_size=5))
    for k in ("scales", "error", "pitchs", "kinds"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


@pytest.mark.parametrize("world,n", [(2, 41), (2, 42), (8, 83)])
def test_sharded_sequence_driver(tmp_path, world, n):
    """offline.run_sequence_sharded on two gloo ranks (CPU oracle behind the estimator interface) equals the
    frame-at-a-time replay of the whole sequence on one process, not-moving / too-few-feature frames included.
    41 frames give 39 processed ones (ragged shards), 42 give 40 (equal shards: the case in which round 1's
    two back-to-back gathers handed out the same cached buffer twice)."""
    script = tmp_path / "worker_seq.py"
    script.write_text(WORKER_SEQ % {"root": ROOT, "n": n})
    _run_ranks(script, world, 29300 + (os.getpid() + 7 * n + world) % 250, 400)
