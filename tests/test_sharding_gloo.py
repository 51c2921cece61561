"""Multi-GPU path on CPU: frame partition + all-gather + window median with world_size 2 (gloo).

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
    assert world == 2
    n = %(n)d
    rng = np.random.default_rng(5)
    raw_all = rng.uniform(0.5, 3.0, n)
    raw_all[17] = np.nan
    st_all = rng.integers(0, 5, n).astype(np.int32)
    a, b = sharding.partition(n, world, rank)

    def median_fn(raw, window, queue=()):
        out, _ = so.window_median(raw.numpy(), window, queue)
        return torch.from_numpy(out)

    filt, raw, st = sharding.gather_and_filter(torch.from_numpy(raw_all[a:b].copy()), torch.from_numpy(st_all[a:b].copy()),
                                               n, 5, median_fn, queue=[1.0, 2.0])
    want, _ = so.window_median(raw_all, 5, [1.0, 2.0])
    assert np.array_equal(raw.numpy(), raw_all, equal_nan=True)
    assert np.array_equal(st.numpy(), st_all)
    assert np.array_equal(filt.numpy(), want, equal_nan=True)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


@pytest.mark.parametrize("n", [1000, 1001])
def test_world_size_2_gather_and_filter(tmp_path, n):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "n": n})
    port = 29600 + (os.getpid() + n) % 300
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out
