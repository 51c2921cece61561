#!/bin/bash
# round 6, first GPU call: the suite, the host replay's latency A/B, the qhull kernel with / without the prefetch of the partition's operands
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q > $OUT/r06_gputest_a.log 2>&1
tail -3 $OUT/r06_gputest_a.log
: > $OUT/r06_latency_a.txt
for hr in 1 0; do HOST_REPLAY=$hr timeout 120 python profiles/latency_probe.py exact 200 2000 2>&1 | sed "s/^/host_replay=$hr /" >> $OUT/r06_latency_a.txt; done
HOST_REPLAY=1 timeout 120 python profiles/latency_probe.py exact 200 900 2>&1 | sed "s/^/host_replay=1 n=900 /" >> $OUT/r06_latency_a.txt
python - >> $OUT/r06_latency_a.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
from mvoscalerecovery_amd import packing, synth
for n in (900, 2000):
    f2 = synth.synth_frame(1, n, base_seed=5000)[1]
    for fn in (packing.qhull_rows_host, packing.delaunay_simplices):
        fn(f2)
        t = time.perf_counter()
        for _ in range(100): fn(f2)
        print("host triangulation alone, %d points: %s %.3f ms" % (n, fn.__name__, (time.perf_counter() - t) / 100 * 1e3))
PY
cat $OUT/r06_latency_a.txt
: > $OUT/r06_qhull_ab.txt
for rep in 1 2; do
  for lib in "" profiles/ab/libmvosr_noprefetch.so; do
    MVOSR_LIB_PATH=${lib:+$R/$lib} QH_FRAMES=4096,16384 timeout 300 python profiles/qhull_gpu_check.py 2048 2000 2>&1 | grep -i "sets/s\|different" | sed "s|^|${lib:-product(prefetch)} |" >> $OUT/r06_qhull_ab.txt
  done
done
cat $OUT/r06_qhull_ab.txt
