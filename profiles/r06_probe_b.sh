#!/bin/bash
# round 6, second GPU call: the suite (mode matrix included), the per-frame soak against the pure SciPy estimator, the gridded workload
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q > $OUT/r06_gputest_b.log 2>&1
tail -4 $OUT/r06_gputest_b.log
timeout 600 python profiles/soak_single_exact.py 6000 300 2000 > $OUT/r06_soak_single_exact.txt 2>&1
tail -2 $OUT/r06_soak_single_exact.txt
timeout 900 python bench.py --workload gridded --no-cpu-baseline > $OUT/r06_bench_gridded.json 2> $OUT/r06_bench_gridded.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r06_bench_gridded.json').read().strip().splitlines()[-1])
    for k in ('e2e_gpu_exact', 'e2e_gpu_triangulation', 'e2e'):
        e = d.get(k, {})
        print(k, {x: e.get(x) for x in ('value', 'frames', 'declined_total', 'declined_fraction', 'error')})
except Exception as exc:
    print("gridded bench:", exc); print(open('gpurun_out/r06_bench_gridded.err').read()[-2000:])
PY
