#!/bin/bash
# round 6: the deferred re-run of declined frames — the suite, then the few-declines workload (both estimators' legs)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q -x > $OUT/r06_gputest_c.log 2>&1
tail -3 $OUT/r06_gputest_c.log
timeout 900 python bench.py --workload gridded --snap-fraction 0.005 --no-cpu-baseline > $OUT/r06_bench_gridded_0005.json 2> $OUT/r06_bench_gridded_0005.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06_bench_gridded_0005.json').read().splitlines() if l.startswith('{')][-1])
for k in ('e2e_gpu_exact', 'e2e_gpu_triangulation', 'e2e'):
    e = d.get(k, {})
    print(k, {x: e.get(x) for x in ('value', 'frames', 'declined_total', 'declined_fraction', 'error')})
PY
