#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q -x > $OUT/r06_gputest_d.log 2>&1
tail -3 $OUT/r06_gputest_d.log
timeout 900 python bench.py --workload gridded --snap-fraction 0.005 --no-cpu-baseline > $OUT/r06_bench_gridded_0005.json 2> $OUT/r06_bench_gridded_0005.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06_bench_gridded_0005.json').read().splitlines() if l.startswith('{')][-1])
for k in ('e2e_gpu_exact', 'e2e_gpu_triangulation', 'e2e'):
    e = d.get(k, {})
    print(k, {x: e.get(x) for x in ('value', 'frames', 'declined_total', 'declined_fraction', 'error')})
PY
SNAP_FRACTION=0.005 python profiles/e2e_gpu_profile.py 16384 2000 scale 2>&1 | head -30 > $OUT/r06_redo_profile_fixed.txt; head -22 $OUT/r06_redo_profile_fixed.txt
