#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q -x > $OUT/r06_gputest_g.log 2>&1
tail -3 $OUT/r06_gputest_g.log
python profiles/lazy_level_probe.py 16384 2>&1 | tail -5
for F in 16384 32768; do python profiles/e2e_gpu_profile.py $F 2000 exact 2>&1 | grep "frames/s"; done
python profiles/e2e_gpu_profile.py 32768 900 exact 2>&1 | grep "frames/s"
