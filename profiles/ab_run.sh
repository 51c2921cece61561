#!/bin/bash
# Same-box A/B of library builds (profiles/ab_build.sh): profiles/ab_run.sh "<bench args>" NAME1 NAME2 ...   ("main" = the product build)
ARGS=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
for NAME in "$@"; do
  if [ $NAME = main ]; then unset MVOSR_LIB_PATH; else export MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_$NAME.so; fi
  python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-12s kernel %.4f ms  frac %.4f  road %.4f ms  step %.4f  value %.0f' % ('$NAME', r['kernel_ms_avg'], r['frac'], r['road_model_kernel_ms_avg'], r['step_frac'], d['value']))"
done
done
