#!/bin/bash
# Time the fused kernel with phases switched off (runtime flag, same binary): which phase costs what.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for M in 0 1 2 4 8 3 7 15 6 14; do
  MVOSR_DEBUG_SKIP=$M python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>&1 | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('SKIP=$M', 'kernel_ms=%.4f'%d['roofline']['kernel_ms_avg'], d['status_histogram'])"
done
