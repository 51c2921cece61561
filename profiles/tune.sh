#!/bin/bash
# Build-and-time sweep on the GPU box: bash profiles/tune.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for V in "$@"; do
  make -C mvoscalerecovery_amd/csrc clean >/dev/null
  make -C mvoscalerecovery_amd/csrc -j8 EXTRA="$V" 2>&1 | grep -E "error" 
  for W in ${WAVES_LIST:-8}; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --waves $W 2>&1 | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('VARIANT [$V] waves=$W', 'frames/s=%.3e'%d['value'], 'frac=%.3f'%d['roofline']['frac'], 'scale_ms=%.4f'%d['roofline']['kernel_ms_avg'], 'road_ms=%.4f'%d['roofline']['road_model_kernel_ms_avg'], 'step_frac=%.3f'%d['roofline']['step_frac'], d['status_histogram'])"
  done
done
