#!/bin/bash
# Same-box A/B of the scale kernels: the product library against libraries under profiles/ab (profiles/ab_build.sh <tag> <flags>),
# three alternating passes.   AB_LIBS="vb0" [AB_ARGS="--workload kitti"] bash profiles/ab_kernel.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
one() {
  python bench.py --no-cpu-baseline --no-e2e $AB_ARGS 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 %s: %s %.4f ms, %.4f of the HBM peak, step %.4f ms, crc %s' % ('$AB_ARGS', r['kernel'], r['kernel_ms_avg'], r['frac'], d['ms_per_step'], d['raw_scale_crc32']))"
}
for i in 1 2 3; do
  one prod
  for l in ${AB_LIBS}; do MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_$l.so one $l; done
done
