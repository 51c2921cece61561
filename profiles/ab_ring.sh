#!/bin/bash
# Same-box A/B of the tiled dense kernel's LDS ring: three tiles / one barrier per tile (product) against two tiles / two barriers
# (profiles/ab_build.sh ring2 -DMVOSR_TILED_RING=2).  bash profiles/ab_ring.sh > gpurun_out/r03_ab_ring.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
one() { python bench.py --features ${N:-20000} --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 features ${N:-20000}: %s %.4f ms, %.4f of the HBM peak, step %.4f ms' % (r['kernel'], r['kernel_ms_avg'], r['frac'], d['ms_per_step']))"; }
for i in 1 2 3 4; do
  one ring3
  MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_ring2.so one ring2
done
for N in 8000 12000 32000; do
  export N
  one ring3
  MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_ring2.so one ring2
done
