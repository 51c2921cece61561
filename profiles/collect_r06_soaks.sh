#!/bin/bash
# Round 6: the longer soaks on the final source (the suite's and collect_r06.sh's shorter ones aside): bash profiles/collect_r06_soaks.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd $R
timeout 500 python profiles/soak_gpu_path.py 300 > $OUT/r06_soak_fixed.txt 2>&1
timeout 900 python profiles/soak_rescale_device.py 12 64 > $OUT/r06_soak_rescale_device.txt 2>&1
SOAK_CHUNK=8192 timeout 900 python profiles/soak_rescale_device.py 2 600 >> $OUT/r06_soak_rescale_device.txt 2>&1
timeout 900 python profiles/soak_qhull.py 60000 > $OUT/r06_soak_qhull.txt 2>&1
timeout 700 python profiles/soak_gpu_path.py 420 exact > $OUT/r06_soak_exact.txt 2>&1
# delaunay_kernel against SciPy on coordinates quantised to 1/64, 1/16, 1/4 and 1 px (collinear triples, cocircular quadruples, repeated sites:
# what is accepted must be SciPy's triangle set) and on unquantised ones; the kernel as it was before the collinearity change on the same sets
: > $OUT/r06_soak_delaunay_quantised.txt
for q in 64 16 4 1 0; do SOAK_QUANT=$q SOAK_NMAX=2600 timeout 600 python profiles/soak_delaunay.py 12 2>&1 | tail -1 >> $OUT/r06_soak_delaunay_quantised.txt; done
if [ -f profiles/ab/libmvosr_dtold.so ]; then
  echo "## the kernel before round 6's collinearity change (profiles/ab/libmvosr_dtold.so), same sets" >> $OUT/r06_soak_delaunay_quantised.txt
  for q in 64 16 4; do MVOSR_LIB_PATH=profiles/ab/libmvosr_dtold.so SOAK_QUANT=$q SOAK_NMAX=2600 timeout 600 python profiles/soak_delaunay.py 12 2>&1 | tail -1 >> $OUT/r06_soak_delaunay_quantised.txt; done
fi
timeout 900 python profiles/dt_decline_rate.py 16384 2000 > $OUT/r06_dt_decline_rate.txt 2>&1
for f in r06_soak_fixed r06_soak_rescale_device r06_soak_qhull r06_soak_exact r06_soak_delaunay_quantised; do tail -n 2 $OUT/$f.txt; done
