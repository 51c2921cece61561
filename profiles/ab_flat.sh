#!/bin/bash
# Same-box A/B of flat_selection_kernel<device> builds: the product library against profiles/ab/libmvosr_<tag>.so — the kernel's
# time under rocprofv3 and the end-to-end rate of the rescale estimator.   AB_LIBS="flatbase" bash profiles/ab_flat.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp MVOSR_DELAUNAY_WORKERS=0
for l in prod ${AB_LIBS:-flatbase}; do
  if [ $l = prod ]; then unset MVOSR_LIB_PATH; else export MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_$l.so; fi
  rm -rf /tmp/abflat_$l
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abflat_$l -o e2e -- python3 $R/profiles/e2e_gpu_profile.py 16384 2000 rescale > /tmp/abflat_$l.log 2>&1
  echo "$l: $(grep flat_selection /tmp/abflat_$l/e2e_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120)   $(grep 'frames/s' /tmp/abflat_$l.log)"
done
