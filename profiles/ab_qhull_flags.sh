#!/bin/bash
# Same-box A/B of builds of mvosr_qhull.hip (ONLY=mvosr_qhull profiles/ab_build.sh qf_<tag> <flags>) against the product build: rows against SciPy
# and the launch times of 4 096 / 16 384 sets of 2000 points, two alternating passes.   AB_LIBS="qf_ilp qf_O2" bash profiles/ab_qhull_flags.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export MVOSR_DELAUNAY_WORKERS=0
for pass in 1 2; do
  for l in main $AB_LIBS; do
    if [ $l = main ]; then unset MVOSR_LIB_PATH; else export MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_$l.so; fi
    QH_FRAMES=4096,16384 timeout 200 python profiles/qhull_gpu_check.py 64 2000 2>&1 | awk -v t=$l '/different [1-9]|MISMATCH/ {bad=1} /launch of/ {printf "%s: %s ms; ", t, $8} END {print (bad ? "ROWS DIFFER" : "rows ok")}'
  done
done
