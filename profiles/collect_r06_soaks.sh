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
tail -2 $OUT/r06_soak_fixed.txt $OUT/r06_soak_rescale_device.txt $OUT/r06_soak_qhull.txt $OUT/r06_soak_exact.txt
