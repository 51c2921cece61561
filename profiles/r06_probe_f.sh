#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp MVOSR_DELAUNAY_WORKERS=0
cd /tmp
F=16384
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06_e2e_exact_lazy_$F -o e2e -- python3 $R/profiles/e2e_gpu_profile.py $F 2000 exact > $OUT/r06_e2e_exact_lazy_$F.log 2>&1
python3 $R/profiles/e2e_gpu_busy.py $OUT/r06_e2e_exact_lazy_$F/e2e_kernel_trace.csv > $OUT/r06_e2e_exact_lazy_busy_$F.txt 2>&1
grep "frames/s" $OUT/r06_e2e_exact_lazy_$F.log >> $OUT/r06_e2e_exact_lazy_busy_$F.txt
python3 - <<PY
import csv
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open("$OUT/r06_e2e_exact_lazy_$F/e2e_kernel_trace.csv"))]
rows.sort(); rows=rows[len(rows)//2:]
t0=rows[0][0]
for s,e,k in rows:
    k=k.replace("(anonymous namespace)::","").split("(")[0][-50:]
    if e-s>200000 or "qhull" in k: print("%8.2f %8.2f  %s"%((s-t0)/1e6,(e-s)/1e6,k))
PY
rm -f $OUT/r06_e2e_exact_lazy_$F/e2e_kernel_trace.csv
cat $OUT/r06_e2e_exact_lazy_busy_$F.txt
