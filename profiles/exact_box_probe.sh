#!/bin/bash
# Which kind of box is this for the exact path (LABNOTES 10.7: 95 k or 118 k frames/s at 16 384 frames, by box)?  The e2e call three times,
# then its kernel timeline under rocprofv3 (the union of the replays, head, tail, the gaps).   bash profiles/exact_box_probe.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=$R/gpurun_out
export TMPDIR=/tmp MVOSR_DELAUNAY_WORKERS=0 GPU_MAX_HW_QUEUES=${HWQ:-8}
cd /tmp
SIDE=${SIDE:-1} python3 $R/profiles/exact_host_trace.py 16384 2000 2>&1
SIDE=${SIDE:-1} python3 $R/profiles/exact_host_trace.py 16384 2000 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/boxprobe -o e2e -- python3 $R/profiles/e2e_gpu_profile.py 16384 2000 exact > $OUT/boxprobe.log 2>&1
grep "frames/s" $OUT/boxprobe.log
python3 $R/profiles/e2e_gpu_busy.py $OUT/boxprobe/e2e_kernel_trace.csv | head -16
python3 - $OUT/boxprobe/e2e_kernel_trace.csv <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(); rows = rows[len(rows) // 2:]
t0 = rows[0][0]
for s, e, k in rows:
    if e - s > 300e3:
        print("%8.2f .. %8.2f ms (%7.2f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, k.replace("(anonymous namespace)::", "").split("(")[0][-60:]))
PY
rm -f $OUT/boxprobe/e2e_kernel_trace.csv
rocminfo 2>/dev/null | grep -m3 "Marketing Name\|Compute Unit\|Max Clock" ; nproc; grep -m1 "model name" /proc/cpuinfo
