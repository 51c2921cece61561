R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
export TMPDIR=/tmp MVOSR_DELAUNAY_WORKERS=0
cd /tmp
for tag in none one; do
  if [ $tag = one ]; then export SNAP_ONE=8000; else unset SNAP_ONE; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s2_tl_$tag -o e2e -- python3 $R/profiles/e2e_gpu_profile.py 16384 2000 exact > $OUT/s2_tl_$tag.log 2>&1
  python3 $R/profiles/e2e_gpu_busy.py $OUT/s2_tl_$tag/e2e_kernel_trace.csv > $OUT/s2_tl_busy_$tag.txt 2>&1
  grep "frames/s" $OUT/s2_tl_$tag.log >> $OUT/s2_tl_busy_$tag.txt
  python3 - $OUT/s2_tl_$tag/e2e_kernel_trace.csv >> $OUT/s2_tl_busy_$tag.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(); rows = rows[len(rows) // 2:]
t0 = rows[0][0]
for s, e, k in rows:
    if e - s > 300e3:
        print("%8.2f .. %8.2f ms (%7.2f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, k.replace("(anonymous namespace)::", "").split("(")[0][-60:]))
PY
  rm -f $OUT/s2_tl_$tag/e2e_kernel_trace.csv
done
cat $OUT/s2_tl_busy_none.txt $OUT/s2_tl_busy_one.txt
