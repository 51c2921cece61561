#!/bin/bash
# Round 4: everything the tables of BASELINE.md / README.md / DESIGN.md are rendered from (profiles/make_tables.py), one GPU call:
#   bash profiles/collect_r04.sh      -> gpurun_out/r04_*   (copy the summaries into profiles/ afterwards: make_tables.py --import)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 600 python bench.py > $OUT/r04_bench.json 2> $OUT/r04_bench.err
timeout 600 python bench.py --workload kitti > $OUT/r04_bench_kitti.json 2> $OUT/r04_bench_kitti.err
: > $OUT/r04_delaunay_bench.jsonl
for a in "" "--seeded --keep 0.95" "--seeded --keep 0.95 --no-carry" "--seeded --keep 0.85" "--seeded --keep 0.85 --no-carry" \
         "--points 900 --sets 8192" "--points 900 --sets 8192 --seeded --keep 0.95" "--ragged 300:1500 --sets 8192" "--ragged 300:1500 --sets 8192 --seeded --keep 0.95"; do
  timeout 120 python profiles/bench_delaunay.py $a 2>/dev/null | tail -1 >> $OUT/r04_delaunay_bench.jsonl
done
export TMPDIR=/tmp MVOSR_DELAUNAY_WORKERS=0
cd /tmp
for w in scale rescale; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_e2e_$w -o e2e -- python3 $R/profiles/e2e_gpu_profile.py 32768 2000 $w > $OUT/r04_e2e_$w.log 2>&1
  python3 $R/profiles/e2e_gpu_busy.py $OUT/r04_e2e_$w/e2e_kernel_trace.csv > $OUT/r04_e2e_${w}_busy.txt 2>&1
  grep "frames/s" $OUT/r04_e2e_$w.log >> $OUT/r04_e2e_${w}_busy.txt
  rm -f $OUT/r04_e2e_$w/e2e_kernel_trace.csv
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_dt1 -o dt -- python3 $R/profiles/bench_delaunay.py --steps 10 > $OUT/r04_dt1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_dt2 -o dt -- python3 $R/profiles/bench_delaunay.py --steps 10 --seeded --keep 0.95 > $OUT/r04_dt2.log 2>&1
rm -f $OUT/r04_dt1/dt_kernel_trace.csv $OUT/r04_dt2/dt_kernel_trace.csv
ls $OUT | grep r04_
