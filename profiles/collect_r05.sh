#!/bin/bash
# Round 5: everything the round's tables are rendered from (profiles/make_tables.py r05), one GPU call:
#   bash profiles/collect_r05.sh        -> gpurun_out/r05_*   (then, in the build container: bash profiles/import_r05.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python bench.py > $OUT/r05_bench.json 2> $OUT/r05_bench.err
timeout 900 python bench.py --workload kitti > $OUT/r05_bench_kitti.json 2> $OUT/r05_bench_kitti.err
timeout 900 python bench.py --features 20000 > $OUT/r05_bench_dense.json 2> $OUT/r05_bench_dense.err
# the Qhull-rows kernel alone (2000 points, ragged 300-1500) and its phase shares
QH_FRAMES=512,4096,16384 timeout 300 python profiles/qhull_gpu_check.py 2048 2000 > $OUT/r05_qhull_check.txt 2>&1
QH_FRAMES=8192,32768 timeout 300 python profiles/qhull_gpu_check.py 2048 0 | tail -2 >> $OUT/r05_qhull_check.txt 2>&1
timeout 300 python profiles/e2e_exact_probe.py 32768 2000 > $OUT/r05_e2e_exact_probe.txt 2>&1
# per-frame calls: the three estimators' chains (median wall time), the one-frame triangulation launches with and without PARTS
: > $OUT/r05_latency_probe.txt
for w in rescale scale exact; do timeout 120 python profiles/latency_probe.py $w 200 2000 >> $OUT/r05_latency_probe.txt 2>&1; done
SINGLE_FAST=0 timeout 120 python profiles/latency_probe.py exact 200 2000 | sed 's/^exact:/exact (two SciPy calls, the path before):/' >> $OUT/r05_latency_probe.txt 2>&1
for n in 900 2000 4000; do timeout 60 python profiles/dt_parts_probe.py $n | grep MVOSR >> $OUT/r05_latency_probe.txt 2>&1; MVOSR_DT_PARTS=0 timeout 60 python profiles/dt_parts_probe.py $n | grep MVOSR >> $OUT/r05_latency_probe.txt 2>&1; done
: > $OUT/r05_delaunay_bench.jsonl
for a in "" "--seeded --keep 0.95" "--seeded --keep 0.85" "--points 900 --sets 8192" "--points 900 --sets 8192 --seeded --keep 0.95" \
         "--ragged 300:1500 --sets 8192" "--ragged 300:1500 --sets 8192 --seeded --keep 0.95"; do
  timeout 120 python profiles/bench_delaunay.py $a 2>/dev/null | tail -1 >> $OUT/r05_delaunay_bench.jsonl
done
# rocprofv3: the headline step, the KITTI-sized and the dense workloads (stats + PMC passes; traffic.json entries)
bash profiles/collect.sh r05 > /dev/null 2>&1
bash profiles/collect.sh r05_kitti --workload kitti > /dev/null 2>&1
bash profiles/collect.sh r05_dense --features 20000 > /dev/null 2>&1
# rocprofv3: qhull_rows_kernel (stats, then SQ counters and the HBM bytes)
export TMPDIR=/tmp MVOSR_DELAUNAY_WORKERS=0
cd /tmp
QH_FRAMES=4096 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05_qhull_stats -o qh -- python3 $R/profiles/qhull_gpu_check.py 2048 2000 > $OUT/r05_qhull_stats.log 2>&1
rm -f $OUT/r05_qhull_stats/qh_kernel_trace.csv
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  QH_FRAMES=4096 timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/r05_qhull_pmc$i -o qh -- python3 $R/profiles/qhull_gpu_check.py 2048 2000 > $OUT/r05_qhull_pmc$i.log 2>&1
  rm -f $OUT/r05_qhull_pmc$i/qh_kernel_trace.csv
done
# the end-to-end call of the exact path on the GPU's timeline
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05_e2e_exact -o e2e -- python3 $R/profiles/e2e_gpu_profile.py 32768 2000 exact > $OUT/r05_e2e_exact.log 2>&1
python3 $R/profiles/e2e_gpu_busy.py $OUT/r05_e2e_exact/e2e_kernel_trace.csv > $OUT/r05_e2e_exact_busy.txt 2>&1
grep "frames/s" $OUT/r05_e2e_exact.log >> $OUT/r05_e2e_exact_busy.txt
rm -f $OUT/r05_e2e_exact/e2e_kernel_trace.csv
ls $OUT | grep r05_
