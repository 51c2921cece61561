#!/bin/bash
# Round 6: everything the round's tables are rendered from (profiles/make_tables.py r06), one GPU call:
#   bash profiles/collect_r06.sh        -> gpurun_out/r06_*   (then, in the build container: bash profiles/import_r06.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
# the suite as shipped (no MVOSR_TRIANGULATION anywhere: constructions without the keyword ARE the default), its pass count
python -m pytest tests -m gpu -q > $OUT/r06_gputest.log 2>&1
tail -1 $OUT/r06_gputest.log > $OUT/r06_gputest_count.txt
timeout 900 python bench.py > $OUT/r06_bench.json 2> $OUT/r06_bench.err
timeout 900 python bench.py --workload kitti > $OUT/r06_bench_kitti.json 2> $OUT/r06_bench_kitti.err
timeout 900 python bench.py --features 20000 > $OUT/r06_bench_dense.json 2> $OUT/r06_bench_dense.err
timeout 900 python bench.py --workload gridded --no-cpu-baseline > $OUT/r06_bench_gridded.json 2> $OUT/r06_bench_gridded.err
timeout 900 python bench.py --workload gridded --snap-fraction 0.005 --no-cpu-baseline > $OUT/r06_bench_gridded_0005.json 2> $OUT/r06_bench_gridded_0005.err
# coordinates on a 1/16 px grid: collinear triples everywhere, few cocircular quadruples (the fast kernel declined two frames in three of these; r06_bench_grid16_before.json is that kernel on the same box, profiles/ab/libmvosr_dtold.so, when it is there)
timeout 900 python bench.py --workload gridded --snap-grid 0.0625 --no-cpu-baseline > $OUT/r06_bench_grid16.json 2> $OUT/r06_bench_grid16.err
# (only while that library still has every symbol of the current ABI — it does not since mvosr_memcpy_d2h_kernel: the committed r06_bench_grid16_before.json is the run of the round's third collection)
[ -f profiles/ab/libmvosr_dtold.so ] && MVOSR_LIB_PATH=profiles/ab/libmvosr_dtold.so timeout 900 python bench.py --workload gridded --snap-grid 0.0625 --no-cpu-baseline > $OUT/r06_bench_grid16_before.tmp 2> /dev/null && [ -s $OUT/r06_bench_grid16_before.tmp ] && mv $OUT/r06_bench_grid16_before.tmp $OUT/r06_bench_grid16_before.json
# the N-rank paths as dry runs on this one GPU (gloo, the ranks share the device): a driver-side 8-GPU run then fails for hardware reasons only
timeout 900 python bench.py --gpus 2 --share-gpu --frames 16384 --steps 5 --warmup 1 > $OUT/r06_bench_share2.json 2> $OUT/r06_bench_share2.err
timeout 900 python bench.py --gpus 2 --share-gpu --c4 --total-frames 100000 --steps 5 --warmup 1 > $OUT/r06_bench_share2_c4.json 2> $OUT/r06_bench_share2_c4.err
# the Qhull-rows kernel alone (2000 points, ragged 300-1500)
QH_FRAMES=512,4096,16384 timeout 300 python profiles/qhull_gpu_check.py 2048 2000 > $OUT/r06_qhull_check.txt 2>&1
QH_FRAMES=8192,32768 timeout 300 python profiles/qhull_gpu_check.py 2048 0 | tail -2 >> $OUT/r06_qhull_check.txt 2>&1
# per-frame calls: the three estimators' chains (median wall time); the default estimator with SciPy instead of the host replay (round 5's path)
: > $OUT/r06_latency_probe.txt
for w in rescale scale exact; do timeout 120 python profiles/latency_probe.py $w 200 2000 >> $OUT/r06_latency_probe.txt 2>&1; done
HOST_REPLAY=0 timeout 120 python profiles/latency_probe.py exact 200 2000 | sed 's/^exact:/exact (SciPy for the first triangulation, round 5):/' >> $OUT/r06_latency_probe.txt 2>&1
timeout 120 python profiles/latency_probe.py exact 200 900 >> $OUT/r06_latency_probe.txt 2>&1
python - >> $OUT/r06_latency_probe.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.')
from mvoscalerecovery_amd import packing, synth
for n in (300, 900, 2000, 4000):
    f2 = synth.synth_frame(1, n, base_seed=5000)[1]
    for fn in (packing.qhull_rows_host, packing.delaunay_simplices):
        fn(f2)
        t = time.perf_counter()
        for _ in range(100): fn(f2)
        print("host triangulation alone, %d points: %s %.3f ms" % (n, fn.__name__, (time.perf_counter() - t) / 100 * 1e3))
PY
# soaks: the per-frame call of the default estimator against the pure SciPy estimator; the Qhull-rows kernel against SciPy's rows
timeout 900 python profiles/soak_single_exact.py 6000 300 2000 > $OUT/r06_soak_single_exact.txt 2>&1
timeout 1500 python profiles/soak_qhull.py 20000 > $OUT/r06_soak_qhull.txt 2>&1
: > $OUT/r06_delaunay_bench.jsonl
for a in "" "--seeded --keep 0.95" "--seeded --keep 0.85" "--points 900 --sets 8192" "--points 900 --sets 8192 --seeded --keep 0.95" \
         "--ragged 300:1500 --sets 8192" "--ragged 300:1500 --sets 8192 --seeded --keep 0.95"; do
  timeout 120 python profiles/bench_delaunay.py $a 2>/dev/null | tail -1 >> $OUT/r06_delaunay_bench.jsonl
done
# rocprofv3: the headline step, the KITTI-sized and the dense workloads (stats + PMC passes; traffic.json entries)
bash profiles/collect.sh r06 > /dev/null 2>&1
bash profiles/collect.sh r06_kitti --workload kitti > /dev/null 2>&1
bash profiles/collect.sh r06_dense --features 20000 > /dev/null 2>&1
# rocprofv3: qhull_rows_kernel (stats, then SQ counters and the HBM bytes)
export TMPDIR=/tmp MVOSR_DELAUNAY_WORKERS=0
cd /tmp
QH_FRAMES=4096 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06_qhull_stats -o qh -- python3 $R/profiles/qhull_gpu_check.py 2048 2000 > $OUT/r06_qhull_stats.log 2>&1
rm -f $OUT/r06_qhull_stats/qh_kernel_trace.csv
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  QH_FRAMES=4096 timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/r06_qhull_pmc$i -o qh -- python3 $R/profiles/qhull_gpu_check.py 2048 2000 > $OUT/r06_qhull_pmc$i.log 2>&1
  rm -f $OUT/r06_qhull_pmc$i/qh_kernel_trace.csv
done
# the end-to-end call of the exact path on the GPU's timeline, at the bench leg's size and at twice that
for F in 16384 32768; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06_e2e_exact_$F -o e2e -- python3 $R/profiles/e2e_gpu_profile.py $F 2000 exact > $OUT/r06_e2e_exact_$F.log 2>&1
  python3 $R/profiles/e2e_gpu_busy.py $OUT/r06_e2e_exact_$F/e2e_kernel_trace.csv > $OUT/r06_e2e_exact_busy_$F.txt 2>&1
  grep "frames/s" $OUT/r06_e2e_exact_$F.log >> $OUT/r06_e2e_exact_busy_$F.txt
  rm -f $OUT/r06_e2e_exact_$F/e2e_kernel_trace.csv
done
# the same-process A/B of the side downloads (LABNOTES 10.14): every e2e path, two shapes
: > $OUT/r06_side_downloads_ab.txt
for a in "exact 16384 2000" "exact 16384 300:1500" "fixed 32768 2000" "fixed 32768 300:1500" "rescale 32768 2000" "rescale 32768 300:1500"; do timeout 300 python $R/profiles/side_downloads_ab.py $a 5 >> $OUT/r06_side_downloads_ab.txt 2>&1; done
ls $OUT | grep r06_
