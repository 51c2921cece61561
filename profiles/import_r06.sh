#!/bin/bash
# Round 6: gpurun_out/r06_* (profiles/collect_r06.sh) -> the committed summaries under profiles/ and the tables of the documents.
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
for f in r06_bench.json r06_bench_kitti.json r06_bench_dense.json r06_bench_gridded.json r06_bench_gridded_0005.json r06_bench_grid16.json r06_bench_grid16_before.json r06_bench_share2.json r06_bench_share2_c4.json r06_delaunay_bench.jsonl \
         r06_qhull_check.txt r06_latency_probe.txt r06_soak_single_exact.txt r06_gputest_count.txt r06_e2e_exact_busy_16384.txt r06_e2e_exact_busy_32768.txt r06_side_downloads_ab.txt; do
  [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/$f
done
cp gpurun_out/r06_e2e_exact_16384/e2e_kernel_stats.csv profiles/r06_e2e_exact_kernel_stats.csv 2>/dev/null
python profiles/summarize.py r06 --features 2000 --entry c2_2000 --kernel "scale_frames_kernel<8, 4, 0" > /dev/null
python profiles/summarize.py r06_kitti --features 900 --entry kitti_2000 --kernel "scale_frames_kernel<8, 4, 0, true>;scale_frames_kernel<4, 4, 0, true>;scale_frames_kernel<1, 8, 0, true>" > /dev/null
python profiles/summarize.py r06_dense --features 20000 --entry c2_20000 --kernel scale_frames_tiled_kernel > /dev/null
python profiles/summarize_qhull.py r06 > /dev/null
python profiles/make_tables.py r06
