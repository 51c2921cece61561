#!/bin/bash
# Round 5: gpurun_out/r05_* (profiles/collect_r05.sh) -> the committed summaries under profiles/ and the tables of the documents.
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
for f in r05_bench.json r05_bench_kitti.json r05_bench_dense.json r05_delaunay_bench.jsonl r05_qhull_check.txt r05_e2e_exact_probe.txt r05_e2e_exact_busy.txt r05_latency_probe.txt; do cp gpurun_out/$f profiles/$f; done
cp gpurun_out/r05_e2e_exact/e2e_kernel_stats.csv profiles/r05_e2e_exact_kernel_stats.csv
python profiles/summarize.py r05 --features 2000 --entry c2_2000 --kernel "scale_frames_kernel<8, 4, 0" > /dev/null
python profiles/summarize.py r05_kitti --features 900 --entry kitti_2000 --kernel "scale_frames_kernel<8, 4, 0, true>;scale_frames_kernel<4, 4, 0, true>;scale_frames_kernel<1, 8, 0, true>" > /dev/null
python profiles/summarize.py r05_dense --features 20000 --entry c2_20000 --kernel scale_frames_tiled_kernel > /dev/null
python profiles/summarize_qhull.py r05 > /dev/null
python profiles/make_tables.py r05
