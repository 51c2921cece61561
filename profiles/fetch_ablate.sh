R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for M in 0 1 2 4 24; do
  MVOSR_DEBUG_SKIP=$M rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/fa_$M -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv,statistics
v=[float(r["Counter_Value"]) for r in csv.DictReader(open("$R/gpurun_out/fa_$M/bench_counter_collection.csv")) if "scale_frames" in r["Kernel_Name"]]
print("SKIP=$M fetch bytes/frame (x2 corrected): %.0f"%(2*1024*statistics.median(v)/16384))
PY
done
