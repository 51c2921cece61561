#!/bin/bash
# quick SQ counter pass for the fused kernel: [KSEL=<kernel substring>] [FR=<frames per launch>] [PASSES="2 5"] [BENCH=profiles/bench_rescale.py BENCH_ARGS="--steps 3"] bash profiles/pmc_quick.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out; mkdir -p $OUT
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  if [ -n "$PASSES" ] && [[ " $PASSES " != *" $i "* ]]; then continue; fi
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/${TAG}_q$i -o bench -- python3 $R/${BENCH:-bench.py} ${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu-baseline} "$@" > $OUT/${TAG}_q$i.log 2>&1
done
python3 - <<PY
import csv,glob,statistics,os
c={}
for f in glob.glob("$OUT/${TAG}_q*/bench_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if os.environ.get("KSEL","scale_frames") in r["Kernel_Name"]:
            c.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
F=int(os.environ.get('FR','16384'))
for k in sorted(c): print("%-28s %14.6g per-frame %10.1f"%(k,statistics.median(c[k]),statistics.median(c[k])/F))
PY
