#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box:
#   bash profiles/collect.sh r01 [extra bench.py args]
# Writes raw CSVs under gpurun_out/<tag>_*/ (scratch); profiles/summarize.py turns them into the
# committed summaries profiles/<tag>_*.{md,json}.  Counter passes are separate from the
# kernel-trace/stats pass (PMC serialises dispatches; every pass under `timeout`: a counter list the
# hardware cannot schedule makes rocprofv3 abort and hang), one --pmc list per pass (TCC has 4 slots,
# SQ 8: MI355X_MICROARCH.md "rocprofv3 PMC slots").
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
# no forked Delaunay workers under the profiler: its preloaded library lives in them too and has hung their exit
export MVOSR_DELAUNAY_WORKERS=0
# (bench.py's two-stream probe overlaps kernels of two streams: it would pollute the per-kernel averages of the stats pass)
export MVOSR_BENCH_NO_TWO_STREAMS=1
cd /tmp
OUT=$R/gpurun_out
mkdir -p $OUT
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e "$@" > $OUT/${TAG}_stats.log 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B" \
           "TCC_HIT TCC_MISS TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/${TAG}_pmc$i -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $OUT/${TAG}_pmc$i.log 2>&1
done
grep -h '^{' $OUT/${TAG}_stats.log > $OUT/${TAG}_bench_under_rocprof.json
ls $OUT
