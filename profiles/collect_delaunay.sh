#!/bin/bash
# rocprofv3 evidence for delaunay_kernel (profiles/bench_delaunay.py):
#   bash profiles/collect_delaunay.sh r03 [--points 2000 --sets 4096]   -> gpurun_out/<tag>_dt_* ; then  python profiles/summarize_delaunay.py r03
TAG=${1:-r03}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_dt_stats -o bench -- python3 $R/profiles/bench_delaunay.py --steps 10 "$@" > $OUT/${TAG}_dt_stats.log 2>&1
i=0
for PMC in "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/${TAG}_dt_pmc$i -o bench -- python3 $R/profiles/bench_delaunay.py --steps 2 "$@" > $OUT/${TAG}_dt_pmc$i.log 2>&1
done
grep -h '^{' $OUT/${TAG}_dt_stats.log > $OUT/${TAG}_dt_bench.json
ls $OUT | grep _dt_
