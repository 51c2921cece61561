#!/bin/bash
# rocprofv3 counters of the device-resident rescale path's kernels (one 32 768-frame end-to-end call, twice):
#   bash profiles/collect_rescale_device.sh r04   -> gpurun_out/<tag>_rsd_pmc*/ ; python profiles/summarize_rescale_device.py r04
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp MVOSR_DELAUNAY_WORKERS=0
cd /tmp
OUT=$R/gpurun_out
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/${TAG}_rsd_pmc$i -o e2e -- python3 $R/profiles/e2e_gpu_profile.py 8192 2000 rescale > $OUT/${TAG}_rsd_pmc$i.log 2>&1
  rm -f $OUT/${TAG}_rsd_pmc$i/e2e_kernel_trace.csv
done
ls $OUT | grep _rsd_
