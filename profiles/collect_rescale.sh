#!/bin/bash
# rocprofv3 evidence for the `rescale` variant's kernels and triangle_batch (profiles/bench_rescale.py):
#   bash profiles/collect_rescale.sh r02      -> gpurun_out/<tag>_rescale_* ; then  python profiles/summarize_rescale.py r02
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_rescale_stats -o bench -- python3 $R/profiles/bench_rescale.py --steps 10 "$@" > $OUT/${TAG}_rescale_stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/${TAG}_rescale_pmc1 -o bench -- python3 $R/profiles/bench_rescale.py --steps 2 "$@" > $OUT/${TAG}_rescale_pmc1.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_rescale_pmc2 -o bench -- python3 $R/profiles/bench_rescale.py --steps 2 "$@" > $OUT/${TAG}_rescale_pmc2.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_rescale_pmc3 -o bench -- python3 $R/profiles/bench_rescale.py --steps 2 "$@" > $OUT/${TAG}_rescale_pmc3.log 2>&1
grep -h '^{' $OUT/${TAG}_rescale_stats.log > $OUT/${TAG}_rescale_bench.json
ls $OUT | grep rescale
