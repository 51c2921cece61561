#!/bin/bash
# Review item 4 (ii): LDS slot swizzle of the vertex records, and two 8-byte planes an odd stride apart — timing A/B and the LDS conflict counters, one box.
#   for v in 3 5; do bash profiles/ab_build.sh swz$v -DMVOSR_PSWZ=$v; done; bash profiles/ab_build.sh psplit -DMVOSR_PSPLIT; gpurun -- bash profiles/ab_swizzle.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
{
echo "== the experiment builds compute the same results (LDS-resident variants: stage goldens + seeded batches against the oracle)"
for NAME in swz3 swz5 psplit; do
  MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_$NAME.so python3 -m pytest tests/test_gpu_kernels.py -q -x -k "stage_goldens_fused or test_seeded_batches or ragged_batch" 2>&1 | tail -1 | sed "s/^/$NAME: /"
done
echo "== timing (bench.py --steps 10, two alternating repeats)"
bash profiles/ab_run.sh "" main swz3 swz5 psplit
echo "== LDS counters, HOT kernel, 65536 frames per launch (medians; per-frame in the last column)"
for NAME in main swz3 swz5 psplit; do
  if [ $NAME = main ]; then unset MVOSR_LIB_PATH; else export MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_$NAME.so; fi
  echo "-- $NAME"
  KSEL="scale_frames_kernel<8, 4, 0" FR=65536 PASSES="2" bash profiles/pmc_quick.sh swz_$NAME --no-e2e | grep -E "LDS|VALU "
done
} 2>&1 | tee gpurun_out/r03_ab_swizzle.txt
