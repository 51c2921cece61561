#!/bin/bash
# Review item 4 (i), bounded from measurements: what a two-frame pipeline inside ONE 16-wavefront workgroup per CU could reach.
# Its sweeping half would be 8 wavefronts alone on the CU's LDS/VALU, its loading half 8 more that only stream.  The same
# kernel with its LDS request padded so that only 1 (or 2) workgroups fit a CU gives the time of each phase when a frame has
# the CU to itself; the pipeline's time per frame is at least the longer of {load} and {everything else}.
#   bash profiles/ab_build.sh occ -DMVOSR_ABLATE; bash profiles/ab_build.sh occst "-DMVOSR_ABLATE -DMVOSR_STAMPS"
#   gpurun -- bash profiles/ab_occupancy.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
{
for PAD in 0 20000 60000; do
  export MVOSR_LDS_PAD=$PAD
  echo "== MVOSR_LDS_PAD=$PAD  (53 KB + pad per workgroup of 8 wavefronts; 160 KB per CU)"
  MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_occ.so python3 bench.py --no-cpu-baseline --no-e2e --steps 10 --warmup 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('kernel %.4f ms per %d frames = %.3f us per frame and CU;  frac %.4f' % (r['kernel_ms_avg'], d['config']['frames_per_step_per_gpu'], 1e3 * r['kernel_ms_avg'] / (d['config']['frames_per_step_per_gpu'] / 256.0), r['frac']))"
  MVOSR_DEBUG_SKIP=16 MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_occst.so python3 profiles/stamps.py 65536 2000 | head -7
done
} 2>&1 | tee gpurun_out/r03_ab_occupancy.txt
