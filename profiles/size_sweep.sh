#!/bin/bash
# Throughput of the step across frame sizes (auto wave count): bash profiles/size_sweep.sh > gpurun_out/size_sweep.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for n in 256 320 512 768 1024 1500 2000 2500 3000 4000 6000; do
  fr=$((65536*2000/n)); fr=$((fr/128*128))
  python bench.py --features $n --frames $fr --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | grep "^{" | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('| %d | %d | %.3g | %.3f | %.3f | %.3f | %.3f |' % ($n, $fr, d['value'], r['kernel_ms_avg'], r['road_model_kernel_ms_avg'], r['frac'], r['step_frac']))"
done
