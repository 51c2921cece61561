#!/bin/bash
# Same-box A/B around the arena-out variant (three four-wavefront frames per CU up to ~2 100 points): product against
# profiles/ab/libmvosr_head.so (bash profiles/ab_build_rev.sh head HEAD) at the sizes it changes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for n in ${AB_POINTS:-1600 1800 2000 2100}; do
 for a in "--points $n --sets 6144" "--points $n --sets 6144 --seeded --keep 0.95" "--points $n --sets 6144 --seeded --keep 0.85"; do
  echo "$a"
  for rep in 1 2; do
  for l in prod ${AB_LIBS:-head}; do
      if [ $l = prod ]; then r=$(timeout 120 python profiles/bench_delaunay.py $a 2>&1 | tail -1)
      else r=$(MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_$l.so timeout 120 python profiles/bench_delaunay.py $a 2>&1 | tail -1); fi
      echo "$l $(echo $r | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sets/s, declined %d' % (d['sets_per_s'], d['declined']))")"
  done; done
 done
done
