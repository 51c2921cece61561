#!/bin/bash
# Same-box A/B on ragged batches (size classes): product against profiles/ab/libmvosr_head.so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for a in "--ragged 300:1500 --sets 7680" "--ragged 300:1500 --sets 7680 --seeded --keep 0.95" "--ragged 100:2000 --sets 6144" "--ragged 100:2000 --sets 6144 --seeded --keep 0.95" "--points 2000 --sets 6144" "--points 2000 --sets 6144 --seeded --keep 0.95" "--points 900 --sets 8192"; do
  echo "$a"
  for rep in 1 2; do
    for l in prod ${AB_LIBS:-head}; do
      if [ $l = prod ]; then r=$(timeout 120 python profiles/bench_delaunay.py $a 2>&1 | tail -1)
      else r=$(MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_$l.so timeout 120 python profiles/bench_delaunay.py $a 2>&1 | tail -1); fi
      echo "$l $(echo $r | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sets/s, declined %d' % (d['sets_per_s'], d['declined']))")"
    done
  done
done
