#!/bin/bash
# Same-box A/B of delaunay_kernel variants: the product library against libraries under profiles/ab
# (profiles/ab_build.sh <tag> <flags>, ONLY=mvosr_delaunay), two alternating passes per size.
#   AB_LIBS="col0 nn0" [AB_SIZES="2000:4096 600:8192 r300:1500"] bash profiles/ab_dt_order.sh      (r<lo>:<hi> = a ragged batch)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for sz in ${AB_SIZES:-2000:4096 600:8192}; do
  for mode in "" "--seeded"; do
    if [[ $sz == r* ]]; then a="--ragged ${sz#r} --sets ${AB_SETS:-8192} $mode"; else a="--points ${sz%%:*} --sets ${sz##*:} $mode"; fi
    echo "$a"
    for rep in 1 2; do
      for l in prod ${AB_LIBS:-col0}; do
        if [ $l = prod ]; then r=$(timeout 120 python profiles/bench_delaunay.py $a 2>&1 | tail -1)
        else r=$(MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_$l.so timeout 120 python profiles/bench_delaunay.py $a 2>&1 | tail -1); fi
        echo "$l $(echo $r | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sets/s, declined %d' % (d['sets_per_s'], d['declined']))")"
      done
    done
  done
done
