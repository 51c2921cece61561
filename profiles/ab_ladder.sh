cd $GRAFT_REPO_ROOT
for n in 1000 1200 1400 1500; do
 for a in "--points $n --sets 8192" "--points $n --sets 8192 --seeded --keep 0.95"; do
  echo "$a"
  for l in prod noladder; do
      if [ $l = prod ]; then r=$(timeout 120 python profiles/bench_delaunay.py $a 2>&1 | tail -1)
      else r=$(MVOSR_LIB_PATH=$GRAFT_REPO_ROOT/profiles/ab/libmvosr_$l.so timeout 120 python profiles/bench_delaunay.py $a 2>&1 | tail -1); fi
      echo "$l $(echo $r | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sets/s' % (d['sets_per_s']))")"
  done
 done
done
