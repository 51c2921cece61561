R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for l in prod flat8; do
  if [ $l = prod ]; then unset MVOSR_LIB_PATH; else export MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_$l.so; fi
  MVOSR_DELAUNAY_WORKERS=0 timeout 300 python profiles/e2e_chunk_ab.py "10000000:2" 2>&1 | grep rescale | sed "s/^/$l /"
done; done
