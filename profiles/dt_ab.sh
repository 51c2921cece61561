#!/bin/bash
# Same-box A/B of Delaunay-kernel builds (profiles/ab_build.sh NAME -D...): profiles/dt_ab.sh "<n> <sets>" main NAME1 NAME2 ...
ARGS=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for NAME in "$@"; do
  if [ $NAME = main ]; then unset MVOSR_LIB_PATH; else export MVOSR_LIB_PATH=$R/profiles/ab/libmvosr_$NAME.so; fi
  python3 $R/profiles/dt_check.py $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); t = d['timing']
print('%-14s n=%d  %.0f sets/s  declined %d  mismatch %d  random bad %d' % ('$NAME', t['n'], t['sets_per_s'], t['declined'], len(d['mismatch']), d['random_frames_bad']))"
done
