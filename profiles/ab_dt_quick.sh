#!/bin/bash
# Same-box check of a Delaunay-kernel change: the product library against profiles/ab/libmvosr_base.so (built from HEAD) at two
# sizes, first and seeded triangulation, then the Delaunay parity tests.   bash profiles/ab_dt_quick.sh [more lib tags]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
AB_LIBS="base $*" bash profiles/ab_dt_order.sh
timeout 600 python -m pytest tests -m gpu -x -q -k 'delaunay or triang or seeded' 2>&1 | tail -3
