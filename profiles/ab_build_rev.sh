#!/bin/bash
# A library with mvosr_delaunay.hip as of a git revision (the other objects from the product build), for same-box A/B runs:
#   bash profiles/ab_build_rev.sh <tag> [<rev>=HEAD]   ->  profiles/ab/libmvosr_<tag>.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; REV=${2:-HEAD}
mkdir -p $R/profiles/ab/obj_$TAG
cd $R/mvoscalerecovery_amd/csrc
git show $REV:mvoscalerecovery_amd/csrc/mvosr_delaunay.hip > _rev_dt.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -c _rev_dt.hip -o $R/profiles/ab/obj_$TAG/mvosr_delaunay.o
rm -f _rev_dt.hip
cp mvosr_kernels.o mvosr_rescale.o mvosr_capi.o $R/profiles/ab/obj_$TAG/
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $R/profiles/ab/obj_$TAG/*.o -o $R/profiles/ab/libmvosr_$TAG.so
echo built $R/profiles/ab/libmvosr_$TAG.so
