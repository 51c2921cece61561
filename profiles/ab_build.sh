#!/bin/bash
# A/B builds of libmvosr.so with extra -D flags, for same-box comparisons on the GPU:
#   profiles/ab_build.sh NAME "-DFOO=1 -DBAR"   ->  gpurun_out/ab/libmvosr_NAME.so   (use with MVOSR_LIB_PATH=...)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
#   ONLY=mvosr_rescale profiles/ab_build.sh NAME -DFOO   rebuilds that one source with the flags and takes the other objects
#                                                         from the product build (make -C mvoscalerecovery_amd/csrc first)
NAME=$1; shift
mkdir -p $R/profiles/ab/obj_$NAME
cd $R/mvoscalerecovery_amd/csrc
for f in mvosr_kernels mvosr_rescale mvosr_delaunay mvosr_qhull mvosr_capi; do
  if [ -n "$ONLY" ] && [ $f != "$ONLY" ]; then cp $f.o $R/profiles/ab/obj_$NAME/$f.o; continue; fi
  if [ -n "$ONLY" ] || [ $f = mvosr_kernels ] || [ ! -f $R/profiles/ab/obj_$NAME/$f.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function $@ -c $f.hip -o $R/profiles/ab/obj_$NAME/$f.o &
  fi
done
wait
cp mvosr_qhull_host.o $R/profiles/ab/obj_$NAME/mvosr_qhull_host.o      # (plain C, no flags of interest: from the product build)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $R/profiles/ab/obj_$NAME/*.o -o $R/profiles/ab/libmvosr_$NAME.so
echo built $R/profiles/ab/libmvosr_$NAME.so
