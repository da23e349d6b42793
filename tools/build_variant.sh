#!/bin/bash
# Build a second copy of the library with extra compile flags, for A/B runs inside one gpurun call:
#   tools/build_variant.sh wt -DIMT_WT_STORES=1   ->  build/libimt_hip_wt.so   (select with IMT_LIB=<path>)
# objects, the library and any offload-bundler debris stay under build/ (git-ignored; travels to the GPU box)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/build/variant_$name; mkdir -p $obj
cd $root/imagetranslate_amd/csrc
pids=()
# ONLY="gemm rowops": recompile just those with the extra flags, take the other objects from the base build
all="core gemm gemm_ln rowops attention loss optim decode batch model comm"
for f in $all; do
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $f "; then cp $f.o $obj/$f.o; continue; fi
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable "$@" -c $f.hip -o $obj/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/libimt_hip_$name.so $obj/*.o -ldl
echo built $root/build/libimt_hip_$name.so
