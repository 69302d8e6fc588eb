#!/bin/bash
# exp/build_variant.sh NAME [extra hipcc flags]: the working tree's library with extra flags on ONE translation unit
# (TU=svoxt_kernels by default; TU=svoxt_bwd for the backward), the others taken from the last in-tree build's objects
# -> exp/libsvoxt_NAME.so, for A/B runs (exp/ab_libs.sh, SVOXT_LIB).
set -e
name=$1; shift
tu=${TU:-svoxt_kernels}
root=$(cd $(dirname $0)/.. && pwd)
cd $root/svox_t_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" -c -o /tmp/${tu}_$name.o $tu.hip
objs=""
for o in svoxt_kernels svoxt_bwd svoxt_build svoxt_motion svoxt_order svoxt_step; do
  if [ $o = $tu ]; then objs="$objs /tmp/${tu}_$name.o"; else objs="$objs build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/exp/libsvoxt_$name.so $objs
echo $root/exp/libsvoxt_$name.so
