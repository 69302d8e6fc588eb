#!/bin/bash
# exp/build_variant.sh NAME [extra hipcc flags]: the working tree's library with extra flags on svoxt_kernels.hip (the other
# translation units taken from the last in-tree build's objects) -> exp/libsvoxt_NAME.so, for A/B runs (exp/ab_libs.sh, SVOXT_LIB).
set -e
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
cd $root/svox_t_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" -c -o /tmp/svoxt_kernels_$name.o svoxt_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/exp/libsvoxt_$name.so /tmp/svoxt_kernels_$name.o build/svoxt_bwd.o build/svoxt_build.o build/svoxt_motion.o build/svoxt_order.o build/svoxt_step.o
echo $root/exp/libsvoxt_$name.so
