#!/bin/bash
# exp/build_all_variant.sh NAME [extra hipcc flags]: the working tree's library with extra flags on EVERY translation unit
# -> exp/libsvoxt_NAME.so (for switches in headers that several units share), for A/B runs (exp/ab_libs.sh, SVOXT_LIB).
set -e
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
cd $root/svox_t_amd/csrc
objs=""
for tu in svoxt_kernels svoxt_bwd svoxt_build svoxt_motion svoxt_order svoxt_step; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" -c -o /tmp/${tu}_$name.o $tu.hip &
  objs="$objs /tmp/${tu}_$name.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/exp/libsvoxt_$name.so $objs
echo $root/exp/libsvoxt_$name.so
