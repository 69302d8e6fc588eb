#!/bin/bash
# exp/build_rev.sh REV [extra hipcc flags]: the library as of git revision REV -> exp/libsvoxt_REV$SUFFIX.so (for A/B runs with SVOXT_LIB;
# SUFFIX from the environment names a build with extra flags: SUFFIX=_nocsum exp/build_rev.sh HEAD -DSVOXT_ROLES_CSUM=0)
set -e
rev=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
tmp=$(mktemp -d)
git -C $root archive $rev svox_t_amd/csrc include | tar -x -C $tmp
cd $tmp/svox_t_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math "$@" \
    -o $root/exp/libsvoxt_$rev$SUFFIX.so svoxt_kernels.hip $( [ -f svoxt_bwd.hip ] && echo svoxt_bwd.hip ) svoxt_build.hip svoxt_motion.hip svoxt_order.hip $( [ -f svoxt_step.hip ] && echo svoxt_step.hip ) 2>&1 | grep -v hip-link || true
rm -rf $tmp
echo $root/exp/libsvoxt_$rev$SUFFIX.so
