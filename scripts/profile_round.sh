#!/bin/bash
# One pass over everything profiles/rNN_* holds for the bench workloads (run on the GPU box through
# gpurun; ~5 minutes):  per workload the bench line, rocprofv3 --kernel-trace --stats, and the PMC passes
# (scripts/pmc_passes.sh: one rocprofv3 --pmc run per counter group, no trace domains).
#   scripts/profile_round.sh <outdir under gpurun_out> [a|b|all]     then copy what is to be judged into profiles/
#   (a: the headline config and the f-rows; b: configs[3], exact and native math)
set -u
ROUND=r05      # = bench.py's ROUND
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bench() {   # name, args...
  local name=$1; shift
  timeout -k 10 300 python3 bench.py "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_bench.err < /dev/null || echo "bench $name failed"
  echo "bench $name done"
}
stats() {   # name, args...
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -- \
      python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --prewarm-s 0.05 --no-cpu-baseline --no-plain "$@" > /dev/null 2>&1 < /dev/null ) || echo "stats $name failed"
  local f=$(find $OUT/${name}_trace -name "*kernel_stats.csv" | sort | tail -1)
  [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv && cp "$f" profiles/${ROUND}_${name}_kernel_stats.csv   # (the bench lines read it for limits.forward)
  rm -rf $OUT/${name}_trace
  echo "stats $name done"
}
pmc() {     # name, args...
  local name=$1; shift
  bash scripts/pmc_passes.sh $name bench.py --steps 3 --warmup 1 --prewarm-s 0 --no-cpu-baseline --no-plain "$@" > $OUT/${name}_pmc.log 2>&1 < /dev/null
  python3 scripts/pmc_summary.py $name $OUT/${name}_pmc.json > $OUT/${name}_pmc_summary.txt 2>&1
  cp $OUT/${name}_pmc.json profiles/${ROUND}_${name}_pmc.json      # (on the box's copy: the bench lines below read it for roofline.traffic)
  echo "pmc $name done"
}
timing() {  # name, script, args...
  local name=$1; shift
  timeout -k 10 240 python3 "$@" > $OUT/${name}_timing.txt 2> $OUT/${name}_timing.err < /dev/null || echo "timing $name failed"
  echo "timing $name done"
}
PART=${2:-all}
if [ $PART = all ] || [ $PART = a ]; then
# headline (configs[2]) forward+backward, its forward alone, the camera route (render_persp) and the two opt-outs
pmc   d8_sh9_800
stats d8_sh9_800
bench d8_sh9_800
stats d8_sh9_800_fwd --forward-only
bench d8_sh9_800_fwd --forward-only
bench d8_sh9_800_camera --route camera --no-plain
stats d8_sh9_800_plain --route plain
SVOXT_BWD_EXACT=0 bench d8_sh9_800_single_march --no-plain
# SURVEY.md 8(d)'s extra row: fast=True (thresholds 1e-2) -- forward only, and (r05: through the recording forward) forward+backward
bench d8_sh9_800_fast_fwd --fast --forward-only --no-plain
bench d8_sh9_800_fast --fast --no-plain
# configs[0]: the reference's own CPU-runnable case
bench d5_rgba_64_fwd --workload d5_rgba_64 --forward-only
# the rows SURVEY.md 8(f) added around the path, on this round's kernels
timing persp scripts/persp_timing.py
timing motion scripts/motion_timing.py
timing xform scripts/xform_timing.py
timing opacity scripts/opacity_timing.py
timing depth scripts/depth_timing.py
timing query scripts/query_timing.py
timing build scripts/build_timing.py
timing ray_order scripts/ray_order_timing.py
timing generic scripts/generic_timing.py
fi
if [ $PART = all ] || [ $PART = b ]; then
# configs[3]: the one that leaves the Infinity Cache -- exact (the parity suite's mode) and native math (tolerance)
pmc   d9_rgba32_1024 --workload d9_rgba32_1024
stats d9_rgba32_1024 --workload d9_rgba32_1024
bench d9_rgba32_1024 --workload d9_rgba32_1024
pmc   d9_rgba32_1024_fwd --workload d9_rgba32_1024 --forward-only
stats d9_rgba32_1024_fwd --workload d9_rgba32_1024 --forward-only
bench d9_rgba32_1024_fwd --workload d9_rgba32_1024 --forward-only
bench d9_rgba32_1024_fast_fwd --workload d9_rgba32_1024 --fast --forward-only --no-plain
bench d9_rgba32_1024_fast --workload d9_rgba32_1024 --fast --no-plain
SVOXT_NATIVE_MATH=1 stats d9_rgba32_1024_native --workload d9_rgba32_1024
SVOXT_NATIVE_MATH=1 pmc   d9_rgba32_1024_native --workload d9_rgba32_1024
SVOXT_NATIVE_MATH=1 stats d9_rgba32_1024_fwd_native --workload d9_rgba32_1024 --forward-only
fi
ls $OUT
