#!/bin/bash
# Run GPU steps one after another on the GPU box (through gpurun): each under its own `timeout -k 10`, its
# output in gpurun_out/<dir>/<name>.log; an ordinary failure (a failing assert: rc 1) does not stop the
# sequence, a step that was killed or timed out (rc >= 124) does -- nothing further touches the GPU then.
#   source scripts/gpu_steps.sh <dir>;  step <name> <seconds> <command...>
OUTDIR=$GRAFT_REPO_ROOT/gpurun_out/${1:-steps}
mkdir -p $OUTDIR
cd $GRAFT_REPO_ROOT
step() {
  local name=$1 secs=$2; shift 2
  echo "== $name: $*"
  timeout -k 10 $secs "$@" > $OUTDIR/$name.log 2> $OUTDIR/$name.err < /dev/null
  local rc=$?
  echo "== $name rc=$rc"
  if [ $rc -ge 124 ]; then echo "step $name was killed (rc $rc): stopping here"; tail -5 $OUTDIR/$name.err; exit $rc; fi
  return 0
}
