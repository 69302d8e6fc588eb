#!/bin/bash
# PMC counter passes (run on the GPU box through gpurun).  Each pass is its own
# rocprofv3 run with --pmc only (no trace domains), as the pool requires; results land
# in gpurun_out/pmc_<tag>/.
#   scripts/pmc_passes.sh <tag> [python-script args...]   (default: bench.py --steps 3 --warmup 1 --no-cpu-baseline)
#   PMC_GROUPS="sq1 sq2" selects passes (default: all seven)
set -u
TAG=${1:-r02}
shift || true
if [ $# -eq 0 ]; then set -- bench.py --steps 3 --warmup 1 --no-cpu-baseline; fi
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
GROUPS_WANTED=${PMC_GROUPS:-"fetch write tcc sq1 sq2 tcp grbm"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  case " $GROUPS_WANTED " in *" $name "*) ;; *) return;; esac
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- \
     python3 $SCRIPT "${ARGS[@]}" > $OUT/$name.log 2>&1 < /dev/null || echo "pass $name failed"
  echo "pass $name done"
}
ARGS=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TA_BUSY_avr
run grbm GRBM_GUI_ACTIVE
ls $OUT
