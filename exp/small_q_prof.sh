#!/bin/bash
# kernel time per step of the two plain calls at small batch sizes: exp/small_q_prof.sh OUT
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for q in ${QS:-4096 16384}; do for sc in ${SCS:-1 0}; do for sm in 16384 512; do
  export PROBE_Q=$q PROBE_SCRATCH=$sc PROBE_SORT_MIN=$sm
  d=$out/q${q}_s${sc}_m${sm}
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 $GRAFT_REPO_ROOT/exp/small_q_prof.py > $d.log 2>&1 || exit 1
  python3 - $d/t_kernel_stats.csv $q $sc $sm <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r['Calls']) >= 50]
tot = sum(float(r['TotalDurationNs']) for r in rows) / 50 / 1e3
top = sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:5]
print(f"Q {sys.argv[2]} scratch {sys.argv[3]} sort_min {sys.argv[4]}: kernels {tot:.1f} us/step: " + ", ".join(f"{r['Name'].split('<')[0].split('::')[-1][:22]} {float(r['TotalDurationNs'])/50/1e3:.0f}" for r in top), flush=True)
P
done; done; done
