#!/bin/bash
# exp/prof_lib.sh OUT LIB "bench args": kernel stats of one bench run with another build of the library (SVOXT_LIB)
out=$1; lib=$2; args=$3
mkdir -p $out
export SVOXT_LIB=$GRAFT_REPO_ROOT/$lib
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-plain --no-other-configs $args > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 - $out/t_kernel_stats.csv <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:8]:
    print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
P
