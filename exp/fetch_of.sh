#!/bin/bash
# FETCH_SIZE (PMC pass) of the shade kernel for library builds: exp/fetch_of.sh "bench args" lib...   ("-" = in-tree)
args=$1; shift
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset SVOXT_LIB; name=intree; else export SVOXT_LIB=$GRAFT_REPO_ROOT/$lib; name=$(basename $lib .so); fi
  PMC_GROUPS="fetch" bash $GRAFT_REPO_ROOT/scripts/pmc_passes.sh x_$name bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-plain $args > /dev/null 2>&1
  python3 - $GRAFT_REPO_ROOT/gpurun_out/pmc_x_$name $name <<'P'
import csv,glob,sys,collections
tot=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob(sys.argv[1]+"/fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name")=="FETCH_SIZE":
            k=r["Kernel_Name"].split("(")[0][-40:]
            tot[k][0]+=1; tot[k][1]+=float(r["Counter_Value"])
for k,(n,v) in sorted(tot.items(), key=lambda kv:-kv[1][1])[:4]:
    print(sys.argv[2], k, "launches", n, "FETCH_SIZE per launch (KB)", round(v/n,1), "-> MB x2:", round(v/n*2*1024/1e6,1), flush=True)
P
done
