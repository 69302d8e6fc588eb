#!/bin/bash
# A/B of library builds on one box: exp/ab_libs.sh OUTDIR "bench args" lib1 lib2 ...   ("-" = the in-tree library)
# Two rounds over the list so that drift of the box shows.
out=$1; shift; args=$1; shift
mkdir -p $out
for rnd in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset SVOXT_LIB; name=intree; else export SVOXT_LIB=$PWD/$lib; name=$(basename $lib .so); fi
    python bench.py --no-cpu-baseline --no-plain $args > $out/${name}_$rnd.json 2> $out/${name}_$rnd.err || { echo "FAILED $name"; tail -5 $out/${name}_$rnd.err; exit 1; }
    python - $out/${name}_$rnd.json $name <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[2], d['value'], d['ms_per_step'], d['kernel_ms'], flush=True)
P
  done
done
