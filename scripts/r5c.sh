set -x
mkdir -p gpurun_out/r5c
for w in d8_sh9_800 d9_rgba32_1024; do
  for f in "" "--fast"; do
    for m in "" "--forward-only"; do
      n=${w}${f:+_fast}${m:+_fwd}
      timeout -k 10 300 python3 bench.py --workload $w $f $m --no-cpu-baseline --no-plain > gpurun_out/r5c/$n.json 2> gpurun_out/r5c/$n.err || echo "FAILED $n"
    done
  done
done
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r5c/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5c/tests.log
