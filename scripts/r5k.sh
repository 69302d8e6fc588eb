set -x
mkdir -p gpurun_out/r5k
timeout -k 10 600 python3 -m pytest tests/test_gpu_render_parity.py tests/test_gpu_thresholds.py tests/test_gpu_exp_table.py -x -q > gpurun_out/r5k/tests.log 2>&1; echo rc=$?; tail -3 gpurun_out/r5k/tests.log
for f in "" "--fast"; do
timeout -k 10 300 python3 bench.py --workload d9_rgba32_1024 $f --forward-only --no-cpu-baseline --no-plain > gpurun_out/r5k/d9${f:+_fast}_fwd.json 2> gpurun_out/r5k/err.log || echo FAILED
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5k/*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, j['value'], j['kernel_ms'], j['kernels']['forward'])
PY
