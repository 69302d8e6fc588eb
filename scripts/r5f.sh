set -x
mkdir -p gpurun_out/r5f
timeout -k 10 600 python3 -m pytest tests/test_gpu_roles_handoff.py tests/test_gpu_thresholds.py -x -q -s > gpurun_out/r5f/roles.log 2>&1
echo "roles rc=$?"; grep "roles hand-over" gpurun_out/r5f/roles.log | head -30; tail -3 gpurun_out/r5f/roles.log
for i in 1 2; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/r5f/bench_$i.json 2> gpurun_out/r5f/bench_$i.err || echo FAILED
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline --forward-only > gpurun_out/r5f/bench_fwd.json 2> gpurun_out/r5f/bench_fwd.err || echo FAILED
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5f/*.json')):
    j=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, j['value'], j['value_median'], j['kernel_ms'], [(o['route'],o['value']) for o in j.get('other_routes',[])])
PY
