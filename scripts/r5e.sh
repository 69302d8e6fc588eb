set -x
mkdir -p gpurun_out/r5e
timeout -k 10 900 python3 -m pytest tests/test_gpu_step_api.py -x -q -s > gpurun_out/r5e/step.log 2>&1
echo "step rc=$?"; tail -25 gpurun_out/r5e/step.log
