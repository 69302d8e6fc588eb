set -x
mkdir -p gpurun_out/r5b
timeout -k 10 900 python3 -m pytest tests/test_gpu_thresholds.py -x -q > gpurun_out/r5b/thresh.log 2>&1
echo "thresh rc=$?"; tail -15 gpurun_out/r5b/thresh.log
