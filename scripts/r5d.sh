set -x
mkdir -p gpurun_out/r5d
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > gpurun_out/r5d/tests.log 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/r5d/tests.log
