set -x
mkdir -p gpurun_out/r5a
# first process on the fresh box: the bench exactly as the driver runs it
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5a/bench_1.json 2> gpurun_out/r5a/bench_1.err
echo "bench1 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5a/bench_2.json 2> gpurun_out/r5a/bench_2.err
echo "bench2 rc=$?"
timeout -k 10 900 python3 -m pytest tests/test_gpu_nccl_world1.py tests/test_gpu_bench_contract.py tests/test_gpu_pool_hint.py tests/test_gpu_super_tiles.py tests/test_gpu_roles_handoff.py -x -q > gpurun_out/r5a/tests.log 2>&1
echo "tests rc=$?"
tail -5 gpurun_out/r5a/tests.log
