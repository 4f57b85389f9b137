#!/bin/bash
# round 6, GPU call N: the device producer's set-up after its last two numpy
# passes went native (positions of all components at once, contribution sources)
out=gpurun_out
timeout 1200 python -m pytest tests/test_device_producer_gpu.py tests/test_device_producer_plans.py tests/test_multi_gpu_threads.py tests/test_partitioned_device_producer_gpu.py -x -q -m gpu > $out/r06_n_pytest_producers.txt 2>&1
echo "pytest producers rc $?"; tail -2 $out/r06_n_pytest_producers.txt
timeout 1200 python3 bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 20 --warmup 5 --no-cpu-baseline > $out/r06_n_bench_cube_n73_native_plan_passes.json 2> $out/r06_n_bench_cube_n73.err
echo "bench n73 rc $?"
