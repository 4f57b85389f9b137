#!/bin/bash
# round 6, GPU call B
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 1500 python -m pytest tests/test_abi_closed.py tests/test_kernels_random_gpu.py "tests/test_api_gpu.py::test_bench_line_of_the_north_stars_literal_solvers" tests/test_device_producer_gpu.py tests/test_multi_gpu_threads.py -x -q -m gpu --durations=15 > $out/r06_b_pytest.txt 2>&1
echo "pytest rc $? $(( $(date +%s) - t0 )) s" | tee -a $out/r06_b_pytest.txt
export FENAPACK_AMD_RSS_TRACE=1
timeout 900 python3 bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 20 --warmup 5 > $out/r06_b_bench_cube_n73_product.json 2> $out/r06_b_bench_cube_n73_product.err
echo "bench n73 mg rc $? $(( $(date +%s) - t0 )) s"
timeout 600 python3 bench.py --geometry cube --level 0 --n0 48 --inner jacobi --steps 5 --warmup 2 --cpu-seconds 4 > $out/r06_b_bench_cube_n48_inner_jacobi.json 2> $out/r06_b_bench_cube_n48_inner_jacobi.err
echo "bench jacobi n48 rc $? $(( $(date +%s) - t0 )) s"
unset FENAPACK_AMD_RSS_TRACE
timeout 600 bash tools/gpu_pmc.sh r06_b_level6_jacobi --inner jacobi
echo "pmc l6 jacobi rc $? $(( $(date +%s) - t0 )) s"
timeout 1200 bash tools/gpu_pmc.sh r06_b_n73 --geometry cube --level 0 --n0 73 --algebraic
echo "pmc n73 rc $? $(( $(date +%s) - t0 )) s"
tail -3 $out/r06_b_pytest.txt
grep rss $out/r06_b_bench_cube_n48_inner_jacobi.err
