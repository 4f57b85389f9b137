#!/bin/bash
# round 6, GPU call A: new tests, the north star's literal solvers (bench
# --inner jacobi) at level 6 and cube N = 73, counter passes (level 6 jacobi:
# the CG iteration; cube N = 73: every launch of one PCApply)
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_abi_closed.py tests/test_kernels_random_gpu.py "tests/test_api_gpu.py::test_bench_line_of_the_north_stars_literal_solvers" -x -q -m gpu > $out/r06_a_pytest.txt 2>&1
echo "pytest rc $?" | tee -a $out/r06_a_pytest.txt
timeout 600 python3 bench.py --inner jacobi --steps 20 --warmup 5 > $out/r06_a_bench_level6_inner_jacobi.json 2> $out/r06_a_bench_level6_inner_jacobi.err
echo "bench jacobi l6 rc $?"
timeout 900 bash tools/gpu_pmc.sh r06_a_level6_jacobi --inner jacobi
echo "pmc l6 jacobi rc $?"
timeout 900 python3 bench.py --geometry cube --level 0 --n0 73 --inner jacobi --steps 5 --warmup 2 --cpu-seconds 10 > $out/r06_a_bench_cube_n73_inner_jacobi.json 2> $out/r06_a_bench_cube_n73_inner_jacobi.err
echo "bench jacobi n73 rc $?"
timeout 1200 bash tools/gpu_pmc.sh r06_a_n73 --geometry cube --level 0 --n0 73 --algebraic
echo "pmc n73 rc $?"
tail -3 $out/r06_a_pytest.txt
tail -c 400 $out/r06_a_bench_level6_inner_jacobi.json
