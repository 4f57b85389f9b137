#!/bin/bash
# round 6, GPU call Q: the one-launch CG iteration - parity, then the bench line
out=gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_kernels_random_gpu.py "tests/test_api_gpu.py::test_bench_line_of_the_north_stars_literal_solvers" -x -q -m gpu > $out/r06_q_pytest.txt 2>&1
echo "pytest rc $?"; tail -4 $out/r06_q_pytest.txt
timeout 600 python3 bench.py --inner jacobi --steps 20 --warmup 5 > $out/r06_q_jacobi_bench_level6_one_launch_cg.json 2> $out/r06_q_jacobi.err
echo "bench l6 jacobi rc $?"
PCD_CGSR_FUSED_ROWS=0 timeout 600 python3 bench.py --inner jacobi --steps 20 --warmup 5 --no-cpu-baseline > $out/r06_q_jacobi_bench_level6_two_launch_cgsr.json 2> $out/r06_q_jacobi2.err
echo "bench l6 jacobi (two-launch) rc $?"
