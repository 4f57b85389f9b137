#!/bin/bash
# round 6, GPU call C: the rank-local device producer, fixed tests, lane-major
# tiles for long rows A/B at config 5's size
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 1500 python -m pytest tests/test_partitioned_device_producer_gpu.py tests/test_abi_closed.py "tests/test_api_gpu.py::test_bench_line_of_the_north_stars_literal_solvers" tests/test_device_producer_gpu.py -q -m gpu --durations=15 > $out/r06_c_pytest.txt 2>&1
echo "pytest rc $? $(( $(date +%s) - t0 )) s" | tee -a $out/r06_c_pytest.txt
export FENAPACK_AMD_RSS_TRACE=1
PCD_LM_ROW_ENTRIES=0 timeout 900 python3 bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 20 --warmup 5 --no-cpu-baseline > $out/r06_c_bench_cube_n73_direct_tiles_on_level1.json 2> $out/r06_c_bench_cube_n73_direct_tiles_on_level1.err
echo "bench n73 A rc $? $(( $(date +%s) - t0 )) s"
timeout 1200 python3 bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 20 --warmup 5 --cpu-seconds 20 > $out/r06_c_bench_cube_n73_lane_major_on_level1.json 2> $out/r06_c_bench_cube_n73_lane_major_on_level1.err
echo "bench n73 B rc $? $(( $(date +%s) - t0 )) s"
unset FENAPACK_AMD_RSS_TRACE
timeout 1200 bash tools/gpu_pmc.sh r06_c_n73 --geometry cube --level 0 --n0 73 --algebraic
echo "pmc n73 rc $? $(( $(date +%s) - t0 )) s"
tail -5 $out/r06_c_pytest.txt
