#!/bin/bash
# round 6, GPU call H: the partitioned path against the oracle (child process),
# the new rank-local producer tests, the driver's commands on the final tree,
# the single-reduction CG rehearsals
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
timeout 1200 python3 tools/parity_partitioned.py --n0 48 --ranks 8 > $out/r06_h_parity_cube_n48_8_thread_ranks_vs_oracle.json 2> $out/r06_h_parity_cube_n48.err
echo "parity n48 rc $? $(( $(date +%s) - t0 )) s"; tail -c 600 $out/r06_h_parity_cube_n48_8_thread_ranks_vs_oracle.json
timeout 900 python -m pytest tests/test_partitioned_device_producer_gpu.py -q -m gpu --durations=10 > $out/r06_h_pytest_rank_local.txt 2>&1
echo "pytest rank-local rc $? $(( $(date +%s) - t0 )) s"; tail -3 $out/r06_h_pytest_rank_local.txt
bash tools/driver_commands.sh r06_zzz
echo "driver commands done $(( $(date +%s) - t0 )) s"
timeout 600 python3 bench.py --gpus 2 --share-gpu --inner jacobi --steps 5 --warmup 2 --cpu-seconds 2 > $out/r06_zzz_jacobi_2_processes_share_gpu_level6_single_reduction_cg.json 2> $out/r06_zzz_jacobi_2_processes.err
echo "bench 2 procs jacobi rc $? $(( $(date +%s) - t0 )) s"
timeout 600 python3 bench.py --gpus 8 --share-gpu --inner jacobi --level 4 --steps 5 --warmup 2 --cpu-seconds 2 > $out/r06_zzz_jacobi_8_processes_share_gpu_level4_single_reduction_cg.json 2> $out/r06_zzz_jacobi_8_processes.err
echo "bench 8 procs jacobi l4 rc $? $(( $(date +%s) - t0 )) s"
