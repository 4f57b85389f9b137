#!/bin/bash
# round 6, GPU call E: the driver's commands again + config 5 through the
# driver's multi-GPU command on ONE GPU (8 processes, rank-local device producer)
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
bash tools/driver_commands.sh r06_e
echo "driver commands done $(( $(date +%s) - t0 )) s"
PCD_REPLICATE_BELOW=1500 timeout 600 python3 bench.py --gpus 2 --share-gpu --geometry cube --level 0 --n0 16 --algebraic --partitioned-producer --steps 10 --warmup 3 > $out/r06_e_bench_2_processes_cube_n16_rank_local_producer.json 2> $out/r06_e_bench_2_processes_cube_n16_rank_local_producer.err
echo "bench 2 procs n16 rc $? $(( $(date +%s) - t0 )) s"
timeout 1500 python3 bench.py --gpus 8 --share-gpu --geometry cube --level 0 --n0 73 --algebraic --steps 10 --warmup 3 > $out/r06_e_bench_config5_own_mesh_8_processes_rank_local_device_producer.json 2> $out/r06_e_bench_config5_own_mesh_8_processes_rank_local_device_producer.err
echo "bench 8 procs n73 rc $? $(( $(date +%s) - t0 )) s"
tail -c 1500 $out/r06_e_bench_config5_own_mesh_8_processes_rank_local_device_producer.json
