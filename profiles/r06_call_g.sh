#!/bin/bash
# round 6, GPU call G: the driver's commands on the final tree, the rehearsals
# of the multi-rank paths on one GPU, the first-contact kit
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
bash tools/driver_commands.sh r06_zzz
echo "driver commands done $(( $(date +%s) - t0 )) s"
timeout 900 python3 bench.py --gpus 8 --share-gpu --inner jacobi --steps 5 --warmup 2 --cpu-seconds 2 > $out/r06_zzz_jacobi_8_processes_share_gpu_level6_single_reduction_cg.json 2> $out/r06_zzz_jacobi_8_processes.err
echo "bench 8 procs jacobi rc $? $(( $(date +%s) - t0 )) s"
timeout 3000 python3 tools/first_contact.py --share-gpu --ranks 2,8 --out $out/r06_zzz_first_contact > $out/r06_zzz_first_contact_summary.txt 2>&1
echo "first contact rc $? $(( $(date +%s) - t0 )) s"
cat $out/r06_zzz_first_contact_summary.txt | tail -15
