#!/bin/bash
# round 6, GPU call F: the measurement set on the final kernel sources
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
lap() { echo "$1 rc $2 $(( $(date +%s) - t0 )) s"; }
# --- headline size: counters, kernel stats, timeline
timeout 600 bash tools/gpu_pmc.sh r06_zzz; lap "pmc l6" $?
timeout 600 bash tools/gpu_kernel_stats.sh r06_zzz_bench_level6 --steps 20 --warmup 5; lap "stats l6" $?
timeout 600 bash tools/gpu_timeline.sh r06_zzz_one_pcapply; lap "timeline l6" $?
# --- the north star's literal solvers
timeout 600 bash tools/gpu_pmc.sh r06_zzz_jacobi --inner jacobi; lap "pmc l6 jacobi" $?
timeout 600 python3 bench.py --inner jacobi --steps 20 --warmup 5 > $out/r06_zzz_jacobi_bench_level6.json 2> $out/r06_zzz_jacobi_bench_level6.err; lap "bench l6 jacobi" $?
timeout 1200 python3 bench.py --geometry cube --level 0 --n0 73 --inner jacobi --steps 5 --warmup 2 --cpu-seconds 10 > $out/r06_zzz_jacobi_bench_cube_n73.json 2> $out/r06_zzz_jacobi_bench_cube_n73.err; lap "bench n73 jacobi" $?
# --- the reference's own iterative option string at the headline size (S-ref-iter)
timeout 600 python3 bench.py --cycles-p 2 --supg --rediscretise-u --steps 20 --warmup 5 > $out/r06_zzz_sref_bench_level6_cycles_p2_supg.json 2> $out/r06_zzz_sref_bench_level6_cycles_p2_supg.err; lap "bench l6 s-ref-iter" $?
# --- beyond the Infinity Cache
timeout 900 python3 bench.py --level 7 --steps 20 --warmup 5 > $out/r06_zzz_l7_bench_cavity_level7.json 2> $out/r06_zzz_l7.err; lap "bench l7" $?
timeout 900 bash tools/gpu_pmc.sh r06_zzz_l7 --level 7; lap "pmc l7" $?
timeout 900 python3 bench.py --level 7 --re 1000 --supg --rediscretise-u --cycles-u 2 --cycles-p 2 --smooth 3 --steps 20 --warmup 5 > $out/r06_zzz_config3_bench_cavity_level7_re1000_supg.json 2> $out/r06_zzz_config3.err; lap "bench config3" $?
timeout 900 python3 bench.py --geometry cube --level 3 --n0 6 --steps 20 --warmup 5 > $out/r06_zzz_n48_bench_cube_n48.json 2> $out/r06_zzz_n48.err; lap "bench n48" $?
timeout 900 bash tools/gpu_pmc.sh r06_zzz_n48 --geometry cube --level 3 --n0 6; lap "pmc n48" $?
timeout 1500 python3 bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 20 --warmup 5 --cpu-seconds 20 > $out/r06_zzz_n73_bench_cube_n73_config5_own_mesh_one_gpu.json 2> $out/r06_zzz_n73.err; lap "bench n73" $?
timeout 1200 bash tools/gpu_pmc.sh r06_zzz_n73 --geometry cube --level 0 --n0 73 --algebraic; lap "pmc n73" $?
timeout 900 bash tools/gpu_timeline.sh r06_zzz_one_pcapply_cube_n73_gamg --geometry cube --level 0 --n0 73 --algebraic; lap "timeline n73" $?
ls -la $out | grep r06_zzz | head -40
