#!/bin/bash
# round 6, GPU call L: after the host-side gathers (monolithic values, the
# velocity block's extraction) went native - the driver's commands, and config
# 5's own mesh for its set-up time
out=gpurun_out
bash tools/driver_commands.sh r06_zzzzz
export FENAPACK_AMD_RSS_TRACE=1
timeout 1200 python3 bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 20 --warmup 5 --no-cpu-baseline > $out/r06_l_bench_cube_n73_native_host_gathers.json 2> $out/r06_l_bench_cube_n73_native_host_gathers.err
echo "bench n73 rc $?"; grep rss $out/r06_l_bench_cube_n73_native_host_gathers.err
