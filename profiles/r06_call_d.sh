#!/bin/bash
# round 6, GPU call D: the driver's commands (whole suite, smoke, bench) +
# 16-lane rows on config 5's first coarse level A/B
out=gpurun_out; mkdir -p $out
t0=$(date +%s)
bash tools/driver_commands.sh r06_d
echo "driver commands done $(( $(date +%s) - t0 )) s"
PCD_VT_LONG_ROW=0 timeout 900 python3 bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 20 --warmup 5 --no-cpu-baseline > $out/r06_d_bench_cube_n73_rows64_on_level1.json 2> $out/r06_d_bench_cube_n73_rows64_on_level1.err
echo "bench n73 A rc $? $(( $(date +%s) - t0 )) s"
timeout 900 python3 bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 20 --warmup 5 --no-cpu-baseline > $out/r06_d_bench_cube_n73_rows16_on_level1.json 2> $out/r06_d_bench_cube_n73_rows16_on_level1.err
echo "bench n73 B rc $? $(( $(date +%s) - t0 )) s"
