#!/bin/bash
# round 6, GPU call O: where config 5's set-up time goes (cProfile of the bench
# process + the engine's own phase timer)
out=gpurun_out
PCD_SETUP_TIMING=1 timeout 1500 python3 -m cProfile -o /tmp/n73.prof bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 5 --warmup 2 --no-cpu-baseline --no-producer > $out/r06_o_bench_cube_n73_profiled.json 2> $out/r06_o_bench_cube_n73_setup_phases.txt
echo "bench rc $?"
python3 - <<'PY' > gpurun_out/r06_o_cprofile_cube_n73_setup.txt 2>&1
import pstats
st = pstats.Stats("/tmp/n73.prof")
st.sort_stats("cumulative").print_stats(60)
st.sort_stats("tottime").print_stats(40)
PY
grep "pcd set-up" $out/r06_o_bench_cube_n73_setup_phases.txt | sort -k4 -n -r | head -30
